#!/usr/bin/env python3
"""Precision of the pooling contraction with the two cross terms of the f16 hi/lo split carried in fp6 (e2m3) slots of
v_mfma_scale_f32_16x16x128_f8f6f4, simulated in float64 on the CPU before anything is built:

    sum_px s L  ~  sum hs hL                               (f16 x f16, exact products, as today)
                 + 2^-10 sum [fp6(s / S), fp6(1024 r / S)] . [fp6(1024 lL / T), fp6(hL / T)] S T    (one block-scaled instruction)

with hs = f16 truncation of the stream value s, r = s - hs, hL = f16 rounding of the LUT value L, lL = L - hL, and S / T the
per-lane power-of-two block scales (a lane = 8 pixels of one patch row; S from the largest magnitude value of the row
segment, T fixed on the host per LUT lane).  Reports the relative L2 error of the final 128-D descriptor against float64
for: the split as the kernel does it today (three f16 terms), the fp6 form, and the two-term form (no cross terms).
usage: tools/sim_fp6.py [n_patches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def f16_rtz(x):
    """f32 -> f16 by truncation (v_cvt_pkrtz_f16_f32), returned as float64"""
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)                       # nearest
    over = np.abs(h.astype(np.float32)) > np.abs(x)
    h = np.where(over, np.nextafter(h, np.float16(0)), h)
    return h.astype(np.float64)


E2M3 = np.array(sorted({(m / 8.0 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)) for e in range(4) for m in range(8)}))


def fp6(x):
    """nearest e2m3 value (saturating at 7.5), sign kept"""
    a = np.minimum(np.abs(x), 7.5)
    i = np.clip(np.searchsorted(E2M3, a), 1, len(E2M3) - 1)
    lo, hi = E2M3[i - 1], E2M3[i]
    return np.sign(x) * np.where(a - lo <= hi - a, lo, hi)


def pow2_scale(mx):
    """power of two S with mx / S in [2, 4) (the top of e2m3's range is 7.5)"""
    mx = np.maximum(mx, 1e-30)
    return 2.0 ** (np.floor(np.log2(mx)) - 1)


def streams_of(patches, mode="shader"):
    P = np.asarray(patches, np.float64).reshape(-1, 32, 32)
    pv = np.pad(P, ((0, 0), (2, 2), (0, 0)), mode="edge")
    V = sum(gg.BLUR[i] * pv[:, i:i + 32, :] for i in range(5))
    ph = np.pad(V, ((0, 0), (0, 0), (2, 2)), mode="edge")
    B = sum(gg.BLUR[i] * ph[:, :, i:i + 32] for i in range(5))
    bx = np.pad(B, ((0, 0), (0, 0), (1, 1)), mode="edge")
    by = np.pad(B, ((0, 0), (1, 1), (0, 0)), mode="edge")
    gx = bx[:, :, 0:32] - bx[:, :, 2:34]
    gy = by[:, 2:34, :] - by[:, 0:32, :]
    mag = (gx ** 2 + gy ** 2 + 1e-8) ** 0.25
    th = -(gg.atan2_shader(gx, gy) if mode == "shader" else np.arctan2(gy, gx))
    s = [mag] + [mag * np.cos(k * th) for k in (1, 2, 3)] + [mag * np.sin(k * th) for k in (1, 2, 3)]
    return np.stack(s, axis=1).astype(np.float32).astype(np.float64)      # [n, 7, 32, 32], f32 values


def lut_columns():
    """(stream index, LUT [32,32], output slot, sign) per product, the rotation by phi folded into the LUT as in the kernel"""
    phi, ep, ec = gg.luts()
    c = gg.C_N3K8
    prods = []          # (stream, lut, ("p"|"c", in_dim, j), sign)
    for j in range(25):
        prods.append((0, c[0] * ep[j], ("p", 0, j), 1.0))
    for j in range(9):
        prods.append((0, c[0] * ec[j], ("c", 0, j), 1.0))
    for k in (1, 2, 3):
        ck, sk = np.cos(k * phi), np.sin(k * phi)
        for j in range(25):
            prods.append((k, c[k] * ep[j] * ck, ("p", k, j), 1.0))          # relcos = cos x EPc - sin x EPs
            prods.append((3 + k, c[k] * ep[j] * sk, ("p", k, j), -1.0))
            prods.append((3 + k, c[k] * ep[j] * ck, ("p", 3 + k, j), 1.0))  # relsin = sin x EPc + cos x EPs
            prods.append((k, c[k] * ep[j] * sk, ("p", 3 + k, j), 1.0))
        for j in range(9):
            prods.append((k, c[k] * ec[j], ("c", k, j), 1.0))
            prods.append((3 + k, c[k] * ec[j], ("c", 3 + k, j), 1.0))
    return prods


def pooled(streams, how):
    n = len(streams)
    polar, cart = np.zeros((n, 7, 25)), np.zeros((n, 7, 9))
    # stream-side pieces, per (patch, stream, row, segment of 8 pixels)
    s = streams
    hs = f16_rtz(s)
    r = s - hs
    ls = f16_rtz(r)
    seg = np.abs(s[:, 0]).reshape(n, 32, 4, 8).max(axis=3)                 # block scale from the m stream's segment
    S = np.repeat(pow2_scale(seg)[:, None, :, :, None], 8, axis=4).reshape(n, 1, 32, 32)
    qa, qr = fp6(s / S), fp6(1024.0 * r / S)
    for (si, lut, (blk, i, j), sign) in lut_columns():
        L = lut.astype(np.float32).astype(np.float64)
        if how == "f64":
            v = np.einsum("npq,pq->n", s[:, si], L)
        elif how in ("f16x3", "f16x2"):
            hL = f16_rtz(L)
            lL = (L - hL).astype(np.float32).astype(np.float16).astype(np.float64)
            v = np.einsum("npq,pq->n", hs[:, si], hL)
            if how == "f16x3":
                v += np.einsum("npq,pq->n", hs[:, si], lL) + np.einsum("npq,pq->n", ls[:, si], hL)
        else:   # fp6 cross terms
            hL = L.astype(np.float32).astype(np.float16).astype(np.float64)                  # nearest
            lL = L - hL
            both = np.maximum(np.abs(1024.0 * lL), np.abs(hL)).reshape(32, 4, 8).max(axis=2)
            T = np.repeat(pow2_scale(both)[:, :, None], 8, axis=2).reshape(32, 32)
            ql, qh = fp6(1024.0 * lL / T), fp6(hL / T)
            v = np.einsum("npq,pq->n", hs[:, si], hL)
            v += np.einsum("npq,pq->n", qa[:, si] * S[:, 0], ql * T) / 1024.0
            v += np.einsum("npq,pq->n", qr[:, si] * S[:, 0], qh * T) / 1024.0
        (polar if blk == "p" else cart)[:, i, j] += sign * v
    return polar.reshape(n, 175), cart.reshape(n, 63)


def finish(polar, cart, pca):
    mean, eigvals, eigvecs = pca
    polar = polar / np.linalg.norm(polar, axis=1, keepdims=True)
    cart = cart / np.linalg.norm(cart, axis=1, keepdims=True)
    raw = np.concatenate([polar, cart], axis=1)
    raw /= np.linalg.norm(raw, axis=1, keepdims=True)
    W = eigvecs[:, :128] * eigvals[:128] ** float(np.float32(-0.5) * np.float32(0.7))
    d = (raw - mean) @ W
    return d / np.linalg.norm(d, axis=1, keepdims=True), raw


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(3)
    sets = {"uniform random": rng.random((n, 32, 32)).astype(np.float32),
            "structured": np.asarray(gg.structured_patches(), np.float32),
            "smooth (keypoint-like)": np.stack([gg.smooth_image(32, 32, 100 + i) for i in range(min(n, 64))]),
            "low contrast": (0.5 + 0.02 * rng.random((min(n, 64), 32, 32))).astype(np.float32)}
    pca = gg.load_pca("liberty")
    for name, p in sets.items():
        st = streams_of(p)
        ref, ref_raw = finish(*pooled(st, "f64"), pca)
        line = f"{name:24s} ({len(p):4d} patches): relative L2 of the 128-D descriptor vs float64, worst / mean:"
        for how in ("f16x3", "fp6", "f16x2"):
            d, raw = finish(*pooled(st, how), pca)
            e = np.linalg.norm(d - ref, axis=1) / np.linalg.norm(ref, axis=1)
            line += f"   {how} {e.max():.2e} / {e.mean():.2e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
