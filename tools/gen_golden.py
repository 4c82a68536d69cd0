#!/usr/bin/env python3
"""Independent NumPy float64 restatement of the reference MKD path + golden generator.

Purpose: the reference (tnibler/local-features) pins no numerical results for this
path, so two restatements written separately from the same sources pin each other:
the C f32 oracle (oracle/mkd_oracle.c) and this vectorised float64 one.  This script
writes tests/golden/*.npz (inputs + float64-computed outputs stored as f32); the
tests check the C oracle, and the HIP path, against them.

Sources followed (paths relative to /root/reference/local_features/src/):
  mkd_ref.rs:7-9,146-267      LUT builders        vulkan/mod.rs:1594-1619  constants
  vulkan/shaders/mkd/patch_gradients.glsl:20-28,42-104   vulkan/shaders/atan2.glsl:19-46
  vulkan/shaders/mkd/embedding.glsl:34-88  normalize.glsl  whitening.glsl  normalize_final.glsl
  vulkan/patch_pyramid.rs, shaders/blur.glsl, swt.glsl, blur_pyramid.glsl (keypoint mode)

Run:  python tools/gen_golden.py   (needs only numpy; reads the PCA models shipped in
local-features_amd/models/mkd/, byte-identical copies of the reference's data files).
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "local-features_amd", "models", "mkd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

C_N3K8 = np.array([0.37872374, 0.51796234, 0.46882015, 0.39798096], np.float32).astype(np.float64)
C_N1K1 = np.array([0.618176, 0.6934725], np.float32).astype(np.float64)
C_N2K8 = C_N3K8[:3]
BLUR = np.array([0.0096, 0.2054, 0.5699, 0.2054, 0.0096], np.float32).astype(np.float64)


def load_pca(name):
    b = open(os.path.join(MODELS, f"concat-pca-{name}.safetensors"), "rb").read()
    (hl,) = struct.unpack("<Q", b[:8])
    hdr = json.loads(b[8:8 + hl])
    out = {}
    for k, v in hdr.items():
        if k == "__metadata__":
            continue
        s, e = v["data_offsets"]
        out[k] = np.frombuffer(b[8 + hl + s:8 + hl + e], "<f4").reshape(v["shape"]).astype(np.float64)
    return out["mean"], out["eigvals"], out["eigvecs"]


def vm(t, c):
    n = len(c) - 1
    return np.stack([np.full_like(t, c[0])] + [c[k] * np.cos(k * t) for k in range(1, n + 1)]
                    + [c[k] * np.sin(k * t) for k in range(1, n + 1)])


def luts():
    ax = 2.0 * np.arange(32) / 31.0 - 1.0
    gx, gy = np.meshgrid(ax, ax)            # gx varies along x (columns), gy along y (rows)
    phi = -np.arctan2(gy, gx)               # gradient_angle
    rho = np.sqrt(gx ** 2 + gy ** 2 + 1e-8)
    nrm = np.sqrt(gx ** 2 + gy ** 2)
    g = np.exp(-(nrm / nrm.max()) ** 2)
    a, b = vm(gx * np.pi / 2, C_N1K1), vm(gy * np.pi / 2, C_N1K1)
    ec = np.stack([a[i] * b[j] * g for i in range(3) for j in range(3)])
    a, b = vm(-phi, C_N2K8), vm(rho * np.pi / np.sqrt(2.0), C_N2K8)
    ep = np.stack([a[i] * b[j] * g for i in range(5) for j in range(5)])
    return phi, ep, ec


def atan2_shader(x, y):
    """shaders/atan2.glsl evaluated in f64; atan2(x, y) = angle of point (x, y)."""
    A = [0.99997726, -0.33262347, 0.19354346, -0.11643287, 0.05265332, -0.0117212]
    A = [float(np.float32(v)) for v in A]
    PI, HPI = float(np.float32(3.1415927)), float(np.float32(1.5707964))
    swap = np.abs(x) < np.abs(y)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(swap, x / np.where(y == 0, 1, y), y / np.where(x == 0, 1, x))
    s = a * a
    p = a * (A[0] + s * (A[1] + s * (A[2] + s * (A[3] + s * (A[4] + s * A[5])))))
    res = np.where(swap, HPI * np.sign(a) - p, p)
    res = np.where(x < 0, PI * np.where(y < 0, -1.0, 1.0) + res, res)
    return np.where((x == 0) & (y == 0), 0.0, res)


def describe(patches, pca, mode="shader", want_raw=False):
    phi, ep, ec = luts()
    mean, eigvals, eigvecs = pca
    P = np.asarray(patches, np.float64).reshape(-1, 32, 32)
    pv = np.pad(P, ((0, 0), (2, 2), (0, 0)), mode="edge")
    V = sum(BLUR[i] * pv[:, i:i + 32, :] for i in range(5))
    ph = np.pad(V, ((0, 0), (0, 0), (2, 2)), mode="edge")
    B = sum(BLUR[i] * ph[:, :, i:i + 32] for i in range(5))
    bx = np.pad(B, ((0, 0), (0, 0), (1, 1)), mode="edge")
    by = np.pad(B, ((0, 0), (1, 1), (0, 0)), mode="edge")
    gx = bx[:, :, 0:32] - bx[:, :, 2:34]     # left - right
    gy = by[:, 2:34, :] - by[:, 0:32, :]     # down - up
    mag = (gx ** 2 + gy ** 2 + 1e-8) ** 0.25
    th = -(atan2_shader(gx, gy) if mode == "shader" else np.arctan2(gy, gx))

    def emb(t):
        return np.stack([C_N3K8[0] * mag] + [C_N3K8[k] * np.cos(k * t) * mag for k in (1, 2, 3)]
                        + [C_N3K8[k] * np.sin(k * t) * mag for k in (1, 2, 3)], axis=1)
    polar = np.einsum("nipq,jpq->nij", emb(th + phi), ep).reshape(len(P), 175)
    cart = np.einsum("nipq,jpq->nij", emb(th), ec).reshape(len(P), 63)
    polar /= np.linalg.norm(polar, axis=1, keepdims=True)
    cart /= np.linalg.norm(cart, axis=1, keepdims=True)
    raw = np.concatenate([polar, cart], axis=1)
    raw /= np.linalg.norm(raw, axis=1, keepdims=True)
    W = eigvecs[:, :128] * eigvals[:128] ** float(np.float32(-0.5) * np.float32(0.7))
    d = (raw - mean) @ W
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (d, raw) if want_raw else d


# ---------------------------------------------------------------- keypoint mode
def mirror(i, n):
    m = np.mod(i, 2 * n)
    return np.where(m < n, m, 2 * n - 1 - m)


def tex(img, u, v):
    """Bilinear, MirroredRepeat, unnormalised coords with texel centres at i+0.5."""
    h, w = img.shape
    fu, fv = u - 0.5, v - 0.5
    x0, y0 = np.floor(fu), np.floor(fv)
    ax, ay = fu - x0, fv - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    xa, xb, ya, yb = mirror(x0, w), mirror(x0 + 1, w), mirror(y0, h), mirror(y0 + 1, h)
    top = img[ya, xa] * (1 - ax) + img[ya, xb] * ax
    bot = img[yb, xa] * (1 - ax) + img[yb, xb] * ax
    return top * (1 - ay) + bot * ay


def sep3(img, w0, w1, off):
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w]
    cx, cy = xx + 0.5, yy + 0.5
    t = tex(img, cx, cy) * w0 + (tex(img, cx - off, cy) + tex(img, cx + off, cy)) * w1
    return tex(t, cx, cy) * w0 + (tex(t, cx, cy - off) + tex(t, cx, cy + off)) * w1


def pyramid(img):
    img = np.asarray(img, np.float64)
    h, w = img.shape
    L = int(np.ceil(np.log2(min(w, h))))
    W0, W1, OFF = (float(np.float32(v)) for v in (0.66381836, 0.16809084, 0.015267163))
    lv = [sep3(img, W0, W1, 1.0 + OFF)]
    k = np.array([1, 4, 6, 4, 1]) / 16.0
    a = lv[0]
    t = sum(k[i] * a[:, mirror(np.arange(w) + i - 2, w)] for i in range(5))
    c1 = sum(k[i] * t[mirror(np.arange(h) + i - 2, h), :] for i in range(5))
    lw, lh = max(w >> 1, 1), max(h >> 1, 1)
    sx = np.minimum(np.floor((np.arange(lw) + 0.5) * w / (w // 2)).astype(int), w - 1)
    sy = np.minimum(np.floor((np.arange(lh) + 0.5) * h / (h // 2)).astype(int), h - 1)
    lv.append(c1[np.ix_(sy, sx)])
    for l in range(2, L):
        p = lv[-1]
        ph, pw = p.shape
        lw, lh = max(w >> l, 1), max(h >> l, 1)
        yy, xx = np.mgrid[0:ph, 0:pw]
        cx, cy = xx + 0.5, yy + 0.5
        t = tex(p, cx, cy) * 0.375 + (tex(p, cx - 1.2, cy) + tex(p, cx + 1.2, cy)) * 0.3125
        yy, xx = np.mgrid[0:lh, 0:lw]
        cx, cy = 2.0 * xx + 0.5, 2.0 * yy + 0.5
        lv.append(tex(t, cx, cy) * 0.375 + (tex(t, cx, cy - 1.2) + tex(t, cx, cy + 1.2)) * 0.3125)
    return lv


def sample(lv, kps, psf=24.0):
    out = []
    ly, lx = np.mgrid[0:32, 0:32]
    dx, dy = lx - 16.0, ly - 16.0
    for x, y, size, ang in np.asarray(kps, np.float64):
        ls = np.log2(size * psf / 32.0)
        l = int(min(max(np.floor(ls), 0), len(lv) - 1))
        r = 2.0 ** (ls - l)
        a = np.deg2rad(ang)
        xx = dx * np.cos(a) - dy * np.sin(a)
        yy = dx * np.sin(a) + dy * np.cos(a)
        out.append(tex(lv[l], xx * r + x / 2 ** l + 0.5, yy * r + y / 2 ** l + 0.5))
    return np.stack(out)


# ---------------------------------------------------------------- keypoint orientation
def coarse_stack(img, n_scales=4):
    """vulkan/mod.rs:1093-1130: layer 0 = sigma-0.6 blur, layer l+1 = a-trous [1 4 6 4 1]/16 with dilation 2^l."""
    img = np.asarray(img, np.float64)
    h, w = img.shape
    W0, W1, OFF = (float(np.float32(v)) for v in (0.66381836, 0.16809084, 0.015267163))
    layers = [sep3(img, W0, W1, 1.0 + OFF)]
    k = np.array([1, 4, 6, 4, 1]) / 16.0
    for l in range(n_scales + 2):
        d, a = 1 << l, layers[-1]
        t = sum(k[i] * a[:, mirror(np.arange(w) + (i - 2) * d, w)] for i in range(5))
        layers.append(sum(k[i] * t[mirror(np.arange(h) + (i - 2) * d, h), :] for i in range(5)))
    return np.stack(layers)


def orient(stack, extrema):
    """shaders/keypoint_orientation.glsl:36-171 in float64; returns [m,5] (x, y, size, angle_deg, response)."""
    nl, h, w = stack.shape
    out = []
    for x, y, size, resp in np.asarray(extrema, np.float64):
        level = int(np.clip(np.floor(np.log2(size / (float(np.float32(0.82)) * np.sqrt(2.0))) + 0.5), 0, nl - 1))
        step = 1 << level
        radius = int(np.floor(3 * 1.5 * size / np.sqrt(2.0) + 0.5))
        sigma = 1.5 * size / np.sqrt(2.0)
        kx, ky = int(x), int(y)
        off = (np.arange(15) - 7) * step
        xi, yi = kx + off[None, :], ky + off[:, None]
        valid = (xi >= 0) & (xi < w) & (yi >= 0) & (yi <= h)
        patch = np.where(valid & (yi < h), stack[level][np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)], 0.0)
        ingrad = valid & (np.abs(off[None, :]) <= radius) & (np.abs(off[:, None]) <= radius)
        raw = np.zeros(40)
        for ly in range(1, 14):
            for lx in range(1, 14):
                if not ingrad[ly, lx]:
                    continue
                gx = patch[ly, lx + 1] - patch[ly, lx - 1]
                gy = patch[ly - 1, lx] - patch[ly + 1, lx]
                if gx == 0 and gy == 0:
                    continue
                dist = float(off[lx]) ** 2 + float(off[ly]) ** 2
                wgt = np.exp(-dist / (2 * sigma * sigma)) * np.hypot(gx, gy)
                ang = float(atan2_shader(np.float64(gx), np.float64(gy)))
                rb = int(np.floor(ang * 36 / (2 * float(np.float32(3.1415927))) + 0.5))
                raw[2 + (rb + 36 if rb < 0 else rb - 36 if rb >= 36 else rb)] += wgt
        raw[1], raw[0], raw[38], raw[39] = raw[37], raw[36], raw[2], raw[3]
        hist = np.array([(raw[b] + raw[b + 4]) / 16 + (raw[b + 1] + raw[b + 3]) * 4 / 16 + raw[b + 2] * 6 / 16
                         for b in range(36)])
        th = hist.max() * float(np.float32(0.8))
        for b in range(36):
            hv, le, ri = hist[b], hist[b - 1], hist[(b + 1) % 36]
            if le < hv and ri < hv and th <= hv:
                rbin = b + (le - ri) / (le - 2 * hv + ri) / 2
                rbin = rbin + 36 if rbin < 0 else rbin - 36 if rbin > 36 else rbin
                out.append((x, y, size, 360.0 - 10.0 * rbin, resp))
    return np.array(out, np.float64).reshape(-1, 5)


def scan_extrema(fine, border=5, skip=0, thr=float(np.float32(0.035))):
    """shaders/scan_extrema.glsl in float64 on a given DoG volume; order: 4x4x4 cubes raster, then local index;
    at most 8 candidates per cube.  Returns [m,4] (x, y, size, contrast)."""
    fine = np.asarray(fine, np.float64)
    nf, h, w = fine.shape
    b1 = max(border, 1)
    out = []
    gx, gy, gz = -(-(w - 2 * border) // 4), -(-(h - 2 * border) // 4), -(-(nf - 2 - skip) // 4)
    # candidates, vectorised: |v| > thr and sign*v >= sign*neighbour for all 26 neighbours
    core = fine[1:-1, 1:-1, 1:-1]
    sg = np.sign(core)
    ok = np.abs(core) > thr
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dz or dy or dx:
                    nb = fine[1 + dz:nf - 1 + dz, 1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]
                    ok &= sg * core >= sg * nb
    cand = np.zeros(fine.shape, bool)
    cand[1:-1, 1:-1, 1:-1] = ok
    cand[:1 + skip] = False
    cand[:, :b1] = False; cand[:, h - b1:] = False; cand[:, :, :b1] = False; cand[:, :, w - b1:] = False
    zs, ys, xs = np.nonzero(cand)
    cube = (((zs - 1 - skip) // 4) * gy + (ys - border) // 4) * gx + (xs - border) // 4
    local = (xs - border) % 4 + 4 * ((ys - border) % 4) + 16 * ((zs - 1 - skip) % 4)
    order = np.lexsort((local, cube))
    seen = {}
    A = lambda z, y, x: fine[z, y, x]
    for i in order:
        z, y, x = int(zs[i]), int(ys[i]), int(xs[i])
        seen[cube[i]] = seen.get(cube[i], 0) + 1
        if seen[cube[i]] > 8:
            continue
        g = np.array([(A(z + 1, y, x) - A(z - 1, y, x)) / 2, (A(z, y + 1, x) - A(z, y - 1, x)) / 2,
                      (A(z, y, x + 1) - A(z, y, x - 1)) / 2])
        v2 = 2 * A(z, y, x)
        h11 = A(z + 1, y, x) + A(z - 1, y, x) - v2
        h22 = A(z, y + 1, x) + A(z, y - 1, x) - v2
        h33 = A(z, y, x + 1) + A(z, y, x - 1) - v2
        h12 = (A(z + 1, y + 1, x) - A(z - 1, y + 1, x) - A(z + 1, y - 1, x) + A(z - 1, y - 1, x)) / 4
        h13 = (A(z + 1, y, x + 1) - A(z - 1, y, x + 1) - A(z + 1, y, x - 1) + A(z - 1, y, x - 1)) / 4
        h23 = (A(z, y + 1, x + 1) - A(z, y + 1, x - 1) - A(z, y - 1, x + 1) + A(z, y - 1, x - 1)) / 4
        H = np.array([[h11, h12, h13], [h12, h22, h23], [h13, h23, h33]])
        if np.linalg.det(H) == 0:
            off = np.full(3, np.nan)
        else:
            off = -np.linalg.solve(H, g)          # the shader spells out the adjugate; same thing
        if (np.abs(off) > 0.5).any():
            continue
        contrast = abs(A(z, y, x) + off @ g / 2)
        denom = (h22 + h33) ** 2
        if denom == 0:
            continue
        cm = 1 - 4 * (h22 * h33 - h23 * h23) / denom
        if 0.7 <= cm <= 1.5:
            continue
        size = float(np.float32(0.82)) * np.sqrt(2.0) * 2.0 ** (z + off[0])
        out.append((x + off[2], y + off[1], size, contrast))
    return np.array(out, np.float64).reshape(-1, 4)


def topk_filter(extrema, n, min_size=0.0):
    """TopKContrastFilter (vulkan/mod.rs:1753-1786), indexing by position (see the C oracle's note)."""
    idx = np.flatnonzero(extrema[:, 2] >= min_size)
    c = np.abs(extrema[idx, 3])
    if len(idx) <= n:
        return idx
    cutoff = np.sort(c)[::-1][n]
    return idx[c >= cutoff][:n]


def blob_image(w, h, seed, n_blobs=60):
    """Gaussian blobs of both signs and several sizes on a mid-grey ground: DoG extrema well above threshold."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 0.5)
    for _ in range(n_blobs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        s = rng.uniform(1.2, 9.0)
        a = rng.uniform(0.15, 0.45) * rng.choice([-1, 1])
        e = rng.uniform(0.6, 1.0)                      # some elongated ones for the edge test
        img += a * np.exp(-((xx - cx) ** 2 + ((yy - cy) / e) ** 2) / (2 * s * s))
    img += rng.normal(0, 0.004, img.shape)
    return np.clip(img, 0, 1).astype(np.float32)


def random_extrema(n, w, h, seed, n_scales=4):
    """sizes as the detector emits them: 0.82 sqrt2 2^(z+delta), z in [1, n_scales] (scan_extrema.glsl:229)."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(1.0, n_scales + 0.45, n)
    size = 0.82 * np.sqrt(2.0) * 2.0 ** z
    x = rng.uniform(4.0, w - 4.0, n)
    y = rng.uniform(4.0, h - 4.0, n)
    return np.stack([x, y, size, rng.uniform(0.04, 0.4, n)], axis=1).astype(np.float32)


# ---------------------------------------------------------------- inputs
def structured_patches():
    y, x = np.mgrid[0:32, 0:32].astype(np.float64)
    ps = [x / 31.0, y / 31.0, (x + 2 * y) / 93.0]
    ps.append(np.exp(-((x - 14.3) ** 2 + (y - 17.9) ** 2) / 40.0))
    for deg in (17.0, 61.0, 133.0):                       # step edges off the axes
        a = np.deg2rad(deg)
        ps.append(((x - 15.5) * np.cos(a) + (y - 15.5) * np.sin(a) > 0.3).astype(np.float64))
    ps.append(0.5 + 0.5 * np.sin(x * 0.7) * np.cos(y * 0.45))
    ps.append(np.full((32, 32), 0.37))                    # flat: mag==0.01, angle==0 everywhere
    ps.append(((x // 4 + y // 4) % 2).astype(np.float64))  # checkerboard: axis-aligned (gx==0 quirk)
    return np.stack(ps)


def smooth_image(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h + 16, w + 16))
    k = np.exp(-0.5 * (np.arange(-6, 7) / 2.0) ** 2)
    k /= k.sum()
    img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, img)
    img = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, img)
    img = img[8:-8, 8:-8]
    return ((img - img.min()) / (img.max() - img.min())).astype(np.float32)


def random_keypoints(n, w, h, seed, margin=8.0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(margin, w - margin, n)
    y = rng.uniform(margin, h - margin, n)
    size = np.exp(rng.uniform(np.log(1.64), np.log(min(52.0, min(w, h) / 6.0)), n))
    ang = rng.uniform(0.0, 360.0, n)
    return np.stack([x, y, size, ang], axis=1).astype(np.float32)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    rng = np.random.default_rng(0x4D4B44)
    rand = rng.random((16, 32, 32)).astype(np.float32)
    struct_p = structured_patches().astype(np.float32)
    patches = np.concatenate([rand, struct_p])
    for name in ("liberty", "notredame", "yosemite"):
        pca = load_pca(name)
        d_s, raw_s = describe(patches, pca, "shader", want_raw=True)
        d_l, raw_l = describe(patches, pca, "libm", want_raw=True)
        np.savez_compressed(
            os.path.join(GOLDEN, f"patches_{name}.npz"), patches=patches,
            desc_shader=d_s.astype(np.float32), raw_shader=raw_s.astype(np.float32),
            desc_libm=d_l.astype(np.float32), raw_libm=raw_l.astype(np.float32))
        print(name, "patch goldens:", patches.shape, "->", d_s.shape)
    # keypoint mode: one small frame, float64 pyramid + sampling + descriptors
    w, h = 200, 136
    img = smooth_image(h, w, 7)
    kps = random_keypoints(24, w, h, 11)
    lv = pyramid(img)
    sp = sample(lv, kps)
    pca = load_pca("liberty")
    d = describe(sp, pca, "shader")
    np.savez_compressed(
        os.path.join(GOLDEN, "keypoints_liberty.npz"), image=img, keypoints=kps,
        level1=lv[1].astype(np.float32), level3=lv[3].astype(np.float32),
        patches=sp.astype(np.float32), desc_shader=d.astype(np.float32))
    print("keypoint goldens:", img.shape, kps.shape, "->", d.shape)
    # keypoint orientation on the same frame: coarse stack + angles for 80 extrema (some touch the border)
    st = coarse_stack(img)
    ex = random_extrema(80, w, h, 13)
    ok = orient(st, ex)
    np.savez_compressed(os.path.join(GOLDEN, "orientation.npz"), image=img, extrema=ex,
                        layer2=st[2].astype(np.float32), layer5=st[5].astype(np.float32),
                        keypoints=ok.astype(np.float32))
    print("orientation goldens:", ex.shape, "->", ok.shape)
    # detector: blobs frame -> a-trous stack (f64) -> DoG; extremum scan in f64 on the f32-rounded DoG volume, so
    # that the fixture checks the scan logic and the refinement arithmetic, not ties between roundings
    bimg = blob_image(224, 160, 17, n_blobs=160)
    bst = coarse_stack(bimg)
    fine32 = (bst[:-1] - bst[1:]).astype(np.float32)
    bex = scan_extrema(fine32)
    keep = topk_filter(bex, 25)
    np.savez_compressed(os.path.join(GOLDEN, "detector.npz"), image=bimg, dog2=fine32[2], dog4=fine32[4],
                        extrema=bex.astype(np.float32), top25=keep.astype(np.uint32))
    print("detector goldens:", bimg.shape, "->", bex.shape, "extrema; top-25 keeps", len(keep))
    phi, ep, ec = luts()
    np.savez_compressed(os.path.join(GOLDEN, "luts.npz"), gradient_angle=phi.astype(np.float32),
                        embedding_polar=ep.astype(np.float32), embedding_cartesian=ec.astype(np.float32))


if __name__ == "__main__":
    sys.exit(main())
