#!/usr/bin/env python3
"""Soak of keypoint mode (the one-launch form: producer waves sample into the describe kernel's LDS ring) against the
oracle, end to end: random frame sizes, frames of different character, keypoints anywhere (borders included), every
detector size and beyond both ends of the pyramid, several frames per call.  Per round: the fused descriptors, the
two-launch form (must be the same bits) and the oracle's descriptors of the oracle's own patches; patches on which the
reference's two readings of the blur disagree are set aside (tests/conftest.py) and counted.  Not part of the test suite.
Usage: soak_keypoints.py [rounds] [keypoints per frame]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
                os.path.join(ROOT, "tools")]
import numpy as np, torch
import local_features_python as lfp
from oracle import ATAN_SHADER, MkdOracle
from conftest import GATE, rel_l2, settled_detail
from gen_golden import smooth_image

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(303)
orc = MkdOracle(lfp.model_path("liberty"))
worst = worst_patch = worst_same_bits = 0.0
tot = aside = over = 0
t0 = time.time()
for r in range(rounds):
    w, h = int(rng.integers(48, 1400)), int(rng.integers(40, 900))
    nf = int(rng.integers(1, 4))
    psf = float(rng.choice([12.0, 24.0, 24.0, 32.0, 48.0]))    # patch_scale_factor (lib.rs:34-52): the default and away from it
    kind = r % 3
    frames = []
    for f in range(nf):
        if kind == 0:
            img = smooth_image(h, w, 1000 + 10 * r + f)
        elif kind == 1:                                  # sharper: less blur, more structure at the fine levels
            g = np.random.default_rng(2000 + 10 * r + f)
            img = g.random((h, w)); img = 0.5 * img + 0.5 * smooth_image(h, w, 3000 + 10 * r + f)
        else:                                            # blocks: piecewise constant plus a little noise
            g = np.random.default_rng(4000 + 10 * r + f)
            img = np.kron(g.random((h // 16 + 1, w // 16 + 1)), np.ones((16, 16)))[:h, :w] + 0.02 * g.random((h, w))
        frames.append(np.ascontiguousarray(img, np.float32))
    ks, fid = [], []
    for f in range(nf):
        k = np.stack([rng.uniform(-2, w + 2, nk), rng.uniform(-2, h + 2, nk),
                      np.exp(rng.uniform(np.log(0.8), np.log(120.0), nk)), rng.uniform(-30, 400, nk), np.zeros(nk)], axis=1)
        ks.append(k.astype(np.float32)); fid.append(np.full(nk, f, np.int32))
    k5, fid = np.concatenate(ks), np.concatenate(fid)
    d_img = torch.from_numpy(np.stack(frames)).cuda()
    d_k, d_f = torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
    outs = []
    for flags in (0, lfp.FLAG_UNFUSED_KEYPOINTS):
        hnd = lfp.MkdHandle(max_features=1024, max_image_width=w, max_image_height=h, max_frames=nf, flags=flags,
                            patch_scale_factor=psf)
        o = torch.empty((len(k5), 128), device="cuda")
        hnd.set_images_device(d_img.data_ptr(), nf, w, h)
        hnd.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), len(k5), o.data_ptr())
        outs.append(o.cpu().numpy())
    # (requests of at most 4096 keypoints take the row-split form when fused: its sums round differently from the patch
    #  kernel's, which the two-launch form uses -- same bits above that size, within 1e-5 below)
    same = np.array_equal(outs[0], outs[1]) or float(rel_l2(outs[0], outs[1]).max()) < 1e-5
    errs = []
    one = lfp.MkdHandle(max_features=1024, max_image_width=w, max_image_height=h, patch_scale_factor=psf)
    for f in range(nf):
        sel = fid == f
        ref_p = orc.sample_patches(orc.build_pyramid(frames[f]), w, h, k5[sel, :4], psf)
        one.set_image(frames[f])                        # the GPU's own patches of these keypoints (the verification tap)
        d_p = torch.empty((int(sel.sum()), 32, 32), device="cuda")
        d_ks = torch.from_numpy(k5[sel]).cuda()
        one.sample_patches_device(d_ks.data_ptr(), int(sel.sum()), d_p.data_ptr())
        got_p = d_p.cpu().numpy()
        worst_patch = max(worst_patch, float(np.abs(got_p - ref_p).max()))
        ok_ref, _, ref, _, _ = settled_detail(orc, ref_p, ATAN_SHADER)
        ok_got, ref_c, _, _, _ = settled_detail(orc, got_p, ATAN_SHADER)
        # describe stage on the GPU's own patch bits: every patch against the reading the kernel implements
        worst_same_bits = max(worst_same_bits, float(rel_l2(outs[0][sel], ref_c).max()))
        ok = ok_ref & ok_got                            # end to end: patches on which either side's readings differ set aside
        e = rel_l2(outs[0][sel], ref)
        errs.append(e[ok]); tot += int(sel.sum()); aside += int((~ok).sum()); over += int((e[ok] >= GATE).sum())
    e = np.concatenate(errs)
    worst = max(worst, float(e.max()))
    print(f"round {r:2d}: {nf} frame(s) {w}x{h} kind {kind}, patch_scale_factor {psf:g}, {len(k5)} keypoints: fused ~ two-launch {same}; settled max {e.max():.2e} "
          f"p99.9 {np.quantile(e, 0.999):.2e}; finite {bool(np.isfinite(outs[0]).all())}", flush=True)
print(f"soak_keypoints: {tot} keypoints, {aside} set aside (the reference's own two readings of the blur differ on either side's "
      f"patch), worst settled relative L2 end to end {worst:.2e}, >= 1e-4: {over}; describe stage on the GPU's own patch bits, "
      f"every patch: worst {worst_same_bits:.2e}; sampled values vs the oracle's: worst |diff| {worst_patch:.1e}; {time.time() - t0:.0f} s")
