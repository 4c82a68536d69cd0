#!/usr/bin/env python3
"""Soak of the pyramid and a-trous kernels against the oracle, bit for bit, over random frame shapes: widths that are and are
not multiples of 4 (the staged 16-byte-request kernels / the dword kernels), even and odd heights, frames smaller than a
tile, single frames and small batches; every level with its mirrored apron, every a-trous layer, and (single frames) the
list of extrema the scan finds in them.  Not part of the test suite.  Usage: soak_pyramid.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"),
                os.path.join(ROOT, "tests")]
import numpy as np, torch
import local_features_python as lfp
from oracle import MkdOracle
from gen_golden import smooth_image
from test_gpu_detector import blob_image, assert_same_extrema

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
orc = MkdOracle(lfp.model_path("liberty"))
t0 = time.time()
bad = n_ext = 0
for r in range(rounds):
    w, h = int(rng.integers(8, 900)), int(rng.integers(8, 600))
    if r % 2 == 0: w = max(8, w // 4 * 4)
    if r % 3 == 0: h = max(8, h // 2 * 2)
    frames = (1, 1, 3, 9)[r % 4]
    n_scales = int(rng.integers(3, 6))
    img = np.ascontiguousarray(blob_image(w, h, 500 + r, max(4, w * h // 400)) if r % 5 == 4 else
                               smooth_image(h, w, 500 + r) + 0.05 * rng.random((h, w)), np.float32)
    hnd = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=h, max_frames=frames, n_scales=n_scales)
    if frames == 1:
        hnd.set_image(img)
    else:
        d = torch.from_numpy(np.stack([img] + [img[::-1].copy()] * (frames - 1))).cuda()
        hnd.set_images_device(d.data_ptr(), frames, w, h)
        hnd.synchronize()
    pyr = orc.split_pyramid(orc.build_pyramid(img), w, h)
    ok = True
    for l, lv in enumerate(pyr):
        padded, a = hnd.pyramid_level_apron(l)
        ok &= np.array_equal(padded, np.pad(lv, a, mode="symmetric"))
    # the detector's path: a-trous stack (level 1 then comes with layer 1 or by the blit) and the pyramid again
    st = orc.build_coarse_stack(img, n_scales)
    if frames == 1:
        got, dropped = hnd.detect_extrema()
        want, total = orc.scan_extrema(orc.dog(st))
        try:
            assert dropped == 0 and total == len(want)
            assert_same_extrema(got, want, (w, h))
            n_ext += len(want)
        except AssertionError as e:
            ok = False
            print("   extrema differ:", str(e)[:200])
        for l in range(n_scales + 3):
            ok &= np.array_equal(hnd.coarse_layer(l, w, h), st[l])
        hnd.set_image(img)
        for l, lv in enumerate(pyr):
            padded, a = hnd.pyramid_level_apron(l)
            ok &= np.array_equal(padded, np.pad(lv, a, mode="symmetric"))
    bad += 0 if ok else 1
    print(f"round {r:3d}: {frames} frame(s) {w}x{h} (w%4={w % 4}, h%2={h % 2}) n_scales {n_scales}: {len(pyr)} levels, bit-exact {ok}", flush=True)
print(f"soak_pyramid: {rounds} shapes, {bad} with a difference; {n_ext} extrema compared with the oracle's; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
