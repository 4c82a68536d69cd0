#!/usr/bin/env python3
"""Soak of the orientation stage against the oracle: random frame shapes, random extrema anywhere (windows leave the
image), every detector size; the same keypoint list (x, y, size, response bit-equal; the angle within 1e-3 degrees), in the
same order.  Not part of the test suite.  Usage: soak_orient.py [rounds] [extrema per round]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import local_features_python as lfp
from oracle import MkdOracle
from test_gpu_orientation import smooth_image, random_extrema, assert_same_keypoints

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(99)
orc = MkdOracle(lfp.model_path("liberty"))
t0 = time.time()
bad = total = 0
for r in range(rounds):
    w, h = int(rng.integers(40, 1000)), int(rng.integers(40, 700))
    if r % 2 == 0: w = w // 4 * 4
    n_scales = int(rng.integers(3, 6))
    img = smooth_image(w, h, 700 + r)
    if r % 3 == 1:   # sharper
        img = np.ascontiguousarray(0.5 * img + 0.5 * np.random.default_rng(r).random((h, w)), np.float32)
    ex = random_extrema(n, w, h, 900 + r, n_scales)
    hnd = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=h, n_scales=n_scales)
    hnd.set_image(img)
    want = orc.orient(orc.build_coarse_stack(img, n_scales), ex)
    got, dropped = hnd.orient_keypoints(ex)
    ok = dropped == 0
    try:
        assert_same_keypoints(got, want, (w, h))
    except AssertionError as e:
        ok = False
        print("   differ:", str(e)[:200])
    total += len(want)
    bad += 0 if ok else 1
    print(f"round {r:3d}: {w}x{h} n_scales {n_scales}: {len(ex)} extrema -> {len(want)} keypoints, same list {ok}", flush=True)
print(f"soak_orient: {rounds} rounds, {total} keypoints compared, {bad} rounds with a difference; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
