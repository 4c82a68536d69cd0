#!/usr/bin/env python3
"""The row-split form of the keypoint kernel (requests of at most 4096 keypoints) against the whole-patch form on the same
keypoints: agreement (<= 2e-6 relative L2: R partial chains instead of one), determinism within a form, and the launch's
duration in a queue (HIP events).  LF_MKD_KP_SPLIT selects the form per launch.  Usage: check_split.py [out.txt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints

side = torch.cuda.Stream(); torch.cuda.set_stream(side); s = side.cuda_stream
w, h = 1920, 1080
g = torch.Generator(device="cuda").manual_seed(3)
img = torch.rand((h, w), device="cuda", generator=g)
lines = []
def say(x):
    print(x, flush=True); lines.append(x)
for n in (1, 31, 32, 33, 255, 500, 2000, 2048, 2049, 3000, 4096, 4097, 8192):
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h)
    kps = torch.from_numpy(np.concatenate([random_keypoints(n, w, h, 5, margin=0.0), np.zeros((n, 1), np.float32)], axis=1)).cuda()
    hnd.set_image_device(img.data_ptr(), w, h, s)
    res, us = {}, {}
    for form in ("1", "2", "4", "auto"):
        if form == "auto": os.environ.pop("LF_MKD_KP_SPLIT", None)
        else: os.environ["LF_MKD_KP_SPLIT"] = form
        o = torch.full((n, 128), float("nan"), device="cuda")
        for _ in range(3):
            hnd.describe_keypoints_device(kps.data_ptr(), n, o.data_ptr(), s)
        side.synchronize()
        first = o.cpu().numpy().copy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(40):
            hnd.describe_keypoints_device(kps.data_ptr(), n, o.data_ptr(), s)
        e1.record(side); side.synchronize()
        assert np.array_equal(first, o.cpu().numpy()), (n, form, "not deterministic")
        res[form], us[form] = first, e0.elapsed_time(e1) / 40 * 1e3
    ref = res["1"]
    assert np.isfinite(ref).all()
    def rel(a): return float((np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)).max())
    say(f"n={n:5d}: whole-patch {us['1']:6.1f} us | R=2 {us['2']:6.1f} us (rel {rel(res['2']):.1e}) | R=4 {us['4']:6.1f} us (rel {rel(res['4']):.1e}) | "
        f"default {us['auto']:6.1f} us (rel {rel(res['auto']):.1e})")
    assert max(rel(res["2"]), rel(res["4"]), rel(res["auto"])) < 8e-6
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
