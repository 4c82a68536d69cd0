#!/usr/bin/env python3
"""One-round keypoint-mode launches: the describe call alone on one 1920x1080 frame for n given keypoints (the reference's own
settings are top_n 2000 / max_features 3000), per call in a queue (HIP events).  LF_MKD_LIB selects the build (A/B runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import hashlib
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints

side = torch.cuda.Stream(); torch.cuda.set_stream(side); s = side.cuda_stream
w, h = 1920, 1080
g = torch.Generator(device="cuda").manual_seed(3)
img = torch.rand((h, w), device="cuda", generator=g)
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(lfp.LIB_PATH)
out_line = []
for n in (500, 2000, 3000, 6000, 8192, 10000):
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h)
    kps = torch.from_numpy(np.concatenate([random_keypoints(n, w, h, 5, margin=64.0), np.zeros((n, 1), np.float32)], axis=1)).cuda()
    o = torch.empty((n, 128), device="cuda")
    hnd.set_image_device(img.data_ptr(), w, h, s)
    for _ in range(3):
        hnd.describe_keypoints_device(kps.data_ptr(), n, o.data_ptr(), s)
    side.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    for _ in range(50):
        hnd.describe_keypoints_device(kps.data_ptr(), n, o.data_ptr(), s)
    e1.record(side); side.synchronize()
    out_line.append(f"n={n}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (md5 {hashlib.md5(o.cpu().numpy().tobytes()).hexdigest()[:8]})")
print(f"{label:8s} " + "; ".join(out_line), flush=True)
