#!/usr/bin/env python3
"""Profiling target for keypoint mode: BASELINE configs[2] as one batch (256 frames 640x480, 2000 keypoints each; default)
or configs[1] (one 1920x1080 frame, 10 000 keypoints; argument "configs1") or configs[3] in its own form, the share of one GPU
(128 frames 1920x1080, 8192 keypoints each; argument "configs3") or the reference's own settings on a 1080p frame (3000 keypoints:
examples/match_images/src/main.rs:62-76; argument "refdefaults"), three calls of set_images + describe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints
which = sys.argv[1] if len(sys.argv) > 1 else "configs2"
w, h, nk, frames, margin = {"configs1": (1920, 1080, 10000, 1, 64.0), "configs3": (1920, 1080, 8192, 128, 64.0),
                            "refdefaults": (1920, 1080, 3000, 1, 64.0)}.get(
    which, (640, 480, 2000, 256, 8.0))
n = nk * frames
hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, max_frames=frames)
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
s = side.cuda_stream
imgs = torch.rand((frames, h, w), device="cuda")
base = [np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=margin), np.zeros((nk, 1), np.float32)], axis=1)
        for f in range(min(frames, 8))]
kps = torch.from_numpy(np.concatenate([base[f % len(base)] for f in range(frames)]).astype(np.float32)).cuda()
fid = torch.arange(frames, device="cuda", dtype=torch.int32).repeat_interleave(nk).contiguous()
out = torch.empty((n, 128), device="cuda")
for _ in range(3):
    hnd.set_images_device(imgs.data_ptr(), frames, w, h, s)
    hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print("done", which)
