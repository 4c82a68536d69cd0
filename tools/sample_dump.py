#!/usr/bin/env python3
"""Samples the patches of a fixed set of keypoints (interior, border-straddling, tiny, huge and non-finite ones) on a
smooth frame and writes them to an .npy file.  tests/test_gpu_parity.py runs it twice -- LF_MKD_SAMPLER=lds and
LF_MKD_SAMPLER=gather -- and compares the two files.  Usage: sample_dump.py OUT.npy [W H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints, smooth_image


def keypoints(w, h):
    g = np.random.default_rng(5)
    k = [random_keypoints(3000, w, h, 1, margin=0.0)]                       # anywhere, border included
    edge = random_keypoints(600, w, h, 2, margin=0.0)
    edge[:, 0] = g.choice([0.0, 0.4, w - 1.0, w - 0.3, w / 2], 600)         # on and next to the vertical borders
    edge[:300, 1] = g.choice([0.0, h - 1.0, h - 0.2], 300)
    k.append(edge)
    big = random_keypoints(200, w, h, 3, margin=0.0)
    big[:, 2] = g.uniform(40.0, 400.0, 200)                                 # footprints larger than the frame's levels
    k.append(big)
    tiny = random_keypoints(200, w, h, 4, margin=0.0)
    tiny[:, 2] = g.uniform(0.01, 1.4, 200)                                  # below level 0
    k.append(tiny)
    odd = random_keypoints(8, w, h, 6, margin=0.0)
    odd[0, 2], odd[1, 2], odd[2, 0], odd[3, 3], odd[4, 2], odd[5, 1] = np.nan, np.inf, np.nan, np.inf, 0.0, -1e30
    k.append(odd)
    k = np.concatenate(k).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))


if __name__ == "__main__":
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (333, 257)
    img = np.ascontiguousarray(smooth_image(h, w, 9), np.float32)
    k5 = keypoints(w, h)
    hnd = lfp.MkdHandle(max_features=len(k5), max_image_width=w, max_image_height=h)
    hnd.set_image(img)
    d_k = torch.from_numpy(k5).cuda()
    d_p = torch.full((len(k5), 32, 32), -7.0, device="cuda")
    hnd.sample_patches_device(d_k.data_ptr(), len(k5), d_p.data_ptr())
    hnd.synchronize()
    np.save(sys.argv[1], d_p.cpu().numpy())
    print("sampled", len(k5), "keypoints with", os.environ.get("LF_MKD_SAMPLER", "lds"))
