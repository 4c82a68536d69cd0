#!/usr/bin/env python3
"""Profiling target: a few describe passes in one (angle, pool) configuration (for rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pool = int(sys.argv[2]) if len(sys.argv) > 2 else lfp.POOL_F16X3   # lf_mkd_pool_mode: 1 = f16x3 split, 2 = f32
angle = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
p = torch.rand((n, 32, 32), device="cuda")
out = torch.empty((n, 128), device="cuda")
h = lfp.MkdHandle(max_features=n, angle_mode=angle, pool_mode=pool)
side = torch.cuda.Stream()
torch.cuda.synchronize()
s = side.cuda_stream
for _ in range(iters):
    h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
torch.cuda.synchronize()
print("done", float(out[0, 0]))
