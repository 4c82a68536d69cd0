#!/usr/bin/env python3
"""Quick GPU timing of the describe path (development aid; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
p = torch.rand((n, 32, 32), device="cuda")
out = torch.empty((n, 128), device="cuda")
for angle in (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT):
    for pool in (lfp.POOL_F32, lfp.POOL_F16X3, lfp.POOL_F16_FP6):
        h = lfp.MkdHandle(max_features=n, angle_mode=angle, pool_mode=pool)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 5
        for _ in range(it):
            h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        print(f"angle={angle} pool={pool} n={n}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} M desc/s  "
              f"{n*4608/dt/1e9:.1f} GB/s algorithmic", flush=True)
