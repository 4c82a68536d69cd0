#!/usr/bin/env python3
"""Times the f16 describe kernel (shader and exact angle) of several library builds in ONE process-per-library sweep:
tools/ab_quick.py libA.so libB.so ...   (timing-only ablation builds produce wrong descriptors by design)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, os
sys.path.insert(0, os.path.join(%r, "local-features_amd"))
import torch
import local_features_python as lfp
n = 1 << 20
p = torch.rand((n, 32, 32), device="cuda"); out = torch.empty((n, 128), device="cuda")
res = []
for angle in (0, 1):
    h = lfp.MkdHandle(max_features=n, angle_mode=angle)
    side = torch.cuda.Stream(); torch.cuda.synchronize(); s = side.cuda_stream
    for _ in range(2): h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 5 * 1e3)
print("%%-50s shader %%.3f ms   exact %%.3f ms" %% (os.path.basename(os.environ["LF_MKD_LIB"]), res[0], res[1]), flush=True)
''' % ROOT
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LF_MKD_LIB=os.path.abspath(lib)), timeout=300)
