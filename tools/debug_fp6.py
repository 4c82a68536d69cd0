#!/usr/bin/env python3
"""Where the fp6 cross-term mode's error sits: raw 238-D descriptors (before whitening) of the two f16 pool modes against
the oracle, per block of the descriptor (polar / cartesian x in-dim)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import local_features_python as lfp
from oracle import ATAN_SHADER, BLUR_CONTRACT, MkdOracle
o = MkdOracle(lfp.model_path("liberty"))
rng = np.random.default_rng(5)
p = rng.random((512, 32, 32)).astype(np.float32)
ref, raw_ref = o.describe_patches(p, atan_mode=ATAN_SHADER | BLUR_CONTRACT, nthreads=8, want_raw=True)
dp = torch.from_numpy(p).cuda()
for name, pool in (("f16x3", lfp.POOL_F16X3), ("fp6", lfp.POOL_F16_FP6)):
    h = lfp.MkdHandle(max_features=512, pool_mode=pool)
    raw = torch.empty((512, 238), device="cuda")
    h.raw_descriptors_device(dp.data_ptr(), 512, raw.data_ptr()); h.synchronize()
    r = raw.cpu().numpy()
    d = h.describe_patches(p)
    e = np.linalg.norm(d - ref, axis=1) / np.linalg.norm(ref, axis=1)
    print(f"{name}: final worst {e.max():.2e} mean {e.mean():.2e}; raw worst {np.abs(r - raw_ref).max():.2e}")
    err = np.abs(r - raw_ref)
    for i in range(7):
        pol = err[:, i * 25:(i + 1) * 25]
        car = err[:, 175 + i * 9:175 + (i + 1) * 9]
        print(f"   in-dim {i}: polar[0:16] {pol[:, :16].max():.1e} polar[16:25] {pol[:, 16:].max():.1e} | cart[0:7] {car[:, :7].max():.1e} cart[7:9] {car[:, 7:].max():.1e}")
