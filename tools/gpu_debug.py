#!/usr/bin/env python3
"""Development aid: per-block error of the raw 238-D descriptor for every (angle, pool) variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import local_features_python as lfp
from oracle import MkdOracle, ATAN_LIBM, ATAN_SHADER
o = MkdOracle(lfp.model_path("liberty"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
p = np.random.default_rng(1).random((n, 32, 32)).astype(np.float32)
dp = torch.from_numpy(p).cuda()
for a in (0, 1):
    ref, raw_ref = o.describe_patches(p, atan_mode=ATAN_SHADER if a == 0 else ATAN_LIBM, nthreads=8, want_raw=True)
    for pm in (lfp.POOL_F32, lfp.POOL_F16X3):
        h = lfp.MkdHandle(max_features=1024, angle_mode=a, pool_mode=pm)
        raw = torch.empty((n, 238), device="cuda"); out = torch.empty((n, 128), device="cuda")
        for rep in range(3):
            h.raw_descriptors_device(dp.data_ptr(), n, raw.data_ptr()); h.describe_patches_device(dp.data_ptr(), n, out.data_ptr()); h.synchronize()
            r = raw.cpu().numpy(); d = out.cpu().numpy()
            e = np.linalg.norm(d - ref, axis=1) / np.linalg.norm(ref, axis=1)
            blocks = [np.abs(r[:, i*25:(i+1)*25] - raw_ref[:, i*25:(i+1)*25]).max() for i in range(7)] + \
                     [np.abs(r[:, 175+i*9:175+(i+1)*9] - raw_ref[:, 175+i*9:175+(i+1)*9]).max() for i in range(7)]
            bad = np.where(e > 1e-4)[0]
            print(f"angle={a} pool={pm} rep={rep}: desc relL2 max {e.max():.2e}  bad patches {len(bad)} {bad[:12]}  "
                  f"raw block max-abs polar{np.round(blocks[:7], 6)} cart{np.round(blocks[7:], 6)}", flush=True)
