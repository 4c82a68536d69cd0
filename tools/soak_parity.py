#!/usr/bin/env python3
"""One-off soak: 2^18 patches per configuration (3 PCA models x 2 pooling modes x 3 angle modes, mixed patch
statistics) against the oracle.  Prints the worst relative L2 per configuration.  Not part of the test suite."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import local_features_python as lfp
from oracle import ATAN_LIBM, ATAN_SHADER, BLUR_CONTRACT, MkdOracle

n = 1 << 18
rng = np.random.default_rng(2026)
parts = [rng.random((n // 4, 32, 32)),                                             # white noise
         np.clip(rng.normal(0.5, 0.05, (n // 4, 32, 32)), 0, 1),                   # low contrast
         rng.random((n // 4, 1, 1)) + 0.3 * rng.random((n // 4, 32, 32)).cumsum(axis=2) / 32,   # ramps + noise
         (rng.random((n // 4, 32, 32)) > 0.5) * rng.random((n // 4, 1, 1))]        # binary texture
p = np.ascontiguousarray(np.concatenate(parts), np.float32)
for model in lfp.PCA_NAMES:
    orc = MkdOracle(lfp.model_path(model))
    refs = {}
    for name, mode in (("shader", ATAN_SHADER | BLUR_CONTRACT), ("libm", ATAN_LIBM | BLUR_CONTRACT)):
        t = time.time(); refs[name] = orc.describe_patches(p, atan_mode=mode, nthreads=16)
        print(f"  oracle {model} {name}: {time.time()-t:.1f} s", flush=True)
    for pool in (lfp.POOL_F16X3, lfp.POOL_F32, lfp.POOL_F16_FP6):      # (3 = the fp6 cross-term experiment: inside the gate, not the default)
        for amode, ref in ((lfp.ANGLE_SHADER, "shader"), (lfp.ANGLE_EXACT, "libm"), (lfp.ANGLE_EXACT_ZERO, "shader")):
            d = lfp.MkdHandle(pca=model, max_features=1 << 16, angle_mode=amode, pool_mode=pool).describe_patches(p)
            e = np.linalg.norm(d.astype(np.float64) - refs[ref], axis=1) / np.linalg.norm(refs[ref], axis=1)
            bad = int((e >= 1e-4).sum())
            print(f"{model:10s} pool={pool} angle={amode} vs {ref:6s}: max {e.max():.2e}  p99.9 {np.quantile(e, 0.999):.2e}  "
                  f">=1e-4: {bad}  finite {bool(np.isfinite(d).all())}", flush=True)
