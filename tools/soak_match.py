#!/usr/bin/env python3
"""Randomised soak of the matcher against the CPU oracle: random shapes, ratios, exclusion ranges, planted duplicates and
clusters, every form (LF_MKD_MATCH=small / scan / screen; small: where the problem fits one launch, the scan otherwise).  A differing decision is tolerated only at a near-tie (2e-6), as in
tests/test_gpu_match.py.  Usage: soak_match.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import local_features_python as lfp
from oracle import MkdOracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = MkdOracle(os.path.join(ROOT, "local-features_amd", "models", "mkd", "concat-pca-liberty.safetensors"))
h = lfp.MkdHandle(max_features=64)


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


t_end, cases, worst, redone, near_ties = time.time() + budget, 0, 0.0, 0, 0
while time.time() < t_end:
    na = int(rng.choice([1, 7, 63, 500, 2000, 5000, 20000, 40000]))
    nb = int(rng.choice([2, 33, 700, 4000, 30000, 120000]))
    if na * nb > 1.5e9:
        continue
    base = unit(rng.normal(size=(max(8, nb // int(rng.choice([1, 1, 4, 50]))), 128)))
    b = unit(base[rng.integers(0, len(base), nb)] + rng.choice([0.0, 1e-4, 0.05, 0.3]) * rng.normal(size=(nb, 128)))
    a = unit(b[rng.integers(0, nb, na)] + rng.choice([0.0, 0.02, 0.25, 1.0]) * rng.normal(size=(na, 128)) / np.sqrt(128) * 4)
    if rng.random() < 0.3:                                   # planted exact duplicates
        k = min(nb // 2, 50)
        b[rng.integers(0, nb, k)] = b[rng.integers(0, nb, k)]
    ratio = float(rng.choice([0.0, 0.8, 0.8, 0.95, 1.0]))
    excl = None
    if rng.random() < 0.5 and nb > 8:
        lo = rng.integers(0, nb - 4, na).astype(np.uint32)
        hi = np.minimum(nb, lo + rng.integers(0, max(2, nb // 3), na)).astype(np.uint32)
        full = (lo == 0) & (hi >= nb - 1)                    # keep at least two candidates
        hi[full] = nb - 2
        excl = (lo, hi)
    want, s1, s2 = oracle.match(a, b, ratio=ratio, exclude=excl)
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d_lo = torch.from_numpy(excl[0].view(np.int32)).cuda() if excl else None
    d_hi = torch.from_numpy(excl[1].view(np.int32)).cuda() if excl else None
    for form in ("small", "scan", "screen"):
        os.environ["LF_MKD_MATCH"] = form
        d_m = torch.empty(na, dtype=torch.int32, device="cuda")
        d_1, d_2 = torch.empty(na, device="cuda"), torch.empty(na, device="cuda")
        h.match_device(d_a.data_ptr(), na, d_b.data_ptr(), nb, d_m.data_ptr(), ratio, d_lo.data_ptr() if excl else None,
                       d_hi.data_ptr() if excl else None, d_1.data_ptr(), d_2.data_ptr(), None)
        h.synchronize()
        if form == "screen":
            redone += h.match_overflowed()
        got, g1, g2 = d_m.cpu().numpy(), d_1.cpu().numpy(), d_2.cpu().numpy()
        fin = np.isfinite(s2)
        e = max(np.abs(g1 - s1).max(), np.abs(g2[fin] - s2[fin]).max(initial=0.0))
        assert e < 2e-6 and np.array_equal(np.isfinite(g2), fin), (form, na, nb, ratio, e)
        worst = max(worst, float(e))
        for i in np.flatnonzero(got != want):
            tie = abs(s1[i] - s2[i]) < 2e-6 or abs(s1[i] * ratio - s2[i]) < 2e-6
            assert tie, (form, na, nb, ratio, int(i), int(got[i]), int(want[i]), float(s1[i]), float(s2[i]))
            near_ties += 1
    cases += 1
print(f"soak_match: {cases} cases x 3 forms, worst similarity error {worst:.2e}, decisions differing at near-ties "
      f"{near_ties}, rows redone by the full scan {redone}: all decisions equal the oracle's")
