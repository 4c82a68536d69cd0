#!/usr/bin/env python3
"""Per-phase wall clock of the describe kernel's patch-row loop (development aid).
Needs a library built with -DLF_PHASE_TIMING (LF_MKD_LIB=...): workgroup 0 leaves, per wave, the s_memtime cycles it
spent in each phase in the first descriptors of the output."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp

NAMES = ["sync wait", "blur", "gradient+direction", "m stream", "harmonics (cos / sin streams)", "(unused stamp)", "epilogue", "loop"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
p = torch.rand((n, 32, 32), device="cuda")
out = torch.empty((n, 128), device="cuda")
for angle in (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT_ZERO):
    h = lfp.MkdHandle(max_features=n, angle_mode=angle, pool_mode=lfp.POOL_F16X3)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = out[:8, :8].double().cpu().numpy()
    tot = t.sum(axis=1)
    print(f"angle={angle}: {dt*1e3:.3f} ms; cycles per wave {tot.mean():.3e} (= {tot.mean()/dt/1e6:.1f} MHz counter)")
    for i, name in enumerate(NAMES):
        print(f"  {name:30s} {t[:, i].mean() / tot.mean() * 100:5.1f} %   per wave: " + " ".join(f"{v/1e3:8.0f}k" for v in t[:, i]))
