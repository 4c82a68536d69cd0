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
    side = torch.cuda.Stream(); torch.cuda.synchronize(); s = side.cuda_stream
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
    # The whitening projection on its own (north_star: "MFMA utilisation on the projection"): the epilogue of a batch is
    # finish_descriptors -- norms, the 238 -> 128 projection as 11 steps x 8 row tiles x 3 split terms = 264
    # v_mfma_f32_16x16x32_f16 per wave (16 descriptors), final L2, stores.  Two waves share a SIMD's matrix pipe, one MFMA
    # occupies it for 16 cycles.
    batches = (n // 128 + 255) // 256                      # batches of workgroup 0
    epi = t[:, 6].mean() / batches                          # shader cycles of one epilogue, per wave
    util = 2 * 264 * 16 / epi
    clock = tot.mean() / dt / 1e6
    issued = 264 * 16 * 16 * 32 * 2 / 16                    # f16 MFMA flop per descriptor in the projection (three-term split)
    print(f"  projection stage: {epi:.0f} cycles per wave and batch ({epi / clock:.1f} us at {clock:.0f} MHz); its 264 MFMAs keep "
          f"the SIMD's matrix pipe busy {util:.1%} of that (two waves per SIMD); {issued:.0f} f16 flop per descriptor = "
          f"{issued * 16 * 2 * 4 * 256 / (epi / (clock * 1e6)) / 1e15:.3f} PFLOP/s chip-wide while in the stage "
          f"({issued * 16 * 2 * 4 * 256 / (epi / (clock * 1e6)) / 2.5e15:.1%} of the 2.5 PFLOP/s f16 peak); useful f32 work: 2 x 238 x 128 = 60928 flop")
    if angle == lfp.ANGLE_SHADER and len(sys.argv) > 2:
        import json
        json.dump({"epilogue_cycles_per_wave_batch": epi, "shader_clock_mhz": clock, "mfma_per_wave_batch": 264,
                   "matrix_pipe_busy_frac_in_stage": util, "stage_share_of_kernel": float(t[:, 6].mean() / tot.mean()),
                   "f16_pflops_in_stage": issued * 16 * 2 * 4 * 256 / (epi / (clock * 1e6)) / 1e15,
                   "what": "whitening projection = finish_descriptors (norms + 264 MFMAs per 16 descriptors + L2 + stores), "
                           "phase clocks of a -DLF_PHASE_TIMING build (tools/phase_timing.py)"}, open(sys.argv[2], "w"), indent=1)
