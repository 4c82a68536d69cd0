#!/usr/bin/env python3
"""Matcher throughput: pairs per second and the f16 MFMA rate of the screening pass (one 256-flop term per pair; with
LF_MKD_MATCH=scan the three-term scan, 768 flop per pair).  Development aid; bench.py is the contract for the headline metric."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp

_side = torch.cuda.Stream()
torch.cuda.set_stream(_side)
h = lfp.MkdHandle(max_features=64)
s = torch.cuda.current_stream().cuda_stream
for na, nb in ((2000, 2000), (10000, 10000), (65536, 65536), (65536, 1 << 20), (1 << 18, 1 << 18), (1 << 20, 1 << 20)):
    g = torch.Generator(device="cuda").manual_seed(na + nb)
    a = torch.nn.functional.normalize(torch.randn((na, 128), device="cuda", generator=g), dim=1)
    b = torch.nn.functional.normalize(torch.randn((nb, 128), device="cuda", generator=g), dim=1)
    m = torch.empty(na, dtype=torch.int32, device="cuda")
    h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, stream=s)
    torch.cuda.synchronize()
    it = 3 if na * nb > 1 << 36 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, stream=s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    sims = na * nb / (ms * 1e-3)
    terms = 3 if os.environ.get("LF_MKD_MATCH", "")[:1] == "s" else 1
    print(f"{na} x {nb}: {ms:.3f} ms  {sims/1e12:.2f} T pairs/s  = {sims*128*2*terms/1e15:.2f} PFLOP/s of f16 MFMA "
          f"({sims*128*2*terms/2.5e15*100:.0f} % of 2.5 PF dense), rows redone by the full scan: {h.match_overflowed(s)}", flush=True)
