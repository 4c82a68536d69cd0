#!/usr/bin/env python3
"""Times lf_mkd_match_device on unit-norm random descriptors (the bench's match stage), for A/B runs of the matcher
(LF_MKD_MATCH=scan selects the three-term scan alone).  Usage: ab_match.py [na nb] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp

shapes = [(65536, 65536), (1 << 20, 1 << 20)]
if len(sys.argv) > 2:
    v = [int(x) for x in sys.argv[1:]]
    shapes = list(zip(v[0::2], v[1::2]))
h = lfp.MkdHandle(max_features=64)
s = torch.cuda.current_stream().cuda_stream
for na, nb in shapes:
    g = torch.Generator(device="cuda").manual_seed(na + nb)
    b = torch.nn.functional.normalize(torch.randn((nb, 128), device="cuda", generator=g), dim=1)
    a = torch.nn.functional.normalize(torch.randn((na, 128), device="cuda", generator=g), dim=1)
    m = torch.empty(na, dtype=torch.int32, device="cuda")
    for it in range(3):
        torch.cuda.synchronize()                    # (a null stream handle means the library's own stream: wall clock)
        t0 = time.perf_counter()
        h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, None, None, None, None, s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    print(f"{os.environ.get('LF_MKD_MATCH', 'two-pass'):9s} {na} x {nb}: {ms:9.3f} ms  {na * nb / ms / 1e9:7.3f} T sims/s  "
          f"accepted {(m >= 0).float().mean().item():.4f}  checksum {int(m.to(torch.int64).clamp(min=0).sum().item())}", flush=True)
