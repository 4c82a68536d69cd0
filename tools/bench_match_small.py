#!/usr/bin/env python3
"""Small matching problems (the reference's own use is 2000 x 2000, examples/match_images/src/main.rs:62-76): the
one-launch form (match_small) against the split / scan / merge chain, per call in a queue of calls (HIP events) and as the
latency of a lone call (host clock around a synchronised call), with the decisions of the two forms compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import torch
import local_features_python as lfp

torch.cuda.set_stream(torch.cuda.Stream())
h = lfp.MkdHandle(max_features=64)
s = torch.cuda.current_stream().cuda_stream
for na, nb in ((2000, 2000), (500, 500), (256, 256), (1000, 4000), (3000, 2500), (4000, 4000), (8192, 2048), (64, 8000), (10000, 10000)):
    g = torch.Generator(device="cuda").manual_seed(na + nb)
    b = torch.nn.functional.normalize(torch.randn((nb, 128), device="cuda", generator=g), dim=1)
    a = torch.nn.functional.normalize(b[torch.randint(0, nb, (na,), device="cuda", generator=g)]
                                      + 0.08 * torch.randn((na, 128), device="cuda", generator=g), dim=1)
    res = {}
    for form in ("small", "scan"):
        os.environ["LF_MKD_MATCH"] = form
        m = torch.empty(na, dtype=torch.int32, device="cuda")
        for _ in range(3):
            h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, stream=s)
        e1.record(); torch.cuda.synchronize()
        queued = e0.elapsed_time(e1) / 50 * 1e3
        lone = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.match_device(a.data_ptr(), na, b.data_ptr(), nb, m.data_ptr(), 0.8, stream=s)
            torch.cuda.synchronize(); lone.append((time.perf_counter() - t0) * 1e6)
        lone.sort()
        res[form] = (queued, lone[len(lone) // 2], m.clone())
    same = bool((res["small"][2] == res["scan"][2]).all())
    fits = na * nb <= 1 << 23 and nb <= 4096
    print(f"{na:6d} x {nb:6d}: one launch {res['small'][0]:7.1f} us queued / {res['small'][1]:7.1f} us lone"
          f"{'' if fits else ' (does not fit: scan)'};  split+scan+merge {res['scan'][0]:7.1f} / {res['scan'][1]:7.1f};  "
          f"accepted {float((res['scan'][2] >= 0).float().mean()):.3f}, same decisions: {same}", flush=True)

# both directions of the example (examples/match_images/src/main.rs:113-116): two calls against lf_mkd_match_both_device
os.environ.pop("LF_MKD_MATCH", None)
for na, nb in ((2000, 2000), (1000, 1000), (3000, 2500), (500, 4000)):
    g = torch.Generator(device="cuda").manual_seed(na + nb)
    b = torch.nn.functional.normalize(torch.randn((nb, 128), device="cuda", generator=g), dim=1)
    a = torch.nn.functional.normalize(b[torch.randint(0, nb, (na,), device="cuda", generator=g)]
                                      + 0.08 * torch.randn((na, 128), device="cuda", generator=g), dim=1)
    m_ab, m_ba = torch.empty(na, dtype=torch.int32, device="cuda"), torch.empty(nb, dtype=torch.int32, device="cuda")
    o_ab, o_ba = torch.empty_like(m_ab), torch.empty_like(m_ba)
    def two():
        h.match_device(a.data_ptr(), na, b.data_ptr(), nb, o_ab.data_ptr(), 0.8, stream=s)
        h.match_device(b.data_ptr(), nb, a.data_ptr(), na, o_ba.data_ptr(), 0.8, stream=s)
    def both():
        h.match_both_device(a.data_ptr(), na, b.data_ptr(), nb, m_ab.data_ptr(), m_ba.data_ptr(), 0.8, s)
    out = {}
    for tag, fn in (("two calls", two), ("one call", both)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        lone = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); lone.append((time.perf_counter() - t0) * 1e6)
        lone.sort()
        out[tag] = (e0.elapsed_time(e1) / 50 * 1e3, lone[len(lone) // 2])
    same = bool((m_ab == o_ab).all()) and bool((m_ba == o_ba).all())
    print(f"both directions {na:5d} x {nb:5d}: two calls {out['two calls'][0]:6.1f} us queued / {out['two calls'][1]:6.1f} us lone;  "
          f"lf_mkd_match_both_device {out['one call'][0]:6.1f} / {out['one call'][1]:6.1f};  same decisions: {same}", flush=True)
