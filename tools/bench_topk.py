#!/usr/bin/env python3
"""Top-n selection of one list of extrema (lf_mkd_filter_extrema_device): the one-workgroup form (topk_one, list in the
registers of 1024 threads) against the five-launch form spread over the chip (LF_MKD_TOPK=multi), per call in a queue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np
import torch
import local_features_python as lfp

torch.cuda.set_stream(torch.cuda.Stream())
s = torch.cuda.current_stream().cuda_stream
h = lfp.MkdHandle(max_features=64)
rng = np.random.default_rng(1)
for n in (9000, 12000, 16000, 24000, 32768):
    ex = np.stack([rng.uniform(5, 3800, n), rng.uniform(5, 2100, n), 0.82 * np.sqrt(2) * 2 ** rng.uniform(1, 4.4, n),
                   0.035 + rng.exponential(0.03, n)], axis=1).astype(np.float32)
    d_ex = torch.from_numpy(ex).cuda()
    top_n = 6000
    d_out = torch.zeros((top_n, 4), device="cuda")
    d_idx = torch.zeros((top_n,), dtype=torch.int32, device="cuda")
    res = {}
    for form in ("", "multi"):
        os.environ["LF_MKD_TOPK"] = form
        for _ in range(3):
            h.filter_extrema_device(d_ex.data_ptr(), n, top_n, 0.0, d_out.data_ptr(), d_idx.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            h.filter_extrema_device(d_ex.data_ptr(), n, top_n, 0.0, d_out.data_ptr(), d_idx.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        res[form] = (e0.elapsed_time(e1) / 30 * 1e3, d_idx.clone())
    print(f"n = {n:6d}: one workgroup {res[''][0]:6.1f} us, five launches {res['multi'][0]:6.1f} us per call "
          f"(incl. the count's read-back); same selection: {bool((res[''][1] == res['multi'][1]).all())}", flush=True)
