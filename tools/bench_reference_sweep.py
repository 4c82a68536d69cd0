#!/usr/bin/env python3
"""The reference's own criterion benchmark (local_features/benches/bench.rs) on this path: detect_top_n on one image at
scales 0.25 .. 1.0 of 4096 x 3072 (its houses.jpg; a synthetic frame of the same size stands in -- the photograph's
licence does not allow redistribution), n_scales 3 and 5 with 3000 features, and 100 .. 2000 features at full size.
Host image in, host results out (lf_mkd_detect), as `lf.detect_top_n(&image.view(), n, 0.)` does; median of 20 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np, torch
import local_features_python as lfp


def frame(h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((1, 1, h, w), device="cuda", generator=g)
    k = torch.exp(-0.5 * (torch.arange(-6, 7, device="cuda") / 2.0) ** 2); k /= k.sum()
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1), padding=(0, 6))
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1), padding=(6, 0))
    x = (x - x.min()) / (x.max() - x.min())
    return np.ascontiguousarray(x[0, 0].cpu().numpy())


def bench(n_scales, scale, max_features, group):
    w, h = round(4096 * scale), round(3072 * scale)
    img = frame(h, w, 5)
    lf = lfp.MkdHandle(max_features=max_features, max_image_width=w, max_image_height=h, n_scales=n_scales,
                       max_blobs=5 * max_features, pool_mode=lfp.POOL_F16X3)       # bench.rs:57-64
    for _ in range(3):
        k, _, db, df = lf.detect(img, max_features, 0.0)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); lf.detect(img, max_features, 0.0); ts.append(time.perf_counter() - t0)
    print(f"{group}/{scale}x-{max_features}feats: {w}x{h}: {np.median(ts)*1e3:8.3f} ms  ({len(k)} keypoints, dropped blobs {db}, "
          f"features {df})", flush=True)


for ns in (3, 5):                                  # do_benches_scale
    for s in (0.25, 0.5, 0.75, 1.0):
        bench(ns, s, 3000, f"scale_scales={ns}")
for nf in (100, 500, 1000, 2000):                  # the feature-count sweep at full size
    bench(4, 1.0, nf, "n_features")
