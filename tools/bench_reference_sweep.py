#!/usr/bin/env python3
"""The reference's own criterion benchmark (local_features/benches/bench.rs) on this path, on its own input:
`detect_top_n` on sample_data/houses.jpg (here tests/golden/houses.jpg, 4096 x 3072) -- 8-bit luma, Lanczos-resized to
0.25 .. 1.0 of full size, f32 / 255 (bench.rs:9-19,47-49; Pillow's "L" conversion and LANCZOS filter stand in for the
`image` crate's) -- groups scale_scales={3,5} (3000 features at the four scales) and feats_scales={3,5} (100 .. 2000
features at full size), max_blobs = 5 x max_features (bench.rs:57-64).  Host image in, host results out
(lf_mkd_detect), as `lf.detect_top_n(&image.view(), n, 0.)` does; median of 20 calls after 3 warm-up calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np
from PIL import Image
import local_features_python as lfp

PHOTO = Image.open(os.path.join(ROOT, "tests", "golden", "houses.jpg")).convert("L")


def open_image(scale):
    w, h = round(PHOTO.width * scale), round(PHOTO.height * scale)
    im = PHOTO if scale == 1.0 else PHOTO.resize((w, h), Image.LANCZOS)
    return np.ascontiguousarray(np.asarray(im, np.float32) / 255.0)


def bench(n_scales, scale, max_features, group):
    img = open_image(scale)
    h, w = img.shape
    lf = lfp.MkdHandle(max_features=max_features, max_image_width=w, max_image_height=h, n_scales=n_scales,
                       max_blobs=5 * max_features)                                  # bench.rs:57-64
    for _ in range(3):
        k, _, db, df = lf.detect(img, max_features, 0.0)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); lf.detect(img, max_features, 0.0); ts.append(time.perf_counter() - t0)
    print(f"{group}/{scale}x-{max_features}feats: {w}x{h}: {np.median(ts)*1e3:8.3f} ms  ({len(k)} keypoints, dropped blobs {db}, "
          f"features {df})", flush=True)


for ns in (3, 5):                                  # do_benches_nfeats
    for nf in (100, 500, 1000, 2000):
        bench(ns, 1.0, nf, f"feats_scales={ns}")
for ns in (3, 5):                                  # do_benches_scale
    for s in (0.25, 0.5, 0.75, 1.0):
        bench(ns, s, 3000, f"scale_scales={ns}")
