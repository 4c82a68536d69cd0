#!/usr/bin/env python3
"""lf_mkd_detect / lf_mkd_detect_u8 on the reference benchmark's frame (houses.jpg, 4096 x 3072, n_scales 3, top 2000,
max_blobs = 5 x: benches/bench.rs:41-112), a few calls one after the other: run under `rocprofv3 --kernel-trace` to see a
call's kernel timeline (tools/detect_timeline.sh).  Usage: prof_detect_host.py [u8|f32] [scale] [n_scales] [top_n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np
import local_features_python as lfp
from bench_reference_sweep import open_image

form = sys.argv[1] if len(sys.argv) > 1 else "u8"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nf = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
u8, f32 = open_image(scale)
h, w = f32.shape
lf = lfp.MkdHandle(max_features=nf, max_image_width=w, max_image_height=h, n_scales=ns, max_blobs=5 * nf)
kps, desc = np.empty((nf, 5), np.float32), np.empty((nf, 128), np.float32)
for _ in range(6):
    m, db, df = lf.detect_into(u8 if form == "u8" else f32, nf, 0.0, kps, desc)
print("done", form, w, h, m, db, df)
