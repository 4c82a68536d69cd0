#!/usr/bin/env python3
"""A/B of the banded upload of lf_mkd_detect / lf_mkd_detect_u8 on the reference benchmark's frame (houses.jpg 4096 x 3072,
top 2000, max_blobs 5x: benches/bench.rs:41-112): one piece (LF_MKD_DETECT_BANDS=0) against the default plan and against
forced plans (LF_MKD_BAND_PIECES=k / LF_MKD_BAND_SPLIT=f), n_scales 3 and 5, 8-bit and f32 frames.  Wall ms per call
(median of 30 after 4 warm-up calls) and the library's own split.  Usage: ab_bands.py [out.txt] [variants...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import local_features_python as lfp
from bench_reference_sweep import open_image

out = sys.argv[1] if len(sys.argv) > 1 else None
variants = sys.argv[2:] or ["BANDS=0", "default", "PIECES=2", "PIECES=3", "PIECES=4", "PIECES=6"]
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def run(ns, img, w, h, env):
    for k in ("LF_MKD_DETECT_BANDS", "LF_MKD_BAND_SPLIT", "LF_MKD_BAND_PIECES"):
        os.environ.pop(k, None)
    if env != "default":
        k, v = env.split("=")                      # BANDS=0 | PIECES=k | SPLIT=f1,f2,..
        os.environ["LF_MKD_DETECT_" + k if k == "BANDS" else "LF_MKD_BAND_" + k] = v
    lf = lfp.MkdHandle(max_features=2000, max_image_width=w, max_image_height=h, n_scales=ns, max_blobs=10000,
                       flags=lfp.FLAG_KERNEL_TIMING)
    kps, desc = np.empty((2000, 5), np.float32), np.empty((2000, 128), np.float32)
    for _ in range(4):
        m = lf.detect_into(img, 2000, 0.0, kps, desc)
    ts, sp = [], []
    for _ in range(30):
        t0 = time.perf_counter(); lf.detect_into(img, 2000, 0.0, kps, desc); ts.append(time.perf_counter() - t0)
        sp.append(lf.detect_times())
    lf.close()
    return np.median(ts) * 1e3, np.median(np.array(sp), axis=0), m, desc[:m[0]].copy()


u8, f32 = open_image(1.0)
h, w = f32.shape
say(f"lf_mkd_detect[_u8], houses.jpg {w}x{h}, top 2000: wall ms per call (upload until the whole frame is there + pipeline after it + results)")
for ns in (3, 5):
    for name, img in (("u8", u8), ("f32", f32)):
        ref = None
        for env in variants:
            ms, sp, m, d = run(ns, img, w, h, env)
            if ref is None:
                ref = d
            assert np.array_equal(ref, d), (ns, name, env)            # every plan returns the same bits
            say(f"n_scales {ns} {name:>3} {env:>24}: {ms:7.3f} ms  ({sp[0]:.3f} + {sp[1]:.3f} + {sp[2]:.3f})")
if out:
    open(out, "w").write("\n".join(lines) + "\n")
