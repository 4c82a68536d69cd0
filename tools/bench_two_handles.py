#!/usr/bin/env python3
"""BASELINE configs[2] (256 frames 640x480 per batch, detect top 1400 + orient + describe) as a STREAM of batches: one handle
after the other, and two handles on two streams driven by two host threads -- the describe launch (bound by the SIMDs) of one
batch beside the pyramid / a-trous / scan kernels (bound by HBM) of the next.  Development aid; bench.py is the contract."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import local_features_python as lfp
from bench_detect import frame

w, h, top_n, frames = 640, 480, 1400, 256
cap = 2 * top_n * frames
batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class Lane:
    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.h = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                               max_blobs=8000, max_frames=frames)
        with torch.cuda.stream(self.stream):
            self.imgs = torch.stack([frame(h, w, 100 + (f % 8), 1.8) for f in range(frames)]).contiguous()
            self.kps = torch.empty((cap, 5), device="cuda")
            self.fo = torch.empty((cap,), dtype=torch.int32, device="cuda")
            self.out = torch.empty((cap, 128), device="cuda")
        self.stream.synchronize()
        self.n = 0

    def one(self):
        m, _, _ = self.h.detect_frames_device(self.imgs.data_ptr(), frames, w, h, top_n, 0.0, self.kps.data_ptr(),
                                              self.fo.data_ptr(), self.out.data_ptr(), cap, self.stream.cuda_stream)
        self.n += m


def timed(lanes, per_lane):
    for l in lanes:
        l.one(); l.n = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda l=l: [l.one() for _ in range(per_lane)]) for l in lanes]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(l.n for l in lanes)
    return dt, n


a, b = Lane(), Lane()
dt1, n1 = timed([a], batches)
dt2, n2 = timed([a, b], batches // 2)
print(f"one handle : {batches} batches, {n1} keypoints in {dt1 * 1e3:.2f} ms = {dt1 / batches * 1e3:.3f} ms per batch, {n1 / dt1 / 1e6:.1f} M desc/s")
print(f"two handles: {batches} batches, {n2} keypoints in {dt2 * 1e3:.2f} ms = {dt2 / batches * 1e3:.3f} ms per batch, {n2 / dt2 / 1e6:.1f} M desc/s")
