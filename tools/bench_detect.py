#!/usr/bin/env python3
"""Whole-pipeline timings (detect + orient + describe) for BASELINE.json configs[4] (4K stream, ~8k keypoints per
frame) and configs[1]-sized frames.  Development aid; bench.py is the contract."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np, torch
import local_features_python as lfp

_side = torch.cuda.Stream()        # NULL stream = "the handle's own stream" for the library: use a side stream
torch.cuda.set_stream(_side)


def frame(h, w, seed, sigma=2.5):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((1, 1, h, w), device="cuda", generator=g)
    r = int(3 * sigma)
    k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device="cuda") / sigma) ** 2); k /= k.sum()
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1), padding=(0, r))
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1), padding=(r, 0))
    x = (x - x.min()) / (x.max() - x.min())
    return x[0, 0].contiguous()


def run(w, h, top_n, frames, tag, sigma=2.5):
    cap = 2 * top_n
    hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                        max_blobs=1 << 16)
    s = torch.cuda.current_stream().cuda_stream
    imgs = [frame(h, w, 100 + f, sigma) for f in range(4)]
    ex = torch.empty((1 << 17, 4), device="cuda")
    sel = torch.empty((top_n, 4), device="cuda")
    kps = torch.empty((cap, 5), device="cuda")
    out = torch.empty((cap, 128), device="cuda")
    e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    acc = np.zeros(5); cnt = np.zeros(3)
    t0 = None
    for f in range(frames + 3):
        if f == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        e[0].record(); hnd.set_image_device(imgs[f % 4].data_ptr(), w, h, s)
        e[1].record(); n_ex, _ = hnd.detect_extrema_device(ex.data_ptr(), None, 1 << 17, s)
        e[2].record(); n_sel = hnd.filter_extrema_device(ex.data_ptr(), n_ex, top_n, 0.0, sel.data_ptr(), None, s)
        e[3].record(); m, _ = hnd.orient_keypoints_device(sel.data_ptr(), None, n_sel, kps.data_ptr(), None, cap, s)
        e[4].record(); hnd.describe_keypoints_device(kps.data_ptr(), m, out.data_ptr(), s)
        e[5].record(); torch.cuda.synchronize()
        if f >= 3:
            acc += [e[i].elapsed_time(e[i + 1]) for i in range(5)]; cnt += [n_ex, n_sel, m]
    dt = (time.perf_counter() - t0) / frames
    acc /= frames; cnt /= frames
    print(f"{tag}: {w}x{h}: {cnt[0]:.0f} extrema -> top {cnt[1]:.0f} -> {cnt[2]:.0f} keypoints; {dt*1e3:.3f} ms/frame wall "
          f"({cnt[2]/dt/1e6:.1f} M desc/s): pyramid {acc[0]:.3f}, a-trous + scan {acc[1]:.3f}, top-K {acc[2]:.3f}, "
          f"orientation {acc[3]:.3f}, sample+describe {acc[4]:.3f} ms", flush=True)


def run_graph(w, h, top_n, frames, tag, sigma=2.5):
    """the same frames through lf_mkd_stream_*: one hipGraph launch per frame, no host round trip"""
    cap = 2 * top_n
    hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                        max_blobs=1 << 16)
    s = torch.cuda.current_stream().cuda_stream
    imgs = [frame(h, w, 100 + f, sigma) for f in range(4)]
    d_img = torch.empty((h, w), device="cuda")
    kps = torch.empty((cap, 5), device="cuda")
    out = torch.empty((cap, 128), device="cuda")
    cnt = torch.zeros((8,), dtype=torch.int64, device="cuda")
    hnd.stream_create(w, h, top_n, 0.0, cap, d_img.data_ptr(), kps.data_ptr(), out.data_ptr(), cnt.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in range(3):
        d_img.copy_(imgs[f % 4]); hnd.stream_frame(s)
    torch.cuda.synchronize()
    # latency: one frame at a time, waited for
    lat = []
    for f in range(frames):
        d_img.copy_(imgs[f % 4])
        e0.record(); hnd.stream_frame(s); e1.record(); torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1))
    n_kp = int(cnt[3].item())
    # throughput: frames queued back to back (the copy into d_image is part of the stream)
    t0 = time.perf_counter()
    for f in range(frames):
        d_img.copy_(imgs[f % 4]); hnd.stream_frame(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    print(f"{tag} hipGraph: {w}x{h}: {int(cnt[0].item())} extrema -> top {int(cnt[2].item())} -> {n_kp} keypoints; "
          f"latency {np.median(lat):.3f} ms/frame (graph launch to done), queued {dt*1e3:.3f} ms/frame "
          f"({n_kp/dt/1e6:.1f} M desc/s)", flush=True)


def run_batch(w, h, top_n, frames, tag, sigma=1.8):
    """configs[2]: all frames in one lf_mkd_detect_frames_device call"""
    cap = 2 * top_n * frames
    hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3,
                        max_blobs=8000, max_frames=frames)
    s = torch.cuda.current_stream().cuda_stream
    imgs = torch.stack([frame(h, w, 100 + (f % 8), sigma) for f in range(frames)]).contiguous()
    kps = torch.empty((cap, 5), device="cuda")
    fo = torch.empty((cap,), dtype=torch.int32, device="cuda")
    out = torch.empty((cap, 128), device="cuda")
    def one():
        return hnd.detect_frames_device(imgs.data_ptr(), frames, w, h, top_n, 0.0, kps.data_ptr(), fo.data_ptr(),
                                        out.data_ptr(), cap, s)
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 5
    for _ in range(it):
        m, db, df = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"{tag}: {frames} frames {w}x{h}, top {top_n} per frame, ONE batch: {m} keypoints in {dt*1e3:.3f} ms = "
          f"{dt/frames*1e3:.4f} ms/frame, {m/dt/1e6:.1f} M desc/s (dropped blobs {db}, features {df})", flush=True)


if __name__ == "__main__":
    run_batch(640, 480, 1400, 256, "configs[2] batched")
    run(3840, 2160, 6000, 20, "configs[4]")
    run_graph(3840, 2160, 6000, 20, "configs[4]")
    run_graph(1920, 1080, 7000, 20, "configs[1]-sized")
    run_graph(640, 480, 1400, 20, "configs[2]-sized frame", sigma=1.8)
    run(1920, 1080, 7000, 20, "configs[1]-sized")
    run(640, 480, 1400, 20, "configs[2]-sized frame", sigma=1.8)
