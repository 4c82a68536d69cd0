#!/usr/bin/env python3
"""Keypoint-mode timings for BASELINE.json configs[1] (10k keypoints, one 1920x1080 frame) and configs[2]
(256 frames of 640x480, 2k keypoints each).  Development aid; bench.py is the contract."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints

# The library treats a NULL stream argument as "the handle's own stream", and torch's default stream IS the NULL
# stream: run everything on a side stream so that torch's events and the library's launches share one queue.
_side = torch.cuda.Stream()
torch.cuda.set_stream(_side)


def frame(h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((1, 1, h, w), device="cuda", generator=g)
    k = torch.exp(-0.5 * (torch.arange(-6, 7, device="cuda") / 2.0) ** 2); k /= k.sum()
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1), padding=(0, 6))
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1), padding=(6, 0))
    x = (x - x.min()) / (x.max() - x.min())
    return x[0, 0].contiguous()


def run(w, h, nk, frames, tag):
    hnd = lfp.MkdHandle(max_features=max(nk, 64), max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3)
    s = torch.cuda.current_stream().cuda_stream
    imgs = [frame(h, w, 100 + f) for f in range(min(frames, 8))]
    kps = [torch.from_numpy(np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=64.0 if min(w, h) > 200 else 8.0),
                                            np.zeros((nk, 1), np.float32)], axis=1)).cuda() for f in range(min(frames, 8))]
    out = torch.empty((nk, 128), device="cuda")
    def one(f):
        hnd.set_image_device(imgs[f % len(imgs)].data_ptr(), w, h, s)
        hnd.describe_keypoints_device(kps[f % len(kps)].data_ptr(), nk, out.data_ptr(), s)
    for f in range(3): one(f)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    pyr_ms = desc_ms = 0.0
    for f in range(frames):
        e[0].record(); hnd.set_image_device(imgs[f % len(imgs)].data_ptr(), w, h, s)
        e[1].record(); hnd.describe_keypoints_device(kps[f % len(kps)].data_ptr(), nk, out.data_ptr(), s)
        e[2].record()
        if f < 16:
            torch.cuda.synchronize(); pyr_ms += e[0].elapsed_time(e[1]); desc_ms += e[1].elapsed_time(e[2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = min(frames, 16)
    print(f"{tag}: {frames} frames {w}x{h}, {nk} kpts/frame: {dt/frames*1e3:.3f} ms/frame  {frames*nk/dt/1e6:.2f} M desc/s  "
          f"(pyramid {pyr_ms/m:.3f} ms, sample+describe {desc_ms/m:.3f} ms per frame)", flush=True)


def run_batched(w, h, nk, frames, tag):
    """all frames in one set_images + one describe_keypoints_frames call"""
    n = nk * frames
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3, max_frames=frames,
                        flags=int(os.environ.get("LF_KP_FLAGS", "0")))
    s = torch.cuda.current_stream().cuda_stream
    imgs = torch.stack([frame(h, w, 100 + (f % 8)) for f in range(frames)]).contiguous()
    base = [np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=8.0), np.zeros((nk, 1), np.float32)], axis=1) for f in range(8)]
    kps = torch.from_numpy(np.concatenate([base[f % 8] for f in range(frames)]).astype(np.float32)).cuda()
    fid = torch.arange(frames, device="cuda", dtype=torch.int32).repeat_interleave(nk).contiguous()
    out = torch.empty((n, 128), device="cuda")
    def one():
        hnd.set_images_device(imgs.data_ptr(), frames, w, h, s)
        hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, out.data_ptr(), s)
    one(); torch.cuda.synchronize()
    t0 = time.perf_counter(); it = 5
    for _ in range(it): one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"{tag}: {frames} frames {w}x{h} x {nk} kpts in ONE batch: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} M desc/s", flush=True)


def run_orient(w, h, ne, frames, tag):
    """extract graph from the detector's extrema: pyramid + a-trous stack + orientation + sample + describe"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_orientation import random_extrema
    cap = 2 * ne
    hnd = lfp.MkdHandle(max_features=cap, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3)
    s = torch.cuda.current_stream().cuda_stream
    imgs = [frame(h, w, 100 + f) for f in range(4)]
    exs = [torch.from_numpy(random_extrema(ne, w, h, 300 + f, border=8.0)).cuda() for f in range(4)]
    kps = torch.empty((cap, 5), device="cuda")
    out = torch.empty((cap, 128), device="cuda")
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    acc = np.zeros(3); nk = 0
    for f in range(frames + 3):
        e[0].record(); hnd.set_image_device(imgs[f % 4].data_ptr(), w, h, s)
        e[1].record(); m, dropped = hnd.orient_keypoints_device(exs[f % 4].data_ptr(), None, ne, kps.data_ptr(), None, cap, s)
        e[2].record(); hnd.describe_keypoints_device(kps.data_ptr(), m, out.data_ptr(), s)
        e[3].record(); torch.cuda.synchronize()
        if f >= 3:
            acc += [e[i].elapsed_time(e[i + 1]) for i in range(3)]; nk += m
    acc /= frames
    print(f"{tag}: {w}x{h}, {ne} extrema -> {nk // frames} keypoints/frame: pyramid {acc[0]:.3f} ms, a-trous stack + orientation "
          f"{acc[1]:.3f} ms (incl. host sync for the count), sample+describe {acc[2]:.3f} ms", flush=True)


def locality_order(k5, fid, psf=24.0, tile=64):
    """host-side sort key (frame, pyramid level, 64 x 64 tile of that level, row-major): what a device binning pass
     would produce -- patch_gradients.glsl:42-50 picks the level"""
    scale = k5[:, 2] * psf / 32.0
    lvl = np.clip(np.floor(np.log2(scale)), 0, 15).astype(np.int64)
    tx = (k5[:, 0] / 2.0 ** lvl).astype(np.int64) // tile
    ty = (k5[:, 1] / 2.0 ** lvl).astype(np.int64) // tile
    return np.lexsort((tx, ty, lvl, fid))


def run_locality(w, h, nk, frames, tag, order, reps=5, counters_only=False):
    """describe-only time of the SAME keypoints in the given order: 'random' (as generated) or 'sorted'
    (frame, level, 64 x 64 tile).  Pyramids are built once, outside the timed region."""
    n = nk * frames
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, pool_mode=lfp.POOL_F16X3, max_frames=frames)
    s = torch.cuda.current_stream().cuda_stream
    imgs = torch.stack([frame(h, w, 100 + (f % 8)) for f in range(frames)]).contiguous()
    base = [np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=64.0), np.zeros((nk, 1), np.float32)], axis=1) for f in range(8)]
    k5 = np.concatenate([base[f % 8] for f in range(frames)]).astype(np.float32)
    fid = np.repeat(np.arange(frames, dtype=np.int32), nk)
    if order == "sorted":
        o = locality_order(k5, fid)
        k5, fid = np.ascontiguousarray(k5[o]), np.ascontiguousarray(fid[o])
    kps, d_fid = torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
    out = torch.empty((n, 128), device="cuda")
    hnd.set_images_device(imgs.data_ptr(), frames, w, h, s)
    for _ in range(2):
        hnd.describe_keypoints_frames_device(kps.data_ptr(), d_fid.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    if counters_only:
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); hnd.describe_keypoints_frames_device(kps.data_ptr(), d_fid.data_ptr(), n, out.data_ptr(), s); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{tag} [{order:6s}]: {frames} x {w}x{h} x {nk} keypoints: describe-only {np.median(ts):.4f} ms (min {min(ts):.4f}) "
          f"= {n/np.median(ts)/1e3:.1f} M desc/s", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "locality":
    # usage: bench_keypoints.py locality [configs3|configs1] [random|sorted] [counters]
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    orders = [sys.argv[3]] if len(sys.argv) > 3 else ["random", "sorted", "random", "sorted"]
    co = len(sys.argv) > 4
    for o in orders:
        if which in ("both", "configs3"): run_locality(1920, 1080, 8192, 128, "configs[3] own form", o, counters_only=co)
        if which in ("both", "configs1"): run_locality(1920, 1080, 10000, 1, "configs[1]", o, reps=20, counters_only=co)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "orient":
    run_orient(1920, 1080, 7000, 20, "configs[1] from extrema")
    run_orient(640, 480, 1400, 20, "configs[2] frame from extrema")
    run_orient(3840, 2160, 5600, 20, "configs[4] from extrema")
    sys.exit(0)
run(1920, 1080, 10000, 50, "configs[1]")
run_batched(640, 480, 2000, 256, "configs[2] batched")
run(640, 480, 2000, 256, "configs[2]")
run(3840, 2160, 8000, 30, "configs[4] (describe part)")
