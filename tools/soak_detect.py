#!/usr/bin/env python3
"""Soak of the host-image detect call (round 5): lf_mkd_detect / lf_mkd_detect_u8 through their recorded pipeline -- banded
upload included, at its default cut and at random cuts -- against the stage-by-stage form (LF_MKD_FLAG_DETECT_STEPWISE), bit for
bit, over random frame shapes (widths that are and are not multiples of 4, even and odd heights, frames from a few rows to
several megapixels), a-trous stack depths, top_n / min_size / capacity, both pixel types, several calls per handle.
Not part of the test suite.  Usage: soak_detect.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import local_features_python as lfp

import torch


def frame_u8(w, h, seed):
    """blobs of several sizes on a noisy ground, synthesised on the GPU (the CPU generator of the tests takes a minute at 10 MP)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.zeros((1, 1, h, w), device="cuda")
    for sigma, amp in ((1.5, 1.0), (3.0, 0.8), (6.0, 0.6), (12.0, 0.5)):
        n = (torch.rand((1, 1, h, w), device="cuda", generator=g) < 40.0 / (sigma * sigma * 200)).float() * \
            (torch.rand((1, 1, h, w), device="cuda", generator=g) * 2 - 1)
        rad = int(3 * sigma)
        k = torch.exp(-0.5 * (torch.arange(-rad, rad + 1, device="cuda") / sigma) ** 2)
        n = torch.nn.functional.conv2d(n, k.view(1, 1, 1, -1), padding=(0, rad))
        n = torch.nn.functional.conv2d(n, k.view(1, 1, -1, 1), padding=(rad, 0))
        x += amp * n
    x = 0.5 + 0.35 * x / x.abs().max() + 0.02 * torch.rand((1, 1, h, w), device="cuda", generator=g)
    return np.ascontiguousarray((x[0, 0].clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy())


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0, bad, total_kp, banded_rounds = time.time(), 0, 0, 0
for r in range(rounds):
    kind = r % 4
    if kind == 0:      # small and odd
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 500))
    elif kind == 1:    # aligned, mid-sized
        w, h = 4 * int(rng.integers(100, 500)), 2 * int(rng.integers(100, 600))
    elif kind == 2:    # large: the banded upload as planned (>= 6 MB: as f32 from 1.5 MP, as u8 from 6 MP)
        w, h = 4 * int(rng.integers(500, 1050)), 2 * int(rng.integers(500, 1600))
    else:              # slim or flat shapes
        w, h = (4 * int(rng.integers(10, 60)), int(rng.integers(1500, 3000))) if rng.random() < 0.5 else (
            4 * int(rng.integers(400, 1000)), int(rng.integers(30, 140)))
    n_scales = int(rng.integers(2, 7))
    max_blobs = int(rng.choice([256, 2048, 20000]))
    cap = int(rng.choice([64, 1000, 30000]))
    u8 = frame_u8(w, h, int(rng.integers(1 << 30)))
    f32 = np.ascontiguousarray(u8.astype(np.float32) / np.float32(255))
    kw = dict(max_features=cap, max_image_width=w, max_image_height=h, max_blobs=max_blobs, n_scales=n_scales)
    frac = None if rng.random() < 0.4 else f"{rng.uniform(0.08, 0.93):.3f}"
    os.environ.pop("LF_MKD_BAND_SPLIT", None)
    if frac is not None:
        os.environ["LF_MKD_BAND_SPLIT"] = frac
    ref = lfp.MkdHandle(flags=lfp.FLAG_DETECT_STEPWISE, **kw)
    rec = lfp.MkdHandle(**kw)
    ok = True
    nk = 0
    for call in range(3):
        top_n = int(rng.choice([0, 50, 700, 5000]))
        min_size = float(rng.choice([0.0, 0.0, 2.5]))
        want = ref.detect(f32, top_n, min_size, cap)
        for img in ((f32, u8) if call else (u8, f32)):
            got = rec.detect(img, top_n, min_size, cap)
            ok &= got[2:] == want[2:] and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        nk += len(want[0])
    total_kp += nk
    bad += not ok
    banded_rounds += frac is not None or w * h * 4 >= 12000000
    print(f"round {r:3d}: {w}x{h} n_scales {n_scales} max_blobs {max_blobs} capacity {cap} cut {frac or 'default'}: "
          f"{nk} keypoints over 3 requests x 2 pixel types, recorded == stage by stage {ok}", flush=True)
    ref.close(); rec.close()
print(f"soak_detect: {rounds} shapes ({banded_rounds} with a banded upload), {bad} with a difference; {total_kp} keypoints compared; "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
