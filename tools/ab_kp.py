#!/usr/bin/env python3
"""Same-box timing of keypoint mode (BASELINE configs[2]: 256 frames 640x480 x 2000 keypoints; configs[1]: 10 000 keypoints on
a 1080p frame) for one build of the library (LF_MKD_LIB=...): the describe call alone, and pyramid + describe.
Usage: ab_kp.py [label]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints

side = torch.cuda.Stream()
torch.cuda.set_stream(side)
s = side.cuda_stream
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(lfp.LIB_PATH)


def frames(count, h, w, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((count, 1, h, w), device="cuda", generator=g)
    k = torch.exp(-0.5 * (torch.arange(-6, 7, device="cuda") / 2.0) ** 2); k /= k.sum()
    x = torch.nn.functional.conv2d(x, k.view(1, 1, 1, -1), padding=(0, 6))
    x = torch.nn.functional.conv2d(x, k.view(1, 1, -1, 1), padding=(6, 0))
    return x[:, 0].contiguous()


for tag, w, h, nk, nf, iters in (("configs2", 640, 480, 2000, 256, 10), ("configs1", 1920, 1080, 10000, 1, 50)):
    n = nk * nf
    hnd = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=h, max_frames=nf,
                        flags=int(os.environ.get("LF_KP_FLAGS", "0")))
    imgs = frames(nf, h, w, 11)
    base = [np.concatenate([random_keypoints(nk, w, h, 200 + f, margin=float(os.environ.get("LF_KP_MARGIN", "64" if nf == 1 else "8"))), np.zeros((nk, 1), np.float32)],
                           axis=1) for f in range(min(nf, 8))]
    kps = torch.from_numpy(np.concatenate([base[f % len(base)] for f in range(nf)]).astype(np.float32)).cuda()
    fid = torch.arange(nf, device="cuda", dtype=torch.int32).repeat_interleave(nk).contiguous()
    o = torch.empty((n, 128), device="cuda")
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    side.synchronize()
    tp = td = 0.0
    for it in range(iters + 2):
        e0.record(side)
        hnd.set_images_device(imgs.data_ptr(), nf, w, h, s)
        e1.record(side)
        hnd.describe_keypoints_frames_device(kps.data_ptr(), fid.data_ptr(), n, o.data_ptr(), s)
        e2.record(side)
        side.synchronize()
        if it >= 2:
            tp += e0.elapsed_time(e1); td += e1.elapsed_time(e2)
    tp /= iters; td /= iters
    import hashlib
    digest = hashlib.md5(o.cpu().numpy().tobytes()).hexdigest()[:12]   # builds that claim the same bits show the same digest
    print(f"{label:28s} {tag}: pyramid {tp:.3f} ms, describe {td:.3f} ms = {n / td / 1e3:.1f} M desc/s; "
          f"together {n / (tp + td) / 1e3:.1f} M desc/s; md5 {digest}", flush=True)
    del hnd
