#!/usr/bin/env python3
"""Same-box A/B of the pyramid build (lf_mkd_set_images_device, describe-only handle): the launch sequence of round 4
(LF_MKD_NO_DOWN_STAGED=1) and round 5's staged decimation (the runs with the two fused level-0 + level-1 kernels that were
measured and removed are kept in profiles/r05_ab_pyramid.txt).  HIP-event time of the call,
median of 20, for BASELINE configs[3] own form (128 x 1080p), configs[2] (256 x 640x480), one 1080p frame, one 4K frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np, torch
import local_features_python as lfp

side = torch.cuda.Stream(); torch.cuda.set_stream(side); s = side.cuda_stream
for w, h, frames in ((1920, 1080, 128), (640, 480, 256), (1920, 1080, 1), (3840, 2160, 1)):
    imgs = torch.rand((frames, h, w), device="cuda")
    hnd = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=h, max_frames=frames)
    for label, env in (("round 4 sequence", {"LF_MKD_NO_DOWN_STAGED": "1"}), ("+ staged decimation", {})):
        for k in ("LF_MKD_NO_DOWN_STAGED",):
            os.environ.pop(k, None)
        os.environ.update(env)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for i in range(23):
            e0.record(); hnd.set_images_device(imgs.data_ptr(), frames, w, h, s); e1.record(); torch.cuda.synchronize()
            if i >= 3: ts.append(e0.elapsed_time(e1))
        gb = frames * w * h * 4 * (1 + 1 + 1 / 3) / 1e9          # read the frames, write level 0, write levels >= 1 (aprons not counted)
        print(f"{frames:4d} x {w}x{h}  {label:32s} {np.median(ts)*1e3:9.1f} us  ({gb / np.median(ts) * 1e3 / 1e3:5.2f} TB/s on the minimal {gb:.2f} GB)", flush=True)
