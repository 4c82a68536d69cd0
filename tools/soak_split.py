#!/usr/bin/env python3
"""Soak of the row-split form of the keypoint kernel (requests of at most 4096 keypoints): random request sizes, frame sizes,
frames per handle, keypoints anywhere; two handles launching on two streams at once (their workgroups share the chip: a consumer
workgroup of one launch waits while the other launch's workgroups hold CUs); every result against the whole-patch form of the same
request (<= 1e-5 relative L2), repeated launches bit for bit, the handle's error word (a partial sum that never arrived) zero.
Not part of the test suite.  Usage: soak_split.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import local_features_python as lfp
from gen_golden import random_keypoints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(77)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
t0 = time.time()
rounds = launches = 0
worst = 0.0
while time.time() - t0 < budget:
    w, h = int(rng.integers(64, 1500)), int(rng.integers(48, 1000))
    nf = int(rng.integers(1, 4))
    hs = [lfp.MkdHandle(max_features=4096, max_image_width=w, max_image_height=h, max_frames=nf,
                        angle_mode=int(rng.integers(0, 3))) for _ in range(2)]
    imgs = torch.rand((nf, h, w), device="cuda")
    torch.cuda.synchronize()
    for hd, st in zip(hs, streams):
        hd.set_images_device(imgs.data_ptr(), nf, w, h, st.cuda_stream)
    reqs = []
    for i in range(12):
        n = int(rng.choice([1, 2, 31, 32, 33, 64, int(rng.integers(1, 4097)), int(rng.integers(1, 4097)), 2048, 2049, 4096]))
        k = np.concatenate([random_keypoints(n, w, h, int(rng.integers(1 << 30)), margin=-2.0), np.zeros((n, 1), np.float32)], axis=1)
        k[:, 2] = np.exp(rng.uniform(np.log(0.8), np.log(100.0), n))
        fid = rng.integers(0, nf, n).astype(np.int32)
        reqs.append((n, torch.from_numpy(k.astype(np.float32)).cuda(), torch.from_numpy(fid).cuda()))
    outs = [[torch.full((n, 128), float("nan"), device="cuda") for n, _, _ in reqs] for _ in range(3)]
    torch.cuda.synchronize()
    # the split form, twice, the two handles' launches interleaved on their two streams
    os.environ.pop("LF_MKD_KP_SPLIT", None)
    for rep in range(2):
        for i, (n, dk, df) in enumerate(reqs):
            hd, st = hs[i % 2], streams[i % 2]
            hd.describe_keypoints_frames_device(dk.data_ptr(), df.data_ptr(), n, outs[rep][i].data_ptr(), st.cuda_stream)
            launches += 1
    torch.cuda.synchronize()
    os.environ["LF_MKD_KP_SPLIT"] = "1"
    for i, (n, dk, df) in enumerate(reqs):
        hs[i % 2].describe_keypoints_frames_device(dk.data_ptr(), df.data_ptr(), n, outs[2][i].data_ptr(), streams[i % 2].cuda_stream)
    torch.cuda.synchronize()
    os.environ.pop("LF_MKD_KP_SPLIT", None)
    for i, (n, _, _) in enumerate(reqs):
        a, b, c = (outs[r][i].cpu().numpy() for r in range(3))
        assert np.array_equal(a, b), (rounds, i, n, "not deterministic")
        assert np.isfinite(a).all(), (rounds, i, n)
        e = float((np.linalg.norm(a - c, axis=1) / np.linalg.norm(c, axis=1)).max())
        worst = max(worst, e)
        assert e < 1e-5, (rounds, i, n, e)
    # the error word through the host entry point (it reads it): one more small request per handle
    for hd in hs:
        hd.describe_keypoints(reqs[0][1].cpu().numpy()[:5])
    rounds += 1
print(f"soak_split: {rounds} rounds, {launches} row-split launches on two handles / two streams at once: repeated launches bit for bit, "
      f"worst relative L2 against the whole-patch form {worst:.2e}, no partial sum lost; {time.time() - t0:.0f} s")
