"""BASELINE.json's configurations at their own sizes, against the oracle (GPU).

configs[1]: 10 000 keypoints on one 1920x1080 frame      configs[3]: the per-GPU share, in its own form (128 frames of
1920x1080 with 8192 given keypoints each, keypoint mode) and as 2^20 patches (SURVEY 8(d)'s headline form)
(configs[2] and configs[4] live in test_gpu_detector.py / test_gpu_parity.py next to the pipelines they exercise;
 configs[0] -- the reference's match_images on its own two photographs -- is below.)"""
import os
import sys

import numpy as np
import pytest

from conftest import assert_same_descriptors, kp_form, GATE, GOLDEN, ROOT, assert_keypoint_parity, assert_patch_parity, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lfp():
    import local_features_python as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "these tests need the MI355X"
    return t


def test_configs3_per_gpu_share_of_2pow20_patches(lfp, torch, oracle):
    """2^20 patches (4 GiB in, 512 MiB out) in ONE call, as bench.py times it: size-independent properties over all of
    them, and 4096 rows drawn across the whole range against the oracle."""
    from oracle import ATAN_SHADER
    n = 1 << 20
    gen = torch.Generator(device="cuda").manual_seed(0x4D4B44)
    p = torch.rand((n, 32, 32), device="cuda", generator=gen)
    out = torch.empty((n, 128), device="cuda")
    h = lfp.MkdHandle(max_features=n)
    s = torch.cuda.current_stream().cuda_stream
    h.describe_patches_device(p.data_ptr(), n, out.data_ptr(), s)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    assert float((out.norm(dim=1) - 1).abs().max()) < 1e-5
    # a descriptor depends on its own patch only: the same rows through a small request (the 4-wave form of the kernel)
    idx = torch.randint(0, n, (4096,), device="cuda", generator=gen)
    idx[:3] = torch.tensor([0, n - 1, n // 2], device="cuda")
    sub = p[idx].contiguous()
    out2 = torch.empty((4096, 128), device="cuda")
    h.describe_patches_device(sub.data_ptr(), 4096, out2.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(out2, out[idx])
    # and it is the descriptor the oracle computes
    worst = assert_patch_parity(oracle, sub.cpu().numpy(), out[idx].cpu().numpy(), ATAN_SHADER, what="2^20")
    print(f"2^20 patches: worst relative L2 of 4096 sampled descriptors vs the oracle = {worst:.2e}")
    # a second run gives the same bits (no race in the LDS pipeline at full occupancy)
    out3 = torch.empty_like(out)
    h.describe_patches_device(p.data_ptr(), n, out3.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(out, out3)


def test_configs3_own_form_128_frames_1080p_8192_keypoints_each(lfp, torch, oracle):
    """BASELINE configs[3] as it is written, the share of one GPU: "8M kpts over 1024 1080p frames sharded by image across
    8 x MI355X" = 128 frames of 1920x1080 with 8192 given keypoints each (2^20 descriptors), keypoint mode, through
    set_images_device + describe_keypoints_frames_device in ONE call (the reference's unit of work is an image:
    vulkan/mod.rs:363-453).  Finite and unit-norm over all of them, a 1000-row re-request and a second run give the same
    bits, 256 rows of each of 8 frames against the oracle end to end (pyramid -> sampling -> describe).  The frames a rank
    holds are those of sharding.frames_of_rank: here rank 3 of 8, i.e. global frames 3, 11, 19, ..."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import random_keypoints, smooth_image
    from local_features_python import sharding
    w, hgt, nf, nk, world, rank = 1920, 1080, 128, 8192, 8, 3
    n = nf * nk
    mine = sharding.frames_of_rank(nf * world, rank, world)                # global indices of this rank's frames
    assert len(mine) == nf and mine[0] == rank and mine[1] == rank + world
    # 8 distinct frames (a 1080p frame takes the CPU 0.4 s), each used for 16 of the rank's frames with its own keypoints
    base = [np.ascontiguousarray(smooth_image(hgt, w, 500 + f), np.float32) for f in range(8)]
    k5 = np.concatenate([np.concatenate([random_keypoints(nk, w, hgt, 9000 + g, margin=64.0), np.zeros((nk, 1), np.float32)],
                                        axis=1) for g in mine]).astype(np.float32)
    fid = np.repeat(np.arange(nf, dtype=np.int32), nk)
    d_img = torch.stack([torch.from_numpy(base[g % 8]) for g in mine]).cuda().contiguous()      # 1.06 GB
    d_k, d_f = torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
    out = torch.empty((n, 128), device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    h = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=hgt, max_frames=nf)
    h.set_images_device(d_img.data_ptr(), nf, w, hgt, stream.cuda_stream)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert bool(torch.isfinite(out).all())
    assert float((out.norm(dim=1) - 1).abs().max()) < 1e-5
    # a descriptor depends on its own keypoint and frame only: 1000 rows from all over the batch in a request of their own
    gen = torch.Generator(device="cuda").manual_seed(5)
    idx = torch.sort(torch.randint(0, n, (1000,), device="cuda", generator=gen)).values
    idx[0], idx[-1] = 0, n - 1
    sub_k, sub_f = d_k[idx].contiguous(), d_f[idx].contiguous()
    out2 = torch.empty((1000, 128), device="cuda")
    torch.cuda.synchronize()
    with kp_form(1):       # the form the whole batch took: the same bits; the form a request of 1000 takes by itself: close
        h.describe_keypoints_frames_device(sub_k.data_ptr(), sub_f.data_ptr(), 1000, out2.data_ptr(), stream.cuda_stream)
        stream.synchronize()
    assert torch.equal(out2, out[idx])
    h.describe_keypoints_frames_device(sub_k.data_ptr(), sub_f.data_ptr(), 1000, out2.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert_same_descriptors(out2.cpu().numpy(), out[idx].cpu().numpy(), "1000 rows of a large batch in a request of their own")
    # a second run of the whole call (pyramids rebuilt) gives the same bits
    out3 = torch.empty_like(out)
    h.set_images_device(d_img.data_ptr(), nf, w, hgt, stream.cuda_stream)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out3.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(out, out3)
    got = out.cpu().numpy()
    del h, out, out3, d_img
    torch.cuda.empty_cache()
    # 256 rows of each of 8 of the rank's frames against the oracle, end to end
    one = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt)
    rng = np.random.default_rng(7)
    for f in (0, 1, 9, 42, 64, 77, 126, 127):
        rows = f * nk + np.sort(rng.choice(nk, 256, replace=False))
        one.set_image(base[mine[f] % 8])
        # (coordinates reach 1920: see test_configs1)
        assert_keypoint_parity(oracle, one, base[mine[f] % 8], k5[rows], got[rows], what=f"configs[3] local frame {f}",
                               patch_tol=5e-5)


def test_configs1_10k_keypoints_on_a_1080p_frame(lfp, torch, oracle):
    """10 000 keypoints on one 1920x1080 frame: unit norm and batch-independence over all of them, a 1000-row sample
    against the oracle end to end (pyramid -> sampling -> describe)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, n = 1920, 1080, 10000
    img = np.ascontiguousarray(smooth_image(hgt, w, 11), np.float32)
    k5 = np.concatenate([random_keypoints(n, w, hgt, 12, margin=64.0), np.zeros((n, 1), np.float32)], axis=1)
    k5 = np.ascontiguousarray(k5, np.float32)
    h = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    d = h.describe_keypoints(k5)
    assert d.shape == (n, 128) and np.isfinite(d).all()
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5
    rng = np.random.default_rng(13)
    pick = np.sort(rng.choice(n, 1000, replace=False))
    # the same keypoints in a request of their own give the same bits
    with kp_form(1):
        assert np.array_equal(h.describe_keypoints(k5[pick]), d[pick])
    assert_same_descriptors(h.describe_keypoints(k5[pick]), d[pick], "1000 of 10 000 keypoints in a request of their own")
    # (sample coordinates reach 1920: one ulp there is 1.2e-4 texel, and the two sides round their sin / cos / exp2
    #  differently -- the sampled values agree to that times the local slope)
    assert_keypoint_parity(oracle, h, img, k5[pick], d[pick], what="configs[1]", patch_tol=5e-5)


def test_configs0_match_images_on_the_reference_photographs(lfp, oracle):
    """BASELINE configs[0]: `match_images bird.jpg houses.jpg` (examples/match_images/src/main.rs:36-159):
    detect_top_n(2000, 0) with max_features 3000, max_blobs 8000, n_scales 5 on both photographs, match both ways.
    houses.jpg is the reference's 4096 x 3072 benchmark input (benches/bench.rs:41-74).  The two pictures show
    different scenes, so what is checked is every stage against the oracle on the real 12-megapixel input."""
    sys.path.insert(0, os.path.join(ROOT, "local-features_amd", "examples"))
    import match_images as ex
    img1 = ex.load_gray(os.path.join(GOLDEN, "bird.jpg"))
    img2 = ex.load_gray(os.path.join(GOLDEN, "houses.jpg"))
    assert img2.shape == (3072, 4096)
    kp1, kp2, d1, d2, m12, m21 = ex.match_images(img1, img2)
    assert len(kp2) > 2000 and d2.shape == (len(kp2), 128)
    # keypoints of the large photograph = the oracle's detect (same extrema, same top-2000, same orientations)
    want_k, _ = oracle.detect(img2, n_scales=5, top_n=2000, max_blobs=8000, max_features=3000)
    got = np.array([(k.x, k.y, k.size, k.angle, k.response) for k in kp2], np.float32)
    assert got.shape == want_k.shape
    assert np.abs(got[:, :2] - want_k[:, :2]).max() < 2e-3 and np.abs(got[:, 2] / want_k[:, 2] - 1).max() < 1e-4
    da = np.abs(got[:, 3] - want_k[:, 3])
    assert (np.minimum(da, 360 - da) < 1e-3).mean() > 0.995
    # descriptors of a 512-keypoint sample against the oracle on the photograph
    hgt, w = img2.shape
    h = lfp.MkdHandle(max_features=3000, max_image_width=w, max_image_height=hgt, n_scales=5)
    h.set_image(img2)
    pick = np.arange(0, len(got), max(1, len(got) // 1024))[:1024]
    # (an 8-bit photograph has flat areas -- sky, walls -- where gx is EXACTLY 0 after the blur of equal 8-bit values: more
    #  patches than on synthetic frames sit on the shader's gx == 0 discontinuity and are set aside, 5 % of this
    #  photograph's keypoints (profiles/r04_parity_report.txt); the bar for a photograph is therefore 0.93 on 1024 rows,
    #  for synthetic frames the helper's default 0.97)
    assert_keypoint_parity(oracle, h, img2, got[pick], d2[pick], what="houses", patch_tol=1e-4, min_settled=0.93)
    # the match lists are the oracle's on the same descriptors
    assert m12 == [(i, int(j)) for i, j in enumerate(oracle.match(d1, d2)[0]) if j >= 0]
    assert m21 == [(i, int(j)) for i, j in enumerate(oracle.match(d2, d1)[0]) if j >= 0]


def test_configs2_256_frames_640x480_2000_keypoints_each(lfp, torch, oracle):
    """BASELINE configs[2] at its own size: 256 frames of 640x480 with 2000 given keypoints each (512 000 descriptors)
    through set_images_device + describe_keypoints_frames_device in one call: finite and unit-norm over all of them, a
    1000-row re-request gives the same bits, and 256 rows of each of 8 frames against the oracle end to end (pyramid ->
    sampling -> describe) through the same helper the small keypoint tests use."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, nf, nk = 640, 480, 256, 2000
    n = nf * nk
    # 16 distinct frames, each used 16 times with its own keypoints (generating 256 frames takes the CPU a while)
    base = [np.ascontiguousarray(smooth_image(hgt, w, 300 + f), np.float32) for f in range(16)]
    k5 = np.concatenate([np.concatenate([random_keypoints(nk, w, hgt, 400 + f, margin=64.0), np.zeros((nk, 1), np.float32)],
                                        axis=1) for f in range(nf)]).astype(np.float32)
    fid = np.repeat(np.arange(nf, dtype=np.int32), nk)
    d_img = torch.stack([torch.from_numpy(base[f % 16]) for f in range(nf)]).cuda().contiguous()
    d_k, d_f = torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
    out = torch.empty((n, 128), device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    h = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=hgt, max_frames=nf)
    h.set_images_device(d_img.data_ptr(), nf, w, hgt, stream.cuda_stream)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert bool(torch.isfinite(out).all())
    assert float((out.norm(dim=1) - 1).abs().max()) < 1e-5
    # a descriptor depends on its own keypoint and frame only: 1000 rows from all over the batch in a request of their own
    gen = torch.Generator(device="cuda").manual_seed(5)
    idx = torch.sort(torch.randint(0, n, (1000,), device="cuda", generator=gen)).values
    idx[0], idx[-1] = 0, n - 1
    sub_k, sub_f = d_k[idx].contiguous(), d_f[idx].contiguous()
    out2 = torch.empty((1000, 128), device="cuda")
    torch.cuda.synchronize()
    with kp_form(1):       # the form the whole batch took: the same bits; the form a request of 1000 takes by itself: close
        h.describe_keypoints_frames_device(sub_k.data_ptr(), sub_f.data_ptr(), 1000, out2.data_ptr(), stream.cuda_stream)
        stream.synchronize()
    assert torch.equal(out2, out[idx])
    h.describe_keypoints_frames_device(sub_k.data_ptr(), sub_f.data_ptr(), 1000, out2.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert_same_descriptors(out2.cpu().numpy(), out[idx].cpu().numpy(), "1000 rows of a large batch in a request of their own")
    # and a second run of the whole batch gives the same bits
    out3 = torch.empty_like(out)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out3.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    assert torch.equal(out, out3)
    got = out.cpu().numpy()
    del h, out, out3, d_img
    # 256 rows of each of 8 frames against the oracle (a single-frame handle samples the same bits: one sampling arithmetic);
    # at 256 rows the settled fraction means something: the helper's default bar (0.97) applies
    one = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt)
    rng = np.random.default_rng(6)
    for f in (0, 1, 17, 100, 129, 200, 254, 255):
        rows = f * nk + np.sort(rng.choice(nk, 256, replace=False))
        one.set_image(base[f % 16])
        assert_keypoint_parity(oracle, one, base[f % 16], k5[rows], got[rows], what=f"configs[2] frame {f}")


def test_large_keypoint_batches_fused_and_two_launch_forms_agree(lfp, torch, oracle):
    """80 000 keypoints on 4 frames: the one-launch form (patches sampled inside the describe kernel) and the two-launch form
    through HBM (LF_MKD_FLAG_UNFUSED_KEYPOINTS) give the same bits, repeated calls agree bit for bit (no race between the
    producer and the describe waves of a workgroup), and a sample of every frame is what the oracle gives."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, nf, nk = 320, 240, 4, 20000
    n = nf * nk
    imgs = np.ascontiguousarray(np.stack([smooth_image(hgt, w, 70 + f) for f in range(nf)]), np.float32)
    k5 = np.concatenate([np.concatenate([random_keypoints(nk, w, hgt, 80 + f), np.zeros((nk, 1), np.float32)], axis=1)
                         for f in range(nf)]).astype(np.float32)
    fid = np.repeat(np.arange(nf, dtype=np.int32), nk)
    d_img, d_k, d_f = torch.from_numpy(imgs).cuda(), torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for flags in (0, lfp.FLAG_UNFUSED_KEYPOINTS, 0):
        h = lfp.MkdHandle(max_features=n, max_image_width=w, max_image_height=hgt, max_frames=nf, flags=flags)
        out = torch.zeros((n, 128), device="cuda")
        h.set_images_device(d_img.data_ptr(), nf, w, hgt, s)
        h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out.data_ptr(), s)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
        del h
    a, b, a2 = outs
    assert np.array_equal(a, a2)                                           # deterministic across handles and calls
    assert np.isfinite(a).all() and np.abs(np.linalg.norm(a, axis=1) - 1).max() < 1e-5
    assert np.array_equal(a, b)                                            # one sampling arithmetic: the same bits
    # a sample of every chunk against the oracle, end to end, through the helper: the GPU-sampled patches of the picked
    # rows must meet the gate on every settled row (no allowance)
    pick = np.arange(0, n, 64)                  # 312 rows of every frame: enough for the helper's default bar (0.97)
    one = lfp.MkdHandle(max_features=len(pick), max_image_width=w, max_image_height=hgt)
    for f in range(nf):
        sel = pick[fid[pick] == f]
        one.set_image(imgs[f])
        assert_keypoint_parity(oracle, one, imgs[f], k5[sel], a[sel], what=f"large batch, frame {f}")
