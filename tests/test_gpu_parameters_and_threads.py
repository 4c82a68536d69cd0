"""Round 6: the parts of the boundary no earlier run had varied or executed ("next round" item 1).
 * patch_scale_factor -- the reference's ONLY runtime parameter on this path (FeatureDetectParams, lib.rs:34-52; used at
   shaders/mkd/patch_gradients.glsl:42-50) -- away from its default 24: keypoints move across pyramid-level boundaries and
   into both level clamps; held against the oracle given the same value, through describe_keypoints, the multi-frame form
   and lf_mkd_detect (whose recording bakes the value in).
 * the caller's current HIP device is the same after every entry point as before it.
 * two handles (different PCA model, different pool mode) driven from two host threads at once return the bits of the
   single-threaded run (the documented threading model: one thread per handle, &mut self in the reference, mod.rs:346-367).
Everything goes through the C ABI; the oracle is the checker."""
import os
import sys
import threading

import numpy as np
import pytest

from conftest import GATE, assert_keypoint_parity, assert_same_descriptors, kp_form, rel_l2

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def lfp():
    import local_features_python as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "these tests need the MI355X"
    return t


def _keypoints_across_levels(n, w, h, seed, psf):
    """keypoints whose pyramid level floor(log2(size * psf / 32)) covers every level of the frame and both clamps (below
    level 0, beyond the last level), plus sizes placed a hair either side of every level boundary"""
    from gen_golden import random_keypoints
    rng = np.random.default_rng(seed)
    levels = int(np.ceil(np.log2(min(w, h))))
    k = random_keypoints(n, w, h, seed, margin=4.0)
    # log-uniform in scale = size * psf / 32 from 2^-1.5 (clamped to level 0) upwards; eight beyond (clamped to the last level)
    k[:, 2] = (32.0 / psf) * 2.0 ** rng.uniform(-1.5, levels - 2.0, n)
    k[:8, 2] = (32.0 / psf) * 2.0 ** rng.uniform(levels - 2.0, levels + 0.5, 8)     # the last two levels (a few texels) and beyond
    edge = random_keypoints(4 * (levels + 1), w, h, seed + 1, margin=4.0)
    for i in range(levels + 1):                    # scale = 2^i (1 -+ 2e-3): the level decision's two sides
        edge[4 * i:4 * i + 2, 2] = (32.0 / psf) * 2.0 ** i * (1 - 2e-3)
        edge[4 * i + 2:4 * i + 4, 2] = (32.0 / psf) * 2.0 ** i * (1 + 2e-3)
    k = np.concatenate([k, edge]).astype(np.float32)
    lv = np.floor(np.log2(k[:, 2].astype(np.float64) * psf / 32.0))
    assert (lv < 0).any() and (lv >= levels).any() and len(np.unique(np.clip(lv, 0, levels - 1))) == levels
    return np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))


@pytest.mark.parametrize("psf", [12.0, 32.0, 48.0])
def test_patch_scale_factor_keypoint_entry_points(lfp, torch, oracle, psf):
    from gen_golden import smooth_image
    w, hgt = 400, 304
    img = smooth_image(hgt, w, 61)
    k5 = _keypoints_across_levels(700, w, hgt, 62, psf)
    # (the same keypoints at the default value sample other footprints: the parameter reaches the kernel)
    h24 = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt)
    h24.set_image(img)
    d24 = h24.describe_keypoints(k5)
    outs = {}
    for flags in (0, lfp.FLAG_UNFUSED_KEYPOINTS):
        h = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt, patch_scale_factor=psf, flags=flags)
        h.set_image(img)
        if flags == 0:
            with kp_form(1):                                    # the whole-patch form: the two-launch form's arithmetic
                whole = h.describe_keypoints(k5)
        outs[flags] = h.describe_keypoints(k5)                  # 3 internal batches (the row-split form when fused)
        # (keypoints clamped to the last levels -- a few texels wide -- sample near-constant patches: pixels with gx == 0 are
        #  common there, so more patches than usual are set aside end to end; on the GPU's own patch bits all of them are held)
        assert_keypoint_parity(oracle, h, img, k5, outs[flags], what=f"patch_scale_factor {psf:g}, flags {flags}",
                               patch_scale_factor=psf, min_settled=0.85)
    assert np.array_equal(whole, outs[lfp.FLAG_UNFUSED_KEYPOINTS])            # one launch == sampler + patch kernel
    assert_same_descriptors(outs[0], whole, f"row-split vs whole-patch form, patch_scale_factor {psf:g}")
    assert rel_l2(outs[0], d24).max() > 0.05
    # the oracle at the default value does NOT describe these: the helper above would fail with the wrong parameter
    ref24 = oracle.describe_keypoints(img, k5[:, :4])
    assert np.median(rel_l2(outs[0], ref24)) > 1e-2


@pytest.mark.parametrize("psf", [12.0, 48.0])
def test_patch_scale_factor_multi_frame_form(lfp, torch, oracle, psf):
    from gen_golden import smooth_image
    w, hgt, nf = 208, 160, 3
    frames = np.ascontiguousarray(np.stack([smooth_image(hgt, w, 70 + f) for f in range(nf)]))
    kps = [_keypoints_across_levels(120 + 40 * f, w, hgt, 80 + f, psf) for f in range(nf)]
    allk = np.concatenate(kps).astype(np.float32)
    fid = np.concatenate([np.full(len(k), f, np.int32) for f, k in enumerate(kps)])
    perm = np.random.default_rng(3).permutation(len(allk))                   # frames interleaved within a launch
    allk, fid = np.ascontiguousarray(allk[perm]), np.ascontiguousarray(fid[perm])
    h = lfp.MkdHandle(max_features=128, max_image_width=w, max_image_height=hgt, max_frames=nf, patch_scale_factor=psf)
    d_frames, d_k, d_f = torch.from_numpy(frames).cuda(), torch.from_numpy(allk).cuda(), torch.from_numpy(fid).cuda()
    out = torch.empty((len(allk), 128), device="cuda")
    h.set_images_device(d_frames.data_ptr(), nf, w, hgt)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), len(allk), out.data_ptr())
    h.synchronize()
    got = out.cpu().numpy()
    single = lfp.MkdHandle(max_features=128, max_image_width=w, max_image_height=hgt, patch_scale_factor=psf)
    for f in range(nf):
        sel = fid == f
        single.set_image(frames[f])
        assert np.array_equal(got[sel], single.describe_keypoints(allk[sel])), f
        assert_keypoint_parity(oracle, single, frames[f], allk[sel], got[sel], what=f"multi-frame, patch_scale_factor {psf:g}, frame {f}",
                               patch_scale_factor=psf, min_settled=0.85)


@pytest.mark.parametrize("psf", [12.0, 32.0, 48.0])
def test_patch_scale_factor_detect(lfp, oracle, psf):
    """lf_mkd_detect with the parameter away from its default: stage by stage (first sighting), recording (second) and
    replay (third) return the same bits, the keypoints are the oracle's (the detector does not depend on the parameter),
    and the descriptors are the oracle's at this value."""
    from gen_golden import blob_image
    w, hgt = 480, 352
    img = blob_image(w, hgt, 17, 260)
    kw = dict(max_features=3000, max_image_width=w, max_image_height=hgt, max_blobs=2048, patch_scale_factor=psf)
    h = lfp.MkdHandle(**kw)
    calls = [h.detect(img, 300, 0.0, 3000) for _ in range(3)]
    ref = lfp.MkdHandle(flags=lfp.FLAG_DETECT_STEPWISE, **kw).detect(img, 300, 0.0, 3000)
    for c in calls:
        assert c[2:] == ref[2:] and np.array_equal(c[0], ref[0]) and np.array_equal(c[1], ref[1])
    k, d = calls[2][:2]
    assert len(k) > 250
    want_k, _ = oracle.detect(img, top_n=300, max_blobs=2048, patch_scale_factor=psf)
    assert k.shape == want_k.shape and np.abs(k[:, :3] - want_k[:, :3]).max() < 1e-3
    # the handle holds the frame after the call: the returned rows against the oracle's description of the returned keypoints
    # (at small values the footprints are small and flat: more patches hold a gx == 0 pixel than at the default)
    assert_keypoint_parity(oracle, h, img, k, d, what=f"lf_mkd_detect, patch_scale_factor {psf:g}", patch_scale_factor=psf,
                           min_settled=0.9)
    d24 = lfp.MkdHandle(max_features=3000, max_image_width=w, max_image_height=hgt, max_blobs=2048).detect(img, 300, 0.0, 3000)
    assert np.array_equal(d24[0], k) and np.median(rel_l2(d24[1], d)) > 1e-2


def _hip_current_device():
    """the calling thread's current device as the HIP runtime has it (torch asks the runtime liblf_mkd.so is bound to:
    local_features_python/_lib.py load_library; a second copy of libamdhip64 must not be loaded into the process)"""
    import torch
    return torch.cuda.current_device()


def _every_entry_point_once(lfp, torch, h, device):
    """a representative call of every family: patches, frame, keypoints, orientation, detector, detect (three sightings),
    stream pipeline, matcher, verification taps, error paths"""
    rng = np.random.default_rng(9)
    from gen_golden import blob_image
    w, hgt = 192, 144
    img = blob_image(w, hgt, 3, 80)
    p = rng.random((70, 32, 32)).astype(np.float32)
    d = h.describe_patches(p)
    with torch.cuda.device(device):
        dp, do = torch.from_numpy(p).cuda(), torch.empty((70, 128), device="cuda")
        h.describe_patches_device(dp.data_ptr(), 70, do.data_ptr())
        assert np.array_equal(do.cpu().numpy(), d)
        h.set_image(img)
        ex, _ = h.detect_extrema(1 << 12)
        kps, _ = h.orient_keypoints(ex)
        dk = h.describe_keypoints(kps)
        for _ in range(3):
            k2, d2, _, _ = h.detect(img, 0, 0.0, 4000)
        assert len(k2) == len(kps)
        assert_same_descriptors(d2, dk, "detect vs describe_keypoints of the same keypoints")
        h.match(dk, dk[::-1].copy())
        h.pyramid_level(1)
        h.coarse_layer(2, w, hgt)
        d_img = torch.from_numpy(img).cuda()
        d_k, d_d = torch.zeros((4000, 5), device="cuda"), torch.zeros((4000, 128), device="cuda")
        d_c = torch.zeros((8,), dtype=torch.int64, device="cuda")
        h.stream_create(w, hgt, 0, 0.0, 4000, d_img.data_ptr(), d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr())
        h.stream_frame()
        h.synchronize()
    with pytest.raises(RuntimeError):
        h.set_image(np.zeros((hgt + 1, w), np.float32))        # an error return restores the device too
    return d, dk


def test_entry_points_leave_the_callers_current_device_alone(lfp, torch):
    """One process, one handle per GPU: after any call the calling thread's current device is what
    it was.  With one visible GPU the device cannot differ from the handle's; the check then only pins hipGetDevice."""
    n = torch.cuda.device_count()
    h = lfp.MkdHandle(max_features=64, max_image_width=192, max_image_height=144, max_blobs=1024, device=0)
    before = _hip_current_device()
    _every_entry_point_once(lfp, torch, h, 0)
    assert _hip_current_device() == before
    h.close()
    assert _hip_current_device() == before
    if n < 2:
        pytest.skip("one visible GPU: the caller's device cannot differ from the handle's (restore path not observable)")
    # the caller sits on the LAST device, the handle lives on device 0 -- and one more handle on the caller's own
    torch.cuda.set_device(n - 1)
    assert _hip_current_device() == n - 1
    h0 = lfp.MkdHandle(max_features=64, max_image_width=192, max_image_height=144, max_blobs=1024, device=0)
    assert _hip_current_device() == n - 1                      # creation restores it as well
    a = _every_entry_point_once(lfp, torch, h0, 0)
    assert _hip_current_device() == n - 1
    h1 = lfp.MkdHandle(max_features=64, max_image_width=192, max_image_height=144, max_blobs=1024, device=n - 1)
    b = _every_entry_point_once(lfp, torch, h1, n - 1)
    assert _hip_current_device() == n - 1
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])     # the same bits on either device
    h0.close()
    assert _hip_current_device() == n - 1
    torch.cuda.set_device(0)


def _workload(lfp, h, seed, rounds):
    """what one thread does with its handle: every family of host-pointer entry point, several times over; returns every
    result so that two runs can be compared bit for bit"""
    from gen_golden import blob_image, random_keypoints, smooth_image
    rng = np.random.default_rng(seed)
    w, hgt = 256, 192
    out = []
    for r in range(rounds):
        p = rng.random((150 + 37 * r, 32, 32)).astype(np.float32)
        out.append(h.describe_patches(p))
        img = smooth_image(hgt, w, seed + r)
        k = random_keypoints(260, w, hgt, seed + 100 + r)
        k5 = np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))
        h.set_image(img)
        out.append(h.describe_keypoints(k5))
        blobs = blob_image(w, hgt, seed + 7 * r, 120)
        d_all = None
        for top_n in (0, 150):                      # (the rounds repeat the requests: first sighting, recording, replay --
            kk, dd, db, df = h.detect(blobs, top_n, 0.0, 3000)          #  while the other thread does the same)
            out += [kk, dd, np.array([db, df])]
            d_all = dd if d_all is None else d_all
        u8 = np.ascontiguousarray(np.rint(blobs * 255).astype(np.uint8))
        kk, dd, _, _ = h.detect(u8, 150, 0.0, 3000)
        out += [kk, dd]
        ex, _ = h.detect_extrema(1 << 13)
        kp, _ = h.orient_keypoints(ex)
        assert len(dd) >= 2 and len(d_all) >= 2
        out += [ex, kp, h.match(dd, d_all)]
    return out


def test_two_handles_on_two_host_threads_return_the_single_threaded_bits(lfp, torch, oracles):
    """Two handles that share nothing but the device -- another PCA model, another pool mode (so: other kernels, the
    two-launch keypoint form and the stage-by-stage detect on one side, the fused kernel and recorded pipelines on the other)
    -- each driven by its own host thread (ctypes releases the GIL for the duration of a call), against the same work done
    one handle after the other."""
    cfg = [dict(pca="liberty", pool_mode=lfp.POOL_F16X3), dict(pca="notredame", pool_mode=lfp.POOL_F32)]
    common = dict(max_features=512, max_image_width=256, max_image_height=192, max_blobs=2048)
    rounds = 4
    serial = [_workload(lfp, lfp.MkdHandle(**c, **common), 500 + 50 * i, rounds) for i, c in enumerate(cfg)]
    handles = [lfp.MkdHandle(**c, **common) for c in cfg]
    results, errors = [None, None], []
    gate = threading.Barrier(2)

    def run(i):
        try:
            gate.wait(timeout=60)
            results[i] = _workload(lfp, handles[i], 500 + 50 * i, rounds)
        except BaseException as e:      # noqa: BLE001 -- reported below, in the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for i in range(2):
        assert len(results[i]) == len(serial[i])
        for j, (a, b) in enumerate(zip(results[i], serial[i])):
            assert a.shape == b.shape and np.array_equal(a, b), (i, j)
    # and the two sides are what they claim to be: each one's patch descriptors are its own model's oracle's
    rng = np.random.default_rng(500)
    p = rng.random((150, 32, 32)).astype(np.float32)
    from oracle import BLUR_CONTRACT
    assert rel_l2(results[0][0], oracles["liberty"].describe_patches(p, atan_mode=BLUR_CONTRACT)).max() < GATE
    assert rel_l2(results[0][0], oracles["notredame"].describe_patches(p, atan_mode=BLUR_CONTRACT)).max() > 1e-2
