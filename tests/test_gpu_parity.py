"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(liblf_mkd.so); the oracle is only the checker.  Gate: 1e-4 relative L2 per descriptor
(BASELINE.json north_star); observed values are far below and asserted tighter where stable."""
import numpy as np
import pytest

from conftest import assert_keypoint_parity, assert_patch_parity, assert_same_descriptors, golden, kp_form, rel_l2

pytestmark = pytest.mark.gpu

GATE = 1e-4


@pytest.fixture(scope="module")
def lfp():
    import local_features_python as m
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "these tests need the MI355X"
    return t


def _handles(lfp, **kw):
    return {(a, p): lfp.MkdHandle(angle_mode=a, pool_mode=p, **kw)
            for a in (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT) for p in (lfp.POOL_F32, lfp.POOL_F16X3)}


@pytest.mark.parametrize("name", ["liberty", "notredame", "yosemite"])
def test_patch_goldens(lfp, name):
    g = golden(f"patches_{name}.npz")
    for (a, p), h in _handles(lfp, pca=name, max_features=64).items():
        d = h.describe_patches(g["patches"])
        ref = g["desc_shader"] if a == lfp.ANGLE_SHADER else g["desc_libm"]
        e = rel_l2(d, ref)
        assert e.max() < GATE, (name, a, p, e)
        assert e.max() < 2e-5, (name, a, p, e)
        assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)


def test_fp6_cross_term_mode_stays_inside_the_gate(lfp, torch, oracle):
    """LF_MKD_POOL_F16_FP6 (an experiment kept as a mode, NOTEBOOK.md section 11): the harmonics' cross terms in e2m3.  Its
    error is seven times the default mode's (2.95e-5 worst against 4.2e-6) -- the reason it is not the default -- and must still be inside the gate on every
    patch: goldens of the three models, 4096 random and the structured patches, both request sizes (4- and 8-wave forms),
    ragged tail, every angle mode.  Also: the keypoint entry points work in this mode (two-launch form)."""
    from oracle import ATAN_LIBM, ATAN_SHADER, BLUR_CONTRACT
    worst = 0.0
    for name in ("liberty", "notredame", "yosemite"):
        g = golden(f"patches_{name}.npz")
        h = lfp.MkdHandle(pca=name, max_features=64, pool_mode=lfp.POOL_F16_FP6)
        e = rel_l2(h.describe_patches(g["patches"]), g["desc_shader"])
        worst = max(worst, e.max())
        assert e.max() < GATE, (name, e)
    rng = np.random.default_rng(77)
    p = np.concatenate([rng.random((3000, 32, 32)), np.clip(rng.normal(0.5, 0.05, (1099, 32, 32)), 0, 1)]).astype(np.float32)
    for angle, mode in ((lfp.ANGLE_SHADER, ATAN_SHADER), (lfp.ANGLE_EXACT, ATAN_LIBM), (lfp.ANGLE_EXACT_ZERO, ATAN_SHADER)):
        ref = oracle.describe_patches(p, atan_mode=mode | BLUR_CONTRACT, nthreads=8)
        for cap in (len(p), 1 << 20):                      # one round of 64-patch workgroups / the 8-wave form
            h = lfp.MkdHandle(max_features=cap, angle_mode=angle, pool_mode=lfp.POOL_F16_FP6)
            d = h.describe_patches(p) if cap == len(p) else None
            if d is None:
                dp = torch.from_numpy(np.tile(p, (20, 1, 1))).cuda()          # 81 980 patches: the 8-wave form
                out = torch.empty((len(dp), 128), device="cuda")
                h.describe_patches_device(dp.data_ptr(), len(dp), out.data_ptr())
                h.synchronize()
                d = out[:len(p)].cpu().numpy()
                assert torch.equal(out[:len(p)], out[len(p):2 * len(p)])
            e = rel_l2(d, ref)
            if angle == lfp.ANGLE_EXACT_ZERO:       # (exact direction against the shader's polynomial: the same ~1e-5 rad everywhere)
                pass
            worst = max(worst, e.max())
            assert e.max() < GATE, (angle, cap, e.max(), int(e.argmax()))
            assert np.isfinite(d).all() and np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5
    print(f"fp6 cross-term mode: worst relative L2 over goldens + 4099 patches x 3 angle modes x 2 forms = {worst:.2e}")
    assert worst > 8e-6         # (if this ever fails the mode has become as good as the default: make it the default)
    assert worst < 4e-5         # the figure include/lf_mkd.h, README.md and NOTEBOOK.md section 11 quote: 2.95e-5 measured
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    img = smooth_image(120, 160, 3)
    kps = random_keypoints(50, 160, 120, 4)
    lf = lfp.LocalFeatures(160, 120, 64, pool_mode=lfp.POOL_F16_FP6)
    _, d = lf.describe(img, kps)
    assert rel_l2(d, oracle.describe_keypoints(img, kps)).max() < GATE


def test_random_patches_vs_oracle_and_raw(lfp, torch, oracle):
    rng = np.random.default_rng(0x4D4B44)
    n = 1000                                   # not a multiple of 16 or 64: ragged tail
    p = rng.random((n, 32, 32)).astype(np.float32)
    from oracle import ATAN_LIBM, ATAN_SHADER
    for (a, pm), h in _handles(lfp, max_features=256).items():   # 4 internal batches
        mode = ATAN_SHADER if a == lfp.ANGLE_SHADER else ATAN_LIBM
        ref, ref_raw = oracle.describe_patches(p, atan_mode=mode, nthreads=8, want_raw=True)
        d = h.describe_patches(p)
        assert assert_patch_parity(oracle, p, d, mode, what=(a, pm)) < 2e-5
        dp = torch.from_numpy(p).cuda()
        raw = torch.empty((n, 238), device="cuda")
        h.raw_descriptors_device(dp.data_ptr(), n, raw.data_ptr())
        h.synchronize()
        er = rel_l2(raw.cpu().numpy(), ref_raw)
        assert er.max() < 1e-5, (a, pm, er.max())
        out = torch.empty((n, 128), device="cuda")
        h.describe_patches_device(dp.data_ptr(), n, out.data_ptr())
        h.synchronize()
        assert np.array_equal(out.cpu().numpy(), d)     # host and device entry points agree bitwise


def test_pixels_on_the_atan2_discontinuity_follow_the_contracted_blur(lfp, oracle):
    """2^16 random patches hold ~100 with a pixel whose gx is 0 to the last bit (conftest.assert_patch_parity).  The
    kernel's blur is an fma chain in the shader's tap order, so it must take the same side of the discontinuity as
    the oracle's contracted blur on every one of them."""
    from oracle import ATAN_SHADER, BLUR_CONTRACT
    p = np.random.default_rng(99).random((1 << 16, 32, 32)).astype(np.float32)
    h = lfp.MkdHandle(max_features=1 << 15, pool_mode=lfp.POOL_F16X3)
    d = h.describe_patches(p)
    assert_patch_parity(oracle, p, d, ATAN_SHADER)
    q = oracle.quirk_pixels(p) > 0
    flip = rel_l2(oracle.describe_patches(p[q]), oracle.describe_patches(p[q], atan_mode=BLUR_CONTRACT))
    assert q.sum() >= 50 and (flip > 1e-3).sum() >= 5       # the test has teeth: the two roundings do disagree


def test_exact_zero_angle_mode_stays_inside_the_gate_on_every_patch(lfp, oracle):
    """LF_MKD_ANGLE_EXACT_ZERO: exact direction + the shader's angle 0 at gx == 0.  Against the SHADER reference it must
    meet the 1e-4 gate everywhere, the patches on the atan2 discontinuity included (plain EXACT misses there by 1e-2)."""
    from oracle import ATAN_SHADER, BLUR_CONTRACT
    rng = np.random.default_rng(123)
    p = rng.random((1 << 15, 32, 32)).astype(np.float32)
    half = rng.random((64, 32, 16)).astype(np.float32)      # mirrored about column 15: gx == 0 (or +-1 ulp) down a column
    sym = np.concatenate([half[:, :, :15], half[:, :, 15:16], half[:, :, 14::-1], half[:, :, :1]], axis=2)
    p = np.concatenate([p, sym])
    ref = oracle.describe_patches(p, atan_mode=ATAN_SHADER | BLUR_CONTRACT, nthreads=8)
    quirk = oracle.quirk_pixels(p) > 0
    assert quirk[-64:].all() and quirk[:-64].sum() >= 30
    for pool in (lfp.POOL_F16X3, lfp.POOL_F32):
        e = rel_l2(lfp.MkdHandle(max_features=1 << 14, angle_mode=lfp.ANGLE_EXACT_ZERO, pool_mode=pool).describe_patches(p), ref)
        assert e.max() < GATE and e.max() < 5e-5, (pool, e.max(), int(e.argmax()))
    plain = rel_l2(lfp.MkdHandle(max_features=1 << 14, angle_mode=lfp.ANGLE_EXACT).describe_patches(p[quirk]), ref[quirk])
    assert (plain > 1e-3).sum() >= 20          # what the zero convention is there for


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 129])
def test_ragged_batch_sizes(lfp, oracle, n):
    rng = np.random.default_rng(n)
    p = rng.random((n, 32, 32)).astype(np.float32)
    h = lfp.MkdHandle(max_features=64)
    d = h.describe_patches(p)
    assert rel_l2(d, oracle.describe_patches(p)).max() < 2e-5


def test_empty_and_errors(lfp):
    h = lfp.MkdHandle(max_features=64)
    assert h.describe_patches(np.zeros((0, 32, 32), np.float32)).shape == (0, 128)
    with pytest.raises(RuntimeError, match="set_image"):
        h.describe_keypoints(np.zeros((1, 5), np.float32))
    h2 = lfp.MkdHandle(max_features=64, max_image_width=64, max_image_height=64)
    with pytest.raises(RuntimeError, match="exceeds"):
        h2.set_image(np.zeros((65, 64), np.float32))


def test_structured_edge_cases(lfp, oracle):
    """flat patches (zero gradient everywhere), axis-aligned structure (the gx == 0 atan2 quirk),
    saturated values."""
    y, x = np.mgrid[0:32, 0:32].astype(np.float32)
    p = np.stack([np.full((32, 32), 0.2, np.float32), np.full((32, 32), 0.9, np.float32),
                  ((x // 4 + y // 4) % 2).astype(np.float32), (x > 15).astype(np.float32),
                  (y > 15).astype(np.float32), x / 31.0, y / 31.0, np.zeros((32, 32), np.float32),
                  np.ones((32, 32), np.float32)])
    h = lfp.MkdHandle(max_features=64)
    d = h.describe_patches(p)
    ref = oracle.describe_patches(p)
    assert np.all(np.isfinite(d))
    assert rel_l2(d, ref).max() < 2e-5
    assert np.array_equal(d[0], d[1])       # flat patch -> one fixed descriptor


def test_keypoint_mode_goldens(lfp, torch):
    g = golden("keypoints_liberty.npz")
    img, kps = g["image"], g["keypoints"]
    hgt, w = img.shape
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    assert np.abs(h.pyramid_level(1) - g["level1"]).max() < 1e-6
    assert np.abs(h.pyramid_level(3) - g["level3"]).max() < 1e-6
    k5 = np.concatenate([kps, np.zeros((len(kps), 1), np.float32)], axis=1)
    dk = torch.from_numpy(k5).cuda()
    patches = torch.empty((len(kps), 32, 32), device="cuda")
    h.sample_patches_device(dk.data_ptr(), len(kps), patches.data_ptr())
    h.synchronize()
    assert np.abs(patches.cpu().numpy() - g["patches"]).max() < 1e-5
    d = h.describe_keypoints(k5)
    assert rel_l2(d, g["desc_shader"]).max() < GATE


def test_keypoint_mode_vs_oracle_larger_frame(lfp, oracle):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt = 640, 480
    img = smooth_image(hgt, w, 21)
    kps = random_keypoints(300, w, hgt, 22)
    k5 = np.concatenate([kps, np.zeros((len(kps), 1), np.float32)], axis=1)
    lf = lfp.LocalFeatures(w, hgt, 128)
    _, d = lf.describe(img, k5)
    assert_keypoint_parity(oracle, lf._inner, img, k5, d)


def test_pyramid_levels_are_bit_exact(lfp, oracle):
    """Every level of the patch pyramid equals the oracle's bit for bit (blur, a-trous, blits and the binomial
    decimation are evaluated in the shader's operation order, without fma contraction)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import smooth_image
    # (the shapes after 1920 x 1080 end in partial tiles of the staged kernels)
    for w, hgt in ((640, 480), (333, 257), (97, 64), (1920, 1080), (256, 98), (500, 26), (1000, 50), (252, 24)):
        img = np.ascontiguousarray(smooth_image(hgt, w, w), np.float32)
        h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
        h.set_image(img)
        pyr = oracle.split_pyramid(oracle.build_pyramid(img), w, hgt)
        for l, lv in enumerate(pyr):
            assert np.array_equal(h.pyramid_level(l), lv), (w, hgt, l)


def test_the_staged_decimation_and_8_bit_frames_give_the_bits_of_the_kernels_they_replace(lfp, torch, monkeypatch):
    """Round 5: the decimated pyramid levels are built with their source rows staged in LDS (pyr_down_staged) and must have
    the bits of the gathering kernel they replace (LF_MKD_NO_DOWN_STAGED in the environment brings it back), aprons included,
    for single frames and batches, f32 and 8-bit frames, tiles cut by the levels' edges."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import smooth_image
    for w, hgt, frames in ((1920, 1080, 1), (640, 480, 9), (256, 98, 2), (500, 26, 8), (1000, 50, 1), (4096, 64, 1), (252, 2000, 1)):
        img = np.ascontiguousarray(smooth_image(hgt, w, w + 3), np.float32)
        u8 = np.ascontiguousarray(np.rint(img * 255).astype(np.uint8))
        stack32 = np.stack([img] + [np.ascontiguousarray(img[::-1])] * (frames - 1))
        stack8 = np.stack([u8] + [np.ascontiguousarray(u8[::-1])] * (frames - 1))
        got = {}
        for form in ("new", "old"):
            if form == "old":
                monkeypatch.setenv("LF_MKD_NO_DOWN_STAGED", "1")
            else:
                monkeypatch.delenv("LF_MKD_NO_DOWN_STAGED", raising=False)
            for px, stack in (("f32", stack32), ("u8", stack8)):
                h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt, max_frames=frames)
                d = torch.from_numpy(stack).cuda()
                if px == "f32":
                    h.set_images_device(d.data_ptr(), frames, w, hgt)
                else:
                    h.set_images_u8_device(d.data_ptr(), frames, w, hgt)
                h.synchronize()
                lv, l = [], 0
                while True:
                    try:
                        lv.append(h.pyramid_level_apron(l)[0])
                    except RuntimeError:
                        break
                    l += 1
                # a keypoint of the LAST frame sees that frame's pyramid: describe a few there too
                k = np.array([[w / 2, hgt / 2, 2.0 + i, 30.0 * i, 0] for i in range(8)], np.float32)
                dk, df = torch.from_numpy(k).cuda(), torch.full((8,), frames - 1, dtype=torch.int32, device="cuda")
                out = torch.empty((8, 128), device="cuda")
                h.describe_keypoints_frames_device(dk.data_ptr(), df.data_ptr(), 8, out.data_ptr())
                h.synchronize()
                got[form, px] = (lv, out.cpu().numpy())
        for px in ("f32", "u8"):
            a, b = got["new", px], got["old", px]
            assert len(a[0]) == len(b[0]) >= 4
            for l, (x, y) in enumerate(zip(a[0], b[0])):
                assert np.array_equal(x, y), (w, hgt, frames, px, l)
            assert np.array_equal(a[1], b[1]), (w, hgt, frames, px)
        # the 8-bit frame stands for u8 / 255: its pyramid is that f32 frame's
        h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
        h.set_image(u8.astype(np.float32) / np.float32(255))
        for l in range(len(got["new", "u8"][0])):
            assert np.array_equal(h.pyramid_level_apron(l)[0], got["new", "u8"][0][l]), (w, hgt, l)


def test_full_size_properties(lfp, torch):
    """BASELINE-sized batch (2^18 patches = 1 GiB) checked through size-independent properties:
    unit norm, equality with the small-batch path on a subset, and invariance to batch layout."""
    n = 1 << 18
    gen = torch.Generator(device="cuda").manual_seed(0x4D4B44)
    p = torch.rand((n, 32, 32), device="cuda", generator=gen)
    out = torch.empty((n, 128), device="cuda")
    h = lfp.MkdHandle(max_features=1 << 16)
    h.describe_patches_device(p.data_ptr(), n, out.data_ptr())
    h.synchronize()
    nrm = out.norm(dim=1)
    assert torch.isfinite(out).all()
    assert (nrm - 1).abs().max().item() < 1e-5
    idx = torch.randint(0, n, (512,), device="cuda", generator=gen)
    sub = p[idx].contiguous()
    out2 = torch.empty((512, 128), device="cuda")
    h.describe_patches_device(sub.data_ptr(), 512, out2.data_ptr())
    h.synchronize()
    assert torch.equal(out2, out[idx])      # a descriptor depends on its own patch only


def test_exact_angle_mode_stays_inside_the_gate_of_the_shader_reference(lfp, oracle):
    """ANGLE_EXACT replaces the shader's polynomial atan2 (max error 1e-5 rad) by the exact direction;
    its descriptors must still match the shader-faithful oracle within the north-star tolerance."""
    rng = np.random.default_rng(77)
    p = rng.random((512, 32, 32)).astype(np.float32)
    ref = oracle.describe_patches(p, nthreads=8)          # shader-faithful
    for pool in (lfp.POOL_F32, lfp.POOL_F16X3):
        d = lfp.MkdHandle(max_features=512, angle_mode=lfp.ANGLE_EXACT, pool_mode=pool).describe_patches(p)
        e = rel_l2(d, ref)
        assert e.max() < GATE, (pool, e.max())
        assert e.max() < 5e-5, (pool, e.max())


def test_f16x3_pooling_agrees_with_f32_at_full_size_and_is_deterministic(lfp, torch):
    """2^18 patches: the split-f16 MFMA pooling against the exact-f32 MFMA pooling, per descriptor,
    and two runs of each against each other bit for bit (no data race in the LDS pipeline)."""
    n = 1 << 18
    gen = torch.Generator(device="cuda").manual_seed(123)
    p = torch.rand((n, 32, 32), device="cuda", generator=gen)
    outs = {}
    for angle in (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT):
        for pool in (lfp.POOL_F32, lfp.POOL_F16X3):
            h = lfp.MkdHandle(max_features=n, angle_mode=angle, pool_mode=pool)
            a = torch.empty((n, 128), device="cuda")
            b = torch.empty((n, 128), device="cuda")
            h.describe_patches_device(p.data_ptr(), n, a.data_ptr())
            h.describe_patches_device(p.data_ptr(), n, b.data_ptr())
            h.synchronize()
            assert torch.equal(a, b), (angle, pool)
            outs[(angle, pool)] = a
    for angle in (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT):
        d = outs[(angle, lfp.POOL_F16X3)] - outs[(angle, lfp.POOL_F32)]
        e = d.norm(dim=1) / outs[(angle, lfp.POOL_F32)].norm(dim=1)
        assert e.max().item() < 2e-5, (angle, e.max().item())


def test_multi_frame_entry_points_match_single_frame_and_oracle(lfp, torch, oracle):
    """configs[2]-style batch: several frames of one size, keypoints tagged with their frame."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, nf = 160, 120, 3
    frames = np.ascontiguousarray(np.stack([smooth_image(hgt, w, 40 + f) for f in range(nf)]))   # device API: contiguous
    counts = [37, 0, 90]
    kps = [np.concatenate([random_keypoints(c, w, hgt, 50 + f), np.zeros((c, 1), np.float32)], axis=1)
           for f, c in enumerate(counts)]
    allk = np.concatenate(kps).astype(np.float32)
    fid = np.concatenate([np.full(c, f, np.uint32) for f, c in enumerate(counts)])
    n = len(allk)
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt, max_frames=nf)
    d_frames = torch.from_numpy(frames).cuda().contiguous()
    d_k, d_f = torch.from_numpy(allk).cuda(), torch.from_numpy(fid.astype(np.int32)).cuda()
    out = torch.empty((n, 128), device="cuda")
    h.set_images_device(d_frames.data_ptr(), nf, w, hgt)
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, out.data_ptr())
    h.synchronize()
    got = out.cpu().numpy()
    single = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    ref_parts, one_parts = [], []
    for f in range(nf):
        if counts[f] == 0:
            continue
        single.set_image(frames[f])
        one_parts.append(single.describe_keypoints(kps[f]))
        ref_parts.append(oracle.describe_keypoints(frames[f], kps[f][:, :4]))
    assert np.array_equal(got, np.concatenate(one_parts))
    assert rel_l2(got, np.concatenate(ref_parts)).max() < GATE
    with pytest.raises(RuntimeError, match="max_frames"):
        single.set_images_device(d_frames.data_ptr(), nf, w, hgt)


def test_frame_indices_beyond_the_store_are_clamped(lfp, torch):
    """frame_of_kp is caller's data: an index past the loaded frames reads the last frame's pyramid, never outside the store
    (both forms of keypoint mode)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, nf = 160, 120, 2
    frames = np.ascontiguousarray(np.stack([smooth_image(hgt, w, 60 + f) for f in range(nf)]))
    k5 = np.concatenate([random_keypoints(100, w, hgt, 61), np.zeros((100, 1), np.float32)], axis=1).astype(np.float32)
    d_frames, d_k = torch.from_numpy(frames).cuda(), torch.from_numpy(k5).cuda()
    last = torch.full((100,), nf - 1, dtype=torch.int32, device="cuda")
    wild = last.clone()
    wild[::3] = 99
    wild[1::3] = 0x7FFFFFFF
    for flags in (0, lfp.FLAG_UNFUSED_KEYPOINTS):
        h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt, max_frames=nf, flags=flags)
        h.set_images_device(d_frames.data_ptr(), nf, w, hgt)
        a, b = torch.empty((100, 128), device="cuda"), torch.empty((100, 128), device="cuda")
        h.describe_keypoints_frames_device(d_k.data_ptr(), last.data_ptr(), 100, a.data_ptr())
        h.describe_keypoints_frames_device(d_k.data_ptr(), wild.data_ptr(), 100, b.data_ptr())
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())


def test_keypoint_edge_cases_vs_oracle(lfp, oracle):
    """Keypoints whose sampling window leaves the image (MirroredRepeat), the smallest and largest sizes
    the detector emits, sizes below and above the pyramid range (level clamp), angles 0 / 360 / negative."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import smooth_image
    w, hgt = 640, 480     # large enough that the coarse levels still carry structure (a 6x4-texel level
    img = smooth_image(hgt, w, 9)   # gives near-flat patches whose descriptors are rounding noise in any f32 code)
    kps = np.array([
        [0.0, 0.0, 4.0, 0.0],          # corner: three quarters of the window mirrored
        [639.0, 479.0, 10.0, 45.0],    # opposite corner, rotated
        [320.0, 2.0, 30.0, 200.0],     # top edge, coarse level
        [3.5, 70.25, 1.64, 359.99],    # smallest detector size
        [320.0, 240.0, 52.0, 360.0],   # largest detector size, angle == 360
        [50.0, 50.0, 0.5, 10.0],       # below the pyramid range: level clamped to 0
        [150.0, 90.0, 160.0, -30.0],   # above the detector's range, negative angle
        [10.0, 400.0, 2.6667, 90.0],   # scale exactly 2.0 -> level boundary
    ], np.float32)
    k5 = np.concatenate([kps, np.zeros((len(kps), 1), np.float32)], axis=1)
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    d = h.describe_keypoints(k5)
    ref = oracle.describe_keypoints(img, kps)
    assert np.all(np.isfinite(d))
    assert rel_l2(d, ref).max() < GATE


def test_non_finite_keypoints_do_not_reach_outside_the_pyramid(lfp):
    """Caller-supplied keypoints are data: NaN / inf / negative sizes and positions must at worst give NaN descriptors for
    those rows, never an out-of-range level index (the call returns and the other rows are untouched)."""
    w, hgt = 160, 120
    img = np.random.default_rng(8).random((hgt, w)).astype(np.float32)
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    good = np.array([[80.0, 60.0, 4.0, 30.0, 0.0]], np.float32)
    bad = np.array([[80, 60, np.nan, 0, 0], [80, 60, np.inf, 0, 0], [80, 60, -3.0, 0, 0], [80, 60, 0.0, 0, 0],
                    [np.nan, 60, 4, 0, 0], [80, np.inf, 4, 0, 0], [80, 60, 4, np.nan, 0], [1e30, -1e30, 1e30, 1e30, 0]],
                   np.float32)
    d = h.describe_keypoints(np.concatenate([good, bad, good]))
    assert d.shape == (10, 128)
    assert np.array_equal(d[0], d[9]) and np.isfinite(d[0]).all() and abs(np.linalg.norm(d[0]) - 1) < 1e-5
    assert np.array_equal(d[0], h.describe_keypoints(good)[0])


def test_non_square_and_odd_sized_frames(lfp, oracle):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    for (w, hgt) in ((799, 533), (65, 301)):       # bird.jpg's size; a tall odd frame
        img = smooth_image(hgt, w, w)
        kps = random_keypoints(40, w, hgt, hgt)
        k5 = np.concatenate([kps, np.zeros((len(kps), 1), np.float32)], axis=1)
        h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
        h.set_image(img)
        # pyramid levels of odd sizes: floor halving, as patch_pyramid.rs:179-180
        pyr = oracle.split_pyramid(oracle.build_pyramid(img), w, hgt)
        for l in (1, 2, 4):
            assert np.abs(h.pyramid_level(l) - pyr[l]).max() < 2e-6
        assert rel_l2(h.describe_keypoints(k5), oracle.describe_keypoints(img, kps)).max() < GATE


def test_smaller_image_than_max_and_reuse_of_a_handle(lfp, oracle):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    h = lfp.MkdHandle(max_features=64, max_image_width=320, max_image_height=240)
    for (w, hgt, seed) in ((320, 240, 1), (160, 120, 2), (320, 240, 3)):   # image < max: mirror at the content edge
        img = smooth_image(hgt, w, seed)
        kps = random_keypoints(30, w, hgt, seed + 10)
        k5 = np.concatenate([kps, np.zeros((len(kps), 1), np.float32)], axis=1)
        h.set_image(img)
        assert rel_l2(h.describe_keypoints(k5), oracle.describe_keypoints(img, kps)).max() < GATE


def test_extreme_patch_values(lfp, oracle):
    """Patches far outside [0,1], small contrasts, one hot pixel.  (A contrast of 1e-5 on a 0.5 pedestal is below
    what f32 pixels resolve -- any f32 implementation, the oracle included, returns rounding noise there -- so that
    case is only checked for finiteness and unit norm.)"""
    rng = np.random.default_rng(99)
    base = rng.random((6, 32, 32)).astype(np.float32)
    p = np.stack([base[0] * 255.0, base[1] * 1e-3, 0.5 + base[2] * 0.02, -base[3], base[4] * 1e4,
                  np.zeros((32, 32), np.float32), 0.5 + base[5] * 1e-5])
    p[5, 13, 17] = 1.0
    for pool in (lfp.POOL_F32, lfp.POOL_F16X3):
        d = lfp.MkdHandle(max_features=64, pool_mode=pool).describe_patches(p)
        ref = oracle.describe_patches(p)
        assert np.all(np.isfinite(d))
        assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
        assert rel_l2(d[:6], ref[:6]).max() < GATE, (pool, rel_l2(d, ref))


def test_negative_zero_pixels_take_the_shaders_angle(lfp, oracle):
    """gx = left - right is -0.0 exactly when the blurred value on the left is -0.0 and the one on the right +0.0: patches that
    hold negative zeros (or negative values that underflow in the blur).  The shader's atan2 treats x = -0.0 as x = 0: angle 0,
    no pi branch (atan2.glsl:29-45; `sign(-0.0)` is not -1).  The kernel takes cos's sign from gx's sign bit, so it must never
    see the sign of a zero: its blur ends in +0.0 wherever the value is zero (round 5; the advisor's round-4 finding).  Patches:
    a block of -0.0 beside a block of +0.0 above a textured half, the same transposed, negative values small enough to
    underflow, and all of them scattered into random patches."""
    from oracle import BLUR_CONTRACT, ATAN_SHADER
    rng = np.random.default_rng(41)
    tex = rng.random((32, 32)).astype(np.float32)
    a = tex.copy()
    a[:20, :16] = -0.0
    a[:20, 16:] = 0.0
    b = np.ascontiguousarray(a.T)
    c = tex.copy()
    c[:18, :14] = -1e-44          # negative denormals: the blur's products round to -0.0
    c[:18, 14:] = 0.0
    d = rng.random((32, 32)).astype(np.float32)
    d[4:16, 3:12] = -0.0
    d[4:16, 12:24] = 0.0
    e = np.where(rng.random((32, 32)) < 0.5, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
    e[20:, :] = tex[20:, :]
    # The advisor's round-5 case: negative denormals whose products underflow to -0.0 whether the tap is a multiply or an fma.
    # With the kernel's arithmetic (worked through in an f64 emulation of its two passes) these three column patterns leave
    # out[x-1] = -0.0 beside out[x+1] = +0.0, i.e. gx = -0.0, on every row of the zeroed block, at x = 0, 10 and 20:
    #   column 2 = -1.4e-45 (the left border: taps 0..2 of x = 0 sit on the replicated column 0);
    #   column 10 = -1.4e-45 beside column 11 = -0.0;  column 20 = -2.8e-45 beside column 21 = -0.0
    # (the vertical pass keeps a column of -0.0 as -0.0 -- its first tap is a multiply -- where the oracle's fma onto +0.0 gives
    # +0.0: zeros of either sign must describe the same).
    f = tex.copy()
    f[:22, :] = 0.0
    f[:22, 2] = -1.4e-45
    f[:22, 10], f[:22, 11] = -1.4e-45, -0.0
    f[:22, 20], f[:22, 21] = -2.8e-45, -0.0
    g = np.ascontiguousarray(f.T)
    p = np.stack([a, b, c, d, e, f, g]).astype(np.float32)
    assert (p[5:] < 0).any() and np.float32(0.2054) * p[5:].min() == 0      # (and numpy kept the denormals)
    # (Not covered, by design: gradients whose larger component is a DENORMAL below 1e-30 -- the kernel's direction is then
    #  a = min / max(larger, 1e-30), the shader's min / larger; such pixels need pixel values of ~1e-30 and do not occur in
    #  frames of [0, 1]; lf_mkd.h states the domain.)
    assert np.signbit(p).any()
    ref = oracle.describe_patches(p, atan_mode=ATAN_SHADER | BLUR_CONTRACT)
    for pool in (lfp.POOL_F32, lfp.POOL_F16X3):
        got = lfp.MkdHandle(max_features=64, pool_mode=pool).describe_patches(p)
        err = rel_l2(got, ref)
        print(f"negative-zero patches, pool mode {pool}: relative L2 vs the oracle {err}")
        assert err.max() < GATE, (pool, err)
        # the same patches with every -0.0 replaced by +0.0 describe the same (zeros carry no sign into the descriptor)
        q = np.where(p == 0, np.float32(0.0), p).astype(np.float32)
        q[2][q[2] < 0] = 0.0
        assert rel_l2(lfp.MkdHandle(max_features=64, pool_mode=pool).describe_patches(q)[[0, 1, 3, 4]], got[[0, 1, 3, 4]]).max() < 2e-5


def _adversarial_keypoints(w, h):
    """Interior keypoints, keypoints on and next to the frame's borders, footprints larger than the frame's levels, sizes
    below level 0, non-finite ones."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints
    g = np.random.default_rng(5)
    k = [random_keypoints(3000, w, h, 1, margin=0.0)]                       # anywhere, border included
    edge = random_keypoints(600, w, h, 2, margin=0.0)
    edge[:, 0] = g.choice([0.0, 0.4, w - 1.0, w - 0.3, w / 2], 600)         # on and next to the vertical borders
    edge[:300, 1] = g.choice([0.0, h - 1.0, h - 0.2], 300)
    k.append(edge)
    big = random_keypoints(200, w, h, 3, margin=0.0)
    big[:, 2] = g.uniform(40.0, 400.0, 200)                                 # footprints larger than the frame's levels
    k.append(big)
    tiny = random_keypoints(200, w, h, 4, margin=0.0)
    tiny[:, 2] = g.uniform(0.01, 1.4, 200)                                  # below level 0
    k.append(tiny)
    outside = random_keypoints(100, w, h, 7, margin=0.0)
    outside[:, 0] += g.choice([-3.0 * w, -40.0, 55.0 + w, 2.5 * w], 100)    # centres outside the frame: no apron reaches
    k.append(outside)
    odd = random_keypoints(8, w, h, 6, margin=0.0)
    odd[0, 2], odd[1, 2], odd[2, 0], odd[3, 3], odd[4, 2], odd[5, 1] = np.nan, np.inf, np.nan, np.inf, 0.0, -1e30
    k.append(odd)
    k = np.concatenate(k).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))


def test_pyramid_apron_is_mirrored_repeat(lfp, torch):
    """Every level of the patch pyramid carries an apron of 48 texels holding what MirroredRepeat addressing would fetch there
    (several mirror periods on the small levels).  The kernels that produce a level write its apron with it; the shapes cover
    each writer: levels at least as large as the apron (one reflection, written with the texels), the small end of the
    pyramid (pyr_tail), levels narrower than the apron but too large for pyr_tail (the stand-alone fill: the two slim
    shapes), and a batch of frames (its own level-1 kernel)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import smooth_image
    for w, hgt, frames in ((640, 480, 1), (333, 257, 1), (40, 36, 1), (1920, 1080, 1), (1500, 44, 1), (100, 700, 1),
                           (640, 480, 8), (1500, 44, 8), (640, 480, 3), (334, 258, 8)):
        img = np.ascontiguousarray(smooth_image(hgt, w, w + 1), np.float32)
        h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt, max_frames=frames)
        if frames == 1:
            h.set_image(img)
        else:   # the level readers return frame 0 of the batch
            d = torch.from_numpy(np.stack([img] + [img[::-1].copy()] * (frames - 1))).cuda()
            h.set_images_device(d.data_ptr(), frames, w, hgt)
            h.synchronize()
        l = 0
        while True:
            try:
                lvl = h.pyramid_level(l)
            except RuntimeError:
                break
            padded, a = h.pyramid_level_apron(l)
            assert a == 48
            assert np.array_equal(padded, np.pad(lvl, a, mode="symmetric")), (w, hgt, frames, l)
            l += 1
        assert l >= 4
        if frames > 1:   # the batch's pyramid of frame 0 is the single frame's
            h1 = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
            h1.set_image(img)
            for k in range(l):
                assert np.array_equal(h.pyramid_level_apron(k)[0], h1.pyramid_level_apron(k)[0]), k
        else:            # and so is the pyramid of a handle whose detector shares a-trous layer 1 with it (level 1 is then
            before = [h.pyramid_level_apron(k)[0] for k in range(l)]   # stored by the layer's kernel, or blitted from it)
            h.detect_extrema()
            h.set_image(img)
            for k in range(l):
                assert np.array_equal(h.pyramid_level_apron(k)[0], before[k]), (w, hgt, k)


def test_the_fused_keypoint_kernel_describes_exactly_what_the_sampler_samples(lfp, torch, oracle):
    """Keypoint mode is one launch: producer waves of the describe kernel sample the patches into its LDS ring.  The
    two-launch form (LF_MKD_FLAG_UNFUSED_KEYPOINTS: sampler -> patches in HBM -> patch kernel) and the verification tap
    lf_mkd_sample_patches_device use the same sampling arithmetic (csrc/mkd_sample.h), so all three agree BIT FOR BIT -- on
    interior keypoints, footprints over the level's edge (apron), footprints the apron does not cover (MirroredRepeat per
    tap), sizes beyond both ends of the pyramid, and non-finite keypoints (finite or not, they agree and nothing faults)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import smooth_image
    for (w, hgt) in ((333, 257), (1920, 1080), (40, 36)):
        img = np.ascontiguousarray(smooth_image(hgt, w, 9), np.float32)
        k5 = _adversarial_keypoints(w, hgt)
        n = len(k5)
        d_k = torch.from_numpy(k5).cuda()
        outs = []
        for flags in (0, lfp.FLAG_UNFUSED_KEYPOINTS):
            h = lfp.MkdHandle(max_features=1024, max_image_width=w, max_image_height=hgt, flags=flags)   # several internal batches
            h.set_image(img)
            out = torch.full((n, 128), -7.0, device="cuda")
            with kp_form(1):        # the whole-patch form: one chain of row sums, like the patch kernel the two-launch form uses
                h.describe_keypoints_device(d_k.data_ptr(), n, out.data_ptr())
            outs.append(out.cpu().numpy())
            if flags == 0:          # ... and the form a request of this size takes by itself: the same keypoints, other rounding
                out2 = torch.full((n, 128), -7.0, device="cuda")
                h.describe_keypoints_device(d_k.data_ptr(), n, out2.data_ptr())
                split = out2.cpu().numpy()
        fused, unfused = outs
        fin = np.isfinite(fused).all(axis=1)
        assert np.array_equal(fin, np.isfinite(unfused).all(axis=1)), (w, hgt)
        assert fin[:4100].all() and fin.sum() >= n - 8
        assert np.array_equal(fused[fin], unfused[fin]), (w, hgt, np.abs(fused[fin] - unfused[fin]).max())
        assert np.array_equal(fin, np.isfinite(split).all(axis=1))
        assert_same_descriptors(split[fin], fused[fin], f"adversarial keypoints {w}x{hgt}: row-split vs whole-patch form")
        assert (fused[fin] != -7.0).any(axis=1).all()                      # every row written
        # the tap, described by the patch kernel, is the fused kernel's answer too
        d_p = torch.full((n, 32, 32), -7.0, device="cuda")
        h.sample_patches_device(d_k.data_ptr(), n, d_p.data_ptr())
        p = d_p.cpu().numpy()
        assert (p[fin] != -7.0).all()                                       # every pixel written
        hp = lfp.MkdHandle(max_features=n)
        assert np.array_equal(hp.describe_patches(p[fin]), fused[fin])
        # and the sampled values are the oracle's, to the rounding of the sample coordinates (different libm)
        ok = np.where(fin)[0][::7]
        ref_p = oracle.sample_patches(oracle.build_pyramid(img), w, hgt, k5[ok, :4])
        tol = 6e-5 if w > 1000 else 1e-5    # (coordinates up to 1920: one ulp there is 1.2e-4 texel)
        assert np.abs(p[ok] - ref_p).max() < tol, (w, hgt, np.abs(p[ok] - ref_p).max())


def test_row_split_forms_of_the_keypoint_kernel(lfp, torch, oracle):
    """Round 6: requests of at most 4096 keypoints are described by R = 2 or 4 workgroups per batch of 32 keypoints, each
    pooling its share of the patch rows (partial sums through global memory, added in a fixed order; mkd_describe.hip).  Every
    form is held to the oracle directly; forms agree with each other to the rounding of their sums; a form is deterministic
    (repeated launches, the consumer's counters return to zero); frame ids, batch tails and the sizes at which the default
    form changes are covered; a form that does not fit (more workgroups than CUs) is not taken."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from gen_golden import random_keypoints, smooth_image
    w, hgt, nf = 512, 384, 3
    frames = np.ascontiguousarray(np.stack([smooth_image(hgt, w, 130 + f) for f in range(nf)]))
    d_frames = torch.from_numpy(frames).cuda()
    h = lfp.MkdHandle(max_features=8192, max_image_width=w, max_image_height=hgt, max_frames=nf)
    h.set_images_device(d_frames.data_ptr(), nf, w, hgt)
    rng = np.random.default_rng(8)
    for n in (1, 31, 33, 500, 2047, 2049, 4096, 4097):
        k = random_keypoints(n, w, hgt, 140 + n, margin=0.0)
        k5 = np.ascontiguousarray(np.concatenate([k, np.zeros((n, 1), np.float32)], axis=1))
        fid = np.ascontiguousarray(rng.integers(0, nf, n).astype(np.int32))
        d_k, d_f = torch.from_numpy(k5).cuda(), torch.from_numpy(fid).cuda()
        out = {}
        for form in ("1", "2", "4", None):
            o = torch.full((n, 128), float("nan"), device="cuda")
            runs = []
            for _ in range(3):
                if form is None:
                    h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, o.data_ptr())
                else:
                    with kp_form(form):
                        h.describe_keypoints_frames_device(d_k.data_ptr(), d_f.data_ptr(), n, o.data_ptr())
                runs.append(o.cpu().numpy().copy())
            assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2]), (n, form)
            assert np.isfinite(runs[0]).all(), (n, form)
            out[form] = runs[0]
        for form in ("2", "4", None):
            assert_same_descriptors(out[form], out["1"], f"n = {n}: form {form or 'default'} vs the whole-patch form")
        fits4, fits2 = (n + 31) // 32 <= 64, (n + 31) // 32 <= 128            # 8 x the batches' multiple of 8 <= 256 CUs
        assert np.array_equal(out[None], out["4" if fits4 else ("2" if fits2 else "1")]), n       # the default's choice
        if not fits4:
            assert np.array_equal(out["4"], out["2" if fits2 else "1"]), n     # a form that does not fit is not forced
        if not fits2:
            assert np.array_equal(out["2"], out["1"]), n
    # the other angle modes are separate instantiations of the row-split kernel: same agreement with their whole-patch forms
    for angle in (lfp.ANGLE_EXACT, lfp.ANGLE_EXACT_ZERO):
        ha = lfp.MkdHandle(max_features=4096, max_image_width=w, max_image_height=hgt, angle_mode=angle)
        ha.set_image(frames[1])
        k = random_keypoints(1000, w, hgt, 160 + angle, margin=0.0)
        k5 = np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))
        got = {}
        for form in ("1", "2", "4"):
            with kp_form(form):
                got[form] = ha.describe_keypoints(k5)
            assert np.isfinite(got[form]).all()
        for form in ("2", "4"):
            assert_same_descriptors(got[form], got["1"], f"angle mode {angle}: form {form} vs the whole-patch form")
    # each row-split form against the oracle, end to end (frame 0; the helper needs the handle to hold one frame)
    one = lfp.MkdHandle(max_features=2048, max_image_width=w, max_image_height=hgt)
    one.set_image(frames[0])
    k = random_keypoints(700, w, hgt, 150)
    k5 = np.ascontiguousarray(np.concatenate([k, np.zeros((len(k), 1), np.float32)], axis=1))
    for form in ("2", "4"):
        with kp_form(form):
            d = one.describe_keypoints(k5)
        assert_keypoint_parity(oracle, one, frames[0], k5, d, what=f"row-split form R = {form}")
