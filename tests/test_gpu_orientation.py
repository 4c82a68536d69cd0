"""GPU parity tests of keypoint orientation (keypoint_orientation.glsl:36-171 + the a-trous stack,
mod.rs:1093-1130) through the C ABI.  The output of this stage is discrete (which histogram bins are peaks) plus
one interpolated angle per peak: the tests demand the same keypoint list as the oracle, x/y/size/response
bit-equal and the angle within 1e-3 degrees."""
import numpy as np
import pytest

from conftest import assert_keypoint_parity, golden, rel_l2

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


def smooth_image(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w))
    for _ in range(3):
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5
    img = (img - img.min()) / (img.max() - img.min())
    return np.ascontiguousarray(img, np.float32)


def random_extrema(n, w, h, seed, n_scales=4, border=0.0):
    """Sizes as the detector emits them, 0.82 sqrt2 2^z (scan_extrema.glsl:229); positions may touch the border."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(1.0, n_scales + 0.45, n)
    return np.stack([rng.uniform(border, w - border, n), rng.uniform(border, h - border, n),
                     0.82 * np.sqrt(2.0) * 2.0 ** z, rng.uniform(0.04, 0.4, n)], axis=1).astype(np.float32)


def assert_same_keypoints(got, want, what=""):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(got[:, [0, 1, 2, 4]], want[:, [0, 1, 2, 4]]), what
    d = np.abs(got[:, 3] - want[:, 3])
    assert np.minimum(d, 360 - d).max(initial=0.0) < 1e-3, (what, d.max())


def test_orientation_goldens(lfp):
    g = golden("orientation.npz")
    img = g["image"]
    hgt, w = img.shape
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    assert np.abs(h.coarse_layer(2, w, hgt) - g["layer2"]).max() < 1e-6
    assert np.abs(h.coarse_layer(5, w, hgt) - g["layer5"]).max() < 1e-6
    k, dropped = h.orient_keypoints(g["extrema"])
    assert dropped == 0
    assert_same_keypoints(k, g["keypoints"], "golden")


@pytest.mark.parametrize("w,hgt,n,n_scales", [(640, 480, 3000, 4), (333, 257, 700, 3), (97, 64, 300, 5)])
def test_orientation_vs_oracle(lfp, oracle, w, hgt, n, n_scales):
    img = smooth_image(w, hgt, w)
    ex = random_extrema(n, w, hgt, n, n_scales)          # positions up to the very edge: windows leave the image
    h = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt, n_scales=n_scales)
    h.set_image(img)
    st = oracle.build_coarse_stack(img, n_scales)
    for l in range(n_scales + 3):   # the stack is bit-exact (contraction is off in the stack kernels, as in the oracle)
        assert np.array_equal(h.coarse_layer(l, w, hgt), st[l]), l
    want = oracle.orient(st, ex)
    assert len(want) > n                                  # several peaks per extremum do occur
    got, dropped = h.orient_keypoints(ex)
    assert dropped == 0
    assert_same_keypoints(got, want, (w, hgt))
    # truncation: the first max_out keypoints of the same ordered list, the rest counted
    cut, dropped = h.orient_keypoints(ex, max_out=100)
    assert len(cut) == 100 and dropped == len(want) - 100
    assert np.array_equal(cut, got[:100])


def test_orientation_edge_cases(lfp, oracle):
    w, hgt = 160, 120
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    with pytest.raises(RuntimeError, match="set_image"):
        h.orient_keypoints(np.zeros((1, 4), np.float32))
    # a ramp has one gradient direction -> one peak at 360 - direction; the shader's atan2 returns 0 for
    # exactly vertical gradients (atan2.glsl:33-38), a quirk the port keeps
    yy, xx = np.mgrid[0:hgt, 0:w].astype(np.float32)
    ex = np.array([[80.3, 60.7, s, 0.1] for s in (2.5, 4.0)], np.float32)
    for deg in (0.0, 40.0, 90.0, 200.0, 310.0):
        t = np.deg2rad(deg)
        img = (0.5 + 0.002 * (np.cos(t) * (xx - w / 2) - np.sin(t) * (yy - hgt / 2))).astype(np.float32)
        h.set_image(img)
        k, _ = h.orient_keypoints(ex)
        assert_same_keypoints(k, oracle.orient(oracle.build_coarse_stack(img), ex), deg)
    # flat image: empty histogram, no keypoints; empty input: no keypoints
    h.set_image(np.full((hgt, w), 0.25, np.float32))
    k, dropped = h.orient_keypoints(ex)
    assert k.shape == (0, 5) and dropped == 0
    k, dropped = h.orient_keypoints(np.zeros((0, 4), np.float32))
    assert k.shape == (0, 5) and dropped == 0
    # sizes outside the detector's range clamp to the first / last layer; corners and off-by-one positions
    img = smooth_image(w, hgt, 5)
    h.set_image(img)
    ex = np.array([[0.0, 0.0, 2.3, 0.1], [w - 0.01, hgt - 0.01, 2.3, 0.1], [0.5, hgt - 1.0, 30.0, 0.1],
                   [w / 2, hgt / 2, 0.3, 0.1], [w / 2, hgt / 2, 500.0, 0.1], [w - 1.0, 0.0, 9.0, 0.2]], np.float32)
    k, _ = h.orient_keypoints(ex)
    assert_same_keypoints(k, oracle.orient(oracle.build_coarse_stack(img), ex), "clamps")


def test_orientation_device_multi_frame_then_describe(lfp, torch, oracle):
    """Extrema of several frames in one call, keypoints handed on to the describe entry point on the device."""
    w, hgt, frames, per = 200, 136, 3, 150
    imgs = np.stack([smooth_image(w, hgt, 20 + f) for f in range(frames)])
    ex = np.concatenate([random_extrema(per, w, hgt, 30 + f, border=3.0) for f in range(frames)])
    frame_of = np.repeat(np.arange(frames, dtype=np.uint32), per)
    perm = np.random.default_rng(1).permutation(len(ex))            # interleave the frames
    ex, frame_of = np.ascontiguousarray(ex[perm]), np.ascontiguousarray(frame_of[perm])
    h = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt, max_frames=frames)
    d_img = torch.from_numpy(imgs).cuda().contiguous()
    d_ex, d_fo = torch.from_numpy(ex).cuda(), torch.from_numpy(frame_of.view(np.int32)).cuda()
    cap = 18 * len(ex)
    d_k = torch.empty((cap, 5), device="cuda")
    d_fk = torch.empty((cap,), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    h.set_images_device(d_img.data_ptr(), frames, w, hgt, s)
    m, dropped = h.orient_keypoints_device(d_ex.data_ptr(), d_fo.data_ptr(), len(ex), d_k.data_ptr(), d_fk.data_ptr(),
                                           cap, s)
    stacks = [oracle.build_coarse_stack(imgs[f]) for f in range(frames)]
    want = np.concatenate([oracle.orient(stacks[frame_of[i]], ex[i:i + 1]) for i in range(len(ex))])
    want_frame = np.concatenate([np.full(len(oracle.orient(stacks[frame_of[i]], ex[i:i + 1])), frame_of[i])
                                 for i in range(len(ex))])
    assert dropped == 0
    got = d_k[:m].cpu().numpy()
    assert_same_keypoints(got, want, "frames")
    assert np.array_equal(d_fk[:m].cpu().numpy(), want_frame)
    d_out = torch.empty((m, 128), device="cuda")
    h.describe_keypoints_frames_device(d_k.data_ptr(), d_fk.data_ptr(), m, d_out.data_ptr(), s)
    torch.cuda.synchronize()
    desc = d_out.cpu().numpy()
    one = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt)
    for f in range(frames):
        sel = want_frame == f
        one.set_image(imgs[f])
        assert_keypoint_parity(oracle, one, imgs[f], got[sel], desc[sel], what=f)


def test_local_features_describe_extrema(lfp, oracle):
    w, hgt = 320, 240
    img = smooth_image(w, hgt, 77)
    ex = random_extrema(400, w, hgt, 78, border=2.0)
    lf = lfp.LocalFeatures(w, hgt, 512, pool_mode=lfp.POOL_F16X3)
    kps, desc = lf.describe_extrema(img, ex)
    want = oracle.orient(oracle.build_coarse_stack(img), ex)
    got = np.array([(k.x, k.y, k.size, k.angle, k.response) for k in kps], np.float32)
    assert_same_keypoints(got, want, "class")
    assert desc.shape == (len(want), 128)
    assert_keypoint_parity(oracle, lf._inner, img, got, desc, what="class")
    assert [k.angle for k in lf.orient(img, ex)] == [k.angle for k in kps]


def test_orientation_on_the_reference_buffer_formats(lfp, oracle):
    """ExtremumLocations.data (blocks of 256: x | y | size | contrast, common.glsl:45-81) + FilteredExtrema.indices in,
    KeypointIndices-style (extremum index, orientation) out: what a caller that keeps the reference's detect graph has."""
    w, hgt, n = 320, 240, 700
    img = smooth_image(w, hgt, 9)
    ex = random_extrema(n, w, hgt, 10, border=4.0)
    blocked = np.zeros(((n + 255) // 256) * 4 * 256, np.float32)
    for i in range(n):
        base, off = (i // 256) * 4 * 256, i % 256
        blocked[base + off], blocked[base + 256 + off] = ex[i, 0], ex[i, 1]
        blocked[base + 512 + off], blocked[base + 768 + off] = ex[i, 2], ex[i, 3]
    kept = np.random.default_rng(11).permutation(n)[:300].astype(np.uint32)      # the host filter's choice, any order
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    kei, ori, kps, dropped = h.orient_keypoints_blocked(blocked, n, kept, 18 * 300)
    want = oracle.orient(oracle.build_coarse_stack(img), ex[kept])
    assert dropped == 0
    assert_same_keypoints(kps, want, "blocked")
    assert np.array_equal(ori, kps[:, 3])
    assert np.array_equal(ex[kei][:, [0, 1, 2]], kps[:, :3]) and set(kei) <= set(kept)
