"""CPU tests: the C oracle against the committed float64 goldens (tools/gen_golden.py), against
known values derived from the reference's sources, and against properties of the algorithm.
The reference ships no golden vectors of its own (mkd_ref.rs:393-453 names absent files)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import MODELS, golden, rel_l2
from oracle import ATAN_LIBM, ATAN_SHADER

# sha256 of the reference's model files
SHA = {"liberty": "bb731ff7d9d900184fb1dcf720826cf5e4ce8f9331351c83367a324dc1041773",
       "notredame": "9bf382ad89dd90d40dd06aef3477115e1ec3c090aa2d7e32c88b7d03ca3f8c16",
       "yosemite": "cf2beffc3bc5a78f4b67643db732bf562fc91ce9cbb54d120ff37f111de9fac0"}


@pytest.mark.parametrize("name", sorted(SHA))
def test_model_files_are_the_reference_data(name):
    b = open(os.path.join(MODELS, f"concat-pca-{name}.safetensors"), "rb").read()
    assert len(b) == 228696
    assert hashlib.sha256(b).hexdigest() == SHA[name]


def test_pca_model_properties(oracle):
    # eigvals normalised, descending; eigvec columns unit norm
    assert oracle.eigvals[0] == pytest.approx(1.0)
    assert np.all(np.diff(oracle.eigvals[:128]) <= 0)
    assert np.allclose(np.linalg.norm(oracle.eigvecs, axis=0), 1.0, atol=1e-4)
    w = oracle.eigvecs[:, :128] * oracle.eigvals[:128] ** np.float32(-0.35)
    assert np.allclose(oracle.eigen_vecs, w.T, rtol=2e-6, atol=1e-7)


def test_luts_match_float64_restatement(oracle):
    g = golden("luts.npz")
    assert np.abs(oracle.gradient_angle - g["gradient_angle"]).max() < 2e-6
    assert np.abs(oracle.embedding_polar - g["embedding_polar"]).max() < 2e-6
    assert np.abs(oracle.embedding_cartesian - g["embedding_cartesian"]).max() < 1e-6


def test_lut_known_values(oracle):
    ep, ec = oracle.embedding_polar, oracle.embedding_cartesian
    assert ep.shape == (25, 32, 32) and ec.shape == (9, 32, 32)
    c0p, c0c = np.float32(0.37872374), np.float32(0.618176)
    # kernel 0 = c0*c0*G: corner G = e^-1, centre 4 px G = 0.99895996 (mkd_ref.rs:259-267)
    assert ep[0, 0, 0] == pytest.approx(c0p * c0p * 0.36787944, rel=1e-5)
    assert ec[0, 0, 0] == pytest.approx(c0c * c0c * 0.36787944, rel=1e-5)
    assert ec[0, 15, 15] == pytest.approx(c0c * c0c * 0.99895996, rel=1e-5)
    assert ec[0, 16, 16] == pytest.approx(ec[0, 15, 15], rel=1e-6)
    # gradient_angle = -atan2(y, x): top-left pixel (x=-1,y=-1) -> +3pi/4
    assert oracle.gradient_angle[0, 0] == pytest.approx(0.75 * np.pi, rel=1e-6)
    assert oracle.gradient_angle[31, 31] == pytest.approx(-0.25 * np.pi, rel=1e-6)


def test_shader_atan2_quirks(oracle):
    """shaders/atan2.glsl:19-46."""
    a = oracle.atan2_shader
    assert a(0.0, 0.0) == 0.0
    assert a(0.0, 1.0) == 0.0          # libm: +pi/2
    assert a(0.0, -3.0) == 0.0         # libm: -pi/2
    assert a(-0.0, 0.0) == 0.0
    assert a(-1.0, 0.0) == pytest.approx(np.pi, rel=1e-6)   # y == 0 -> +pi
    rng = np.random.default_rng(1)
    for x, y in rng.normal(size=(2000, 2)).astype(np.float32):
        assert abs(a(x, y) - np.arctan2(y, x)) < 2e-5


@pytest.mark.parametrize("name", ["liberty", "notredame", "yosemite"])
def test_oracle_matches_float64_goldens(oracles, name):
    g = golden(f"patches_{name}.npz")
    for mode, key in ((ATAN_SHADER, "shader"), (ATAN_LIBM, "libm")):
        d, raw = oracles[name].describe_patches(g["patches"], atan_mode=mode, want_raw=True)
        assert rel_l2(raw, g[f"raw_{key}"]).max() < 1e-5
        assert rel_l2(d, g[f"desc_{key}"]).max() < 1e-5
        assert np.abs(d - g[f"desc_{key}"]).max() < 1e-5      # mkd_ref.rs:426-427 asks 1e-4


def test_keypoint_mode_matches_float64_goldens(oracle):
    g = golden("keypoints_liberty.npz")
    img = g["image"]
    h, w = img.shape
    pyr = oracle.build_pyramid(img)
    lv = oracle.split_pyramid(pyr, w, h)
    assert len(lv) == 8 and lv[1].shape == (68, 100) and lv[3].shape == (17, 25)
    assert np.abs(lv[1] - g["level1"]).max() < 1e-6
    assert np.abs(lv[3] - g["level3"]).max() < 1e-6
    p = oracle.sample_patches(pyr, w, h, g["keypoints"])
    assert np.abs(p - g["patches"]).max() < 5e-6
    assert rel_l2(oracle.describe_patches(p), g["desc_shader"]).max() < 5e-5


def test_the_two_readings_of_the_sample_position(oracle):
    """patch_gradients.glsl:60-67 is not `precise`: `dx*ca - dy*sa` and `xx*r + x/2^L` may or may not be fused by the GLSL
    compiler.  The oracle holds both readings (contract=False: mul then add, the default; contract=True: fma -- the reading the
    HIP sampler implements).  Both are the float64 restatement's values to the same tolerance; they differ from each other by
    at most one ulp of the coordinate times the frame's gradient (one of the two causes of the thinnest parity margin:
    profiles/r05_parity_report.txt)."""
    g = golden("keypoints_liberty.npz")
    img = g["image"]
    h, w = img.shape
    pyr = oracle.build_pyramid(img)
    a = oracle.sample_patches(pyr, w, h, g["keypoints"])
    b = oracle.sample_patches(pyr, w, h, g["keypoints"], contract=True)
    assert np.abs(a - g["patches"]).max() < 5e-6 and np.abs(b - g["patches"]).max() < 5e-6
    assert 0 < np.abs(a - b).max() < 2e-5 and (a != b).mean() < 0.5
    # a synthetic frame with large coordinates: the two readings are apart by about one coordinate ulp x the gradient
    rng = np.random.default_rng(3)
    big = np.ascontiguousarray(rng.random((64, 4096)).astype(np.float32))
    k = np.array([[x, 32.0, 2.0, 37.0] for x in (40.3, 600.7, 1500.2, 2900.9, 4000.4)], np.float32)
    pb = oracle.build_pyramid(big)
    d = np.abs(oracle.sample_patches(pb, 4096, 64, k) - oracle.sample_patches(pb, 4096, 64, k, contract=True)).reshape(5, -1).max(1)
    assert 0 < d.max() < 4096 * 2.0 ** -23 * 2.0          # never more than one ulp of the coordinate x the gradient (below 1 here)


def test_properties(oracle):
    rng = np.random.default_rng(3)
    p = rng.random((8, 32, 32)).astype(np.float32)
    d, raw = oracle.describe_patches(p, want_raw=True)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
    assert np.allclose(np.linalg.norm(raw, axis=1), 1.0, atol=1e-5)
    # polar and cartesian halves each carry norm 1/sqrt(2) (normalize.glsl)
    assert np.allclose(np.linalg.norm(raw[:, :175], axis=1), np.sqrt(0.5), atol=1e-5)
    # additive constants vanish in the gradient; positive gain only moves the 1e-8 epsilon
    d2 = oracle.describe_patches(p * np.float32(0.5) + np.float32(0.25))
    assert rel_l2(d2, d).max() < 2e-4
    # flat patch: mag == 0.01, angle == 0 everywhere, whatever the constant
    f1 = oracle.describe_patches(np.full((1, 32, 32), 0.2, np.float32))
    f2 = oracle.describe_patches(np.full((1, 32, 32), 0.9, np.float32))
    assert np.array_equal(f1, f2)
    mag, ang = oracle.patch_gradients(np.full((32, 32), 0.5, np.float32))
    assert np.allclose(mag, 0.01) and np.all(ang == 0)


def test_threads_agree(oracle):
    rng = np.random.default_rng(5)
    p = rng.random((37, 32, 32)).astype(np.float32)
    assert np.array_equal(oracle.describe_patches(p, nthreads=1), oracle.describe_patches(p, nthreads=4))


def test_orientation_matches_float64_goldens(oracle):
    """Coarse a-trous stack + keypoint orientation (vulkan/mod.rs:1093-1130, keypoint_orientation.glsl)
    against the separately written float64 restatement (tools/gen_golden.py)."""
    g = golden("orientation.npz")
    st = oracle.build_coarse_stack(g["image"])
    assert st.shape == (7,) + g["image"].shape
    assert np.abs(st[2] - g["layer2"]).max() < 1e-6
    assert np.abs(st[5] - g["layer5"]).max() < 1e-6
    k = oracle.orient(st, g["extrema"])
    assert k.shape == g["keypoints"].shape                       # same peaks found, same order
    assert np.array_equal(k[:, [0, 1, 2, 4]], g["keypoints"][:, [0, 1, 2, 4]])
    d = np.abs(k[:, 3] - g["keypoints"][:, 3])
    assert np.minimum(d, 360 - d).max() < 1e-3                   # degrees


def test_orientation_properties(oracle):
    """A linear ramp has one gradient direction: the histogram has one peak at that direction, for every size."""
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for deg in (0.0, 40.0, 90.0, 200.0, 270.0, 310.0):                # bin centres (10 degree bins)
        t = np.deg2rad(deg)
        # gx = left... the shader's gradient is (right-left, up-down); its angle is 360 - keypoint angle
        img = (0.5 + 0.002 * (np.cos(t) * (xx - w / 2) - np.sin(t) * (yy - h / 2))).astype(np.float32)
        st = np.broadcast_to(img, (7, h, w)).copy()              # bypass the blur: same ramp on every layer
        ex = np.array([[64.3, 48.7, s, 0.1] for s in (2.5, 5.0, 9.0)], np.float32)
        k = oracle.orient(st, ex)
        assert len(k) == 3
        # atan2.glsl returns 0 when x == 0 (exactly vertical gradient): the reference's quirk, kept
        want = 0.0 if deg in (90.0, 270.0) else (360.0 - deg) % 360.0
        d = np.abs(k[:, 3] - want)
        assert np.minimum(d, 360 - d).max() < 1e-3, (deg, k[:, 3])
    # flat image: no gradient, no histogram mass -> no peaks (left < hist fails on 0 < 0)
    assert len(oracle.orient(np.full((7, h, w), 0.3, np.float32), ex)) == 0
    # empty input
    assert oracle.orient(st, np.zeros((0, 4), np.float32)).shape == (0, 5)


def test_quirk_pixel_detector_and_contracted_blur(oracle):
    """atan2.glsl's x == 0 discontinuity: a patch mirrored about column 15 has gx == 0 (or +-1 ulp, depending on how
    the blur rounds) down that column; the detector flags it, and the two blur roundings the shader allows give
    different descriptors there while agreeing to 1e-5 on ordinary patches."""
    from oracle import BLUR_CONTRACT
    rng = np.random.default_rng(5)
    p = rng.random((200, 32, 32)).astype(np.float32)
    q = oracle.quirk_pixels(p)
    assert (q > 0).sum() <= 3
    a = oracle.describe_patches(p)
    b = oracle.describe_patches(p, atan_mode=ATAN_SHADER | BLUR_CONTRACT)
    assert rel_l2(a, b)[q == 0].max() < 2e-5
    half = rng.random((32, 16)).astype(np.float32)
    sym = np.concatenate([half[:, :15], half[:, 15:16], half[:, 14::-1], half[:, :1]], axis=1)   # cols 0..30 mirrored
    assert sym.shape == (32, 32) and np.array_equal(sym[:, 14], sym[:, 16])
    assert oracle.quirk_pixels(sym)[0] >= 20
    mag, ang = oracle.patch_gradients(sym)
    col = np.abs(ang[:, 15])
    assert ((col == 0) | (np.abs(col - np.pi / 2) < 1e-3)).all() and (col == 0).any()


def test_detector_matches_float64_goldens(oracle):
    """DoG + extremum scan + refinement + edge test (swt_sub.glsl, scan_extrema.glsl) and the top-K blob filter
    (mod.rs:1753-1786) against the float64 restatement of tools/gen_golden.py."""
    g = golden("detector.npz")
    fine = oracle.dog(oracle.build_coarse_stack(g["image"]))
    assert fine.shape == (6,) + g["image"].shape
    assert np.abs(fine[2] - g["dog2"]).max() < 1e-6 and np.abs(fine[4] - g["dog4"]).max() < 1e-6
    ex, total = oracle.scan_extrema(fine)
    assert total == len(ex) == len(g["extrema"])
    assert np.abs(ex[:, :2] - g["extrema"][:, :2]).max() < 1e-3            # pixels
    assert np.abs(ex[:, 2] / g["extrema"][:, 2] - 1).max() < 1e-4          # size
    assert np.abs(ex[:, 3] - g["extrema"][:, 3]).max() < 1e-6              # contrast
    assert np.array_equal(oracle.topk_filter(ex, 25), g["top25"])


def test_detector_properties(oracle):
    # one Gaussian blob of sigma s: one extremum at its centre, size ~ 2 sigma... of the blob; sign does not matter
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for sign in (1, -1):
        for s, (cx, cy) in ((2.0, (40.3, 50.6)), (4.0, (70.8, 41.2))):
            img = (0.5 + sign * 0.4 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))).astype(np.float32)
            ex, total = oracle.scan_extrema(oracle.dog(oracle.build_coarse_stack(img)))
            assert total == 1, (sign, s, ex)
            assert abs(ex[0, 0] - cx) < 0.25 and abs(ex[0, 1] - cy) < 0.25
            assert 1.2 * s < ex[0, 2] < 2.2 * s and ex[0, 3] > 0.035
    # a straight edge has an anisotropic hessian everywhere: rejected by the cm test; flat image: nothing
    edge = np.where(xx > 60.5, 0.9, 0.1).astype(np.float32)
    assert oracle.scan_extrema(oracle.dog(oracle.build_coarse_stack(edge)))[1] == 0
    assert oracle.scan_extrema(oracle.dog(oracle.build_coarse_stack(np.full((h, w), 0.4, np.float32))))[1] == 0
    # nothing closer than `border` to the frame, truncation at max_out keeps the head of the ordered list
    g = golden("detector.npz")
    fine = oracle.dog(oracle.build_coarse_stack(g["image"]))
    ex, total = oracle.scan_extrema(fine)
    assert ex[:, 0].min() >= 4.5 and ex[:, 1].min() >= 4.5
    assert ex[:, 0].max() <= g["image"].shape[1] - 4.5 and ex[:, 1].max() <= g["image"].shape[0] - 4.5
    cut, total2 = oracle.scan_extrema(fine, max_out=10)
    assert total2 == total and np.array_equal(cut, ex[:10])
    # top-K: fewer than n -> all; min_size filters first; kept in index order
    assert np.array_equal(oracle.topk_filter(ex, 1000), np.arange(len(ex)))
    big = oracle.topk_filter(ex, 1000, min_size=6.0)
    assert (ex[big, 2] >= 6.0).all() and len(big) == (ex[:, 2] >= 6.0).sum()
    k = oracle.topk_filter(ex, 10)
    assert len(k) == 10 and (np.diff(k) > 0).all()
    assert np.sort(ex[k, 3]).min() >= np.sort(ex[:, 3])[::-1][10]


def test_matcher_restates_match_features(oracle):
    """examples/match_images/src/main.rs:8-27 against a NumPy stable argsort of the similarity matrix."""
    rng = np.random.default_rng(11)
    b = rng.normal(size=(700, 128)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    a = b[rng.integers(0, 700, 400)] + 0.06 * rng.normal(size=(400, 128)).astype(np.float32)
    a[200:] = rng.normal(size=(200, 128))                   # unrelated queries: best and second close, rejected
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b[650] = b[3]                                           # equal maxima: the higher index is the best, and is rejected
    a[0] = b[3]
    m, s1, s2 = oracle.match(a, b)
    S = a.astype(np.float64) @ b.astype(np.float64).T
    order = np.argsort(S, axis=1, kind="stable")
    first, second = order[:, -1], order[:, -2]
    rows = np.arange(len(a))
    want = np.where(S[rows, first] * 0.8 > S[rows, second], first, -1)
    assert np.array_equal(m[1:], want[1:]) and m[0] == -1
    assert np.abs(s1 - S[rows, first]).max() < 1e-6 and np.abs(s2 - S[rows, second]).max() < 1e-6
    assert 0.2 < (m >= 0).mean() < 0.98
    # exclusion ranges drop candidates; thread count does not matter
    lo = np.zeros(len(a), np.uint32)
    hi = np.full(len(a), 350, np.uint32)
    mx = oracle.match(a, b, exclude=(lo, hi), nthreads=3)[0]
    assert (mx[mx >= 0] >= 350).all()
    assert np.array_equal(oracle.match(a, b, nthreads=1)[0], m)


def test_cpu_baseline_port_agrees_with_the_oracle(oracle):
    """oracle/mkd_cpu_fast.c (what bench.py times as cpu_baseline) is the same arithmetic organised for a CPU: it must give
    the oracle's descriptors (contracted-blur reading) to rounding, in both angle modes, quirk pixels included, and for any
    thread count."""
    from oracle import ATAN_LIBM, ATAN_SHADER, BLUR_CONTRACT
    rng = np.random.default_rng(31)
    flat = np.full((3, 32, 32), 0.25, np.float32)
    flat[1, 10:20, 5:25] = 0.75                                  # exact gx == 0 / null gradients: the shader quirk
    flat[2, :, 16:] = 0.5
    smooth = rng.random((200, 8, 8)).astype(np.float32).repeat(4, 1).repeat(4, 2)
    p = np.concatenate([rng.random((1500, 32, 32), dtype=np.float32), smooth, flat[1:]])
    for mode in (ATAN_SHADER, ATAN_LIBM):
        want = oracle.describe_patches(p, atan_mode=mode | BLUR_CONTRACT, nthreads=8)
        got = oracle.describe_patches_fast(p, atan_mode=mode, nthreads=3)
        assert rel_l2(got, want).max() < 2e-5, (mode, rel_l2(got, want).max())
        assert np.array_equal(got, oracle.describe_patches_fast(p, atan_mode=mode, nthreads=1))
