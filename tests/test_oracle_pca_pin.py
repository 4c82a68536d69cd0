"""Pins the oracle's raw 238-D descriptor against the only numbers the reference holds for this path.

The reference has no golden vectors (mkd_ref.rs:393-453 is commented out and names files that are absent), but its PCA
models `local_features/models/mkd/concat-pca-*.safetensors` (loader mkd_ref.rs:352-391, use vulkan/mod.rs:1594-1612) were
FITTED on raw 238-D MKD descriptors of natural keypoint patches: `mean[238]`, and the eigen-decomposition of their
covariance.  So the statistics of raw descriptors of natural patches are reference-held data: if the oracle computes the
descriptor the models were fitted on -- same feature order (polar block then cartesian block, [in-dim i][kernel j] inside
a block, shaders/common.glsl:114-139), same sign and zero of the gradient angle -- then over many natural patches

  * the mean of its raw descriptors reproduces the model's `mean`, block by block, and
  * the variance of (raw - mean) along the model's k-th eigenvector falls with k the way `eigvals` does.

Both restatements in this repository (oracle/mkd_oracle.c and tools/gen_golden.py) could share a misreading of the
layout; these two checks could not be passed with one, as the second half of the test shows by scoring the obvious
misreadings.  This does not make parity "green" (there are still no reference output vectors), but it is evidence
from the reference's own data, independent of both restatements.

Patches: keypoint-sampled from the reference's sample photographs (tests/golden/bird.jpg, houses.jpg; CREDITS.md) at
the oracle detector's keypoints (scale and orientation normalised, as the patch datasets behind the models are)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, MODELS

N_MIN = 50000


def _gray(name, scale=1.0):
    from PIL import Image
    im = Image.open(os.path.join(GOLDEN, name)).convert("L")
    if scale != 1.0:
        im = im.resize((int(im.width * scale), int(im.height * scale)), Image.BICUBIC)
    return np.asarray(im, np.float32) / 255.0


@pytest.fixture(scope="module")
def natural(oracle):
    """(patches [n,32,32], raw [n,238]) of >= 50 000 detector keypoints on the sample photographs."""
    patches = []
    for name, scale, thr in (("houses.jpg", 1.0, 0.035), ("houses.jpg", 0.5, 0.035), ("houses.jpg", 0.7, 0.035),
                             ("bird.jpg", 1.0, 0.015), ("bird.jpg", 1.4, 0.015)):
        img = _gray(name, scale)
        stack = oracle.build_coarse_stack(img, 4)
        extrema, _ = oracle.scan_extrema(oracle.dog(stack), contrast_threshold=thr)
        kps = oracle.orient(stack, extrema)
        h, w = img.shape
        patches.append(oracle.sample_patches(oracle.build_pyramid(img), w, h, kps[:, :4]))
    patches = np.concatenate(patches)
    assert len(patches) >= N_MIN, len(patches)
    _, raw = oracle.describe_patches(patches, nthreads=8, want_raw=True)
    return patches, raw


def _corr(a, b):
    return float(np.corrcoef(a, b)[0, 1])


def _spearman(a, b):
    from scipy.stats import spearmanr
    return float(spearmanr(a, b).correlation)


def _scores(raw, model):
    """(correlation of the polar block's mean with the model's; same for the cartesian block;
        rank correlation between the variance along eigenvector k and eigvals[k], all 238 directions;
        spread of log(variance along eigenvector k / eigvals[k]) over the 128 directions the whitening uses -- 0 if the
        spectrum is reproduced exactly up to one scale factor)."""
    mean, eigvals, eigvecs = model
    m = raw.mean(0)
    var = ((raw - mean) @ eigvecs).var(0)          # eigvecs[:, k] is the k-th eigenvector (mod.rs:1604-1612)
    spread = float(np.std(np.log(var[:128] / eigvals[:128])))
    return _corr(m[:175], mean[:175]), _corr(m[175:], mean[175:]), _spearman(var, eigvals), spread


@pytest.mark.parametrize("model_name", ["liberty", "notredame", "yosemite"])
def test_raw_descriptor_statistics_follow_the_reference_pca_models(natural, oracles, model_name):
    _, raw = natural
    o = oracles[model_name]
    c_polar, c_cart, rank, spread = _scores(raw.astype(np.float64), (o.mean, o.eigvals, o.eigvecs))
    print(f"{model_name}: {len(raw)} patches, mean correlation polar {c_polar:.4f} cartesian {c_cart:.4f}, "
          f"spectrum rank correlation {rank:.4f}, log-spectrum spread {spread:.3f}")
    assert c_polar > 0.98 and c_cart > 0.98, (c_polar, c_cart)
    assert rank > 0.98, rank
    assert spread < 0.4, spread


def test_whitened_descriptor_spectrum_has_the_attenuation_exponent(natural, oracle):
    """The whitening is W = eigvecs[:, :128] diag(eigvals^-0.35) (t = 0.7, exponent -0.5 t; vulkan/mod.rs:1604-1612,
    mkd_ref.rs:61-75): "attenuated" PCA whitening.  On data whose variance along eigenvector k is proportional to
    eigvals[k] (the test above), output dimension k then has variance proportional to eigvals[k]^(1 - 0.7) = eigvals[k]^0.3
    (the final L2 normalisation rescales a whole descriptor and leaves the ratios).  On these photographs the variance along
    eigenvector k follows eigvals[k] with an exponent slightly below 1 (log-spectrum spread 0.22), so the measured slope is
    0.26; full whitening (exponent -0.5) would give about -0.05, no scaling about 0.95."""
    patches, _ = natural
    desc = oracle.describe_patches(patches[::4], nthreads=8).astype(np.float64)
    var = desc.var(0)
    slope, _ = np.polyfit(np.log(oracle.eigvals[:128].astype(np.float64)), np.log(var), 1)
    print(f"log-variance of the whitened descriptor against log eigvals: slope {slope:.3f} (0.3 expected)")
    assert 0.15 < slope < 0.45, slope
    assert _spearman(var, oracle.eigvals[:128]) > 0.9


def test_the_obvious_misreadings_score_clearly_worse(natural, oracle):
    """Feature order, block order and the sign conventions of the angle are what two restatements by one author could
    get wrong together; each such variant must fit the reference's model clearly worse than the oracle's reading:
    at least 2.5 times the spread of the log-spectrum, a lower rank correlation and lower block-mean correlations."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden as gg
    patches, raw = natural
    model = (oracle.mean.astype(np.float64), oracle.eigvals.astype(np.float64), oracle.eigvecs.astype(np.float64))
    raw = raw.astype(np.float64)
    base = _scores(raw, model)

    variants = {
        "cartesian block before polar": np.concatenate([raw[:, 175:], raw[:, :175]], axis=1),
        "[kernel j][in-dim i] inside the blocks": np.concatenate(
            [raw[:, :175].reshape(-1, 7, 25).transpose(0, 2, 1).reshape(-1, 175),
             raw[:, 175:].reshape(-1, 7, 9).transpose(0, 2, 1).reshape(-1, 63)], axis=1),
        "sin in-dims before cos in-dims": np.concatenate(
            [raw[:, :175].reshape(-1, 7, 25)[:, [0, 4, 5, 6, 1, 2, 3]].reshape(-1, 175),
             raw[:, 175:].reshape(-1, 7, 9)[:, [0, 4, 5, 6, 1, 2, 3]].reshape(-1, 63)], axis=1),
    }
    # variants of the angle itself need the descriptor recomputed: the float64 restatement on a subset of the patches
    sub = patches[:: max(1, len(patches) // 6000)][:6000]
    pca = gg.load_pca("liberty")

    def recompute(**patch):
        saved = {k: getattr(gg, k) for k in patch}
        try:
            for k, v in patch.items():
                setattr(gg, k, v)
            return np.concatenate([gg.describe(sub[i:i + 1000], pca, want_raw=True)[1] for i in range(0, len(sub), 1000)])
        finally:
            for k, v in saved.items():
                setattr(gg, k, v)

    sub_base = _scores(recompute(), model)
    real_atan2, real_luts = gg.atan2_shader, gg.luts
    variants_sub = {
        "angle = +atan2 instead of -atan2": recompute(atan2_shader=lambda x, y: -real_atan2(x, y)),
        "gradient right-left instead of left-right": recompute(atan2_shader=lambda x, y: real_atan2(-x, y)),
        "gradient_angle (phi) with the opposite sign": recompute(luts=lambda: (lambda p, ep, ec: (-p, ep, ec))(*real_luts())),
    }
    fmt = "polar {:.4f} cartesian {:.4f} spectrum rank {:.4f} log-spectrum spread {:.3f}".format
    print("oracle's reading:", fmt(*base))
    print(f"  (float64 restatement, {len(sub)} patches:", fmt(*sub_base), ")")
    assert min(sub_base[0], sub_base[1]) > 0.98 and sub_base[2] > 0.97 and sub_base[3] < 0.4
    for name, r in list(variants.items()) + list(variants_sub.items()):
        ref = base if name in variants else sub_base
        s = _scores(r, model)
        print(f"{name}:", fmt(*s))
        # Clearly worse.  The three layout variants and the sign of phi wreck the block means and the spectrum; the two
        # variants that mirror the patch (sign of the angle = a vertical flip, of the x gradient = a horizontal one) leave
        # the means almost alone -- natural patches are close to mirror symmetric -- but not the spectrum: the model's
        # eigenvectors are not mirror images of themselves, so the variances along them no longer follow eigvals.
        assert s[3] > 2.5 * ref[3], (name, s, ref)
        assert s[2] < ref[2] - 0.015 and min(s[0], s[1]) < min(ref[0], ref[1]), (name, s, ref)
