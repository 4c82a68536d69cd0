"""The reference's match_images example end to end on a real photograph (BASELINE.json configs[0] uses the same
image): detect_top_n on the picture and on a rotated, scaled copy of it, match, and check the matches against the
known transform -- plus parity of every stage against the oracle on the real image."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_keypoint_parity

pytestmark = pytest.mark.gpu


def test_match_images_example_on_a_real_photograph(oracle):
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "local-features_amd", "examples"))
    import match_images as ex
    img1 = ex.load_gray(os.path.join(GOLDEN, "bird.jpg"))
    hgt, w = img1.shape
    assert (w, hgt) == (799, 533)
    # second view: rotate by 17 degrees about the centre (inverse map for PIL: output -> input)
    ang, sc = np.deg2rad(17.0), 1.0
    c, s = np.cos(ang) / sc, np.sin(ang) / sc
    cx, cy = w / 2, hgt / 2
    inv = (c, s, cx - c * cx - s * cy, -s, c, cy + s * cx - c * cy)
    im2 = Image.fromarray((img1 * 255).astype(np.uint8)).transform((w, hgt), Image.AFFINE, inv, resample=Image.BICUBIC)
    img2 = np.asarray(im2, np.float32) / 255.0
    kp1, kp2, d1, d2, m12, m21 = ex.match_images(img1, img2)
    assert len(kp1) > 350 and len(kp2) > 350        # a soft-focus photograph: ~400 blobs above the contrast threshold
    assert len(m12) > 150 and len(m21) > 150 and m12 != [(j, i) for i, j in m21]     # the ratio test is not symmetric
    # forward map image 1 -> image 2 (inverse of `inv`)
    A = np.array([[c, s], [-s, c]])
    t = np.array([inv[2], inv[5]])
    p1 = np.array([[kp1[i].x, kp1[i].y] for i, _ in m12])
    p2 = np.array([[kp2[j].x, kp2[j].y] for _, j in m12])
    pred = (np.linalg.inv(A) @ (p1 - t).T).T
    err = np.linalg.norm(pred - p2, axis=1)
    assert (err < 3.0).mean() > 0.8, (err < 3.0).mean()           # matches are geometrically right
    good = err < 3.0
    sizes = np.array([[kp1[i].size, kp2[j].size] for i, j in m12])[good]
    assert abs(np.median(sizes[:, 1] / sizes[:, 0]) - 1.0) < 0.03  # same scale in both views
    dang = np.array([(kp2[j].angle - kp1[i].angle) % 360 for i, j in m12])[good]
    assert abs(np.median(dang) - 17.0) < 2.5                       # and the assigned orientations turned with the image
    out = os.path.join(ROOT, "gpurun_out", "match_bird.png")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    ex.draw(img1, img2, kp1, kp2, m12, out)
    assert os.path.getsize(out) > 10000
    # every stage against the oracle on the real image (n_scales = 5 as the example sets it)
    want_k, _ = oracle.detect(img1, n_scales=5, top_n=2000, max_blobs=8000, max_features=3000)
    got = np.array([(k.x, k.y, k.size, k.angle, k.response) for k in kp1], np.float32)
    assert got.shape == want_k.shape
    assert np.abs(got[:, :2] - want_k[:, :2]).max() < 1e-3 and np.abs(got[:, 2] / want_k[:, 2] - 1).max() < 1e-4
    da = np.abs(got[:, 3] - want_k[:, 3])
    assert (np.minimum(da, 360 - da) < 1e-3).mean() > 0.995
    import local_features_python as lfp
    h = lfp.MkdHandle(max_features=3000, max_image_width=w, max_image_height=hgt, n_scales=5, pool_mode=lfp.POOL_F16X3)
    h.set_image(img1)
    assert_keypoint_parity(oracle, h, img1, got, d1, what="bird", patch_tol=1e-4)
    want_m = oracle.match(d1, d2)[0]
    assert m12 == [(i, int(j)) for i, j in enumerate(want_m) if j >= 0]
