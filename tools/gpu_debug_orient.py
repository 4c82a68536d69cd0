"""Debug aid: worst keypoint of the multi-frame orientation test; is the error from sampling or describing?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")]
import torch
import local_features_python as lfp
from oracle import MkdOracle
from test_gpu_orientation import smooth_image, random_extrema
from conftest import rel_l2

o = MkdOracle(lfp.model_path("liberty"))
w, hgt = 200, 136
for f in range(3):
    img = smooth_image(w, hgt, 20 + f)
    ex = random_extrema(150, w, hgt, 30 + f, border=3.0)
    h = lfp.MkdHandle(max_features=256, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    k, _ = h.orient_keypoints(ex)
    d = h.describe_keypoints(k)
    pyr = o.build_pyramid(img)
    pat = o.sample_patches(pyr, w, hgt, k[:, :4])
    ref = o.describe_patches(pat)
    e = rel_l2(d, ref)
    i = int(np.argmax(e))
    dk = torch.from_numpy(k).cuda()
    dp = torch.empty((len(k), 32, 32), device="cuda")
    h.sample_patches_device(dk.data_ptr(), len(k), dp.data_ptr())
    h.synchronize()
    gp = dp.cpu().numpy()
    print(f, "worst", i, e[i], "kp", k[i], "patch range", pat[i].min(), pat[i].max(), "std", pat[i].std(),
          "patch maxdiff", np.abs(gp[i] - pat[i]).max(), "all patch maxdiff", np.abs(gp - pat).max())
    print("   oracle(desc of gpu patch) vs oracle:", rel_l2(o.describe_patches(gp[i:i+1]), ref[i:i+1]),
          " gpu desc vs oracle(desc of gpu patch):", rel_l2(d[i:i+1], o.describe_patches(gp[i:i+1])))
    lv = np.log2(k[i, 2] * 24 / 32)
    print("   log2 scale", lv)
