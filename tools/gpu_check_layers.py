"""Debug aid: are the GPU a-trous layers / pyramid levels bitwise equal to the oracle's?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")]
import torch
import local_features_python as lfp
from oracle import MkdOracle
from test_gpu_orientation import smooth_image
o = MkdOracle(lfp.model_path("liberty"))
for (w, hgt) in ((640, 480), (333, 257), (97, 64)):
    img = smooth_image(w, hgt, 5)
    h = lfp.MkdHandle(max_features=64, max_image_width=w, max_image_height=hgt)
    h.set_image(img)
    st = o.build_coarse_stack(img)
    print(w, hgt, "layers max|diff|:", [float(np.abs(h.coarse_layer(l, w, hgt) - st[l]).max()) for l in range(7)])
    pyr = o.split_pyramid(o.build_pyramid(img), w, hgt)
    print("   pyramid levels max|diff|:", [float(np.abs(h.pyramid_level(l) - pyr[l]).max()) for l in range(len(pyr))])
