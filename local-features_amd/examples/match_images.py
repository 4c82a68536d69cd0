#!/usr/bin/env python3
"""match_images IMAGE_1 IMAGE_2 IMAGE_OUT -- the reference's example (examples/match_images/src/main.rs) on the
MI355X path: load two images, detect_top_n(2000, min_size 0) on each, brute-force match 1->2 and 2->1 with the
0.8 ratio test, draw keypoints and the 1->2 matches side by side.

Image decoding follows main.rs:44-60: 8-bit luma, then f32 / 255 (Pillow's "L" conversion stands in for the `image`
crate's grayscale(); they may differ by one LSB).  Needs Pillow."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import local_features_python as lfp  # noqa: E402


def load_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("L"), np.float32) / 255.0


def match_images(img1, img2, top_n=2000, min_size=0.0):
    """Returns (keypoints1, keypoints2, matches 1->2, matches 2->1) as the example computes them (main.rs:62-121)."""
    feats = lfp.LocalFeatures(max(img1.shape[1], img2.shape[1]), max(img1.shape[0], img2.shape[0]), 3000,
                              max_blobs=8000, n_scales=5, pca="liberty", pool_mode=lfp.POOL_F16X3)
    kp1, d1 = feats.detect_top_n(img1, top_n, min_size)
    kp2, d2 = feats.detect_top_n(img2, top_n, min_size)
    m12, m21 = feats.match_both(d1, d2)        # main.rs:113-116: both directions, one launch on the device
    return kp1, kp2, d1, d2, m12, m21


def draw(img1, img2, kp1, kp2, matches, out_path):
    from PIL import Image, ImageDraw
    h = max(img1.shape[0], img2.shape[0])
    canvas = Image.new("L", (img1.shape[1] + img2.shape[1], h))
    canvas.paste(Image.fromarray((img1 * 255).astype(np.uint8)), (0, 0))
    canvas.paste(Image.fromarray((img2 * 255).astype(np.uint8)), (img1.shape[1], 0))
    d = ImageDraw.Draw(canvas)
    for k in kp1:
        d.ellipse([k.x - k.size, k.y - k.size, k.x + k.size, k.y + k.size], outline=255)
    for k in kp2:
        ox = img1.shape[1]
        d.ellipse([ox + k.x - k.size, k.y - k.size, ox + k.x + k.size, k.y + k.size], outline=255)
    for i, j in matches:
        d.line([kp1[i].x, kp1[i].y, img1.shape[1] + kp2[j].x, kp2[j].y], fill=255)
    canvas.save(out_path)


def main():
    if len(sys.argv) != 4:
        print("Required arguments: IMAGE_1 IMAGE_2 IMAGE_OUT", file=sys.stderr)
        return 1
    img1, img2 = load_gray(sys.argv[1]), load_gray(sys.argv[2])
    kp1, kp2, _, _, m12, m21 = match_images(img1, img2)
    print(f"Extracted {len(kp1)} and {len(kp2)} keypoints")
    print(f"Matching 1 -> 2: {len(m12)} matches")
    print(f"Matching 2 -> 1: {len(m21)} matches")
    draw(img1, img2, kp1, kp2, m12, sys.argv[3])
    return 0


if __name__ == "__main__":
    sys.exit(main())
