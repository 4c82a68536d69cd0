"""include/local_features.hpp, the C++ mirror of the reference crate's API (new_vulkan -> new_hip, detect_extract_all,
detect_top_n, detect with a FilterBlobs policy, FeaturesResult, LocalFeaturesError): a small g++ program is built
against liblf_mkd.so, run on the GPU, and its results compared with the Python binding's for the same image."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import MODELS, ROOT

pytestmark = pytest.mark.gpu


def test_cpp_api_program(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gen_golden import blob_image
    import local_features_python as lfp
    exe = str(tmp_path / "demo_local_features")
    lib_dir = os.path.join(ROOT, "local-features_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "demo_local_features.cpp"), "-L", lib_dir, "-llf_mkd",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    w, hgt = 400, 300
    img = blob_image(w, hgt, 7, 400)
    img.tofile(tmp_path / "img.f32")
    out = subprocess.run([exe, MODELS, str(tmp_path / "img.f32"), str(w), str(hgt), str(tmp_path / "res")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[-1].startswith("oversize: InvalidParameters"), lines
    assert "same as the f32 frame: yes" in lines[1], lines             # detect_top_n(ImageViewU8) == detect_top_n of u8 / 255
    counts = dict(zip(lines[0].split()[0::2], lines[0].split()[1::2]))

    def load(tag):
        k = np.fromfile(tmp_path / f"res.{tag}.kps", np.float32).reshape(-1, 5)
        return k, np.fromfile(tmp_path / f"res.{tag}.desc", np.float32).reshape(-1, 128)

    h = lfp.MkdHandle(max_features=1500, max_image_width=w, max_image_height=hgt, max_blobs=1000,
                      pool_mode=lfp.POOL_F16X3)
    for tag, top_n in (("all", 0), ("top", 100)):
        k, d = load(tag)
        pk, pd, _, _ = h.detect(img, top_n, 0.0)
        assert len(k) == int(counts[tag]) > 50
        assert np.array_equal(k, pk) and np.array_equal(d, pd), tag
    # the FilterBlobs policy of the program, replayed through the pieces of the ABI
    h.set_image(img)
    ex, _ = h.detect_extrema(max_out=1024)
    big = np.flatnonzero(ex[:, 2] >= 3.0)[::2]
    pk, _ = h.orient_keypoints(ex[big], max_out=1500)
    k, d = load("filt")
    assert 10 < len(k) == int(counts["filt"]) < int(counts["all"])
    assert np.array_equal(k, pk) and np.array_equal(d, h.describe_keypoints(pk))
    assert int(counts["matches"]) > 50       # the top-100 descriptors find themselves among all
