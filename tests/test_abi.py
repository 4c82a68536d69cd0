"""CPU tests of the drop-in boundary: liblf_mkd.so loads, exports every symbol include/lf_mkd.h
declares, builds the same constants as the oracle, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

import local_features_python as lfp


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lf_mkd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_mkd_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = lfp.load_library()
    declared = _declared_symbols()
    assert declared == sorted(lfp.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), f"liblf_mkd.so does not export {s}"
    assert L.lf_mkd_version().decode().startswith("lf_mkd ") and b"gfx950" in L.lf_mkd_version()


# (struct layouts: tests/test_rust_binding.py::test_struct_layouts_match_the_header_field_by_field checks the ctypes and
#  Rust structs against offsetof / sizeof compiled from the header)


def test_host_constants_match_oracle(oracle):
    L = lfp.load_library()
    ga = np.zeros(1024, np.float32)
    ep = np.zeros(25 * 1024, np.float32)
    ec = np.zeros(9 * 1024, np.float32)
    wt = np.zeros(128 * 238, np.float32)
    rc = L.lf_mkd_build_constants(oracle.mean.ctypes.data, oracle.eigvals.ctypes.data, oracle.eigvecs.ctypes.data,
                                  ga.ctypes.data, ep.ctypes.data, ec.ctypes.data, wt.ctypes.data)
    assert rc == 0
    assert np.abs(ga - oracle.gradient_angle.ravel()).max() < 1e-6
    assert np.abs(ep - oracle.embedding_polar.ravel()).max() < 1e-6
    assert np.abs(ec - oracle.embedding_cartesian.ravel()).max() < 1e-6
    assert np.allclose(wt, oracle.eigen_vecs.ravel(), rtol=2e-6, atol=1e-7)


def test_bad_arguments_are_reported_not_fatal():
    L = lfp.load_library()
    h = ctypes.c_void_p()
    assert L.lf_mkd_create_from_file(None, b"x", ctypes.byref(h)) != 0
    p = lfp._lib.Params(device=0)
    assert L.lf_mkd_create_from_file(ctypes.byref(p), b"/nonexistent.safetensors", ctypes.byref(h)) == -3
    assert b"nonexistent" in L.lf_mkd_last_error(None)
    with pytest.raises(RuntimeError, match="Invalid PCA argument"):       # python/src/lib.rs:60-64
        lfp.LocalFeatures(640, 480, 100, pca="oxford")


def test_an_unusable_pca_model_is_an_error_not_an_abort(oracle):
    """include/lf_mkd.h: "they never abort".  The whitening scales by eigvals^-0.35 (vulkan/mod.rs:1604-1612): a model whose
    eigenvalues are not positive, or that holds NaNs, is refused with LF_MKD_ERR_BAD_ARG and a message -- by the
    constants tap (no device needed) and by lf_mkd_create before it looks for a device."""
    L = lfp.load_library()
    mean, vals, vecs = oracle.mean.copy(), oracle.eigvals.copy(), oracle.eigvecs.copy()
    p = lfp._lib.Params(device=0)
    h = ctypes.c_void_p()
    for what, breaker in (("eigvals[5]", lambda: vals.__setitem__(5, 0.0)), ("eigvals[127]", lambda: vals.__setitem__(127, -1e-3)),
                          ("eigvals[0]", lambda: vals.__setitem__(0, np.nan)), ("mean", lambda: mean.__setitem__(17, np.inf)),
                          ("eigvecs", lambda: vecs.reshape(-1).__setitem__(1234, np.nan))):
        mean, vals, vecs = oracle.mean.copy(), oracle.eigvals.copy(), oracle.eigvecs.copy()
        breaker()
        assert L.lf_mkd_build_constants(mean.ctypes.data, vals.ctypes.data, vecs.ctypes.data, None, None, None, None) == -1
        assert what.encode() in L.lf_mkd_last_error(None), (what, L.lf_mkd_last_error(None))
        assert L.lf_mkd_create(ctypes.byref(p), mean.ctypes.data, vals.ctypes.data, vecs.ctypes.data, ctypes.byref(h)) == -1
        assert not h.value and what.encode() in L.lf_mkd_last_error(None)
    # eigenvalues beyond the 128 the whitening uses may be anything non-negative (the models' tails are ~1e-7)
    vals = oracle.eigvals.copy()
    vals[200] = 0.0
    assert L.lf_mkd_build_constants(oracle.mean.ctypes.data, vals.ctypes.data, oracle.eigvecs.ctypes.data, None, None, None, None) == 0
    assert b"abort" not in open(os.path.join(ROOT, "local-features_amd", "csrc", "mkd_consts.cpp"), "rb").read().replace(b"never aborts", b"")


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="Failed to initialize local features"):
        lfp.LocalFeatures(640, 480, 100)


def test_cpp_header_compiles_against_the_library(tmp_path):
    """include/local_features.hpp (C++ mirror of the reference crate's API) builds with plain g++ and links."""
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.join(ROOT, "local-features_amd")
    if not os.path.exists(os.path.join(lib_dir, "liblf_mkd.so")):
        pytest.skip("liblf_mkd.so not built")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "demo_local_features.cpp"), "-L", lib_dir, "-llf_mkd",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(tmp_path / "demo")])


def test_c_header_is_plain_c99():
    """include/lf_mkd.h is a C ABI: it must parse as C99 (no C++-isms) as well as C++."""
    import subprocess
    from conftest import ROOT
    hdr = os.path.join(ROOT, "include", "lf_mkd.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
