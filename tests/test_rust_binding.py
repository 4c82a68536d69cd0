"""The Rust side of the boundary (bindings/rust/) against include/lf_mkd.h, mechanically (no Rust toolchain exists in
the build image, so the binding cannot be compiled here): every `extern "C"` item of ffi.rs has the header's prototype,
every prototype has an item, the #[repr(C)] structs and the ctypes structs have the header's layout (offsetof / sizeof
compiled from the header with gcc), the constants have the header's values, and mod.rs calls nothing ffi.rs lacks."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

import local_features_python as lfp

HDR = os.path.join(ROOT, "include", "lf_mkd.h")
FFI = os.path.join(ROOT, "bindings", "rust", "local_features", "src", "hip", "ffi.rs")
MOD = os.path.join(ROOT, "bindings", "rust", "local_features", "src", "hip", "mod.rs")

C_SCALARS = {"uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "uint8_t": "u8", "float": "f32", "double": "f64", "int": "c_int",
             "char": "c_char", "void": "c_void"}


def _strip_comments(text, rust=False):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text) if rust else text


def _c_type(t):
    """'const float *' -> '*const f32', 'lf_mkd **' -> '*mut *mut lf_mkd', 'uint64_t' -> 'u64'."""
    t = t.strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [w for w in base if w not in ("const", "struct")]
    assert len(base) == 1, t
    name = C_SCALARS.get(base[0], base[0])
    if stars == 0:
        return name
    # only the innermost pointee can be const in this header (const T *, T **)
    out = ("*const " if const else "*mut ") + name
    for _ in range(stars - 1):
        out = "*mut " + out
    return out


def header_prototypes():
    text = _strip_comments(open(HDR).read())
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(lf_mkd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)          # type, then the parameter's name
                params.append(_c_type(mm.group(1)))
        ret = ret.strip()
        protos[name] = (None if ret == "void" else _c_type(ret), params)
    return protos


def rust_items():
    text = _strip_comments(open(FFI).read(), rust=True)
    block = re.search(r'extern\s+"C"\s*\{(.*)\}', text, flags=re.S).group(1)
    items = {}
    for m in re.finditer(r"pub\s+fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        items[name] = (None if ret is None else " ".join(ret.split()), params)
    return items


def test_every_extern_item_has_the_headers_prototype():
    protos, items = header_prototypes(), rust_items()
    assert sorted(protos) == sorted(lfp.SYMBOLS)                       # the parser sees the whole header
    assert sorted(items) == sorted(protos), (set(protos) ^ set(items))
    for name, (ret, params) in protos.items():
        r_ret, r_params = items[name]
        assert r_ret == ret, (name, "return", r_ret, ret)
        assert r_params == params, (name, r_params, params)


def test_ctypes_prototypes_have_the_headers_arity_and_classes():
    """The Python binding declares argtypes by hand too: same number of arguments, pointers where the header has pointers."""
    L = lfp.load_library()
    for name, (ret, params) in header_prototypes().items():
        fn = getattr(L, name)
        if fn.argtypes is None:
            assert params == [], name                                   # lf_mkd_version()
            continue
        assert len(fn.argtypes) == len(params), (name, len(fn.argtypes), len(params))
        for i, (ct, want) in enumerate(zip(fn.argtypes, params)):
            is_ptr = ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, "contents")
            assert is_ptr == want.startswith("*"), (name, i, ct, want)
            if not is_ptr:
                assert {"u64": ctypes.c_uint64, "u32": ctypes.c_uint32, "i32": ctypes.c_int32, "f32": ctypes.c_float}[want] is ct, (name, i)


def test_the_backend_type_only_calls_declared_functions():
    items = rust_items()
    src = _strip_comments(open(MOD).read(), rust=True)
    called = set(re.findall(r"ffi::(lf_mkd_[a-z0-9_]+)\s*\(", src))
    assert called and called <= set(items), called - set(items)
    for must in ("lf_mkd_create", "lf_mkd_destroy", "lf_mkd_detect", "lf_mkd_detect_extrema", "lf_mkd_orient_keypoints",
                 "lf_mkd_describe_keypoints", "lf_mkd_match", "lf_mkd_last_error"):
        assert must in called, must
    # the reference's three public methods, a FeaturesResult each, and no Vulkan object inside
    for method in ("detect_extract_all", "detect_top_n", "detect"):
        assert re.search(r"pub fn %s\s*\(&mut self.*?\)\s*->\s*Result<FeaturesResult, Error>" % method, src, flags=re.S), method
    struct = re.search(r"pub struct LocalFeaturesHip\s*\{(.*?)\}", src, flags=re.S).group(1)
    assert "Vulkan" not in struct and "vulkan" not in struct


def _rust_structs():
    text = _strip_comments(open(FFI).read(), rust=True)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if not f:
                continue
            name, ty = [x.strip() for x in f.replace("pub ", "").split(":")]
            fields.append((name, ty))
        out[m.group(1)] = fields
    return out


def _repr_c_layout(fields):
    """Offsets of a #[repr(C)] struct of u32 / i32 / f32 / [u32; N] fields."""
    off, res, align = 0, {}, 1
    for name, ty in fields:
        m = re.match(r"\[(\w+);\s*(\d+)\]", ty)
        elem, count = (m.group(1), int(m.group(2))) if m else (ty, 1)
        size = {"u32": 4, "i32": 4, "f32": 4, "u64": 8, "f64": 8, "u8": 1}[elem]
        off = (off + size - 1) // size * size
        res[name] = off
        off += size * count
        align = max(align, size)
    return res, (off + align - 1) // align * align


STRUCTS = {
    "lf_mkd_params": ["max_image_width", "max_image_height", "max_features", "patch_scale_factor", "device", "angle_mode",
                      "pool_mode", "flags", "max_frames", "n_scales", "max_blobs", "reserved"],
    "lf_mkd_keypoint": ["x", "y", "size", "angle", "response"],
    "lf_mkd_extremum": ["x", "y", "size", "response"],
}
CONSTANTS = ["LF_MKD_OK", "LF_MKD_ERR_BAD_ARG", "LF_MKD_ERR_HIP", "LF_MKD_ERR_IO", "LF_MKD_ERR_NO_IMAGE",
             "LF_MKD_ERR_NO_DEVICE", "LF_MKD_ERR_COMM", "LF_MKD_COMM_ID_BYTES", "LF_MKD_GATHER_DIRECT", "LF_MKD_GATHER_RING",
             "LF_MKD_ANGLE_SHADER", "LF_MKD_ANGLE_EXACT", "LF_MKD_ANGLE_EXACT_ZERO",
             "LF_MKD_POOL_DEFAULT", "LF_MKD_POOL_F16X3", "LF_MKD_POOL_F32", "LF_MKD_POOL_F16_FP6", "LF_MKD_FLAG_KERNEL_TIMING", "LF_MKD_FLAG_UNFUSED_KEYPOINTS", "LF_MKD_FLAG_DETECT_STEPWISE",
             "LF_MKD_MAX_ANGLES_PER_EXTREMUM", "LF_MKD_PCA_LIBERTY", "LF_MKD_PCA_NOTREDAME", "LF_MKD_PCA_YOSEMITE",
             "LF_MKD_PATCH_SIZE", "LF_MKD_RAW_LEN", "LF_MKD_DESC_LEN"]


@pytest.fixture(scope="module")
def header_facts(tmp_path_factory):
    """offsetof / sizeof of every struct field and the value of every constant, as gcc sees the header."""
    d = tmp_path_factory.mktemp("layout")
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{HDR}"', "int main(void) {"]
    for s, fields in STRUCTS.items():
        lines.append(f'  printf("sizeof {s} %zu\\n", sizeof({s}));')
        for f in fields:
            lines.append(f'  printf("offsetof {s} {f} %zu\\n", offsetof({s}, {f}));')
    for c in CONSTANTS:
        lines.append(f'  printf("const {c} %ld\\n", (long)({c}));')
    lines += ["  return 0;", "}"]
    src = d / "layout.c"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(d / "layout")])
    facts = {"sizeof": {}, "offsetof": {}, "const": {}}
    for line in subprocess.check_output([str(d / "layout")], text=True).splitlines():
        w = line.split()
        if w[0] == "sizeof":
            facts["sizeof"][w[1]] = int(w[2])
        elif w[0] == "offsetof":
            facts["offsetof"][(w[1], w[2])] = int(w[3])
        else:
            facts["const"][w[1]] = int(w[2])
    return facts


def test_struct_layouts_match_the_header_field_by_field(header_facts):
    rust = _rust_structs()
    for s, fields in STRUCTS.items():
        assert [n for n, _ in rust[s]] == fields, s                    # same fields, same order
        offs, size = _repr_c_layout(rust[s])
        assert size == header_facts["sizeof"][s], (s, size)
        for f in fields:
            assert offs[f] == header_facts["offsetof"][(s, f)], (s, f)
    # the ctypes binding's structure and the numpy keypoint record, against the same facts
    P = lfp._lib.Params
    assert ctypes.sizeof(P) == header_facts["sizeof"]["lf_mkd_params"]
    assert [n for n, _ in P._fields_] == STRUCTS["lf_mkd_params"]
    for f in STRUCTS["lf_mkd_params"]:
        assert getattr(P, f).offset == header_facts["offsetof"][("lf_mkd_params", f)], f
    K = lfp.KEYPOINT_DTYPE
    assert K.itemsize == header_facts["sizeof"]["lf_mkd_keypoint"] and list(K.names) == STRUCTS["lf_mkd_keypoint"]
    for f in K.names:
        assert K.fields[f][1] == header_facts["offsetof"][("lf_mkd_keypoint", f)] and K.fields[f][0] == np.dtype("<f4")
    assert header_facts["sizeof"]["lf_mkd_extremum"] == 16


def test_constants_match_the_header(header_facts):
    c = header_facts["const"]
    text = _strip_comments(open(FFI).read(), rust=True)
    rust = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (\w+)\s*:\s*[\w:]+\s*=\s*(-?\d+)\s*;", text)}
    for name, value in rust.items():
        assert c[name] == value, name
    assert {"LF_MKD_POOL_DEFAULT", "LF_MKD_POOL_F16X3", "LF_MKD_POOL_F32", "LF_MKD_ERR_NO_DEVICE"} <= set(rust)
    assert (lfp.ANGLE_SHADER, lfp.ANGLE_EXACT, lfp.ANGLE_EXACT_ZERO) == (c["LF_MKD_ANGLE_SHADER"], c["LF_MKD_ANGLE_EXACT"],
                                                                        c["LF_MKD_ANGLE_EXACT_ZERO"])
    assert (lfp.POOL_DEFAULT, lfp.POOL_F16X3, lfp.POOL_F32) == (c["LF_MKD_POOL_DEFAULT"], c["LF_MKD_POOL_F16X3"],
                                                                c["LF_MKD_POOL_F32"])
    assert lfp.FLAG_KERNEL_TIMING == c["LF_MKD_FLAG_KERNEL_TIMING"]
    assert [c["LF_MKD_PCA_LIBERTY"], c["LF_MKD_PCA_NOTREDAME"], c["LF_MKD_PCA_YOSEMITE"]] == [0, 1, 2]
    assert (c["LF_MKD_PATCH_SIZE"], c["LF_MKD_RAW_LEN"], c["LF_MKD_DESC_LEN"]) == (32, 238, 128)
