"""ctypes front end of the CPU oracle (oracle/mkd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (local-features_amd/) never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmkd_oracle.so")

ATAN_SHADER = 0
ATAN_LIBM = 1
BLUR_CONTRACT = 2   # OR into atan_mode: the blur's mul+add contracted to fma (both readings are the reference)

_fp = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _ptr(a):
    return a.ctypes.data_as(_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class MkdOracle:
    """CPU restatement of the reference MKD path for one PCA model."""

    def __init__(self, pca_path):
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        self.L = L
        L.mkd_oracle_sizeof_consts.restype = ctypes.c_ulong
        L.mkd_oracle_pyramid_floats.restype = ctypes.c_long
        L.mkd_oracle_pyramid_floats.argtypes = [ctypes.c_int, ctypes.c_int]
        L.mkd_oracle_pyramid_levels.argtypes = [ctypes.c_int, ctypes.c_int]
        L.mkd_oracle_atan2_shader.restype = ctypes.c_float
        L.mkd_oracle_atan2_shader.argtypes = [ctypes.c_float, ctypes.c_float]
        L.mkd_oracle_describe_patches.argtypes = [
            ctypes.c_void_p, _fp, ctypes.c_long, _fp, _fp, ctypes.c_int, ctypes.c_int]
        L.mkd_cpu_fast_describe_patches.argtypes = [ctypes.c_void_p, _fp, ctypes.c_long, _fp, ctypes.c_int, ctypes.c_int]
        L.mkd_oracle_sample_patches.argtypes = [
            _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_long, ctypes.c_float, _fp]
        L.mkd_oracle_sample_patches_reading.argtypes = [
            _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_long, ctypes.c_float, ctypes.c_int, _fp]
        L.mkd_oracle_build_pyramid.argtypes = [_fp, ctypes.c_int, ctypes.c_int, _fp]
        L.mkd_oracle_patch_gradients.argtypes = [_fp, _fp, _fp, ctypes.c_int]
        L.mkd_oracle_dog.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
        L.mkd_oracle_scan_extrema.restype = ctypes.c_long
        L.mkd_oracle_scan_extrema.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_float, _fp, ctypes.c_long, ctypes.POINTER(ctypes.c_long)]
        L.mkd_oracle_topk_filter.restype = ctypes.c_long
        L.mkd_oracle_topk_filter.argtypes = [_fp, ctypes.c_long, ctypes.c_long, ctypes.c_float,
                                             ctypes.POINTER(ctypes.c_uint)]
        L.mkd_oracle_match.argtypes = [_fp, ctypes.c_long, _fp, ctypes.c_long, ctypes.c_float, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, _fp, _fp, ctypes.c_int]
        L.mkd_oracle_quirk_pixels.argtypes = [_fp, ctypes.c_float]
        L.mkd_oracle_build_coarse_stack.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
        L.mkd_oracle_orient.restype = ctypes.c_long
        L.mkd_oracle_orient.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_long, _fp,
                                        ctypes.c_long]
        self.mean = np.zeros(238, np.float32)
        self.eigvals = np.zeros(238, np.float32)
        self.eigvecs = np.zeros((238, 238), np.float32)
        rc = L.mkd_oracle_load_pca(pca_path.encode(), _ptr(self.mean), _ptr(self.eigvals),
                                   _ptr(self.eigvecs))
        if rc != 0:
            raise RuntimeError(f"oracle: cannot load PCA model {pca_path} (rc={rc})")
        self._consts = np.zeros(L.mkd_oracle_sizeof_consts() // 4, np.float32)
        L.mkd_oracle_build_consts(_ptr(self.mean), _ptr(self.eigvals), _ptr(self.eigvecs),
                                  self._consts.ctypes.data_as(ctypes.c_void_p))

    # ConstantData fields (shaders/common.glsl:34-40)
    @property
    def gradient_angle(self):
        return self._consts[0:1024].reshape(32, 32)

    @property
    def embedding_polar(self):
        return self._consts[1024:1024 + 25 * 1024].reshape(25, 32, 32)

    @property
    def embedding_cartesian(self):
        o = 1024 + 25 * 1024
        return self._consts[o:o + 9 * 1024].reshape(9, 32, 32)

    @property
    def mean_vec(self):
        o = 1024 + 34 * 1024
        return self._consts[o:o + 238]

    @property
    def eigen_vecs(self):
        o = 1024 + 34 * 1024 + 238
        return self._consts[o:o + 128 * 238].reshape(128, 238)

    def atan2_shader(self, x, y):
        return self.L.mkd_oracle_atan2_shader(float(x), float(y))

    def patch_gradients(self, patch, atan_mode=ATAN_SHADER):
        p = _f32(patch).reshape(32, 32)
        mag = np.zeros((32, 32), np.float32)
        ang = np.zeros((32, 32), np.float32)
        self.L.mkd_oracle_patch_gradients(_ptr(p), _ptr(mag), _ptr(ang), atan_mode)
        return mag, ang

    def describe_patches(self, patches, atan_mode=ATAN_SHADER, nthreads=1, want_raw=False):
        p = _f32(patches).reshape(-1, 32, 32)
        n = p.shape[0]
        desc = np.zeros((n, 128), np.float32)
        raw = np.zeros((n, 238), np.float32) if want_raw else None
        self.L.mkd_oracle_describe_patches(
            self._consts.ctypes.data_as(ctypes.c_void_p), _ptr(p), n, _ptr(desc),
            _ptr(raw) if want_raw else None, atan_mode, nthreads)
        return (desc, raw) if want_raw else desc

    def describe_patches_fast(self, patches, atan_mode=ATAN_SHADER, nthreads=1):
        """The CPU-organised port (mkd_cpu_fast.c): what bench.py times as cpu_baseline.  Not a parity reference; held to
        the oracle by tests/test_oracle.py."""
        p = _f32(patches).reshape(-1, 32, 32)
        desc = np.zeros((p.shape[0], 128), np.float32)
        self.L.mkd_cpu_fast_describe_patches(self._consts.ctypes.data_as(ctypes.c_void_p), _ptr(p), p.shape[0], _ptr(desc),
                                             atan_mode, nthreads)
        return desc

    def quirk_pixels(self, patches, tol=2e-7):
        """Per patch: pixels with gy != 0 and |gx| <= tol, i.e. on the x == 0 discontinuity of atan2.glsl."""
        p = _f32(patches).reshape(-1, 32, 32)
        return np.array([self.L.mkd_oracle_quirk_pixels(_ptr(p[i]), tol) for i in range(len(p))], np.int32)

    def pyramid_levels(self, w, h):
        return self.L.mkd_oracle_pyramid_levels(w, h)

    def build_pyramid(self, img):
        img = _f32(img)
        h, w = img.shape
        pyr = np.zeros(self.L.mkd_oracle_pyramid_floats(w, h), np.float32)
        self.L.mkd_oracle_build_pyramid(_ptr(img), w, h, _ptr(pyr))
        return pyr

    def split_pyramid(self, pyr, w, h):
        out, off = [], 0
        for l in range(self.pyramid_levels(w, h)):
            lw, lh = max(w >> l, 1), max(h >> l, 1)
            out.append(pyr[off:off + lw * lh].reshape(lh, lw))
            off += lw * lh
        return out

    def sample_patches(self, pyr, w, h, kps, patch_scale_factor=24.0, contract=False):
        """kps: [n,4] (x, y, size, angle_deg).  contract: the reference's other reading of the sample position
        (patch_gradients.glsl:60-67 without `precise`: mul+add fused into fma), up to one coordinate ulp away."""
        k = _f32(kps).reshape(-1, 4)
        n = k.shape[0]
        patches = np.zeros((n, 32, 32), np.float32)
        self.L.mkd_oracle_sample_patches_reading(_ptr(pyr), w, h, _ptr(k), n, patch_scale_factor, int(bool(contract)),
                                                 _ptr(patches))
        return patches

    def build_coarse_stack(self, img, n_scales=4):
        """[n_scales+3][h][w]: the a-trous stack keypoint orientation reads (vulkan/mod.rs:1093)."""
        img = _f32(img)
        h, w = img.shape
        stack = np.zeros((n_scales + 3, h, w), np.float32)
        self.L.mkd_oracle_build_coarse_stack(_ptr(img), w, h, n_scales + 3, _ptr(stack))
        return stack

    def orient(self, stack, extrema):
        """extrema [n,4] (x, y, size, response) -> keypoints [m,5] (x, y, size, angle_deg, response)."""
        e = _f32(extrema).reshape(-1, 4)
        nl, h, w = stack.shape
        out = np.zeros((36 * max(len(e), 1), 5), np.float32)
        m = self.L.mkd_oracle_orient(_ptr(stack), w, h, nl, _ptr(e), len(e), _ptr(out), len(out))
        return out[:m].copy()

    def dog(self, stack):
        """fine[l] = coarse[l] - coarse[l+1] (swt_sub.glsl)."""
        st = _f32(stack)
        nl, h, w = st.shape
        fine = np.zeros((nl - 1, h, w), np.float32)
        self.L.mkd_oracle_dog(_ptr(st), w, h, nl, _ptr(fine))
        return fine

    def scan_extrema(self, fine, border=5, skip_layers=0, contrast_threshold=0.035, max_out=None):
        """DoG volume -> (extrema [m,4] (x, y, size, contrast) in cube-raster order, total found)."""
        f = _f32(fine)
        nf, h, w = f.shape
        cap = int(max_out) if max_out is not None else 1 << 20
        out = np.zeros((max(cap, 1), 4), np.float32)
        total = ctypes.c_long()
        m = self.L.mkd_oracle_scan_extrema(_ptr(f), w, h, nf, border, skip_layers, contrast_threshold, _ptr(out), cap,
                                           ctypes.byref(total))
        return out[:m].copy(), total.value

    def topk_filter(self, extrema, n, min_size=0.0):
        """TopKContrastFilter (mod.rs:1753-1786): indices of the blobs kept, in index order."""
        e = _f32(extrema).reshape(-1, 4)
        idx = np.zeros(max(len(e), 1), np.uint32)
        m = self.L.mkd_oracle_topk_filter(_ptr(e), len(e), n, min_size, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint)))
        return idx[:m].copy()

    def detect(self, img, n_scales=4, top_n=None, min_size=0.0, max_blobs=8000, max_features=None, **kw):
        """LocalFeaturesVulkan::detect (mod.rs:363-593) end to end: (keypoints [m,5], descriptors [m,128])."""
        img = _f32(img)
        st = self.build_coarse_stack(img, n_scales)
        max_extrema = 256 * ((max_blobs + 255) // 256)            # mod.rs:279-286
        ex, _ = self.scan_extrema(self.dog(st), max_out=max_extrema)
        if top_n is not None:
            ex = ex[self.topk_filter(ex, top_n, min_size)]
        kps = self.orient(st, ex)
        if max_features is not None:
            kps = kps[:max_features]
        return kps, self.describe_keypoints(img, kps[:, :4], **kw)

    def match(self, a, b, ratio=0.8, exclude=None, nthreads=8):
        """match_features (examples/match_images/src/main.rs:8-27): (match [na] int32 (-1 = none), best, second).
        exclude = (lo, hi) uint32 arrays: b[lo[i]:hi[i]] is skipped for a[i]."""
        a, b = _f32(a).reshape(-1, 128), _f32(b).reshape(-1, 128)
        m = np.zeros(len(a), np.int32)
        s1, s2 = np.zeros(len(a), np.float32), np.zeros(len(a), np.float32)
        lo = hi = None
        if exclude is not None:
            lo, hi = (np.ascontiguousarray(x, np.uint32) for x in exclude)
        self.L.mkd_oracle_match(_ptr(a), len(a), _ptr(b), len(b), ratio, lo.ctypes.data if lo is not None else None,
                                hi.ctypes.data if hi is not None else None, m.ctypes.data, _ptr(s1), _ptr(s2), nthreads)
        return m, s1, s2

    def describe_keypoints(self, img, kps, patch_scale_factor=24.0, **kw):
        img = _f32(img)
        h, w = img.shape
        pyr = self.build_pyramid(img)
        return self.describe_patches(self.sample_patches(pyr, w, h, kps, patch_scale_factor), **kw)
