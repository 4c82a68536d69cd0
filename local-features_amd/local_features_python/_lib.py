"""ctypes binding of liblf_mkd.so (include/lf_mkd.h).  No fallback: if the HIP library is
missing or no gfx950 device is visible, construction raises."""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("LF_MKD_LIB", os.path.join(_ROOT, "liblf_mkd.so"))   # override: A/B builds
MODEL_DIR = os.path.join(_ROOT, "models", "mkd")

FLAG_KERNEL_TIMING = 1
FLAG_UNFUSED_KEYPOINTS = 2
FLAG_DETECT_STEPWISE = 4
ANGLE_SHADER, ANGLE_EXACT, ANGLE_EXACT_ZERO = 0, 1, 2
POOL_DEFAULT, POOL_F16X3, POOL_F32, POOL_F16_FP6 = 0, 1, 2, 3   # lf_mkd_pool_mode; the default is the f16x3 split
PCA_NAMES = ("liberty", "notredame", "yosemite")   # enum MKDPCA, lib.rs:26-32

# every symbol include/lf_mkd.h declares
SYMBOLS = (
    "lf_mkd_create", "lf_mkd_create_from_file", "lf_mkd_destroy", "lf_mkd_last_error",
    "lf_mkd_describe_patches", "lf_mkd_describe_patches_device", "lf_mkd_raw_descriptors_device",
    "lf_mkd_set_image", "lf_mkd_set_image_device", "lf_mkd_set_images_device", "lf_mkd_describe_keypoints",
    "lf_mkd_set_image_u8", "lf_mkd_set_images_u8_device", "lf_mkd_detect_u8", "lf_mkd_detect_times",
    "lf_mkd_describe_keypoints_frames_device",
    "lf_mkd_describe_keypoints_device", "lf_mkd_sample_patches_device", "lf_mkd_get_pyramid_level",
    "lf_mkd_get_pyramid_level_apron",
    "lf_mkd_build_constants", "lf_mkd_kernel_times", "lf_mkd_kernel_clock", "lf_mkd_synchronize", "lf_mkd_version",
    "lf_mkd_orient_keypoints", "lf_mkd_orient_keypoints_device", "lf_mkd_get_coarse_layer",
    "lf_mkd_detect_extrema", "lf_mkd_detect_extrema_device", "lf_mkd_filter_extrema_device", "lf_mkd_detect",
    "lf_mkd_match", "lf_mkd_match_device", "lf_mkd_match_both_device", "lf_mkd_match_overflowed", "lf_mkd_stream_create", "lf_mkd_stream_frame",
    "lf_mkd_detect_frames_device", "lf_mkd_orient_keypoints_blocked",
    "lf_mkd_comm_unique_id", "lf_mkd_comm_create", "lf_mkd_comm_destroy", "lf_mkd_comm_info", "lf_mkd_allgather_descriptors",
    "lf_mkd_comm_loopback", "lf_mkd_comm_last_form", "lf_mkd_plan_upload", "lf_mkd_detect_recordings",
)
COMM_ID_BYTES = 128
GATHER_DIRECT, GATHER_RING = 0, 1


class Params(ctypes.Structure):
    _fields_ = [
        ("max_image_width", ctypes.c_uint32), ("max_image_height", ctypes.c_uint32),
        ("max_features", ctypes.c_uint32), ("patch_scale_factor", ctypes.c_float),
        ("device", ctypes.c_int32), ("angle_mode", ctypes.c_int32), ("pool_mode", ctypes.c_int32),
        ("flags", ctypes.c_uint32), ("max_frames", ctypes.c_uint32), ("n_scales", ctypes.c_uint32),
        ("max_blobs", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 1),
    ]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4")])

_lib = None


def model_path(pca):
    if pca not in PCA_NAMES:
        raise RuntimeError("Invalid PCA argument")   # python/src/lib.rs:60-64
    return os.path.join(MODEL_DIR, f"concat-pca-{pca}.safetensors")


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `make -C local-features_amd` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64, and a
    # process that loads the system copy first and torch's second ends up with two HSA runtimes (torch
    # then sees no device).  Callers hand us torch device pointers and streams, so let torch load its
    # runtime first; liblf_mkd.so's libamdhip64.so.7 dependency then binds to the copy already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32
    L.lf_mkd_version.restype = ctypes.c_char_p
    L.lf_mkd_last_error.restype = ctypes.c_char_p
    L.lf_mkd_last_error.argtypes = [vp]
    L.lf_mkd_create.argtypes = [ctypes.POINTER(Params), vp, vp, vp, ctypes.POINTER(vp)]
    L.lf_mkd_create_from_file.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.POINTER(vp)]
    L.lf_mkd_destroy.argtypes = [vp]
    L.lf_mkd_destroy.restype = None
    L.lf_mkd_describe_patches.argtypes = [vp, vp, u64, vp]
    L.lf_mkd_describe_patches_device.argtypes = [vp, vp, u64, vp, vp]
    L.lf_mkd_raw_descriptors_device.argtypes = [vp, vp, u64, vp, vp]
    L.lf_mkd_set_image.argtypes = [vp, vp, u32, u32]
    L.lf_mkd_set_image_device.argtypes = [vp, vp, u32, u32, vp]
    L.lf_mkd_set_images_device.argtypes = [vp, vp, u32, u32, u32, vp]
    L.lf_mkd_set_image_u8.argtypes = [vp, vp, u32, u32]
    L.lf_mkd_set_images_u8_device.argtypes = [vp, vp, u32, u32, u32, vp]
    L.lf_mkd_describe_keypoints_frames_device.argtypes = [vp, vp, vp, u64, vp, vp]
    L.lf_mkd_describe_keypoints.argtypes = [vp, vp, u64, vp]
    L.lf_mkd_describe_keypoints_device.argtypes = [vp, vp, u64, vp, vp]
    L.lf_mkd_sample_patches_device.argtypes = [vp, vp, u64, vp, vp]
    L.lf_mkd_get_pyramid_level.argtypes = [vp, u32, vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.lf_mkd_get_pyramid_level_apron.argtypes = [vp, u32, vp, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.lf_mkd_kernel_clock.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.lf_mkd_synchronize.argtypes = [vp]
    L.lf_mkd_orient_keypoints.argtypes = [vp, vp, u64, vp, u64, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.lf_mkd_orient_keypoints_device.argtypes = [vp, vp, vp, u64, vp, vp, u64, ctypes.POINTER(u64),
                                                 ctypes.POINTER(u64), vp]
    L.lf_mkd_get_coarse_layer.argtypes = [vp, u32, vp]
    pu64 = ctypes.POINTER(u64)
    L.lf_mkd_detect_extrema.argtypes = [vp, vp, u64, pu64, pu64]
    L.lf_mkd_detect_extrema_device.argtypes = [vp, vp, vp, u64, pu64, pu64, vp]
    L.lf_mkd_filter_extrema_device.argtypes = [vp, vp, u64, u32, ctypes.c_float, vp, vp, pu64, vp]
    L.lf_mkd_detect.argtypes = [vp, vp, u32, u32, u32, ctypes.c_float, vp, vp, u64, pu64, pu64, pu64]
    L.lf_mkd_detect_u8.argtypes = L.lf_mkd_detect.argtypes
    L.lf_mkd_detect_times.argtypes = [vp] + [ctypes.POINTER(ctypes.c_double)] * 3
    L.lf_mkd_match_device.argtypes = [vp, vp, u64, vp, u64, vp, vp, ctypes.c_float, vp, vp, vp, vp]
    L.lf_mkd_match.argtypes = [vp, vp, u64, vp, u64, ctypes.c_float, vp]
    L.lf_mkd_match_both_device.argtypes = [vp, vp, u64, vp, u64, ctypes.c_float, vp, vp, vp]
    L.lf_mkd_match_overflowed.argtypes = [vp, vp, ctypes.POINTER(u64)]
    L.lf_mkd_stream_create.argtypes = [vp, u32, u32, u32, ctypes.c_float, u64, vp, vp, vp, vp]
    L.lf_mkd_stream_frame.argtypes = [vp, vp]
    L.lf_mkd_orient_keypoints_blocked.argtypes = [vp, vp, u64, u32, vp, u64, vp, vp, vp, u64, pu64, pu64]
    L.lf_mkd_detect_frames_device.argtypes = [vp, vp, u32, u32, u32, u32, ctypes.c_float, vp, vp, vp, u64, pu64, pu64,
                                              pu64, vp]
    L.lf_mkd_build_constants.argtypes = [vp] * 7
    i32 = ctypes.c_int32
    L.lf_mkd_comm_unique_id.argtypes = [vp]
    L.lf_mkd_comm_create.argtypes = [vp, vp, i32, i32, ctypes.POINTER(vp)]
    L.lf_mkd_comm_destroy.argtypes = [vp]
    L.lf_mkd_comm_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.lf_mkd_allgather_descriptors.argtypes = [vp, vp, vp, vp, i32, vp]
    L.lf_mkd_comm_loopback.argtypes = [vp, vp, vp, vp, u64, vp]
    L.lf_mkd_comm_last_form.argtypes = [vp]
    L.lf_mkd_detect_recordings.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.lf_mkd_plan_upload.argtypes = [u32, u32, u32, u32, vp, u32, ctypes.POINTER(u32), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double)]
    L.lf_mkd_kernel_times.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                      ctypes.POINTER(u64)]
    _lib = L
    return L


def plan_upload(width, height, bytes_per_pixel=4, n_scales=4):
    """(cuts, modelled_us, one_piece_us): the pieces lf_mkd_detect[_u8] would upload a frame of this size in (lf_mkd_plan_upload;
    needs no device)."""
    cuts = (ctypes.c_uint32 * 16)()
    n, a, b = ctypes.c_uint32(), ctypes.c_double(), ctypes.c_double()
    rc = load_library().lf_mkd_plan_upload(width, height, bytes_per_pixel, n_scales, cuts, 16, ctypes.byref(n), ctypes.byref(a),
                                           ctypes.byref(b))
    if rc != 0:
        raise RuntimeError(f"lf_mkd_plan_upload failed ({rc})")
    return [int(cuts[i]) for i in range(min(n.value, 16))], a.value, b.value


def comm_unique_id():
    """Rank 0: the 128-byte RCCL identifier the other ranks need for Comm(...)."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    rc = load_library().lf_mkd_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"lf_mkd_comm_unique_id failed ({rc}): RCCL is not loadable")
    return bytes(buf)


class Comm:
    """One rank's RCCL communicator on a handle's device (lf_mkd_comm_create): the all-gather of descriptor shards, the
    path's one collective, through the C boundary."""

    def __init__(self, handle, unique_id, n_ranks, rank):
        self._c = None
        self.handle, self.n_ranks, self.rank = handle, n_ranks, rank
        self.calls = 0                  # lf_mkd_allgather_descriptors calls made through this communicator
        c = ctypes.c_void_p()
        ident = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        handle._check(handle.L.lf_mkd_comm_create(handle._h, ident, n_ranks, rank, ctypes.byref(c)), "lf_mkd_comm_create")
        self._c = c

    def info(self):
        """(RCCL version code, ranks, this rank)"""
        v, n, r = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self.handle._check(self.handle.L.lf_mkd_comm_info(self._c, ctypes.byref(v), ctypes.byref(n), ctypes.byref(r)),
                           "lf_mkd_comm_info")
        return v.value, n.value, r.value

    def allgather_descriptors(self, d_buf, counts, mode=GATHER_DIRECT, stream=None):
        """d_buf: device pointer of the [sum(counts)][128] f32 gathered set, this rank's rows already at their offset."""
        arr = (ctypes.c_uint64 * len(counts))(*[int(c) for c in counts])
        self.calls += 1
        self.handle._device_call(stream, lambda s: self.handle.L.lf_mkd_allgather_descriptors(
            self.handle._h, self._c, arr, d_buf, mode, s), "lf_mkd_allgather_descriptors")

    def loopback(self, d_src, d_dst, n_rows, stream=None):
        """n_rows descriptors from d_src to this very rank and back into d_dst: one group of ncclSend + ncclRecv through
        the posting routine of the direct gather (lf_mkd_comm_loopback)."""
        self.handle._device_call(stream, lambda s: self.handle.L.lf_mkd_comm_loopback(
            self.handle._h, self._c, d_src, d_dst, n_rows, s), "lf_mkd_comm_loopback")

    def last_form(self):
        """GATHER_DIRECT / GATHER_RING as the latest gather ran (-1: none yet)"""
        return self.handle.L.lf_mkd_comm_last_form(self._c)

    def close(self):
        if self._c is not None:
            self.handle.L.lf_mkd_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        self.close()


class MkdHandle:
    """Thin owner of one lf_mkd handle.  Host arrays are numpy; device arguments are raw
    pointers (ints), e.g. torch.Tensor.data_ptr()."""

    def __init__(self, pca="liberty", max_features=2000, max_image_width=0, max_image_height=0,
                 patch_scale_factor=24.0, device=0, angle_mode=ANGLE_SHADER, pool_mode=POOL_DEFAULT, flags=0,
                 max_frames=1, n_scales=4, max_blobs=8000):
        self._h = None
        self.n_scales = n_scales
        self.max_features = max_features
        self.L = load_library()
        p = Params(max_image_width=max_image_width, max_image_height=max_image_height,
                   max_features=max_features, patch_scale_factor=patch_scale_factor,
                   device=device, angle_mode=angle_mode, pool_mode=pool_mode, flags=flags,
                   max_frames=max_frames, n_scales=n_scales, max_blobs=max_blobs)
        h = ctypes.c_void_p()
        rc = self.L.lf_mkd_create_from_file(ctypes.byref(p), model_path(pca).encode(), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"lf_mkd_create failed ({rc}): {self.L.lf_mkd_last_error(None).decode()}")
        self._h = h

    def close(self):
        if self._h is not None:
            self.L.lf_mkd_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.L.lf_mkd_last_error(self._h).decode()}")

    # --- host-pointer entry points -----------------------------------------------------------
    def describe_patches(self, patches):
        p = np.ascontiguousarray(patches, np.float32).reshape(-1, 32, 32)
        out = np.empty((p.shape[0], 128), np.float32)
        self._check(self.L.lf_mkd_describe_patches(self._h, p.ctypes.data, p.shape[0], out.ctypes.data),
                    "lf_mkd_describe_patches")
        return out

    def set_image(self, img):
        """f32 frame in [0, 1], or the uint8 luma it would be made from (1 B/px to the device, converted there: same bits)"""
        if img.ndim != 2:
            raise RuntimeError("image must be 2-D")
        if img.dtype == np.uint8:
            a = np.ascontiguousarray(img)
            self._check(self.L.lf_mkd_set_image_u8(self._h, a.ctypes.data, a.shape[1], a.shape[0]), "lf_mkd_set_image_u8")
            return
        a = np.ascontiguousarray(img, np.float32)
        self._check(self.L.lf_mkd_set_image(self._h, a.ctypes.data, a.shape[1], a.shape[0]), "lf_mkd_set_image")

    def describe_keypoints(self, kps):
        k = np.ascontiguousarray(kps)
        if k.dtype != KEYPOINT_DTYPE:
            k = np.ascontiguousarray(k, np.float32).reshape(-1, 5)
        n = k.shape[0]
        out = np.empty((n, 128), np.float32)
        self._check(self.L.lf_mkd_describe_keypoints(self._h, k.ctypes.data, n, out.ctypes.data),
                    "lf_mkd_describe_keypoints")
        return out

    def orient_keypoints(self, extrema, max_out=None):
        """extrema [n,4] (x, y, size, response) -> (keypoints [m,5] (x, y, size, angle_deg, response), dropped).
        Ordered by extremum, then histogram bin."""
        e = np.ascontiguousarray(extrema, np.float32).reshape(-1, 4)
        n = e.shape[0]
        cap = 18 * n if max_out is None else int(max_out)
        out = np.empty((max(cap, 1), 5), np.float32)
        m, dropped = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.L.lf_mkd_orient_keypoints(self._h, e.ctypes.data, n, out.ctypes.data, cap, ctypes.byref(m),
                                                   ctypes.byref(dropped)), "lf_mkd_orient_keypoints")
        return out[:m.value].copy(), dropped.value

    def detect_extrema(self, max_out=1 << 16):
        """Extrema of the loaded frame(s): ([m,4] (x, y, size, contrast), dropped).  Ordered by frame, scan cube, lane."""
        out = np.empty((max(max_out, 1), 4), np.float32)
        m, dropped = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.L.lf_mkd_detect_extrema(self._h, out.ctypes.data, max_out, ctypes.byref(m),
                                                 ctypes.byref(dropped)), "lf_mkd_detect_extrema")
        return out[:m.value].copy(), dropped.value

    def detect_into(self, img, top_n, min_size, kps, desc):
        """lf_mkd_detect / lf_mkd_detect_u8 (by the frame's dtype) into caller-held arrays kps [cap,5] f32, desc [cap,128] f32:
        (m, dropped_blobs, dropped_features).  What a caller that reuses its buffers pays: the C call and nothing else."""
        if img.ndim != 2:
            raise RuntimeError("image must be 2-D")
        cap = kps.shape[0]
        assert kps.dtype == np.float32 and desc.dtype == np.float32 and desc.shape[0] == cap and kps.flags.c_contiguous and desc.flags.c_contiguous
        m, db, df = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        if img.dtype == np.uint8:
            a = np.ascontiguousarray(img)
            fn, what = self.L.lf_mkd_detect_u8, "lf_mkd_detect_u8"
        else:
            a = np.ascontiguousarray(img, np.float32)
            fn, what = self.L.lf_mkd_detect, "lf_mkd_detect"
        self._check(fn(self._h, a.ctypes.data, a.shape[1], a.shape[0], top_n, min_size, kps.ctypes.data, desc.ctypes.data, cap,
                       ctypes.byref(m), ctypes.byref(db), ctypes.byref(df)), what)
        return m.value, db.value, df.value

    def detect_recordings(self):
        """(recorded pipelines held, of them with a banded upload, distinct requests remembered) -- lf_mkd_detect_recordings"""
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.L.lf_mkd_detect_recordings(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
                    "lf_mkd_detect_recordings")
        return a.value, b.value, c.value

    def detect_times(self):
        """(upload_ms, pipeline_ms, readback_ms) of the latest detect call; needs FLAG_KERNEL_TIMING."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self.L.lf_mkd_detect_times(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "lf_mkd_detect_times")
        return a.value, b.value, c.value

    def detect(self, img, top_n=0, min_size=0.0, max_out=None):
        """lf_mkd_detect (f32 frame) or lf_mkd_detect_u8 (uint8 frame): (keypoints [m,5], descriptors [m,128], dropped_blobs,
        dropped_features)."""
        cap = int(max_out if max_out is not None else self.max_features)
        kps = np.empty((max(cap, 1), 5), np.float32)
        desc = np.empty((max(cap, 1), 128), np.float32)
        if cap == 0:
            m, db, df = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            a = np.ascontiguousarray(img) if img.dtype == np.uint8 else np.ascontiguousarray(img, np.float32)
            fn = self.L.lf_mkd_detect_u8 if img.dtype == np.uint8 else self.L.lf_mkd_detect
            self._check(fn(self._h, a.ctypes.data, a.shape[1], a.shape[0], top_n, min_size, kps.ctypes.data, desc.ctypes.data, 0,
                           ctypes.byref(m), ctypes.byref(db), ctypes.byref(df)), "lf_mkd_detect")
            return kps[:0].copy(), desc[:0].copy(), db.value, df.value
        m, db, df = self.detect_into(img, top_n, min_size, kps, desc)
        return kps[:m].copy(), desc[:m].copy(), db, df

    def match(self, a, b, ratio=0.8):
        """match_features (examples/match_images/src/main.rs:8-27): int32 [na], index into b or -1."""
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 128)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 128)
        out = np.empty(len(a), np.int32)
        self._check(self.L.lf_mkd_match(self._h, a.ctypes.data, len(a), b.ctypes.data, len(b), ratio, out.ctypes.data),
                    "lf_mkd_match")
        return out

    def orient_keypoints_blocked(self, extremum_data, n_extrema, indices, max_out, block_len=256):
        """The reference's ExtremumLocations.data (blocked) + FilteredExtrema.indices in, KeypointIndices-style arrays out:
        (extremum index per keypoint, orientation per keypoint, keypoints [m,5])."""
        data = np.ascontiguousarray(extremum_data, np.float32)
        idx = np.ascontiguousarray(indices, np.uint32)
        kei = np.empty(max(max_out, 1), np.uint32)
        ori = np.empty(max(max_out, 1), np.float32)
        kps = np.empty((max(max_out, 1), 5), np.float32)
        m, dropped = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.L.lf_mkd_orient_keypoints_blocked(self._h, data.ctypes.data, n_extrema, block_len, idx.ctypes.data,
                                                           len(idx), kei.ctypes.data, ori.ctypes.data, kps.ctypes.data,
                                                           max_out, ctypes.byref(m), ctypes.byref(dropped)),
                    "lf_mkd_orient_keypoints_blocked")
        return kei[:m.value].copy(), ori[:m.value].copy(), kps[:m.value].copy(), dropped.value

    def coarse_layer(self, layer, width, height):
        out = np.empty((height, width), np.float32)
        self._check(self.L.lf_mkd_get_coarse_layer(self._h, layer, out.ctypes.data), "lf_mkd_get_coarse_layer")
        return out

    def pyramid_level(self, level):
        w, h = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.L.lf_mkd_get_pyramid_level(self._h, level, None, ctypes.byref(w), ctypes.byref(h)),
                    "lf_mkd_get_pyramid_level")
        out = np.empty((h.value, w.value), np.float32)
        self._check(self.L.lf_mkd_get_pyramid_level(self._h, level, out.ctypes.data, None, None),
                    "lf_mkd_get_pyramid_level")
        return out

    def pyramid_level_apron(self, level):
        """(level with its mirrored apron [(h + 2a), (w + 2a)], a)."""
        w, h, a = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.L.lf_mkd_get_pyramid_level_apron(self._h, level, None, ctypes.byref(w), ctypes.byref(h),
                                                          ctypes.byref(a)), "lf_mkd_get_pyramid_level_apron")
        out = np.empty((h.value + 2 * a.value, w.value + 2 * a.value), np.float32)
        self._check(self.L.lf_mkd_get_pyramid_level_apron(self._h, level, out.ctypes.data, None, None, None),
                    "lf_mkd_get_pyramid_level_apron")
        return out, a.value

    # --- device-pointer entry points ---------------------------------------------------------
    # `stream`: a HIP stream handle (e.g. torch.cuda.Stream().cuda_stream) on which the work is enqueued, in order with
    # whatever the caller has enqueued there.  None / 0 selects the handle's OWN non-blocking stream (lf_mkd.h), which is
    # NOT ordered with torch's streams -- and 0 is also what torch's default stream reports as its handle.  So that the
    # default is safe for torch callers, a call without a stream first waits for the device (the inputs a torch kernel is
    # still producing are complete) and returns only when its own work is done: synchronous, like the host-pointer
    # entry points.  Callers that want asynchrony pass a real stream.
    def _device_call(self, stream, fn, what):
        own = not stream
        if own:
            import sys
            t = sys.modules.get("torch")
            if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
                t.cuda.synchronize()
        self._check(fn(None if own else stream), what)
        if own:
            self.synchronize()

    def describe_patches_device(self, d_patches, n, d_out, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_describe_patches_device(self._h, d_patches, n, d_out, s),
                          "lf_mkd_describe_patches_device")

    def raw_descriptors_device(self, d_patches, n, d_raw, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_raw_descriptors_device(self._h, d_patches, n, d_raw, s),
                          "lf_mkd_raw_descriptors_device")

    def set_image_device(self, d_image, width, height, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_set_image_device(self._h, d_image, width, height, s),
                          "lf_mkd_set_image_device")

    def set_images_device(self, d_images, n_frames, width, height, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_set_images_device(self._h, d_images, n_frames, width, height, s),
                          "lf_mkd_set_images_device")

    def set_images_u8_device(self, d_images, n_frames, width, height, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_set_images_u8_device(self._h, d_images, n_frames, width, height, s),
                          "lf_mkd_set_images_u8_device")

    def describe_keypoints_frames_device(self, d_kps, d_frame_of_kp, n, d_out, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_describe_keypoints_frames_device(self._h, d_kps, d_frame_of_kp, n,
                                                                                         d_out, s),
                          "lf_mkd_describe_keypoints_frames_device")

    def describe_keypoints_device(self, d_kps, n, d_out, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_describe_keypoints_device(self._h, d_kps, n, d_out, s),
                          "lf_mkd_describe_keypoints_device")

    def orient_keypoints_device(self, d_extrema, d_frame_of_extremum, n, d_out, d_frame_of_kp, max_out, stream=None):
        """Returns (written, dropped); waits for the stream (the count comes back to the host)."""
        m, dropped = ctypes.c_uint64(), ctypes.c_uint64()
        self._device_call(stream, lambda s: self.L.lf_mkd_orient_keypoints_device(
            self._h, d_extrema, d_frame_of_extremum, n, d_out, d_frame_of_kp, max_out, ctypes.byref(m),
            ctypes.byref(dropped), s), "lf_mkd_orient_keypoints_device")
        return m.value, dropped.value

    def detect_extrema_device(self, d_out, d_frame_of, max_out, stream=None):
        m, dropped = ctypes.c_uint64(), ctypes.c_uint64()
        self._device_call(stream, lambda s: self.L.lf_mkd_detect_extrema_device(
            self._h, d_out, d_frame_of, max_out, ctypes.byref(m), ctypes.byref(dropped), s), "lf_mkd_detect_extrema_device")
        return m.value, dropped.value

    def filter_extrema_device(self, d_extrema, n, top_n, min_size, d_out, d_index=None, stream=None):
        m = ctypes.c_uint64()
        self._device_call(stream, lambda s: self.L.lf_mkd_filter_extrema_device(
            self._h, d_extrema, n, top_n, min_size, d_out, d_index, ctypes.byref(m), s), "lf_mkd_filter_extrema_device")
        return m.value

    def match_device(self, d_a, na, d_b, nb, d_match, ratio=0.8, d_exclude_lo=None, d_exclude_hi=None, d_best=None,
                     d_second=None, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_match_device(
            self._h, d_a, na, d_b, nb, d_exclude_lo, d_exclude_hi, ratio, d_match, d_best, d_second, s), "lf_mkd_match_device")

    def match_both_device(self, d_a, na, d_b, nb, d_match_ab, d_match_ba, ratio=0.8, stream=None):
        """both directions in one call: match_ab [na] = match(a, b), match_ba [nb] = match(b, a)"""
        self._device_call(stream, lambda s: self.L.lf_mkd_match_both_device(self._h, d_a, na, d_b, nb, ratio, d_match_ab,
                                                                             d_match_ba, s), "lf_mkd_match_both_device")

    def match_overflowed(self, stream=None):
        """Rows of the latest match call that were redone by the full scan (diagnostic; waits for the call)."""
        n = ctypes.c_uint64()
        self._check(self.L.lf_mkd_match_overflowed(self._h, stream, ctypes.byref(n)), "lf_mkd_match_overflowed")
        return n.value

    def detect_frames_device(self, d_images, n_frames, width, height, top_n, min_size, d_keypoints, d_frame_of_kp,
                             d_descriptors, max_out, stream=None):
        """Batch of frames through the whole pipeline; returns (keypoints written, dropped_blobs, dropped_features)."""
        m, db, df = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._device_call(stream, lambda s: self.L.lf_mkd_detect_frames_device(
            self._h, d_images, n_frames, width, height, top_n, min_size, d_keypoints, d_frame_of_kp, d_descriptors, max_out,
            ctypes.byref(m), ctypes.byref(db), ctypes.byref(df), s), "lf_mkd_detect_frames_device")
        return m.value, db.value, df.value

    def stream_create(self, width, height, top_n, min_size, max_out, d_image, d_keypoints, d_descriptors, d_counts):
        """Records the per-frame detect+describe pipeline as a hipGraph over fixed device buffers."""
        self._check(self.L.lf_mkd_stream_create(self._h, width, height, top_n, min_size, max_out, d_image, d_keypoints,
                                                d_descriptors, d_counts), "lf_mkd_stream_create")

    def stream_frame(self, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_stream_frame(self._h, s), "lf_mkd_stream_frame")

    def sample_patches_device(self, d_kps, n, d_patches, stream=None):
        self._device_call(stream, lambda s: self.L.lf_mkd_sample_patches_device(self._h, d_kps, n, d_patches, s),
                          "lf_mkd_sample_patches_device")

    def kernel_times(self):
        """(kernel_ms, 0.0, batches) summed since the previous call; needs FLAG_KERNEL_TIMING.
        (The second value was the separate whitening kernel before it was fused into mkd_pool.)"""
        a, b, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
        self._check(self.L.lf_mkd_kernel_times(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)),
                    "lf_mkd_kernel_times")
        return a.value, b.value, n.value

    def kernel_clock(self, stream=None):
        """(shader MHz sustained during the latest describe launch, lifetime of its workgroup 0 in ms); FLAG_KERNEL_TIMING."""
        mhz, ms = ctypes.c_double(), ctypes.c_double()
        self._check(self.L.lf_mkd_kernel_clock(self._h, stream, ctypes.byref(mhz), ctypes.byref(ms)), "lf_mkd_kernel_clock")
        return mhz.value, ms.value

    def synchronize(self):
        self._check(self.L.lf_mkd_synchronize(self._h), "lf_mkd_synchronize")
