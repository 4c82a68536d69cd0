"""Python face of the MI355X MKD descriptor path, mirroring the reference's pyo3 module
`local_features_python` (python/src/lib.rs:11-160) for the part of it this build covers.

The reference class exposes `detect` / `detect_top_n` (detector + describe in one call,
python/src/lib.rs:86-149), both provided here with the same signatures and the
`(list[Keypoint], ndarray[n,128] f32)` result.  The hot path of this build is the describe half
(SURVEY.md section 8), so the class also offers it on its own: `describe` takes the keypoint list the
detector would have produced, `orient` / `describe_extrema` take the detector's refined extrema and run
keypoint orientation first (the whole extract graph, mod.rs:1277-1572), `describe_patches` takes patches.
Nothing here falls back to a CPU path.
"""
import threading

import numpy as np

from ._lib import (ANGLE_EXACT, ANGLE_EXACT_ZERO, ANGLE_SHADER, FLAG_DETECT_STEPWISE, FLAG_KERNEL_TIMING, FLAG_UNFUSED_KEYPOINTS, KEYPOINT_DTYPE, LIB_PATH, MODEL_DIR, PCA_NAMES,
                   POOL_DEFAULT, POOL_F16X3, POOL_F32, POOL_F16_FP6, SYMBOLS, COMM_ID_BYTES, GATHER_DIRECT, GATHER_RING, Comm, MkdHandle,
                   comm_unique_id, load_library, model_path, plan_upload)

__all__ = ["Keypoint", "LocalFeatures", "MkdHandle", "ANGLE_SHADER", "ANGLE_EXACT", "ANGLE_EXACT_ZERO", "POOL_DEFAULT", "POOL_F32", "POOL_F16_FP6",
           "POOL_F16X3", "FLAG_KERNEL_TIMING", "FLAG_UNFUSED_KEYPOINTS", "FLAG_DETECT_STEPWISE", "KEYPOINT_DTYPE", "PCA_NAMES", "SYMBOLS", "LIB_PATH", "MODEL_DIR",
           "load_library", "model_path", "plan_upload", "Comm", "comm_unique_id", "COMM_ID_BYTES", "GATHER_DIRECT", "GATHER_RING"]


class Keypoint:
    """python/src/lib.rs:11-23; angle in degrees (keypoint_orientation.glsl:162-167)."""
    __slots__ = ("x", "y", "size", "angle", "response")

    def __init__(self, x, y, size, angle, response=0.0):
        self.x, self.y, self.size, self.angle, self.response = (
            float(x), float(y), float(size), float(angle), float(response))

    def __repr__(self):
        return (f"Keypoint(x={self.x:.3f}, y={self.y:.3f}, size={self.size:.3f}, "
                f"angle={self.angle:.2f}, response={self.response:.4f})")


def _keypoints_to_array(keypoints):
    if isinstance(keypoints, np.ndarray):
        if keypoints.dtype == KEYPOINT_DTYPE:
            return np.ascontiguousarray(keypoints)
        k = np.ascontiguousarray(keypoints, np.float32)
        if k.ndim != 2 or k.shape[1] not in (4, 5):
            raise RuntimeError("keypoints array must be [n,4] (x,y,size,angle) or [n,5] (+response)")
        if k.shape[1] == 4:
            k = np.concatenate([k, np.zeros((k.shape[0], 1), np.float32)], axis=1)
        return np.ascontiguousarray(k)
    return np.array([(k.x, k.y, k.size, k.angle, k.response) for k in keypoints],
                    dtype=np.float32).reshape(-1, 5)


class LocalFeatures:
    """LocalFeatures(max_image_width, max_image_height, max_features, max_blobs, n_scales, pca)
    -- python/src/lib.rs:43-84.  The last detect call's FeaturesResult counters (lib.rs:77-83) are kept in
    `dropped_blobs` / `dropped_features`."""

    def __init__(self, max_image_width, max_image_height, max_features, max_blobs=8000, n_scales=4,
                 pca="liberty", device=0, angle_mode=ANGLE_SHADER, pool_mode=POOL_DEFAULT, max_frames=1, flags=0):
        if pca not in PCA_NAMES:
            raise RuntimeError("Invalid PCA argument")
        try:
            self._inner = MkdHandle(pca=pca, max_features=max_features, max_image_width=max_image_width,
                                    max_image_height=max_image_height, device=device,
                                    angle_mode=angle_mode, pool_mode=pool_mode, n_scales=n_scales,
                                    max_blobs=max_blobs, max_frames=max_frames, flags=flags)
        except RuntimeError as e:   # python/src/lib.rs:77-82
            raise RuntimeError("Failed to initialize local features", str(e)) from e
        self._lock = threading.Lock()   # Mutex<LocalFeaturesVulkan>, python/src/lib.rs:38
        self.max_blobs, self.n_scales, self.max_frames = max_blobs, n_scales, max_frames
        self.max_features, self.device = max_features, device
        self.dropped_blobs = self.dropped_features = 0

    def describe(self, img, keypoints):
        """img: 2-D float32 array in [0,1]; keypoints: list[Keypoint] or [n,4|5] array.
        Returns (list[Keypoint], ndarray[n,128] float32), the reference's result shape."""
        arr = np.asarray(img)
        if arr.ndim != 2:
            raise RuntimeError("Failed to extract features", "image must be 2-dimensional")
        kps = _keypoints_to_array(keypoints)
        with self._lock:
            try:
                self._inner.set_image(arr)
                desc = self._inner.describe_keypoints(kps)
            except RuntimeError as e:   # python/src/lib.rs:97-102
                raise RuntimeError("Failed to extract features", str(e)) from e
        out_k = [Keypoint(*row) for row in kps.reshape(-1, 5)] if not isinstance(keypoints, list) else keypoints
        return out_k, desc

    def orient(self, img, extrema):
        """extrema: [n,4] array (x, y, size, response), the detector's refined extrema.  Returns the
        list[Keypoint] keypoint orientation yields (one per histogram peak, keypoint_orientation.glsl:36-171),
        ordered by extremum then bin."""
        return self.describe_extrema(img, extrema, describe=False)[0]

    def describe_extrema(self, img, extrema, describe=True):
        """The whole extract graph: orientation, then sampling + description of every keypoint found.
        Returns (list[Keypoint], ndarray[m,128] float32)."""
        arr = np.asarray(img)
        if arr.ndim != 2:
            raise RuntimeError("Failed to extract features", "image must be 2-dimensional")
        ex = np.ascontiguousarray(extrema, np.float32)
        if ex.ndim != 2 or ex.shape[1] != 4:
            raise RuntimeError("extrema array must be [n,4] (x, y, size, response)")
        with self._lock:
            try:
                self._inner.set_image(arr)
                kps, _ = self._inner.orient_keypoints(ex)
                desc = self._inner.describe_keypoints(kps) if describe else None
            except RuntimeError as e:
                raise RuntimeError("Failed to extract features", str(e)) from e
        return [Keypoint(*row) for row in kps], desc

    def match(self, desc_a, desc_b, ratio=0.8):
        """match_features of the reference's match_images example (examples/match_images/src/main.rs:8-27):
        list of (i, j) with j the best match of desc_a[i] in desc_b that passes Lowe's ratio test."""
        with self._lock:
            m = self._inner.match(desc_a, desc_b, ratio)
        return [(int(i), int(j)) for i, j in enumerate(m) if j >= 0]

    def match_both(self, desc_a, desc_b, ratio=0.8):
        """Both directions of the example (examples/match_images/src/main.rs:113-116: match_features(f1, f2) and
        match_features(f2, f1)) in one library call -- one launch at the example's own size (lf_mkd_match_both_device).
        Returns (matches a -> b, matches b -> a), each as `match` returns them."""
        import torch
        a = np.ascontiguousarray(desc_a, np.float32).reshape(-1, 128)
        b = np.ascontiguousarray(desc_b, np.float32).reshape(-1, 128)
        if len(a) < 2 or len(b) < 2:
            return self.match(a, b, ratio) if len(a) and len(b) >= 2 else [], self.match(b, a, ratio) if len(b) and len(a) >= 2 else []
        dev = torch.device("cuda", self.device)
        with self._lock, torch.cuda.device(dev):
            d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
            m_ab = torch.empty((len(a),), dtype=torch.int32, device=dev)
            m_ba = torch.empty((len(b),), dtype=torch.int32, device=dev)
            s = torch.cuda.current_stream(dev)
            self._inner.match_both_device(d_a.data_ptr(), len(a), d_b.data_ptr(), len(b), m_ab.data_ptr(), m_ba.data_ptr(),
                                          ratio, s.cuda_stream)
            s.synchronize()
            self._inner.synchronize()       # (torch's default stream is handle 0 = "the library's own stream" to the ABI)
            m_ab, m_ba = m_ab.cpu().numpy(), m_ba.cpu().numpy()
        return ([(int(i), int(j)) for i, j in enumerate(m_ab) if j >= 0], [(int(i), int(j)) for i, j in enumerate(m_ba) if j >= 0])

    def match_ip_distance(self, desc_a, desc_b, factor=0.75):
        """The webcam example's acceptance rule (examples/webcam/src/main.rs:97-104,261-265): nearest and second-nearest
        neighbour of desc_a[i] in desc_b under the inner-product distance d = 1 - <a, b> (usearch MetricKind::IP), accepted
        if d0 < factor * d1.  Built on the same matcher: lf_mkd_match_device with ratio <= 0 returns the best index plus
        the best and second-best similarity, the rule is applied to those.  Returns a list of (i, j)."""
        import torch
        a = np.ascontiguousarray(desc_a, np.float32).reshape(-1, 128)
        b = np.ascontiguousarray(desc_b, np.float32).reshape(-1, 128)
        if len(a) == 0:
            return []
        dev = torch.device("cuda", self.device)
        with self._lock, torch.cuda.device(dev):
            d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
            d_m = torch.empty((len(a),), dtype=torch.int32, device=dev)
            d_1, d_2 = torch.empty((len(a),), device=dev), torch.empty((len(a),), device=dev)
            s = torch.cuda.current_stream(dev)
            self._inner.match_device(d_a.data_ptr(), len(a), d_b.data_ptr(), len(b), d_m.data_ptr(), 0.0, None, None,
                                     d_1.data_ptr(), d_2.data_ptr(), s.cuda_stream)
            s.synchronize()
            self._inner.synchronize()       # (torch's default stream is handle 0 = "the library's own stream" to the ABI)
            keep = (1.0 - d_1) < (1.0 - d_2) * factor
            m, keep = d_m.cpu().numpy(), keep.cpu().numpy()
        return [(int(i), int(m[i])) for i in np.flatnonzero(keep)]

    def describe_patches(self, patches):
        """patches: [n,32,32] float32 -> ndarray[n,128] (the CPU twin's Mkd::patch, mkd_ref.rs:57-77)."""
        with self._lock:
            return self._inner.describe_patches(patches)

    def _detect(self, img, n, min_size):
        arr = np.asarray(img)
        if arr.ndim != 2:
            raise RuntimeError("Failed to extract features", "image must be 2-dimensional")
        with self._lock:
            try:
                kps, desc, self.dropped_blobs, self.dropped_features = self._inner.detect(arr, n, min_size)
            except RuntimeError as e:   # python/src/lib.rs:97-102
                raise RuntimeError("Failed to extract features", str(e)) from e
        return [Keypoint(*row) for row in kps], desc

    def detect(self, img):
        """python/src/lib.rs:86-113 (detect_extract_all): every extremum the detector finds, at most max_blobs."""
        return self._detect(img, 0, 0.0)

    def detect_top_n_batch(self, imgs, n, min_size=0.0):
        """detect_top_n over a batch of equally sized frames ([f, h, w] float32, f <= max_frames) with every stage
        launched once for all frames (lf_mkd_detect_frames_device).  Returns one (list[Keypoint], ndarray[k,128]) per
        frame.  The library's keypoint budget is one number for the whole batch (lf_mkd.h): max_features * f here, with
        every frame then cut to its first max_features keypoints -- so each frame's result equals what detect_top_n gives
        for that frame alone as long as the batch total fits the budget (keypoints beyond it are lost to the LAST frames
        and counted, like the per-frame cuts, in dropped_features)."""
        import torch
        arr = np.ascontiguousarray(imgs, np.float32)
        if arr.ndim != 3:
            raise RuntimeError("Failed to extract features", "images must be [frames, height, width]")
        f, h, w = arr.shape
        cap = self.max_features * f
        dev = torch.device("cuda", self.device)      # the handle's device, not torch's current one
        with self._lock:
            try:
                with torch.cuda.device(dev):
                    d_img = torch.from_numpy(arr).to(dev)
                    d_k = torch.empty((cap, 5), device=dev)
                    d_f = torch.empty((cap,), dtype=torch.int32, device=dev)
                    d_d = torch.empty((cap, 128), device=dev)
                    m, self.dropped_blobs, self.dropped_features = self._inner.detect_frames_device(
                        d_img.data_ptr(), f, w, h, int(n), float(min_size), d_k.data_ptr(), d_f.data_ptr(), d_d.data_ptr(),
                        cap, torch.cuda.current_stream(dev).cuda_stream)
                    torch.cuda.synchronize(dev)
            except RuntimeError as e:
                raise RuntimeError("Failed to extract features", str(e)) from e
        kps, fid, desc = d_k[:m].cpu().numpy(), d_f[:m].cpu().numpy(), d_d[:m].cpu().numpy()
        out = []
        for i in range(f):
            sel = np.flatnonzero(fid == i)
            self.dropped_features += max(0, len(sel) - self.max_features)
            sel = sel[:self.max_features]
            out.append(([Keypoint(*row) for row in kps[sel]], desc[sel]))
        return out

    def detect_top_n(self, img, n, min_size):
        """python/src/lib.rs:115-149: the n extrema of largest contrast among those of size >= min_size."""
        return self._detect(img, int(n), float(min_size))
