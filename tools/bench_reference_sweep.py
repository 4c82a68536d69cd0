#!/usr/bin/env python3
"""The reference's own criterion benchmark (local_features/benches/bench.rs:41-112) on this path, on its own input:
`detect_top_n` on sample_data/houses.jpg (here tests/golden/houses.jpg, 4096 x 3072) -- 8-bit luma, Lanczos-resized to
0.25 .. 1.0 of full size, f32 / 255 (bench.rs:9-19,47-49; Pillow's "L" conversion and LANCZOS filter stand in for the
`image` crate's) -- groups scale_scales={3,5} (3000 features at the four scales) and feats_scales={3,5} (100 .. 2000
features at full size), max_blobs = 5 x max_features (bench.rs:57-64).  Host image in, host results out, as
`lf.detect_top_n(&image.view(), n, 0.)` does, in three forms:
   f32       lf_mkd_detect        the caller's f32 frame (4 B/px over PCIe), one recorded hipGraph per request shape (from its second sighting on)
   u8        lf_mkd_detect_u8     the 8-bit frame the f32 one was made from (1 B/px), same results bit for bit
   stepwise  lf_mkd_detect with LF_MKD_FLAG_DETECT_STEPWISE: the call as it was before round 5 (three host waits)
Median of 20 calls after 3 warm-up calls, into arrays the caller keeps (MkdHandle.detect_into: the C call alone); the split
(upload / recorded pipeline / result copy) is the library's own (lf_mkd_detect_times, HIP events on its stream).
`sweep()` returns the rows; bench.py puts them on its line as pipelines.reference_bench_houses."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "local-features_amd"))
import numpy as np
import local_features_python as lfp

_PHOTO = None


def open_image(scale):
    """(uint8 luma, the f32 frame made from it)"""
    global _PHOTO
    from PIL import Image
    if _PHOTO is None:
        _PHOTO = Image.open(os.path.join(ROOT, "tests", "golden", "houses.jpg")).convert("L")
    w, h = round(_PHOTO.width * scale), round(_PHOTO.height * scale)
    im = _PHOTO if scale == 1.0 else _PHOTO.resize((w, h), Image.LANCZOS)
    u8 = np.ascontiguousarray(np.asarray(im, np.uint8))
    return u8, np.ascontiguousarray(u8.astype(np.float32) / np.float32(255.0))


def bench(n_scales, scale, max_features, group, calls=20, forms=("f32", "u8", "stepwise")):
    u8, f32 = open_image(scale)
    h, w = f32.shape
    row = {"group": group, "scale": scale, "max_features": max_features, "n_scales": n_scales, "width": w, "height": h}
    kps, desc = np.empty((max_features, 5), np.float32), np.empty((max_features, 128), np.float32)
    first = None
    for form in forms:
        lf = lfp.MkdHandle(max_features=max_features, max_image_width=w, max_image_height=h, n_scales=n_scales,
                           max_blobs=5 * max_features,                                  # bench.rs:57-64
                           flags=lfp.FLAG_DETECT_STEPWISE if form == "stepwise" else lfp.FLAG_KERNEL_TIMING)
        img = u8 if form == "u8" else f32
        for _ in range(3):
            m, db, df = lf.detect_into(img, max_features, 0.0, kps, desc)
        ts, split = [], []
        for _ in range(calls):
            t0 = time.perf_counter(); lf.detect_into(img, max_features, 0.0, kps, desc); ts.append(time.perf_counter() - t0)
            if form != "stepwise":
                split.append(lf.detect_times())
        if first is None:
            first = (m, db, df, kps[:m].copy(), desc[:m].copy())
        else:                                                       # every form returns the same bits
            assert (m, db, df) == first[:3] and np.array_equal(kps[:m], first[3]) and np.array_equal(desc[:m], first[4]), form
        row[f"{form}_ms"] = float(np.median(ts) * 1e3)
        if split:
            sp = np.median(np.array(split), axis=0)
            row[f"{form}_upload_ms"], row[f"{form}_compute_ms"], row[f"{form}_readback_ms"] = (float(x) for x in sp)
        lf.close()
    row.update(keypoints=int(first[0]), dropped_blobs=int(first[1]), dropped_features=int(first[2]))
    return row


def fmt(r):
    s = f"{r['group']}/{r['scale']}x-{r['max_features']}feats: {r['width']}x{r['height']}:"
    for form in ("f32", "u8", "stepwise"):
        if f"{form}_ms" in r:
            s += f"  {form} {r[form + '_ms']:7.3f} ms"
            if f"{form}_upload_ms" in r:
                s += f" (upload {r[form + '_upload_ms']:.3f} + pipeline {r[form + '_compute_ms']:.3f} + results {r[form + '_readback_ms']:.3f})"
    return s + f"  [{r['keypoints']} keypoints, dropped blobs {r['dropped_blobs']}, features {r['dropped_features']}]"


def sweep(full=True, calls=20):
    rows = []
    for ns in (3, 5):                                  # do_benches_nfeats
        for nf in ((100, 500, 1000, 2000) if full else (2000,)):
            rows.append(bench(ns, 1.0, nf, f"feats_scales={ns}", calls))
    for ns in (3, 5):                                  # do_benches_scale
        for s in ((0.25, 0.5, 0.75, 1.0) if full else (0.25,)):
            rows.append(bench(ns, s, 3000, f"scale_scales={ns}", calls))
    return rows


if __name__ == "__main__":
    for ns in (3, 5):
        for nf in (100, 500, 1000, 2000):
            print(fmt(bench(ns, 1.0, nf, f"feats_scales={ns}")), flush=True)
    for ns in (3, 5):
        for s in (0.25, 0.5, 0.75, 1.0):
            print(fmt(bench(ns, s, 3000, f"scale_scales={ns}")), flush=True)
    # a 1920 x 1080 frame (BASELINE configs[1]'s size) through the same three forms, the reference's default settings
    from PIL import Image
    ph = Image.open(os.path.join(ROOT, "tests", "golden", "houses.jpg")).convert("L").resize((1920, 1080), Image.LANCZOS)
    _PHOTO = ph
    print(fmt(bench(4, 1.0, 3000, "1080p_defaults")), flush=True)
