#!/usr/bin/env python3
"""What the first calls of lf_mkd_detect cost on a fresh handle (the advisor's round-5 finding: recording on the first call was
never measured).  Per frame size and pixel type, on handles that have never seen the request:
   call 1  first sighting  -- served stage by stage (allocations of the handle's scratch included: a fresh handle)
   call 2  recording       -- capture + hipGraphInstantiate (two of each when the upload is banded), then the launch
   call 3+ replay
and the same with LF_MKD_DETECT_RECORD_AFTER=0 (record at the first sighting, the round-5 behaviour), and a second fresh
handle in the same process (code objects loaded, runtime pools warm: what a second image costs a match_images-style
caller).  Wall time of the C call into arrays the caller keeps.  Usage: bench_first_call.py [out.txt]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "tools")]
import numpy as np
import local_features_python as lfp
from bench_reference_sweep import open_image

lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def calls(img, w, h, n, record_after):
    os.environ["LF_MKD_DETECT_RECORD_AFTER"] = str(record_after)
    lf = lfp.MkdHandle(max_features=3000, max_image_width=w, max_image_height=h, n_scales=3, max_blobs=15000)
    kps, desc = np.empty((3000, 5), np.float32), np.empty((3000, 128), np.float32)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        lf.detect_into(img, 2000, 0.0, kps, desc)
        ts.append((time.perf_counter() - t0) * 1e3)
    lf.close()
    return ts


say("lf_mkd_detect, top 2000 of houses.jpg (tests/golden), n_scales 3: wall ms of calls 1..6 on a fresh handle")
say(f"{'frame':>12} {'px':>4} {'record_after':>12} {'handle':>7} | " + " ".join(f"{'call ' + str(i + 1):>9}" for i in range(6)))
for scale in (0.25, 0.47, 1.0):
    u8, f32 = open_image(scale)
    h, w = f32.shape
    for name, img in (("u8", u8), ("f32", f32)):
        for ra in (1, 0):
            for which in ("1st", "2nd"):
                ts = calls(img, w, h, 6, ra)
                say(f"{w:>7}x{h:<4} {name:>4} {ra:>12} {which:>7} | " + " ".join(f"{t:9.3f}" for t in ts))
os.environ.pop("LF_MKD_DETECT_RECORD_AFTER", None)
say("record_after 1 (default): call 1 = the pipeline's launches, unrecorded (a fresh handle also allocates its scratch), call 2 = recording + launch, call 3.. = replay")
say("record_after 0 (round 5):  call 1 = recording + launch, call 2.. = replay")
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")

# ---- what the reference's example does: ONE handle sized for the largest image, every image detected once at its own size ----
# (examples/match_images/src/main.rs:44-76).  The handle's scratch exists after its first call; what a NEW request then costs:
say("")
say("one handle (max 4096 x 3072), a first call on another frame size to allocate its scratch, then each frame size ONCE (first sighting")
say("of its request), then twice more (recording, replay); the stage-by-stage form (LF_MKD_FLAG_DETECT_STEPWISE) beside it")
for name in ("u8", "f32"):
    for flags, label in ((0, "default"), (lfp.FLAG_DETECT_STEPWISE, "stepwise")):
        lf = lfp.MkdHandle(max_features=3000, max_image_width=4096, max_image_height=3072, n_scales=3, max_blobs=15000, flags=flags)
        kps, desc = np.empty((3000, 5), np.float32), np.empty((3000, 128), np.float32)
        u8, f32 = open_image(0.3)
        lf.detect_into(u8 if name == "u8" else f32, 2000, 0.0, kps, desc)
        row = []
        for scale in (0.25, 0.47, 1.0):
            u8, f32 = open_image(scale)
            img = u8 if name == "u8" else f32
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                lf.detect_into(img, 2000, 0.0, kps, desc)
                ts.append((time.perf_counter() - t0) * 1e3)
            row.append(f"{img.shape[1]}x{img.shape[0]}: " + " / ".join(f"{t:.3f}" for t in ts))
        say(f"{name:>4} {label:>9} | " + "   ".join(row))
        lf.close()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
