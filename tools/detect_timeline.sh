#!/bin/bash
# kernel + copy timeline of one lf_mkd_detect[_u8] call on the reference benchmark's frame: tools/detect_timeline.sh [u8|f32] [scale] [n_scales] [top_n]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/detect_timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_detect_host.py "$@" > $OUT/run.log 2>&1
python3 - "$OUT/trace" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lfmkd" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")[:30]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24] + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last call: from the last large host-to-device copy on
starts = [i for i, r in enumerate(rows) if r[2].startswith("COPY") and "HOST_TO_DEVICE" in r[2].upper()]
big = [i for i in starts if rows[i][1] - rows[i][0] > 20000]
lo = big[-1] if big else 0
fr = rows[lo:]
t0 = fr[0][0]
print(f"{'kernel / copy':42s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}")
prev_end = t0
for s, e, name in fr:
    print(f"{name:42s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
    prev_end = max(prev_end, e)
print(f"call: {(prev_end - t0) / 1e3:.1f} us on the device, {len(fr)} operations")
PY
