#!/bin/bash
# kernel timeline of one recorded frame (lf_mkd_stream_*): tools/graph_timeline.sh W H TOPN   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/graph_timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/prof_graph_frame.py "$@" > $OUT/run.log 2>&1
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lfmkd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: walk back from the last mkd_pool to the previous one
ends = [i for i, r in enumerate(rows) if "mkd_pool" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
fr = rows[lo:ends[-1] + 1]
t0 = int(fr[0]["Start_Timestamp"])
print(f"{'kernel':28s} {'start us':>9s} {'dur us':>8s} {'gap us':>7s}")
prev_end = t0
for r in fr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")[:28]
    print(f"{name:28s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}")
    prev_end = max(prev_end, e)
print(f"frame: {(prev_end - t0) / 1e3:.1f} us, {len(fr)} kernels")
PY
