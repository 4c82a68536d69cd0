#!/bin/bash
# Per-kernel device time of the detector pipelines (tools/bench_detect.py) by kernel and grid size, from rocprofv3 --kernel-trace.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/dkt
rocprofv3 --kernel-trace --output-format csv -d /tmp/dkt -- python3 $R/tools/bench_detect.py > /tmp/dkt.log 2>&1
grep "configs\[2\] batched\|configs\[4\] hipGraph" /tmp/dkt.log
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/dkt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "lfmkd" not in n: continue
        acc[(n.split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 300: continue
    print(f"{k[0]:42s} grid {k[1]:>8}x{k[2]}x{k[3]:<5} calls {len(v):4d}  avg {sum(v)/len(v):9.1f} us  total {sum(v)/1e3:8.2f} ms")
PY
