#!/bin/bash
# Per-kernel device time of one keypoint-mode call (tools/ab_kp.py) by kernel and grid size, from rocprofv3 --kernel-trace.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kpt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kpt -- python3 $R/tools/ab_kp.py prof > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/kpt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "lfmkd" not in n: continue
        acc[(n.split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:42s} grid {k[1]:>8}x{k[2]}x{k[3]:<5} calls {len(v):4d}  avg {sum(v)/len(v):9.1f} us  total {sum(v)/1e3:8.2f} ms")
PY
