#!/bin/bash
# FETCH_SIZE of the keypoint-mode describe kernel with keypoints in random order and sorted by (frame, level, 64x64 tile):
# tools/bench_keypoints.py locality ... counters under rocprofv3 --pmc (on the GPU box).  VERDICT r4 item 3.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for cfg in configs3 configs1; do
  for order in random sorted; do
    rm -rf /tmp/pmc_loc
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_loc -- python3 $R/tools/bench_keypoints.py locality $cfg $order counters > /dev/null 2>&1
    python3 - "$cfg" "$order" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob("/tmp/pmc_loc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mkd_pool" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
if vals:
    print(f"{sys.argv[1]} {sys.argv[2]:6s}: mkd_pool FETCH_SIZE per launch = {sum(vals)/len(vals)*1024/1e6:9.1f} MB as counted ({len(vals)} launches)")
else:
    print(sys.argv[1], sys.argv[2], "no counter rows")
PY
  done
done
