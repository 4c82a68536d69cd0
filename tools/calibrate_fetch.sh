#!/bin/bash
# What FETCH_SIZE / WRITE_SIZE count for the access shapes of the pyramid kernels: tools/micro/tile_copy moves a known
# 314.6 MB each way per launch (MI355X_MICROARCH.md: calibrate other widths than 16-byte streaming on a known byte count).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cal_f /tmp/cal_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/cal_f -- $R/tools/micro/tile_copy > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/cal_w -- $R/tools/micro/tile_copy > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
known = 640 * 480 * 256 * 4
for d, c in (("/tmp/cal_f", "FETCH_SIZE"), ("/tmp/cal_w", "WRITE_SIZE")):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        print(f"{c:10s} {k:28s} launches {cnt[k]:3d}  counted {agg[k] / cnt[k] * 1024 / 1e6:8.1f} MB per launch = {agg[k] / cnt[k] * 1024 / known:.3f} x the {known / 1e6:.1f} MB moved")
PY
