#!/bin/bash
# Per-kernel timeline of one lf_mkd_match_device call at mid sizes (round 6): tools/match_timeline.sh NA NB
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/match_timeline_$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/ab_match.py "$1" "$2" > $OUT/run.log 2>&1
python3 - "$OUT/trace" "$1" "$2" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lfmkd" in r["Kernel_Name"] and "match" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")[:34]))
rows.sort()
# the last call: kernels after the last gap of more than 0.5 ms... calls are back to back; split on match_split of b (first kernel of a call)
starts = [i for i, r in enumerate(rows) if r[2].startswith("match_split") and (i == 0 or not rows[i - 1][2].startswith("match_split"))]
lo = starts[-1] if starts else 0
fr = rows[lo:]
t0 = fr[0][0]
print(f"lf_mkd_match_device {sys.argv[2]} x {sys.argv[3]}: the last call's kernels")
print(f"{'kernel':36s} {'start us':>9s} {'dur us':>9s} {'gap us':>7s}")
prev = t0
tot = {}
for s, e, name in fr:
    print(f"{name:36s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f} {(s - prev) / 1e3:7.1f}")
    tot[name] = tot.get(name, 0) + (e - s) / 1e3
    prev = max(prev, e)
span = (prev - t0) / 1e3
print(f"call: {span:.1f} us on the device; shares: " + ", ".join(f"{k} {v / span:.0%}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
PY
