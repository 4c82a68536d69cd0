#!/bin/bash
# kernel-trace summary of the matcher's passes (tools/ab_match.py); run on the GPU box from the repo root:
#   tools/profile_match.sh [LIBNAME ...] -- [ab_match.py args]     (LIBNAME: an ab/liblf_NAME.so A/B build; none = the product)
R=${GRAFT_REPO_ROOT:-$(pwd)}
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" = "--" ] && shift
[ ${#names[@]} -eq 0 ] && names=(product)
cd /tmp && export TMPDIR=/tmp
for name in "${names[@]}"; do
  OUT=$R/gpurun_out/match_prof/$name
  rm -rf $OUT; mkdir -p $OUT
  if [ $name != product ]; then export LF_MKD_LIB=$R/ab/liblf_$name.so; else unset LF_MKD_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/ab_match.py "$@" > $OUT/stats.log 2>&1
  f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
  echo "== $name"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "match_" in r["Name"]:
        print(f'{r["Name"].split("(")[0].split("::")[-1]:18s} calls {r["Calls"]:>3s}  avg {float(r["AverageNs"]) / 1e6:10.3f} ms')
PY
done
