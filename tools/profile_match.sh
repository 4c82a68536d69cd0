#!/bin/bash
# kernel-trace summary of the matcher's passes (tools/ab_match.py); run on the GPU box from the repo root
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/match_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/ab_match.py "$@" > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cut -d, -f1-7 $f | grep -i -E "Name|match_" > $OUT/summary.csv
cat $OUT/summary.csv
