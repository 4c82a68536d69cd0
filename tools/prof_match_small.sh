#!/bin/bash
# kernel durations of the small-problem matcher (tools/bench_match_small.py) under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_match_small
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_match_small.py > $OUT/run.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
grep -i "match" "$f" | cut -c1-200
