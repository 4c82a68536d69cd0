#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel times of the widened rows (detector, orientation, matcher pipelines).
# Usage: tools/profile_next_rows.sh <tag>   -> gpurun_out/prof_<tag>_{detect,match}/
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_detect -- python3 $R/tools/bench_detect.py > $R/gpurun_out/prof_${TAG}_detect.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_match -- python3 $R/tools/bench_match.py > $R/gpurun_out/prof_${TAG}_match.log 2>&1
for d in detect match; do
  f=$(find $R/gpurun_out/prof_${TAG}_$d -name "*kernel_stats.csv" | head -1)
  echo "== $d: $f"; head -25 "$f" | cut -c1-200
done
