#!/bin/bash
# A/B timing of describe-kernel builds on ONE box (box-to-box noise is ~1 %): tools/ab_describe.sh libA.so libB.so ...
# Each library is timed ROUNDS times, interleaved.  Output: gpurun_out/ab_describe.log
set -e
ROUNDS=${ROUNDS:-3}
mkdir -p gpurun_out
: > gpurun_out/ab_describe.log
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo "== $lib (round $r)" >> gpurun_out/ab_describe.log
    LF_MKD_LIB=$(realpath $lib) timeout -k 10 200 python tools/gpu_quick.py 1048576 2>/dev/null >> gpurun_out/ab_describe.log
  done
done
cat gpurun_out/ab_describe.log
