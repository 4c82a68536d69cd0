#!/bin/bash
# Runs on the GPU box (via gpurun): texture-path counters of the keypoint-mode batch (tools/prof_keypoints.py), one
# rocprofv3 --pmc pass per group (a hardware block takes two counters at a time), each under its own timeout.  -> gpurun_out/pmc_sampler/<group>/
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_sampler
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCP_LATENCY" "TD_TD_BUSY TD_TC_STALL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/prof_keypoints.py > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/g$i.log; exit 1; }
  echo "group $i done"
done
