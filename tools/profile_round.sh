#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + HBM traffic / SQ counters of the default bench.py run, and the
# same for the keypoint-mode batch (tools/prof_keypoints.py).  The program is always given directly after `--`.
# Usage: tools/profile_round.sh <tag>      -> writes gpurun_out/prof_<tag>/{stats,fetch,write,sq,sq2,kp_stats,kp_fetch,kp_write}/
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --cpu-sample 0 --no-extras --no-match --frames-per-gpu 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.log 2>&1
# keypoint mode (configs[2] as one batch): per-kernel times and HBM traffic of sample_patches + mkd_pool
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kp_stats -- python3 $R/tools/prof_keypoints.py > $OUT/kp_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/kp_fetch -- python3 $R/tools/prof_keypoints.py > $OUT/kp_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/kp_write -- python3 $R/tools/prof_keypoints.py > $OUT/kp_write.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ --output-format csv -d $OUT/kp_tcp -- python3 $R/tools/prof_keypoints.py > $OUT/kp_tcp.log 2>&1
# configs[1] (one 1080p frame, 10 000 keypoints): HBM traffic of the same call
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/kp1_fetch -- python3 $R/tools/prof_keypoints.py configs1 > $OUT/kp1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/kp1_write -- python3 $R/tools/prof_keypoints.py configs1 > $OUT/kp1_write.log 2>&1
# configs[3] in its own form, one GPU's share (128 frames 1080p x 8192 keypoints): kernel times and HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kp3_stats -- python3 $R/tools/prof_keypoints.py configs3 > $OUT/kp3_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/kp3_fetch -- python3 $R/tools/prof_keypoints.py configs3 > $OUT/kp3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/kp3_write -- python3 $R/tools/prof_keypoints.py configs3 > $OUT/kp3_write.log 2>&1
# the reference's own settings on one 1080p frame (3000 keypoints): HBM traffic of the same call
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/kpr_fetch -- python3 $R/tools/prof_keypoints.py refdefaults > $OUT/kpr_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/kpr_write -- python3 $R/tools/prof_keypoints.py refdefaults > $OUT/kpr_write.log 2>&1
# the whitening projection on its own (north_star: "MFMA utilisation on the projection"): phase clocks of a -DLF_PHASE_TIMING
# build of the same sources (tools/ab_build.sh pt "-DLF_PHASE_TIMING", built beforehand: ab/ travels with the snapshot)
if [ -f $R/ab/liblf_mkd_pt.so ]; then
  (cd $R && LF_MKD_LIB=ab/liblf_mkd_pt.so python3 tools/phase_timing.py 1048576 gpurun_out/${TAG}_projection.json > $OUT/phase_timing.txt 2>&1) || echo "phase timing failed" >> $OUT/phase_timing.txt
fi
tail -1 $OUT/stats.log | cut -c1-400
