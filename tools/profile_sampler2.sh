#!/bin/bash
# GPU box: kernel trace + SQ counters of the keypoint-mode batch (tools/prof_keypoints.py). -> gpurun_out/prof_sampler2/
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_sampler2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/prof_keypoints.py > $OUT/stats.log 2>&1 || { echo stats failed; tail -3 $OUT/stats.log; exit 1; }
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM" \
           "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/prof_keypoints.py > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/g$i.log; }
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof_sampler2"
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lfmkd" in r["Name"]:
            print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void lfmkd::", "")
        if "sample" in k or "mkd_pool" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):.4g}")
PY
