#!/bin/bash
# SQ counters of the keypoint-mode describe kernel for library variants: tools/pmc_kp.sh VARIANT...  (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
    rm -rf /tmp/pmc_$v
    LF_MKD_LIB=$R/ab/liblf_mkd_$v.so rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$v -- python3 $R/tools/ab_kp.py $v > /dev/null 2>&1
    python3 - "$v" /tmp/pmc_$v <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mkd_pool" in r["Kernel_Name"]:
            acc[(int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid, cs in sorted(acc.items()):
    print(v, "grid", grid, {k: f"{sum(x)/len(x):.4g}" for k, x in sorted(cs.items())})
PY
  done
done
