#!/bin/bash
# SQ / memory counters of the kernels of one lf_mkd_detect_u8 call on the reference benchmark's frame (tools/prof_detect_host.py),
# separate --pmc passes: tools/pmc_detect_host.sh [kernel-name-substring]   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$PWD}
WHAT=${1:-scan_extrema}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_dh
  LF_MKD_DETECT_BANDS=0 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_dh -- python3 $R/tools/prof_detect_host.py u8 > /dev/null 2>&1
  python3 - "$WHAT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_dh/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lfmkd::", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(cs.items())})
PY
done
