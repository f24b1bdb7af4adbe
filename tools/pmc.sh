#!/bin/bash
# usage: tools/pmc.sh <kernel-substring> <cmd...>   -- collects a few PMC passes and prints per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pat=$1; shift
rm -rf gpurun_out/pmc; i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
          "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
          "SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES" \
          "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1)); rocprofv3 --pmc $pm --kernel-trace --output-format csv -d gpurun_out/pmc -o p$i -- "$@" > /dev/null 2>&1
done
python - "$pat" <<PY
import csv, glob, collections, sys
pat = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
