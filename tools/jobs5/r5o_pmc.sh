#!/bin/bash
# round 5, job o: LDS counters of the hot FF2 launch, tiling 21 (four waves of 32 x 160) vs tiling 23 (2 x 2 waves of 64 x 80) -- VERDICT r4 item 3's counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r5o; mkdir -p $out
for pm in SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE; do
  timeout 120 rocprofv3 --pmc $pm --kernel-trace --output-format csv -d $out -o $pm -- python tools/w22_pmc.py > $out/$pm.log 2>&1
done
python - <<'PY'
import csv, collections, glob
res = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r5o/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "tiling 23 (gemm_w22_kernel)" if "gemm_w22" in n else "tiling 21 (gemm_conv_kernel<128,160,4,1,4,0,4,...>)" if "gemm_conv_kernel" in n else None
        if k: res[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in res.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
rm -f $out/*_kernel_trace.csv $out/*agent_info.csv $out/*counter_collection.csv
