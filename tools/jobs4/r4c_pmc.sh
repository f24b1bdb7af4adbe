#!/bin/bash
# round 4: what does the hot FF2 K loop (tiling 21) wait for?  PMC passes over the lab run + the counter list of this box
mkdir -p gpurun_out/r4c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD|SQ|SPI|GRBM|LDS)_[A-Z0-9_a-z]+" | sort -u > gpurun_out/r4c/counters.txt
wc -l gpurun_out/r4c/counters.txt
bash tools/pmc.sh gemm_conv_kernel tools/gemm_lab/lab time 4096,1280,5120,1,br cfgs=21 reps=40 nocold > gpurun_out/r4c/pmc_ff2_t21.txt 2>&1
cat gpurun_out/r4c/pmc_ff2_t21.txt
i=10
for pm in "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" "TA_BUSY_sum TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1)); rocprofv3 --pmc $pm --kernel-trace --output-format csv -d gpurun_out/pmc -o q$i -- tools/gemm_lab/lab time 4096,1280,5120,1,br cfgs=21 reps=40 nocold > gpurun_out/r4c/q$i.log 2>&1; tail -2 gpurun_out/r4c/q$i.log
done
python - <<PY > gpurun_out/r4c/pmc_tcp.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc/*q1*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_conv_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:44s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
cat gpurun_out/r4c/pmc_tcp.txt
