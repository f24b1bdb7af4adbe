#!/bin/bash
# round 5: does any HIP runtime knob move the 596 kernel boundaries of the captured step?  (same box, default first and last)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5aa
run() {
  env "$@" timeout 600 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$*', round(d['ms_per_step'],3), 'first', round(d['first_window']['ms_per_step'],3), 'bound', round(r['kernel_boundaries_ms'],3), 'parity', d['parity_check']['rel_l2'])
except Exception as e:
    print('$*', 'FAILED', e)"
}
run A=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run GPU_FLUSH_ON_EXECUTION=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run ROC_USE_FGS_KERNARG=0
run A=0
