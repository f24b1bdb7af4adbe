#!/bin/bash
# round 5, job y: key-split tail of the self-attention kernel -- parity, hot timing, then in the step (A/B through TMIX_ATTN_NO_SPLIT)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5y
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r5y/tests.log 2>&1
tail -5 gpurun_out/r5y/tests.log
for s in "4 20 1024 1024" "4 10 4096 4096" "8 20 1024 1024" "16 20 1024 1024" "2 20 1024 1024"; do timeout 300 python tools/attn_one.py $s; done 2>&1 | tee gpurun_out/r5y/hot.log
for v in nosplit split nosplit split; do
  if [ $v = nosplit ]; then export TMIX_ATTN_NO_SPLIT=1; else unset TMIX_ATTN_NO_SPLIT; export TMIX_ATTN_SPLIT=1; fi
  timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['ms_per_step'],3), 'attn', round(r['classes']['attn']['sum_launch_ms'],3), r['classes']['attn']['tflops'], 'bound', round(r['kernel_boundaries_ms'],3), 'parity', d['parity_check']['rel_l2'])"
done | tee gpurun_out/r5y/step.log
