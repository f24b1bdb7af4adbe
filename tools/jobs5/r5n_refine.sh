#!/bin/bash
# round 5, job n: graph-timed refine of the shipped table behind the slice-aware XCD remap and the one-launch attn2 (all candidates, heaviest shapes), then the A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout 2400 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r5n/tuned.json --top 16 --cobatch 4 > gpurun_out/r5n/refine.log 2>&1
echo "refine rc=$?"
grep "refine " gpurun_out/r5n/refine.log | awk '{ if ($(NF-6) != $(NF-4)) print }' | head -40
grep "^refined" gpurun_out/r5n/refine.log
for t in old new old new; do
  if [ $t = new ]; then export TMIX_TUNE_FILE=$PWD/gpurun_out/r5n/tuned.json; else export TMIX_TUNE_FILE=$PWD/tweediemix_amd/tuned_gfx950.json; fi
  timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>> gpurun_out/r5n/bench_$t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t', round(d['ms_per_step'],3), 'gemm', round(r['classes']['gemm']['sum_launch_ms'],3), 'conv', round(r['classes']['conv']['sum_launch_ms'],3))" | tee -a gpurun_out/r5n/ab.log
done
