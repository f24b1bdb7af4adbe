#!/bin/bash
# round 5, job d: tiling 23 into the tile table (graph-timed refine of the shipped table with 23 as the only new candidate), then the same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
timeout 2400 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r5d/tuned.json --cands 23 --top 40 --cobatch 4 > gpurun_out/r5d/refine.log 2>&1
echo "refine rc=$?" > gpurun_out/r5d/rc.txt
grep -c "-> 23" gpurun_out/r5d/refine.log
for t in old new old new; do
  if [ $t = new ]; then export TMIX_TUNE_FILE=$PWD/gpurun_out/r5d/tuned.json; else export TMIX_TUNE_FILE=$PWD/tweediemix_amd/tuned_gfx950.json; fi
  timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>> gpurun_out/r5d/bench_$t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t', d['ms_per_step'], 'gemm', r['classes']['gemm']['sum_launch_ms'], 'frac', r['frac'], 'replay', r['graph_replay_ms'], d['config']['tilings']['used'])" | tee -a gpurun_out/r5d/ab.log
done
