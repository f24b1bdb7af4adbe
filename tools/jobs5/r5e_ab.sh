#!/bin/bash
# round 5, job e: same-box A/B of the shipped table vs the table with tiling 23 (tools/tables/r5d_tuned.json), full JSON + per-shape stderr kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
for t in old new old new; do
  if [ $t = new ]; then export TMIX_TUNE_FILE=$PWD/tools/tables/r5d_tuned.json; else export TMIX_TUNE_FILE=$PWD/tweediemix_amd/tuned_gfx950.json; fi
  n=$(ls gpurun_out/r5e | grep -c "^$t")
  TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5e/${t}_$n.json 2> gpurun_out/r5e/${t}_$n.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5e/${t}_$n.json")); r=d['roofline']
print("$t", round(d['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3), 'replay', round(r['graph_replay_ms'],3), round(r['uninstrumented_graph_replay_ms'],3), 'busy', round(r['instrumented_busy_ms'],3))
PY
done
