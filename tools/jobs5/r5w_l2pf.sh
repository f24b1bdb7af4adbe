#!/bin/bash
# round 5, job w: tiling 25 = tiling 23 with an L2 prefetcher wave (touches the tile's operand lines 8 K-tiles ahead): tests, hot / cold, then IN SITU through a table variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5w
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "w22" > gpurun_out/r5w/tests.log 2>&1; tail -3 gpurun_out/r5w/tests.log
timeout 600 python tools/w22_bench.py > gpurun_out/r5w/bench.log 2>&1; cat gpurun_out/r5w/bench.log | tail -8
python - <<'PY'
import json
t = json.load(open("tools/tables/r5d_tuned.json"))          # the table with tiling 23 on the N = 1280 launches
s = json.load(open("tweediemix_amd/tuned_gfx950.json"))
for k, v in t.items():
    if v == 23: s[k] = 25
json.dump(s, open("gpurun_out/r5w/tuned25.json", "w"), indent=0)
for k, v in t.items():
    if v == 23: s[k] = 23
json.dump(s, open("gpurun_out/r5w/tuned23.json", "w"), indent=0)
PY
run() {
  TMIX_TUNE_FILE=$2 TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5w/$1.json 2> gpurun_out/r5w/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5w/$1.json")); r=d['roofline']
print("$1", round(d['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3))
PY
  grep -E "'gemm', (1|4), (1024|4096), 1280, (1280|5120), 0, (19|20|21|23|25)\)" gpurun_out/r5w/$1.err | head -4
}
for i in 1 2; do
run shipped_$i $PWD/tweediemix_amd/tuned_gfx950.json
run t25_$i $PWD/gpurun_out/r5w/tuned25.json
done
run t23_1 $PWD/gpurun_out/r5w/tuned23.json
