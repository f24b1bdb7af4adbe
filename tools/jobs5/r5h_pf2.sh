#!/bin/bash
# round 5, job h: next-weight prefetch touches spread over the K loop (loader-wave tilings + tiling 23) vs in front of K-tile 0 (tools/ab/pf0), both tables
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -x -q -k "w22 or prefetch or (plain and (21 or 20 or 19 or 23)) or tiny" > gpurun_out/r5h/tests.log 2>&1; tail -2 gpurun_out/r5h/tests.log
run() {  # name lib table
  TMIX_LIB=$2 TMIX_TUNE_FILE=$3 TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5h/$1.json 2> gpurun_out/r5h/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5h/$1.json")); r=d['roofline']
print("$1", round(d['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3), 'replay', round(r['graph_replay_ms'],3))
PY
}
OLD=$PWD/tools/ab/pf0/libtmix_hip.so; NEW=$PWD/tweediemix_amd/lib/libtmix_hip.so
T0=$PWD/tweediemix_amd/tuned_gfx950.json; T1=$PWD/tools/tables/r5d_tuned.json
for i in 1 2; do
run pf0_t0_$i $OLD $T0
run pf2_t0_$i $NEW $T0
run pf0_t1_$i $OLD $T1
run pf2_t1_$i $NEW $T1
done
TMIX_TUNE_FILE=$T1 timeout 600 python tools/insitu_phases.py lora gemm > gpurun_out/r5h/phases_pf2_t1.txt 2>/dev/null; head -8 gpurun_out/r5h/phases_pf2_t1.txt
TMIX_TUNE_FILE=$T0 timeout 600 python tools/insitu_phases.py lora gemm > gpurun_out/r5h/phases_pf2_t0.txt 2>/dev/null; head -8 gpurun_out/r5h/phases_pf2_t0.txt
