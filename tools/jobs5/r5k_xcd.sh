#!/bin/bash
# round 5, job k: slice-aware XCD remap (whole-grid linear id) vs the remap of blockIdx.x alone (tools/ab/xcdx), LoRA (routed, per-slice weights) and custom (shared);
# then the persistent-stage microbenchmark
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "periodic or batched or cross_att or w22" > gpurun_out/r5k/tests.log 2>&1; tail -2 gpurun_out/r5k/tests.log
run() {  # name lib kind
  TMIX_LIB=$2 TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind $3 --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5k/$1.json 2> gpurun_out/r5k/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5k/$1.json")); r=d['roofline']
print("$1", round(d['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3))
PY
}
OLD=$PWD/tools/ab/xcdx/libtmix_hip.so; NEW=$PWD/tweediemix_amd/lib/libtmix_hip.so
for i in 1 2; do
run xonly_lora_$i $OLD lora
run grid_lora_$i $NEW lora
done
run xonly_custom $OLD custom
run grid_custom $NEW custom
grep -E "'gemm', 4, 1024, (1280|3840), 1280" gpurun_out/r5k/xonly_lora_1.err gpurun_out/r5k/grid_lora_1.err
(cd tools/ubench && make persist >/dev/null 2>&1; for T in 15000 40000; do ./persist 64 $T 128 40; done; ./persist 64 15000 512 40) > gpurun_out/r5k/persist.txt 2>&1; cat gpurun_out/r5k/persist.txt
