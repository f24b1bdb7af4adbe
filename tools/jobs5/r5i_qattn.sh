#!/bin/bash
# round 5, job i: attn2 in one launch -- unit tests, hot A/B, then the step A/B (TMIX_NO_QATTN=1 = the two-launch form)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "cross_att" > gpurun_out/r5i/tests.log 2>&1; tail -12 gpurun_out/r5i/tests.log
timeout 600 python tools/qattn_bench.py > gpurun_out/r5i/bench.log 2>&1; cat gpurun_out/r5i/bench.log | tail -8
run() {
  TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5i/$1.json 2> gpurun_out/r5i/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5i/$1.json")); r=d['roofline']
print("$1", round(d['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3), 'launches', r['launches_per_step_all_classes'], 'parity', d['parity_check']['rel_l2'])
PY
}
for i in 1 2; do
TMIX_NO_QATTN=1 run two_$i
run fused_$i
done
tail -3 gpurun_out/r5i/fused_1.err
