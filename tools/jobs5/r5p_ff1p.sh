#!/bin/bash
# round 5, job p: FF1 on persistent workgroups (tiling 24): unit tests, hot A/B, then the step with FF1 forced to 24 through a table variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "persistent_geglu or (geglu and 24)" > gpurun_out/r5p/tests.log 2>&1; tail -6 gpurun_out/r5p/tests.log
timeout 600 python tools/ff1p_bench.py > gpurun_out/r5p/bench.log 2>&1; tail -5 gpurun_out/r5p/bench.log
python - <<'PY'
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
n = 0
for k in list(t):
    if k.replace("shared|", "").startswith("('gemm'") and t[k] in (14, 22) and ", 1, True" not in k:
        pass
for k, v in t.items():
    if "('gemm'" in k and v == 14: t[k] = 24; n += 1
json.dump(t, open("gpurun_out/r5p/tuned24.json", "w"), indent=0)
print("entries moved 14 -> 24:", n)
PY
run() {
  TMIX_TUNE_FILE=$2 TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r5p/$1.json 2> gpurun_out/r5p/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r5p/$1.json")); r=d['roofline']
print("$1", round(d['ms_per_step'],3), round(d['first_window']['ms_per_step'],3), {k:(round(v['sum_launch_ms'],3), v['launches']) for k,v in r['classes'].items()}, 'bound', round(r['kernel_boundaries_ms'],3), 'parity', d['parity_check']['rel_l2'])
PY
  grep -E "'gemm', 1, (4096|16384), (10240|5120), (1280|640)" gpurun_out/r5p/$1.err
}
for i in 1 2; do
run t14_$i $PWD/tweediemix_amd/tuned_gfx950.json
run t24_$i $PWD/gpurun_out/r5p/tuned24.json
done
