#!/bin/bash
# round 4: fp8 kernels (lock-step tilings, attention e4m3 output) parity, fp8 plans vs oracle, then the fp8 bench leg
mkdir -p gpurun_out/r4h; rm -f gpurun_out/r4h/*
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "fp8 or e4m3 or attention" > gpurun_out/r4h/tests_ops.log 2>&1; tail -3 gpurun_out/r4h/tests_ops.log
timeout 1200 python -m pytest tests/test_unet_gpu.py -q -x -s -m gpu -k "fp8 or hostile" > gpurun_out/r4h/tests_unet.log 2>&1; grep -E "rel_l2|rel-L2|passed|failed|Error" gpurun_out/r4h/tests_unet.log | tail -12
timeout 600 python bench.py --dtype fp8 --kind lora --no-trajectory --steps 30 --warmup 5 > gpurun_out/r4h/bench_fp8.json 2> gpurun_out/r4h/bench_fp8.err; tail -c 600 gpurun_out/r4h/bench_fp8.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4h/bench_fp8.json"))
print(d["value"], d["ms_per_step"], d["dtype"])
for k,v in d["roofline"]["classes"].items(): print(k, v["launches"], round(v["sum_launch_ms"],3), round(v["tflops"],1))
PY
