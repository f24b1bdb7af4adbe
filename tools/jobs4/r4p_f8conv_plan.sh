#!/bin/bash
# round 4: fp8 plans with the convolutions on e4m3 operands: oracle parity (friendly and hostile weights), then the fp8 bench leg with and without them (same box)
mkdir -p gpurun_out/r4p; rm -f gpurun_out/r4p/*
timeout 1500 python -m pytest tests/test_unet_gpu.py -q -x -s -m gpu -k "fp8 or hostile" > gpurun_out/r4p/tests_unet.log 2>&1; grep -E "rel_l2|rel-L2|passed|failed|Error" gpurun_out/r4p/tests_unet.log | tail -14
for v in conv noconv; do
  if [ $v = noconv ]; then export TMIX_FP8_NO_CONV=1; else unset TMIX_FP8_NO_CONV; fi
  timeout 600 python bench.py --dtype fp8 --kind lora --no-trajectory --no-cpu-baseline --no-video --steps 40 --warmup 10 > gpurun_out/r4p/bench_fp8_$v.json 2> gpurun_out/r4p/bench_fp8_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r4p/bench_fp8_$v.json"))
print("$v", d["value"], d["ms_per_step"], d["dtype"], {k:(v["launches"], round(v["sum_launch_ms"],2), round(v["tflops"])) for k,v in d["roofline"]["classes"].items()})
PY
done
