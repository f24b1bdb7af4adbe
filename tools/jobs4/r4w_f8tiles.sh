#!/bin/bash
# round 4: e4m3 lock-step loop with one / two / four loader waves: kernel tests, then the fp8 step with each as the N = 1280 / 640 tiling (same box)
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "fp8" > gpurun_out/r4w/tests.log 2>&1; tail -2 gpurun_out/r4w/tests.log
for i in 1 2; do
for t in 21 20 19; do
  TMIX_FP8_TILE=$t timeout 600 python bench.py --dtype fp8 --kind lora --no-trajectory --no-cpu-baseline --no-video --steps 50 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp8 tile $t', d['value'], d['ms_per_step'], {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()})"
done
done
