#!/bin/bash
# round 4: periodic weight sets (tmix_gemm_desc.w_period) -- the new tests, then the B = 16 fusion call (4 seeds co-batched) per shape, bf16 and fp8
mkdir -p gpurun_out/r4zd; rm -f gpurun_out/r4zd/*
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "periodic or period or batched_weights" 2>&1 | tail -5 | tee gpurun_out/r4zd/tests.log
for dt in bf16 fp8; do
  timeout 600 python tools/step_shapes.py fusion --dtype $dt --seeds-per-gpu 4 > gpurun_out/r4zd/$dt.out 2> gpurun_out/r4zd/$dt.err
done
cat gpurun_out/r4zd/bf16.out; grep -h "total=" gpurun_out/r4zd/bf16.err | head -8; tail -3 gpurun_out/r4zd/bf16.err; cat gpurun_out/r4zd/fp8.out; grep -h "total=" gpurun_out/r4zd/fp8.err | head -8
nvidia-smi 2>/dev/null; rocm-smi --showmemuse 2>/dev/null | head -8
