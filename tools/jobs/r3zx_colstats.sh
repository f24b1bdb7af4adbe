# round 3, call ZX: GroupNorm column statistics from the producing GEMM / conv epilogue -- the new kernel tests, then the neighbours
mkdir -p gpurun_out/r3zx
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "column_statistics or producer_partials or through_the_partials" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "groupnorm or gemm_epilogues or conv3x3 or fused_layernorm or e4m3_copy" 2>&1 | tail -3
