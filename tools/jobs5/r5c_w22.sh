#!/bin/bash
# round 5, job c: tiling 23 (2 x 2 waves of 64 x 80): unit tests, hot / cold A/B against tilings 12 / 20 / 21
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "w22 or (plain and 23) or (epilogues and 23) or (layernorm_pair and 23) or (periodic and 23)" > gpurun_out/r5c/tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r5c/rc.txt
tail -5 gpurun_out/r5c/tests.log
timeout 600 python tools/w22_bench.py > gpurun_out/r5c/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r5c/rc.txt
cat gpurun_out/r5c/bench.log
