#!/bin/bash
# round 5, job a: the new parity tests (every timed call kind at SDXL size; full-size trajectory tolerance; w_period == batch)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 2400 python -m pytest tests/test_trajectory_fullsize_gpu.py -x -q -s > gpurun_out/r5a/traj.log 2>&1
echo "traj rc=$?" >> gpurun_out/r5a/rc.txt
timeout 2400 python -m pytest tests/test_unet_gpu.py -x -q -s -k "every_timed_call_kind" > gpurun_out/r5a/callkinds.log 2>&1
echo "callkinds rc=$?" >> gpurun_out/r5a/rc.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "period" > gpurun_out/r5a/period.log 2>&1
echo "period rc=$?" >> gpurun_out/r5a/rc.txt
tail -3 gpurun_out/r5a/*.log
