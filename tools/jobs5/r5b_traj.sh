#!/bin/bash
# round 5, job b (re-run on the final tree as r5t): the full-size trajectory tolerance test + the 1024^2 n = 50 runs of the same routine (bf16, fp8), recorded under gpurun_out/r5t
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
timeout 1800 python -m pytest tests/test_trajectory_fullsize_gpu.py -q -s > gpurun_out/r5t/traj.log 2>&1
echo "traj rc=$?" >> gpurun_out/r5t/rc.txt
timeout 1500 python tests/trajectory_parity.py --res 1024 --n 50 --kind lora --out gpurun_out/r5t/traj_1024_n50_lora_bf16.json > gpurun_out/r5t/t1.log 2>&1
echo "bf16 rc=$?" >> gpurun_out/r5t/rc.txt
timeout 1500 python tests/trajectory_parity.py --res 1024 --n 50 --kind lora --fp8 --out gpurun_out/r5t/traj_1024_n50_lora_fp8.json > gpurun_out/r5t/t2.log 2>&1
echo "fp8 rc=$?" >> gpurun_out/r5t/rc.txt
timeout 1500 python tests/trajectory_parity.py --res 1024 --n 50 --kind custom --out gpurun_out/r5t/traj_1024_n50_custom_bf16.json > gpurun_out/r5t/t3.log 2>&1
echo "custom rc=$?" >> gpurun_out/r5t/rc.txt
cat gpurun_out/r5t/rc.txt
