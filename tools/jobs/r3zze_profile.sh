# round 3, call ZZE: the profile set r3c of the final library (kernel trace bf16 + fp8, FETCH / WRITE / MfmaUtil passes)
mkdir -p gpurun_out/r3zze
bash tools/collect_profile.sh r3c > gpurun_out/r3zze/collect.log 2>&1; tail -6 gpurun_out/r3zze/collect.log
