# round 3, call R: whole -m gpu suite with the shipped library + table, then the profile set (kernel trace + PMC passes) of the bench command
mkdir -p gpurun_out/r3r
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3r/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r3r/pytest_full.log
bash tools/collect_profile.sh r3a > gpurun_out/r3r/collect.log 2>&1; echo "collect rc=$?"; tail -6 gpurun_out/r3r/collect.log
