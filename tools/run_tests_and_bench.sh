mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -40 gpurun_out/pytest_full.log
timeout 900 python bench.py > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_r2a.json; tail -5 gpurun_out/bench_r2a.err
