mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -30 gpurun_out/pytest_full.log
timeout 2400 python tools/make_tune_table.py gpurun_out/tuned_r2.json > gpurun_out/tune_r2.log 2>&1
echo "tune rc=$?"; tail -5 gpurun_out/tune_r2.log
TMIX_TUNE_FILE=gpurun_out/tuned_r2.json timeout 900 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err
echo "bench rc=$?"; head -c 600 gpurun_out/bench_r2b.json; tail -3 gpurun_out/bench_r2b.err
