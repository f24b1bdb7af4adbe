timeout 2700 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -18 gpurun_out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_r2c.json
bash tools/collect_profile.sh r2c 2>&1 | tail -5
