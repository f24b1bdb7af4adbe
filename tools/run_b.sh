mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sampler_gpu.py tests/test_i2vgen_gpu.py tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/pytest_b.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_b.log
TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline > gpurun_out/bench_s2.json 2> gpurun_out/bench_s2.err
TMIX_BENCH_SHAPES=1 timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --streams 1 > gpurun_out/bench_s1.json 2> gpurun_out/bench_s1.err
timeout 900 python bench.py --kind custom --no-trajectory --no-cpu-baseline --seeds-per-gpu 2 > gpurun_out/bench_s2x2.json 2> gpurun_out/bench_s2x2.err
head -c 400 gpurun_out/bench_s2.json; echo; head -c 400 gpurun_out/bench_s1.json; echo; head -c 400 gpurun_out/bench_s2x2.json
