timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -3 gpurun_out/pytest_full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_r2e.json 2> gpurun_out/bench_r2e.err
echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/bench_r2e.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['images_per_s'], d['roofline']['achieved'], d['roofline']['frac'], d['other_configs']['custom']['value'], d['other_configs']['fp8']['value'])"
