timeout 2700 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_full.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_full.log
tail -4 gpurun_out/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_r2d.json 2> gpurun_out/bench_r2d.err
echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/bench_r2d.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['images_per_s'], d['roofline']['achieved'], d['roofline']['frac'], d['other_configs']['custom']['value'], d['other_configs']['fp8']['value'])"
bash tools/collect_profile.sh r2d 2>&1 | tail -4
