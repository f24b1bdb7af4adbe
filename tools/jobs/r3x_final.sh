# round 3, call X: the driver's round-end sequence on the final tree: build check, -m gpu, smoke, default bench
mkdir -p gpurun_out/r3x
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3x/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3x/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok"
timeout 1500 python bench.py > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/bench.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r3x/bench.json)"
python -c "
import json; d=json.loads(open('gpurun_out/r3x/bench.json').read()); print(d['value'], d['ms_per_step'], d['images_per_s'], d['config']['tilings']['follow_shipped_table'], d['dist']['ranks_seen'])"
