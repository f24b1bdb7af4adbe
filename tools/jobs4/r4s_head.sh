#!/bin/bash
# round 4: HEAD check -- the plan / sampler / kernel suites that the last commits touch, smoke(), and the default bench line
mkdir -p gpurun_out/r4s
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r4s/tests.log 2>&1; tail -3 gpurun_out/r4s/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4s/smoke.log 2>&1; tail -1 gpurun_out/r4s/smoke.log
timeout 1200 python bench.py > gpurun_out/r4s/bench_default.json 2> gpurun_out/r4s/bench_default.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4s/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value","ms_per_step","trajectory_steps_per_s","images_per_s")})
print(d["images_per_s"], d["trajectory_steps_per_s"], d["config"]["tilings"])
PY
