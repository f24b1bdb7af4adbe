#!/bin/bash
# round 4: the whole -m gpu suite, smoke(), and the default bench line
mkdir -p gpurun_out/r4k
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r4k/tests.log 2>&1; tail -4 gpurun_out/r4k/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4k/smoke.log 2>&1; tail -2 gpurun_out/r4k/smoke.log
timeout 900 python bench.py > gpurun_out/r4k/bench.json 2> gpurun_out/r4k/bench.err; tail -c 200 gpurun_out/r4k/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r4k/bench.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "launches", d["roofline"]["launches_per_step_all_classes"], "boundaries", d["roofline"]["kernel_boundaries_ms"])
print(d["roofline"]["conv"])
print({k:(v.get("value"), v.get("ms_per_step"), v.get("plan_build_s")) for k,v in d.get("other_configs",{}).items() if isinstance(v,dict)})
print(d["trajectory"].get("per_rank_images_per_s"), d["trajectory"].get("gather_s"), d["images_per_s"], d["trajectory_steps_per_s"])
PY
