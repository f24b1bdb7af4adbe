#!/bin/bash
# round 5: whole -m gpu suite, smoke(), default bench line, then the profiles/ set (tag = $1)
tag=${1:-r5a}
mkdir -p gpurun_out/$tag
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/$tag/tests.log 2>&1; tail -4 gpurun_out/$tag/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$tag/smoke.log 2>&1; tail -1 gpurun_out/$tag/smoke.log
timeout 1200 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; tail -c 300 gpurun_out/$tag/bench_default.err
timeout 2400 bash tools/collect_profile.sh $tag > gpurun_out/$tag/collect.log 2>&1; tail -3 gpurun_out/$tag/collect.log
timeout 2400 bash tools/collect_profile_extra.sh $tag > gpurun_out/$tag/collect_extra.log 2>&1; tail -4 gpurun_out/$tag/collect_extra.log
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "launches", d["roofline"]["launches_per_step_all_classes"], "boundaries", d["roofline"]["kernel_boundaries_ms"])
print(d["roofline"]["conv"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value","ms_per_step","plan_build_s","trajectory_steps_per_s","images_per_s","launches_per_step")})
print(d["trajectory"].get("per_rank_images_per_s"), d["trajectory"].get("gather_s"), d["images_per_s"], d["trajectory_steps_per_s"])
PY
