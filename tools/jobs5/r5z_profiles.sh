#!/bin/bash
# round 5, last job: default bench line + the profiles/ set on the final tree (tag = $1)
tag=${1:-r5c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 1200 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; tail -c 300 gpurun_out/$tag/bench_default.err
timeout 2400 bash tools/collect_profile.sh $tag > gpurun_out/$tag/collect.log 2>&1; tail -3 gpurun_out/$tag/collect.log
timeout 2400 bash tools/collect_profile_extra.sh $tag > gpurun_out/$tag/collect_extra.log 2>&1; tail -4 gpurun_out/$tag/collect_extra.log
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], "first", d["first_window"]["ms_per_step"], "frac", d["roofline"]["frac"], "launches", d["roofline"]["launches_per_step_all_classes"], "boundaries", d["roofline"]["kernel_boundaries_ms"])
print({k:(round(v['sum_launch_ms'],3), round(v['tflops'])) for k,v in d["roofline"]["classes"].items()})
for k,v in d.get("other_configs",{}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value","ms_per_step","plan_build_s","trajectory_steps_per_s","images_per_s","launches_per_step")})
print(d["trajectory"].get("per_rank_images_per_s"), d["trajectory"].get("gather_s"), d["images_per_s"], d["trajectory_steps_per_s"], d["trajectory"]["single_image"])
PY
