#!/bin/bash
# round 6: whole -m gpu suite (with durations), smoke(), default bench line, the profiles/ set (tag = $1), the 1024^2 n = 50 trajectory tolerance runs
tag=${1:-r6a}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/$tag/tests.log 2>&1; tail -22 gpurun_out/$tag/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$tag/smoke.log 2>&1; tail -1 gpurun_out/$tag/smoke.log
timeout 1200 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; tail -c 300 gpurun_out/$tag/bench_default.err
timeout 2400 bash tools/collect_profile.sh $tag > gpurun_out/$tag/collect.log 2>&1; tail -3 gpurun_out/$tag/collect.log
timeout 2400 bash tools/collect_profile_extra.sh $tag > gpurun_out/$tag/collect_extra.log 2>&1; tail -4 gpurun_out/$tag/collect_extra.log
for v in "lora bf16" "lora fp8" "custom bf16"; do
  set -- $v
  timeout 900 python tests/trajectory_parity.py --res 1024 --n 50 --kind $1 $([ $2 = fp8 ] && echo --fp8) --out gpurun_out/$tag/traj_1024_n50_${1}_$2.json > gpurun_out/$tag/traj_${1}_$2.log 2>&1
  python - gpurun_out/$tag/traj_1024_n50_${1}_$2.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if not isinstance(v, list)})
PY
done
timeout 900 python tests/trajectory_parity.py --res 1024 --n 50 --kind lora --masks overlap --no-teacher --out gpurun_out/$tag/traj_1024_n50_lora_bf16_overlap.json > gpurun_out/$tag/traj_overlap.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "launches", d["roofline"]["launches_per_step_all_classes"], "boundaries", d["roofline"]["kernel_boundaries_ms"])
print(d["roofline"]["conv"]); print(d["roofline"]["classes"])
print("first window", d["config"]["first_window_ms_per_step"], "other masks", d["config"]["other_mask_kind_ms_per_step"], d["max_abs_latent_at_end_of_window"], d["other_mask_kind_window"]["max_abs_latent_at_end_of_window"], d["chip_state_under_load"], d["other_mask_kind_window"]["chip_state_under_load"])
for k,v in d.get("other_configs",{}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value","ms_per_step","plan_build_s","trajectory_steps_per_s","images_per_s","launches_per_step")})
print(d["trajectory"].get("per_rank_images_per_s"), d["trajectory"].get("gather_s"), d["images_per_s"], d["trajectory_steps_per_s"], d["trajectory"]["single_image"])
PY
