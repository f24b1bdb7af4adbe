#!/bin/bash
# round 4: the default bench line + the profiles/ set (kernel trace, PMC traffic, MfmaUtil) of the current library; tag = $1
tag=${1:-r4a}
mkdir -p gpurun_out/$tag
timeout 900 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; tail -c 300 gpurun_out/$tag/bench_default.err
timeout 2400 bash tools/collect_profile.sh $tag > gpurun_out/$tag/collect.log 2>&1; tail -5 gpurun_out/$tag/collect.log
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/bench_default.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in d.get("other_configs",{}).items() if isinstance(v,dict)})
PY
