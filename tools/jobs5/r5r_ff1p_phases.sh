#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
python - <<'PY'
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
for k, v in t.items():
    if "('gemm'" in k and v == 14: t[k] = 24
json.dump(t, open("gpurun_out/r5r/tuned24.json", "w"), indent=0)
PY
TMIX_TUNE_FILE=$PWD/gpurun_out/r5r/tuned24.json timeout 600 python tools/insitu_phases.py lora "10240" 2>/dev/null | tail -3
timeout 600 python tools/insitu_phases.py lora "10240" 2>/dev/null | tail -2
