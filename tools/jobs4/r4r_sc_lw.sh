#!/bin/bash
# round 4: shortcut taps in the two-loader convolution (tiling 20): kernel test, then re-time the 72 shortcut-conv entries of the tile table with 20 as a real candidate
mkdir -p gpurun_out/r4r; rm -f gpurun_out/r4r/*
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "shortcut_taps" > gpurun_out/r4r/tests.log 2>&1; tail -3 gpurun_out/r4r/tests.log
python - <<PY
import json
t=json.load(open("tweediemix_amd/tuned_gfx950.json"))
n=0
for k in list(t):
    kk=k.split("|")[-1]
    if kk.startswith("('conv'") and len(eval(kk))==8: del t[k]; n+=1
json.dump(t, open("gpurun_out/r4r/table_in.json","w"), indent=0); print("dropped", n)
PY
TMIX_TUNE_FILE=gpurun_out/r4r/table_in.json timeout 1500 python tools/extend_table_missing.py gpurun_out/r4r/tuned_gfx950.json > gpurun_out/r4r/extend.log 2>&1; tail -2 gpurun_out/r4r/extend.log | cut -c1-300
