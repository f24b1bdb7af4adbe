#!/bin/bash
# round 6: how many bytes of the next launch's weights should a launch touch?  (TMIX_PF_CAP_MB: bytes per hint; the routed q/k/v weights are 39 MB, one loader wave reaches 16.8 MB, tiling 23's four reach all)
out=gpurun_out/r6z; mkdir -p $out
python - <<'PY' > $out/ff2_23.json
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
t["routed|('gemm', 4096, 1280, 5120, 1, 0, False, True, False, True, False)"] = 23
print(json.dumps(t))
PY
{
for r in 1 2; do
for cap in 0 4 10 17 26; do
  echo -n "cap $cap MB, shipped table: "; TMIX_PF_CAP_MB=$cap python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "cap $cap MB, routed FF2 on 23: "; TMIX_PF_CAP_MB=$cap TMIX_TUNE_FILE=$out/ff2_23.json python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
done
done
} 2>&1 | tee $out/pfcap.txt
