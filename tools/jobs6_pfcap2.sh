#!/bin/bash
# round 6: the byte cap on the routed q/k/v weights' hint only (tensors over 30 MB), and on everything over 20 MB (q/k/v + FF1)
out=gpurun_out/r6z2; mkdir -p $out
python - <<'PY' > $out/ff2_23.json
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
t["routed|('gemm', 4096, 1280, 5120, 1, 0, False, True, False, True, False)"] = 23
print(json.dumps(t))
PY
{
for r in 1 2; do
echo -n "shipped: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
for over in 30 20; do
for cap in 1 8 12 17 22; do
  echo -n "over $over cap $cap MB, shipped table: "; TMIX_PF_CAP_OVER_MB=$over TMIX_PF_CAP_MB=$cap python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "over $over cap $cap MB, routed FF2 on 23: "; TMIX_PF_CAP_OVER_MB=$over TMIX_PF_CAP_MB=$cap TMIX_TUNE_FILE=$out/ff2_23.json python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
done
done
done
} 2>&1 | tee $out/pfcap2.txt
