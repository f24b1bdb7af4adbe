#!/bin/bash
# round 6: tiling 23's next-launch weight touches issued by the MATH waves (dev variant -DTMIX_W22_PF_MATH) instead of in front of the loader waves' first K-tile;
# in the captured LoRA step, with the routed FF2 on tiling 19 (shipped table) and on tiling 23
out=gpurun_out/r6x; mkdir -p $out
python - <<'PY' > $out/ff2_23.json
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
k = "routed|('gemm', 4096, 1280, 5120, 1, 0, False, True, False, True, False)"
assert k in t, k
t[k] = 23
print(json.dumps(t))
PY
V=tools/ab/w22pfm/libtmix_hip.so
{
for r in 1 2 3; do
  echo -n "shipped lib, shipped table: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "math-wave touches, shipped table: "; TMIX_LIB=$V python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "shipped lib, routed FF2 on 23: "; TMIX_TUNE_FILE=$out/ff2_23.json python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "math-wave touches, routed FF2 on 23: "; TMIX_TUNE_FILE=$out/ff2_23.json TMIX_LIB=$V python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
done
echo "== per shape, math-wave touches + routed FF2 on 23"
TMIX_TUNE_FILE=$out/ff2_23.json TMIX_LIB=$V python tools/step_shapes.py fusion --kind lora 2>/dev/null | grep "gemm" | head -12
echo "== custom step"
for r in 1 2; do
  echo -n "shipped: "; python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
  echo -n "math-wave touches: "; TMIX_LIB=$V python tools/step_shapes.py fusion --kind custom 2>/dev/null | tail -1
done
TMIX_LIB=$V python -m pytest tests/test_ops_gpu.py -q -x -k "w22 or tiling_23 or tile_cfg or every_tiling" 2>&1 | tail -2
} 2>&1 | tee $out/pfmath.txt
