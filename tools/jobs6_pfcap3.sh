#!/bin/bash
# round 6: the weight-hint policy "tensors over 20 MB: touch 8 MB (or 1 MB) of them" across the plans the bench times
out=gpurun_out/r6z3; mkdir -p $out
python - <<'PY' > $out/ff2_23.json
import json
t = json.load(open("tweediemix_amd/tuned_gfx950.json"))
t["routed|('gemm', 4096, 1280, 5120, 1, 0, False, True, False, True, False)"] = 23
print(json.dumps(t))
PY
run() { echo -n "$1 | shipped: "; python tools/step_shapes.py $2 2>/dev/null | tail -1
        echo -n "$1 | over 20 cap 8: "; TMIX_PF_CAP_OVER_MB=20 TMIX_PF_CAP_MB=8 python tools/step_shapes.py $2 2>/dev/null | tail -1
        echo -n "$1 | over 20 cap 1: "; TMIX_PF_CAP_OVER_MB=20 TMIX_PF_CAP_MB=1 python tools/step_shapes.py $2 2>/dev/null | tail -1
        echo -n "$1 | over 12 cap 8: "; TMIX_PF_CAP_OVER_MB=12 TMIX_PF_CAP_MB=8 python tools/step_shapes.py $2 2>/dev/null | tail -1; }
{
for r in 1 2; do
run "custom fusion" "fusion --kind custom"
run "lora plain B=2" "plain --kind lora"
run "lora start" "start --kind lora"
run "lora fusion fp8" "fusion --kind lora --dtype fp8"
done
run "lora fusion 8 seeds" "fusion --kind lora --seeds-per-gpu 8"
run "lora plain 8 seeds" "plain --kind lora --seeds-per-gpu 8"
for v in "" "TMIX_PF_CAP_OVER_MB=20 TMIX_PF_CAP_MB=8" "TMIX_PF_CAP_OVER_MB=20 TMIX_PF_CAP_MB=1"; do
  echo -n "video | $v: "; env $v python tools/video_one.py 2>/dev/null | tail -1
done
} 2>&1 | tee $out/pfcap3.txt
