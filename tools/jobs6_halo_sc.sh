#!/bin/bash
# round 6: the halo-patch kernel with shortcut taps -- parity, then the captured step with the shortcut convolutions of the 64 x 64 / 32 x 32 levels on it (same box)
out=gpurun_out/r6m; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "halo or shortcut" 2>&1 | tail -3
python tools/table_with_halo.py tweediemix_amd/tuned_gfx950.json $out/halo_sc.json --small-tiles
for r in 1 2; do
for v in shipped halo_sc; do
  if [ $v = halo_sc ]; then export TMIX_TUNE_FILE=$out/halo_sc.json; else export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json; fi
  echo -n "$v: "; python tools/step_shapes.py fusion --kind lora 2>/dev/null | tail -1
  echo -n "$v plain: "; python tools/step_shapes.py plain --kind lora 2>/dev/null | tail -1
done; done 2>&1 | tee $out/halo_sc_ab.txt
