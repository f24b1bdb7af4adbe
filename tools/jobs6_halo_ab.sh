#!/bin/bash
# round 6: the halo-patch convolution in the captured step -- same box, shipped table vs the table with tiling 26 on every eligible convolution
mkdir -p gpurun_out/r6d
python tools/table_with_halo.py tweediemix_amd/tuned_gfx950.json gpurun_out/r6d/halo_all.json
python tools/table_with_halo.py tweediemix_amd/tuned_gfx950.json gpurun_out/r6d/halo_128_64.json --only "4,128,128;4,64,64"
for rep in 1 2; do
for v in shipped halo_all halo_128_64; do
  if [ $v = shipped ]; then export TMIX_TUNE_FILE=tweediemix_amd/tuned_gfx950.json; else export TMIX_TUNE_FILE=gpurun_out/r6d/$v.json; fi
  python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline > gpurun_out/r6d/bench_${v}_$rep.json 2> gpurun_out/r6d/bench_${v}_$rep.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r6d/bench_${v}_$rep.json"))
c=d["roofline"]["classes"]
print("$v rep $rep: ms_per_step %.3f first %.3f | conv %.3f ms (%d launches, %.0f TF) gemm %.3f attn %.3f norm %.3f | boundaries %.3f | sclk %s power %s" % (d["ms_per_step"], d["first_window"]["ms_per_step"], c["conv"]["sum_launch_ms"], c["conv"]["launches"], c["conv"]["tflops"], c["gemm"]["sum_launch_ms"], c["attn"]["sum_launch_ms"], c["norm"]["sum_launch_ms"], d["roofline"]["kernel_boundaries_ms"], d["chip_state_under_load"].get("sclk_mhz"), d["chip_state_under_load"].get("power_raw")))
print("   other mask kind: %.3f ms; max|x| %.3g / %.3g" % (d["other_mask_kind_window"]["ms_per_step"], d["max_abs_latent_at_end_of_window"], d["other_mask_kind_window"]["max_abs_latent_at_end_of_window"]))
PY
done; done
