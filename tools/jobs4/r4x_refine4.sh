#!/bin/bash
# round 4: graph-timed refine of the 4-seed (images/s) entries of the table, lora plans
mkdir -p gpurun_out/r4x
timeout 3000 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r4x/refined4.json --cobatch 4 --only-cobatch --top 40 --kinds lora > gpurun_out/r4x/refine.log 2>&1
grep -E "refined|wrote" gpurun_out/r4x/refine.log | tail -8
grep -E "refine .*-> " gpurun_out/r4x/refine.log | grep -v -E ": ([0-9]+) -> \1 " | head -30
