#!/bin/bash
# round 4: second refine pass from the refined table: more shapes (top 60), both concept kinds
mkdir -p gpurun_out/r4v
timeout 3000 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r4v/refined2.json --top 60 --kinds lora,custom > gpurun_out/r4v/refine.log 2>&1
grep -E "refined|wrote" gpurun_out/r4v/refine.log | tail -8
grep -E "refine .*-> " gpurun_out/r4v/refine.log | grep -v -E ": ([0-9]+) -> \1 " | head -30
