#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 600 python tools/insitu_phases.py lora "1280, 1280" > gpurun_out/r5j/fused.txt 2> gpurun_out/r5j/fused.err
TMIX_NO_QATTN=1 timeout 600 python tools/insitu_phases.py lora "1280, 1280" > gpurun_out/r5j/two.txt 2> gpurun_out/r5j/two.err
cat gpurun_out/r5j/fused.txt gpurun_out/r5j/two.txt
timeout 600 python tools/insitu_phases.py lora "attn" > gpurun_out/r5j/attn.txt 2>/dev/null; cat gpurun_out/r5j/attn.txt
