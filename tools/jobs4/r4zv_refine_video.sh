#!/bin/bash
# round 4: graph-timed refine of the video step's tile choices (two chains; one chain found nothing to change), A/B against the shipped table on the same box,
# and a short bench run (the in-situ profile on the host clock)
mkdir -p gpurun_out/r4zv; rm -f gpurun_out/r4zv/*
timeout 1500 python tools/refine_video.py gpurun_out/r4zv/t2.json 2 14 > gpurun_out/r4zv/refine2.log 2>&1; echo "refine2 rc=$?"; grep -v amdgpu gpurun_out/r4zv/refine2.log | tail -18 | cut -c1-220
T=gpurun_out/r4zv/t2.json
[ -f $T ] && for r in 1 2; do
  echo "streams 2 shipped: $(timeout 600 python tools/video_bench.py --streams 2 2>/dev/null | tail -1 | cut -c232-300)"
  echo "streams 2 refined: $(TMIX_TUNE_FILE=$T timeout 600 python tools/video_bench.py --streams 2 2>/dev/null | tail -1 | cut -c232-300)"
done | tee gpurun_out/r4zv/ab.txt
timeout 900 python bench.py --kind lora --steps 20 --warmup 5 --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("bench", d["ms_per_step"], r["frac"], r["graph_replay_ms"], r["uninstrumented_graph_replay_ms"], r["kernel_boundaries_ms"])'
