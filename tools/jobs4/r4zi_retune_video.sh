#!/bin/bash
# round 4: the video step's tilings re-timed with this round's kernels (in situ autotune), then the step with the shipped table and with the new one, same box
mkdir -p gpurun_out/r4zi; rm -f gpurun_out/r4zi/*
timeout 1500 python tools/retune_video.py gpurun_out/r4zi/video_table.json > gpurun_out/r4zi/retune.log 2>&1; tail -40 gpurun_out/r4zi/retune.log
for s in 1 2; do
  echo "== streams $s shipped"; timeout 600 python tools/video_bench.py --streams $s 2>/dev/null | tail -1
  echo "== streams $s retuned"; TMIX_TUNE_FILE=gpurun_out/r4zi/video_table.json timeout 600 python tools/video_bench.py --streams $s 2>/dev/null | tail -1
done 2>&1 | tee gpurun_out/r4zi/ab.txt
