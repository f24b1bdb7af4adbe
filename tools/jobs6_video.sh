#!/bin/bash
# round 6: the video step (BASELINE configs[4]) per shape, and the halo-patch convolutions' share of it (shipped table vs round 5's)
out=gpurun_out/r6k; mkdir -p $out
python tools/video_step_shapes.py 2 > $out/video_shapes.txt 2>&1; head -50 $out/video_shapes.txt
for v in r5table shipped r5table shipped; do
  if [ $v = r5table ]; then export TMIX_TUNE_FILE=tools/ab/table_r5.json; else unset TMIX_TUNE_FILE; fi
  echo -n "$v: "; python tools/video_one.py 2>/dev/null | tail -1
done 2>&1 | tee $out/video_ab.txt
