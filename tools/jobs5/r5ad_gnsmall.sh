#!/bin/bash
# round 5: one-launch GroupNorm for small images -- tests, then the video step with / without it
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_i2vgen_gpu.py tests/test_video_gpu.py tests/test_vae_gpu.py -x -q 2>&1 | tail -3
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export TMIX_GN_NO_SMALL=1; else unset TMIX_GN_NO_SMALL; fi
  echo -n "TMIX_GN_NO_SMALL=$v  "; timeout 900 python tools/video_one.py 2>/dev/null | tail -1
done
