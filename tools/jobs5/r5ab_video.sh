#!/bin/bash
# round 5: the video plan's first-level clip-wide norms on the producers' partials (2688 blocks per channel) instead of the statistics kernel
cd $GRAFT_REPO_ROOT
for v in 32768 131072 32768 131072; do
  echo -n "TMIX_CLIP_COLSTATS_MAX=$v  "; TMIX_CLIP_COLSTATS_MAX=$v timeout 900 python tools/video_one.py 2>/dev/null | tail -1
done
