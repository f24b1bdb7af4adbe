#!/bin/bash
# round 5, job u: is the one-launch attn2's K loop (15.6 us against 12.6 for the four-wave to_q) long because of the doubly-loaded SIMD or because of its 3-deep ring?
cd $GRAFT_REPO_ROOT
for lib in tweediemix_amd/lib/libtmix_hip.so tools/ab/qabl4/libtmix_hip.so; do
  echo "== $lib"; TMIX_LIB=$PWD/$lib timeout 600 python tools/insitu_phases.py lora "1280, 1280, 0, False, True" 2>/dev/null | tail -2
done
