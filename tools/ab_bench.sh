#!/bin/bash
# same-box A/B of whole-step time: tools/ab_bench.sh a.so b.so  (alternates the two builds, three rounds each)
cp tweediemix_amd/lib/libtmix_hip.so /tmp/orig.so
for r in 1 2 3; do for so in "$@"; do cp $so tweediemix_amd/lib/libtmix_hip.so
  echo "$so $(python bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"; done; done
cp /tmp/orig.so tweediemix_amd/lib/libtmix_hip.so
