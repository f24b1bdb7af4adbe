#!/bin/bash
# A/B two builds of libtmix_hip.so on the same box, interleaved: tools/ab.sh tools/ab/old.so tools/ab/new.so [bench args]
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for so in $a $b; do
    cp $so tweediemix_amd/lib/libtmix_hip.so
    v=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
    echo "$so $v"
  done
done
