#!/bin/bash
# A/B two source trees (python + .so) on the same box, interleaved: tools/ab_tree.sh <treeA> <treeB> [bench args]
a=$(realpath $1); b=$(realpath $2); shift 2
for i in 1 2 3; do
  for t in $a $b; do
    v=$(cd $t && python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],2))")
    echo "$t $v"
  done
done
