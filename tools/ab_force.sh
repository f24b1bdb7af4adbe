#!/bin/bash
# A/B two trees with every GEMM/conv forced to one tiling: tools/ab_force.sh <treeA> <treeB> <cfg...>
a=$(realpath $1); b=$(realpath $2); shift 2
for cfg in "$@"; do
  for t in $a $b $a $b; do
    v=$(cd $t && TMIX_FORCE_TILE=$cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 2>&1 | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
    echo "cfg$cfg $t $v"
  done
done
