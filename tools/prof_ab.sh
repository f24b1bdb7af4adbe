#!/bin/bash
# per-kernel stats of two trees on the same box, 1 stream, no graphs: tools/prof_ab.sh <treeA> <treeB>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
root=$(pwd)
for t in "$@"; do
  name=$(basename $(realpath $t))
  out=$root/gpurun_out/prof_ab_$name; rm -rf $out; mkdir -p $out
  (cd $t && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --no-cpu-baseline --streams 1 --steps 6 --warmup 2 --no-graphs > $out/bench.log 2>&1)
  rm -f $out/*kernel_trace.csv $out/*agent_info.csv
done
