#!/bin/bash
# round 5, job x: GroupNorm-apply work per workgroup (GN_APPLY_ITEMS 1024 -> 2048 / 4096 vectors): the 32 x 32 maps' apply launches stage 10 KB of scale / shift for 16 KB of data
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5x
for v in base gn2048 gn4096 base gn2048 gn4096; do
  lib=$PWD/tweediemix_amd/lib/libtmix_hip.so; [ $v != base ] && lib=$PWD/tools/ab/$v/libtmix_hip.so
  TMIX_LIB=$lib timeout 900 python bench.py --kind lora --no-trajectory --no-video --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', round(d['ms_per_step'],3), 'norm', round(r['classes']['norm']['sum_launch_ms'],3), 'bound', round(r['kernel_boundaries_ms'],3))"
done
