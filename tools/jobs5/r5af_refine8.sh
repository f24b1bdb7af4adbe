#!/bin/bash
# round 5: graph-timed refine of the 8-seed plans' heaviest shapes (the images/s path), then images/s with the old and the new table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5af
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r5af/tuned.json --top 20 --cobatch 8 --only-cobatch --kinds lora > gpurun_out/r5af/refine.log 2>&1
echo "refine rc=$?"
grep "refine " gpurun_out/r5af/refine.log | awk '{ if ($(NF-6) != $(NF-4)) print }' | head -40
grep "^refined" gpurun_out/r5af/refine.log
for t in old new old new; do
  if [ $t = new ]; then export TMIX_TUNE_FILE=$PWD/gpurun_out/r5af/tuned.json; else export TMIX_TUNE_FILE=$PWD/tweediemix_amd/tuned_gfx950.json; fi
  timeout 900 python bench.py --kind lora --no-video --no-cpu-baseline 2>> gpurun_out/r5af/bench_$t.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$t', 'images/s', round(d['images_per_s'],4), 'traj steps/s', round(d['trajectory_steps_per_s'],2), 'headline', round(d['ms_per_step'],3))" | tee -a gpurun_out/r5af/ab.log
done
