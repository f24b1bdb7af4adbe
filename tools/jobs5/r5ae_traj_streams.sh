#!/bin/bash
# round 5: 8 co-batched seeds per UNet launch as ONE chain of B = 32 / 16 against TWO chains of half the rows each (images/s incl. VAE decode)
cd $GRAFT_REPO_ROOT
for st in 1 2 1 2; do
  timeout 1500 python bench.py --kind lora --streams $st --no-video --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['trajectory']
print('streams $st', 'images/s', round(d['images_per_s'],4), 'traj steps/s', round(d['trajectory_steps_per_s'],2), 'single image', round(t['single_image']['seconds_loop'],3), 'headline ms', round(d['ms_per_step'],3), 'not in table', d['config']['tilings']['shapes_not_in_table'])"
done
