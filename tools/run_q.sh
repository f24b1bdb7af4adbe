python tools/extend_table_seeds8.py gpurun_out/tuned_ext.json 2>&1 | tail -4
for cb in 4 8; do
TMIX_TUNE_FILE=gpurun_out/tuned_ext.json python bench.py --kind lora --steps 5 --warmup 2 --no-cpu-baseline --traj-cobatch $cb --traj-images 8 2>gpurun_out/q_$cb.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cobatch $cb', d.get('images_per_s'), d['trajectory']['seconds'])"
done
