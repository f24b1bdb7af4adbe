timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "linear_small" 2>&1 | tail -3
for cb in 8; do
python bench.py --kind lora --steps 5 --warmup 2 --no-cpu-baseline --traj-cobatch $cb --traj-images 8 2>gpurun_out/p_$cb.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cobatch $cb', d.get('images_per_s'), json.dumps(d.get('trajectory'))[:600])"
done
