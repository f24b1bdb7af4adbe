"""one I2VGen-XL denoising step (bench.video_step_bench) alone: ms per step and kernels per step.   python tools/video_one.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
r = bench.video_step_bench()
print(json.dumps({k: r[k] for k in ("ms_per_step", "launches_per_step", "achieved_tflops")}))
