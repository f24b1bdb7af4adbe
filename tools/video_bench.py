#!/usr/bin/env python3
"""BASELINE config #5 timing alone (the same measurement as bench.py's other_configs.video):
python tools/video_bench.py [--steps 10] [--res_w 768 --res_h 448] [--streams 1|2]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--res_w", type=int, default=768); ap.add_argument("--res_h", type=int, default=448)
ap.add_argument("--frames", type=int, default=16); ap.add_argument("--streams", type=int, default=2)
a = ap.parse_args()
print(json.dumps(bench.video_step_bench(a.steps, a.warmup, a.res_w, a.res_h, a.frames, a.streams)))
