#!/usr/bin/env python3
"""extend the shipped tile table in place with the launch shapes of 8 co-batched seeds per GPU (B = 32 / 16 rows):
python tools/extend_table_seeds8.py [out.json]   (shapes already in the table keep their entries)"""
import os, sys, argparse
os.environ.setdefault("TMIX_TUNE_REPS", "5")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U

out = sys.argv[1] if len(sys.argv) > 1 else U._TUNE_FILE
dev = torch.device("cuda:0")
n0 = len(U._TUNE_CACHE)
for kind in ("lora", "custom"):
    args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=1, seeds_per_gpu=8, dtype="bf16")
    tw, _ = bench.build_sampler(args, kind, dev, seed=7)
    for name in ("fusion", "fusion_base", "plain", "start"):
        tw.plan(name)
    tw.plans.clear()
    print(kind, "->", len(U._TUNE_CACHE), "shapes", flush=True)
    del tw
    torch.cuda.empty_cache()
U.save_tune_table(out)
print("wrote", out, n0, "->", len(U._TUNE_CACHE))
