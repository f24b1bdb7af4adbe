#!/usr/bin/env python3
"""re-time the tilings of the single-seed, one-chain plans only (the headline step and the B = 2 calls) from an empty table:
python tools/quick_tune.py out.json [kinds]   -- minutes instead of the half hour of make_tune_table.py; for kernel A/B work"""
import os, sys, argparse
os.environ["TMIX_TUNE_FILE"] = ""
os.environ.setdefault("TMIX_TUNE_REPS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U
out = sys.argv[1]
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lora"]
for kind in kinds:
    args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=1, seeds_per_gpu=1, dtype="bf16")
    tw, _ = bench.build_sampler(args, kind, torch.device("cuda:0"), seed=7)
    for name in ("fusion", "plain"):
        tw.plan(name)
    print(kind, "->", len(U._TUNE_CACHE), "shapes", flush=True)
    del tw
    torch.cuda.empty_cache()
U.save_tune_table(out)
import collections
print(sorted(collections.Counter(v for k, v in U._TUNE_CACHE.items() if not k.startswith(U.SHARED)).items()))
