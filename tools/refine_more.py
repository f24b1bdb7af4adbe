#!/usr/bin/env python3
"""extend the shipped tile table in place: re-rank the chains of the co-batched-seed groups (2 and 4 seeds per GPU) under
two-chain load, starting from the entries tools/make_tune_table.py wrote.  python tools/refine_more.py [out.json]"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U

out = sys.argv[1] if len(sys.argv) > 1 else U._TUNE_FILE
dev = torch.device("cuda:0")
for kind, seeds in (("custom", 2), ("custom", 4), ("lora", 2)):
    args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=2, seeds_per_gpu=seeds)
    tw, _ = bench.build_sampler(args, kind, dev, seed=7)
    pl = tw.plan("fusion")
    print(kind, seeds, "refined group step:", pl.refine(verbose=True, top=24), "ms", flush=True)
    tw.plans.clear()
    del tw, pl
    torch.cuda.empty_cache()
U.save_tune_table(out)
print("wrote", out, len(U._TUNE_CACHE))
