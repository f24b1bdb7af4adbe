#!/usr/bin/env python3
"""time the launch shapes the shipped tile table does not hold yet (e.g. after a tune-key change) and write the extended table:
python tools/extend_table_missing.py [out.json]   -- starts from the shipped table, so only missing keys are measured (in situ, TMIX_TUNE_REPS passes)"""
import os, sys, argparse
os.environ.setdefault("TMIX_TUNE_REPS", "5")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U

out = sys.argv[1] if len(sys.argv) > 1 else U._TUNE_FILE
dev = torch.device("cuda:0")
n0 = len(U._TUNE_CACHE)
before = set(U._TUNE_CACHE)
for kind in ("custom", "lora"):
    for streams, seeds in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
        args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=streams, seeds_per_gpu=seeds, dtype="bf16")
        tw, _ = bench.build_sampler(args, kind, dev, seed=7)
        for name in ("fusion", "fusion_base", "plain", "start"):
            tw.plan(name)
        tw.plans.clear()
        print(kind, streams, seeds, "->", len(U._TUNE_CACHE), "shapes", flush=True)
        del tw
        torch.cuda.empty_cache()
# keep what the shipped table said (or did not say) about shapes it already held: only shapes that were missing get entries
for k in list(U._TUNE_CACHE):
    if k not in before and k.startswith(U.SHARED) and k[len(U.SHARED):] in before:
        del U._TUNE_CACHE[k]
print("new entries:", sorted(k for k in U._TUNE_CACHE if k not in before))
U.save_tune_table(out)
print("wrote", out, n0, "->", len(U._TUNE_CACHE))
