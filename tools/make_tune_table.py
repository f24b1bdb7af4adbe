#!/usr/bin/env python3
"""Measure the GEMM / conv tilings of every launch shape the sampler meets at SDXL scale and write the table the
package ships (tweediemix_amd/tuned_gfx950.json).  Run on an MI355X:  python tools/make_tune_table.py [out.json]
Each candidate tiling is timed in situ (UNetPlan.autotune) with TMIX_TUNE_REPS passes; the plans built here cover
Custom-Diffusion and LoRA routing, 1 and 2 launch chains, 1 and 2 co-batched seeds, and every phase's batch."""
import os, sys, argparse
os.environ["TMIX_TUNE_FILE"] = ""            # start from an empty cache
os.environ.setdefault("TMIX_TUNE_REPS", "5")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U, vae as V

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(U.__file__)), "tuned_gfx950.json")
dev = torch.device("cuda:0")
for kind in ("custom", "lora"):
    for streams, seeds in ((1, 1), (2, 1), (2, 2)):
        args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=streams, seeds_per_gpu=seeds)
        tw, _ = bench.build_sampler(args, dev, seed=7)
        for name in ("fusion", "fusion_base", "plain", "start"):          # every phase's plan (B = K+1, K+1, 2, K+1 rows)
            tw.plan(name)
        tw.plans.clear()
        print(kind, streams, seeds, "->", len(U._TUNE_CACHE), "shapes", flush=True)
        del tw
        torch.cuda.empty_cache()
U.save_tune_table(out)
print("wrote", out, len(U._TUNE_CACHE))
