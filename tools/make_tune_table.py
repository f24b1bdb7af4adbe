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

REFINE = "--refine" in sys.argv            # second pass for chains that share the chip (two-stream groups): ~15 GPU-minutes
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
out = argv[0] if argv else os.path.join(os.path.dirname(os.path.abspath(U.__file__)), "tuned_gfx950.json")
dev = torch.device("cuda:0")
for kind in ("custom", "lora"):
    for streams, seeds in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
        args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=streams, seeds_per_gpu=seeds, dtype="bf16")
        tw, _ = bench.build_sampler(args, kind, dev, seed=7)
        for name in ("fusion", "fusion_base", "plain", "start"):          # every phase's plan (B = K+1, K+1, 2, K+1 rows)
            pl = tw.plan(name)
            if REFINE and name in ("fusion", "plain") and hasattr(pl, "refine") and seeds == 1:   # chains that share the chip: re-rank under two-chain load
                print("refined group step:", name, pl.refine(verbose=True, top=40 if name == "fusion" else 24), "ms", flush=True)
        tw.plans.clear()
        print(kind, streams, seeds, "->", len(U._TUNE_CACHE), "shapes", flush=True)
        del tw
        torch.cuda.empty_cache()
U.save_tune_table(out)
# the video UNet (BASELINE config #5): CFG pair of 16-frame clips at 768x448 and 512x512
from tweediemix_amd import i2vgen as I
from tweediemix_amd.weights import synthetic_i2vgen_state_dict
Wv = I.I2VWeights(I.FULL, {k: v.to(torch.bfloat16) for k, v in synthetic_i2vgen_state_dict(I.FULL).items()})
g = torch.Generator().manual_seed(0)
for hh, ww in ((56, 96), (64, 64)):
    il = torch.randn(2, 4, 16, hh, ww, generator=g)
    fe, ctx, ilf = I.conditioning(Wv, torch.tensor([8.0, 8.0]), il, torch.randn(2, 1024, generator=g), torch.randn(2, 77, 1024, generator=g))
    I.I2VPlan(Wv, 2, 16, hh, ww, fe, ctx, ilf)
    grp = I.I2VPlanGroup(Wv, 2, 16, hh, ww, fe, ctx, ilf)              # run_video's default: one chain per clip
    if REFINE:
        print("refined video step:", grp.refine(verbose=True, reps=5, top=30), "ms", flush=True)
    del grp
    print("video", hh, ww, "->", len(U._TUNE_CACHE), "shapes", flush=True)
    torch.cuda.empty_cache()
U.save_tune_table(out)
print("wrote", out, len(U._TUNE_CACHE))
