#!/usr/bin/env python3
"""second tuning pass over a tile table: for the heaviest launch shapes of the single-seed plans (the headline fusion step and
the B = 2 calls; with --cobatch N also the N-seed plans of the images/s path) every candidate tiling is tried in place and the
whole plan is timed as a captured hipGraph (median of 9 replays) -- the eager, event-timed ranking of make_tune_table.py is noisy
at the +-1 % level.   python tools/refine_table.py in.json out.json [--cobatch 4] [--top 40]"""
import os, sys, argparse
ap = argparse.ArgumentParser()
ap.add_argument("src"); ap.add_argument("dst")
ap.add_argument("--cobatch", type=int, default=0); ap.add_argument("--top", type=int, default=40); ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--cands", default="")       # e.g. 2,12,20,21: only these tilings are tried (default: every candidate)
ap.add_argument("--kinds", default="lora,custom")
ap.add_argument("--only-cobatch", action="store_true")       # skip the single-seed plans (refine the N-seed entries only)
ap.add_argument("--only-kind", default="")                   # "conv" / "gemm": re-rank that class only
a = ap.parse_args()
os.environ["TMIX_TUNE_FILE"] = a.src
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import unet as U
dev = torch.device("cuda:0")
for seeds in (([] if a.only_cobatch else [1]) + ([a.cobatch] if a.cobatch else [])):
    for kind in a.kinds.split(","):
        args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=True, streams=1, seeds_per_gpu=seeds, dtype="bf16")
        tw, _ = bench.build_sampler(args, kind, dev, seed=7)
        for name, top in (("fusion", a.top), ("plain", a.top // 2)):
            ms = tw.plan(name).refine(top=top if seeds == 1 else top // 2, reps=a.reps if seeds == 1 else 5, verbose=True,
                                       cands=[int(c) for c in a.cands.split(',')] if a.cands else None, only_kind=a.only_kind or None)
            print(f"refined {kind} seeds={seeds} {name}: {ms:.3f} ms per UNet call", flush=True)
        del tw
        torch.cuda.empty_cache()
        U.save_tune_table(a.dst)
print("wrote", a.dst, len(U._TUNE_CACHE))
