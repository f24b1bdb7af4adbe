#!/usr/bin/env python3
"""graph-timed second tuning pass over the video step's heaviest launch shapes (unet.refine_group: every candidate tried in place, the WHOLE captured step timed):
python tools/refine_video.py out.json [1|2 chains] [top]   -- starts from the shipped table, writes the refined one"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tweediemix_amd import i2vgen as I, unet as U
from tweediemix_amd.weights import synthetic_i2vgen_state_dict
out = sys.argv[1]
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
h, w, Fr = 56, 96, 16
Wt = I.I2VWeights(I.FULL, synthetic_i2vgen_state_dict(I.FULL, dtype=torch.bfloat16, device="cuda"))
g = torch.Generator().manual_seed(0)
fe, ctx, ilf = I.conditioning(Wt, torch.tensor([8.0, 8.0]), torch.randn(2, 4, Fr, h, w, generator=g), torch.randn(2, 1024, generator=g), torch.randn(2, 77, 1024, generator=g))
plan = (I.I2VPlanGroup if chains == 2 else I.I2VPlan)(Wt, 2, Fr, h, w, fe, ctx, ilf)
before = dict(U._TUNE_CACHE)
t = U.refine_group(plan, top=top, reps=5, verbose=True)
U.save_tune_table(out)
ch = {k: (before.get(k), v) for k, v in U._TUNE_CACHE.items() if before.get(k) != v}
print("step", t, "ms; changed:", ch)
