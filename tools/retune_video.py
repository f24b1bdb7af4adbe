#!/usr/bin/env python3
"""re-time the tilings of the I2VGen-XL step's launch shapes (BASELINE config #5: 2 clips x 16 frames x 56 x 96) with the CURRENT kernels and candidate list,
in situ (UNetPlan.autotune: the whole forward per candidate, events around every tunable launch): python tools/retune_video.py out.json
Starts from the shipped table with the video plans' keys removed; every other entry is kept."""
import os, sys
os.environ.setdefault("TMIX_TUNE_REPS", "4")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tweediemix_amd import i2vgen as I, unet as U
from tweediemix_amd.weights import synthetic_i2vgen_state_dict

out = sys.argv[1]
h, w, Fr = 56, 96, 16
Wt = I.I2VWeights(I.FULL, synthetic_i2vgen_state_dict(I.FULL, dtype=torch.bfloat16, device="cuda"))
g = torch.Generator().manual_seed(0)
fe, ctx, ilf = I.conditioning(Wt, torch.tensor([8.0, 8.0]), torch.randn(2, 4, Fr, h, w, generator=g), torch.randn(2, 1024, generator=g), torch.randn(2, 77, 1024, generator=g))
old = dict(U._TUNE_CACHE)
changed = {}
for clips, shared in ((2, False), (1, True)):
    plan = I.I2VPlan(Wt, clips, Fr, h, w, fe[:clips], ctx[:clips], ilf[:clips], autotune=False, shared=shared)
    keys = {plan._tune_key(kind, d) for _i, kind, d in plan._tunable}
    for k in keys:
        U._TUNE_CACHE.pop(k, None)
        U._TUNE_CACHE.pop(U.SHARED + k, None)
    plan.autotune()
    for k in keys:
        for kk in ((U.SHARED + k,) if shared else (k, U.SHARED + k)):
            if old.get(kk) != U._TUNE_CACHE.get(kk):
                changed[kk] = (old.get(kk), U._TUNE_CACHE.get(kk))
    if shared:                                  # the two-chain plans never run alone: keep what the table said about these shapes without a sibling
        for k in keys:
            if k in old:
                U._TUNE_CACHE[k] = old[k]
            else:
                U._TUNE_CACHE.pop(k, None)
    print(f"clips={clips} shared={shared}: {len(keys)} shapes", flush=True)
    del plan
    torch.cuda.empty_cache()
for k, (a, b) in sorted(changed.items()):
    print(f"  {k}: {a} -> {b}")
U.save_tune_table(out)
print("wrote", out, len(U._TUNE_CACHE), "entries,", len(changed), "changed")
