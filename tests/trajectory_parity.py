"""Whole-trajectory parity at SDXL size: the product sampler against the oracle's sampler driving the fp32 UNet oracle.

TEST INFRASTRUCTURE (imports oracle/): used by tests/test_trajectory_fullsize_gpu.py at 512^2, n = 20 (BASELINE config 1's
schedule: 27 UNet calls at B = 4 + 18 at B = 2) and run by hand at 1024^2, n = 50 (configs 2-3: 75 calls)

    python tests/trajectory_parity.py --res 1024 --n 50 [--fp8] [--kind lora|custom] --out profiles/<name>.json

What is compared (/root/reference/fusion_generation/fusion_sampling.py:490-494 `for t in timesteps: x = denoise_step(x, t)`):
  * free-running: Tweediemix.run_fusion(x_T) vs TweedieOracle.denoise_step iterated from the same x_T -- rel L2 of the FINAL latent,
    and of every intermediate latent (the oracle's trajectory is kept);
  * teacher-forced: every scheduler step of the product started from the ORACLE's latent of that step -- rel L2 per step (the start
    step chains 1 + 2 * resampling_steps UNet calls).
Both samplers see the same bf16-rounded prompt rows, the same fp32 weights (the product rounds them to bf16 / e4m3 itself) and the
same rectangle masks.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def oracle_concepts(kind, con):
    from oracle import unet_oracle as UO
    if kind == "custom":
        return UO.Concepts("custom", kv={tb: [(c[f"{tb}.attn2.to_k.weight"], c[f"{tb}.attn2.to_v.weight"]) for c in con]
                                         for tb in UO.attention_prefixes(UO.SDXL)})
    lo = {}
    for tb in UO.attention_prefixes(UO.SDXL):
        for a in ("attn1", "attn2"):
            lo[f"{tb}.{a}"] = [{nm: (c[f"{tb}.{a}.processor.to_{nm}_lora.down.weight"], c[f"{tb}.{a}.processor.to_{nm}_lora.up.weight"])
                                for nm in ("q", "k", "v", "out")} for c in con]
    return UO.Concepts("lora", lora=lo)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@torch.no_grad()
def trajectory_parity(sd, kind="lora", res=512, n=20, fp8=False, K=3, resampling_steps=10, jumping_steps=5, teacher_forced=True, seed=11):
    """sd: fp32 SDXL state dict on the device.  Returns a dict of the measured numbers (nothing asserted here)."""
    from oracle import tweedie_oracle as TO, unet_oracle as UO
    from tweediemix_amd import masks as M, sampler as S, unet as U, weights as Wt
    cfg = U.SDXL
    h = w = res // 8
    con = Wt.synthetic_concepts(cfg, kind, K, device="cuda")
    W = U.UNetWeights(cfg, sd, "cuda", (kind, con))
    g = torch.Generator().manual_seed(seed)
    te = (torch.randn(K + 2, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K + 2, cfg.pooled_dim, generator=g))
    ts = (torch.randn(K, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K, cfg.pooled_dim, generator=g))
    xT = torch.randn(1, 4, h, w, generator=g)
    imgs = M.random_rectangle_masks(K, res, res, seed=3)
    lora = kind == "lora"
    conf = S.make_config(guidance_scale=0.8, n_timesteps=n, t_cond=0.2, t_stop=0.8, resampling_steps=resampling_steps, jumping_steps=jumping_steps,
                         resolution_h=res, resolution_w=res)
    tw = S.Tweediemix(conf, W, te, ts, lambda x0: M.build_masks(imgs, h, w), concept_num=K, lora=lora, use_graphs=True, fp8=fp8)
    out = tw.run_fusion(xT.clone()).cpu()

    orc = UO.UNetOracle(UO.SDXL, sd, oracle_concepts(kind, con))
    time_ids = torch.tensor([[res, res, 0, 0, res, res]], dtype=torch.float32, device="cuda")
    o = TO.TweedieOracle(K, n, g=0.8, t_cond=0.2, t_stop=0.8 if lora else None, resampling_steps=resampling_steps, jumping_steps=jumping_steps,
                         mask_fn=lambda: TO.build_masks(imgs, h, w))
    src = {"e": (te[0].cuda(), te[1].cuda()), "s": (ts[0].cuda(), ts[1].cuda())}

    def unet_fn(x, t, rows, kindname, routed):
        ehs = torch.stack([src[k][0][r] for k, r in rows])
        pooled = torch.stack([src[k][1][r] for k, r in rows])
        return orc.forward(torch.from_numpy(np.ascontiguousarray(x)).cuda(), t, ehs, pooled, time_ids.repeat(len(rows), 1), routed=routed).cpu().numpy()

    x = xT.numpy()
    traj = [x]
    for t in o.sch.timesteps:
        x = o.denoise_step(x, int(t), unet_fn)
        traj.append(x)
    ref = torch.from_numpy(x)
    same_calls = [(b, t) for _k, b, t in tw.unet_calls] == [(b, t) for b, t, *_ in o.requests]
    res_d = {"kind": kind, "dtype": "fp8" if fp8 else "bf16", "resolution": res, "n_timesteps": n, "K": K,
             "unet_calls": len(o.requests), "calls_BK1": sum(1 for r in o.requests if r[0] == K + 1), "calls_B2": sum(1 for r in o.requests if r[0] == 2),
             "same_unet_call_schedule": bool(same_calls), "finite": bool(torch.isfinite(out).all()),
             "free_running_final_rel_l2": rel(out, ref)}
    if teacher_forced:
        per = []
        for k, t in enumerate(o.sch.timesteps):
            y = tw.denoise_step(torch.from_numpy(traj[k]).cuda(), int(t)).cpu()
            per.append((int(t), rel(y, torch.from_numpy(traj[k + 1]))))
        res_d["teacher_forced_per_step"] = per
        res_d["teacher_forced_start_step"] = per[0][1]
        res_d["teacher_forced_worst_other_step"] = max(r for _t, r in per[1:])
    return res_d


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--kind", default="lora")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--no-teacher", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args(argv)
    from tweediemix_amd import unet as U, weights as Wt
    sd = Wt.synthetic_state_dict(U.SDXL, seed=1234, device="cuda", dtype=torch.float32)
    r = trajectory_parity(sd, a.kind, a.res, a.n, a.fp8, teacher_forced=not a.no_teacher)
    print(json.dumps(r))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
