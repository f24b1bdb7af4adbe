"""Whole-trajectory parity at SDXL size: the product sampler against the oracle's sampler driving the fp32 UNet oracle.

TEST INFRASTRUCTURE (imports oracle/): used by tests/test_trajectory_fullsize_gpu.py at 512^2, n = 20 (BASELINE config 1's
schedule: 27 UNet calls at B = 4 + 18 at B = 2) and run by hand at 1024^2, n = 50 (configs 2-3: 75 calls)

    python tests/trajectory_parity.py --res 1024 --n 50 [--fp8] [--kind lora|custom] [--masks partition|overlap] --out profiles/<name>.json

Reference loop: /root/reference/fusion_generation/fusion_sampling.py:490-494 `for t in timesteps: x = denoise_step(x, t)`; the fusion
branch :376-385, the DDIM move :430, the blend :466-469.

Masks (round 6).  The blend of :466-469 is NOT normalised: where two foreground rectangles overlap the weights sum to 2 and that region of
the latent doubles at every fusion step (max|x| ~ 4e11 at the end of a 50-step run with tweediemix_amd.masks.random_rectangle_masks' 4 %
overlap) -- eps is then invisible beside x, every per-step error decays by 2x per step, and a final-latent tolerance says nothing about the
fusion window.  The DEFAULT case here therefore draws masks that PARTITION the image (masks.partition_rectangle_masks: fg_1 + fg_2 + bg == 1,
what the side-car's overlap rule, text_segment/run_expand.py:62-87, leaves of two intersecting rectangles); the overlapping set stays as a
second, labelled case (`mask_kind="overlap"`): the quirk is the reference's and the product must follow it there too.

What is measured:
  * free-running: Tweediemix.run_fusion(x_T) vs TweedieOracle.denoise_step iterated from the same x_T --
      - rel L2 of the FINAL latent and of every intermediate latent,
      - rel L2 of eps of EVERY UNet call along the two free-running trajectories (call i of the product against call i of the oracle:
        the same (B, t) schedule is asserted), reported per phase (start / plain+jumping / fusion window);
  * teacher-forced: every scheduler step of the product started from the ORACLE's latent of that step --
      - rel L2 of the step's output latent (as in round 5),
      - the same error normalised by the step's UPDATE, || x' - (sqrt(a') / sqrt(a)) x ||, i.e. by the part of x' that eps produced
        (x' = (sqrt(a')/sqrt(a)) x - sqrt(a') sqrt(1-a)/sqrt(a) e_cfg + sqrt(1-a') e_uncond; the last step returns x0: a' := 1),
      - rel L2 of eps of the step's first UNet call (identical input latent on both sides: the per-call error at in-trajectory latents);
  * max|x| along the oracle's trajectory (bounded with partition masks; the explosion with overlapping ones is reported, not hidden).
Both samplers see the same bf16-rounded prompt rows, the same fp32 weights (the product rounds them to bf16 / e4m3 itself) and the same masks.
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


def sdxl_bundle(sd, kind, K=3):
    """(concept state dicts, product UNetWeights, oracle UNetOracle) for one concept kind on SDXL shapes: what every full-size test builds;
    tests/conftest.py keeps one per kind for the session."""
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U, weights as Wt
    con = Wt.synthetic_concepts(U.SDXL, kind, K, device="cuda")
    W = U.UNetWeights(U.SDXL, sd, "cuda", (kind, con))
    return con, W, UO.UNetOracle(UO.SDXL, sd, oracle_concepts(kind, con))


def _tap_eps(tw, store):
    """record eps of every UNet call the product sampler makes (the plan's output buffer right behind the step that ran it)."""
    orig = tw._run_step

    def tapped(kind, mode, t, at, at_next, is_last=False):
        orig(kind, mode, t, at, at_next, is_last)
        store.append(tw.plan(kind).eps.clone())
    tw._run_step = tapped
    return orig


def _phase(o, i):
    """phase of oracle request i: 'start' (the B = K+1 start call and its resampling pairs), 'fusion' (B = K+1 inside t <= t_cond_cur),
    'plain' (the B = 2 CFG steps and the jumping look-ahead)."""
    B, t, _rows, kind, _routed = o.requests[i]
    if kind == "start" or (kind == "plain" and t == o.start_t - o.sch.skip and i < 2 * o.resampling_steps + 1):
        return "start"
    return "fusion" if kind == "fusion" else "plain"


@torch.no_grad()
def trajectory_parity(sd, kind="lora", res=512, n=20, fp8=False, K=3, resampling_steps=10, jumping_steps=5, teacher_forced=True, seed=11,
                      mask_kind="partition", bundle=None):
    """sd: fp32 SDXL state dict on the device.  Returns a dict of the measured numbers (nothing asserted here)."""
    from oracle import tweedie_oracle as TO
    from tweediemix_amd import masks as M, sampler as S, unet as U
    cfg = U.SDXL
    h = w = res // 8
    con, W, orc = bundle if bundle is not None else sdxl_bundle(sd, kind, K)
    g = torch.Generator().manual_seed(seed)
    te = (torch.randn(K + 2, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K + 2, cfg.pooled_dim, generator=g))
    ts = (torch.randn(K, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K, cfg.pooled_dim, generator=g))
    xT = torch.randn(1, 4, h, w, generator=g)
    imgs = M.synthetic_masks(mask_kind, K, res, res, seed=3)
    m_dev = M.build_masks(imgs, h, w)
    weight_sum = m_dev.sum(0)
    lora = kind == "lora"
    conf = S.make_config(guidance_scale=0.8, n_timesteps=n, t_cond=0.2, t_stop=0.8, resampling_steps=resampling_steps, jumping_steps=jumping_steps,
                         resolution_h=res, resolution_w=res)
    tw = S.Tweediemix(conf, W, te, ts, lambda x0: M.build_masks(imgs, h, w), concept_num=K, lora=lora, use_graphs=True, fp8=fp8)
    eps_free = []
    _tap_eps(tw, eps_free)
    out = tw.run_fusion(xT.clone()).cpu()

    time_ids = torch.tensor([[res, res, 0, 0, res, res]], dtype=torch.float32, device="cuda")
    o = TO.TweedieOracle(K, n, g=0.8, t_cond=0.2, t_stop=0.8 if lora else None, resampling_steps=resampling_steps, jumping_steps=jumping_steps,
                         mask_fn=lambda: TO.build_masks(imgs, h, w))
    src = {"e": (te[0].cuda(), te[1].cuda()), "s": (ts[0].cuda(), ts[1].cuda())}
    eps_orc = []

    def unet_fn(x, t, rows, kindname, routed):
        ehs = torch.stack([src[k][0][r] for k, r in rows])
        pooled = torch.stack([src[k][1][r] for k, r in rows])
        e = orc.forward(torch.from_numpy(np.ascontiguousarray(x)).cuda(), t, ehs, pooled, time_ids.repeat(len(rows), 1), routed=routed)
        eps_orc.append(e)
        return e.cpu().numpy()

    x = xT.numpy()
    traj = [x]
    first_call_of_step = []
    for t in o.sch.timesteps:
        first_call_of_step.append(len(o.requests))
        x = o.denoise_step(x, int(t), unet_fn)
        traj.append(x)
    ref = torch.from_numpy(x)
    same_calls = [(b, t) for _k, b, t in tw.unet_calls] == [(b, t) for b, t, *_ in o.requests]
    res_d = {"kind": kind, "dtype": "fp8" if fp8 else "bf16", "resolution": res, "n_timesteps": n, "K": K, "masks": mask_kind,
             "mask_weight_sum_min_max": [float(weight_sum.min()), float(weight_sum.max())],
             "mask_overlap_fraction": float((weight_sum > 1).float().mean()),
             "unet_calls": len(o.requests), "calls_BK1": sum(1 for r in o.requests if r[0] == K + 1), "calls_B2": sum(1 for r in o.requests if r[0] == 2),
             "same_unet_call_schedule": bool(same_calls), "finite": bool(torch.isfinite(out).all()),
             "max_abs_latent_oracle_per_step": [float(np.abs(v).max()) for v in traj],
             "max_abs_latent_oracle": float(max(np.abs(v).max() for v in traj)), "max_abs_latent_product_final": float(out.abs().max()),
             "free_running_final_rel_l2": rel(out, ref)}
    # ---- eps of every UNet call along the two free-running trajectories
    if same_calls and len(eps_free) == len(eps_orc):
        per_call = []
        for i, (ep, eo) in enumerate(zip(eps_free, eps_orc)):
            B, t = o.requests[i][0], o.requests[i][1]
            per_call.append((i, B, int(t), _phase(o, i), rel(ep, eo)))
        res_d["free_running_eps_per_call"] = per_call
        for ph in ("start", "plain", "fusion"):
            v = [r for *_x, p, r in per_call if p == ph]
            if v:
                res_d[f"free_running_eps_worst_{ph}"] = max(v)
                res_d[f"free_running_eps_median_{ph}"] = float(np.median(v))
    del eps_free[:]
    if teacher_forced:
        sch = o.sch
        per, per_upd, per_eps = [], [], []
        for k, t in enumerate(sch.timesteps):
            t = int(t)
            xk = torch.from_numpy(traj[k])
            n_before = len(eps_free)
            y = tw.denoise_step(xk.cuda(), t).cpu()
            want = torch.from_numpy(traj[k + 1])
            at = float(sch.alpha(t))
            a_next = 1.0 if t == 1 else float(sch.alpha(t - sch.skip))
            upd = (want.double() - (a_next / at) ** 0.5 * xk.double()).norm()
            err = (y.double() - want.double()).norm()
            per.append((t, float(err / want.double().norm())))
            per_upd.append((t, float(err / upd), float(upd / want.double().norm())))
            per_eps.append((t, rel(eps_free[n_before], eps_orc[first_call_of_step[k]])))      # same input latent on both sides
            del eps_free[:]
        in_win = [t for t in sch.timesteps if o._in_fusion(int(t))]
        res_d["teacher_forced_per_step"] = per
        res_d["teacher_forced_per_step_update_normalised"] = per_upd          # (t, error / update, update / ||x'||)
        res_d["teacher_forced_eps_first_call_per_step"] = per_eps
        res_d["teacher_forced_start_step"] = per[0][1]
        res_d["teacher_forced_worst_other_step"] = max(r for _t, r in per[1:])
        res_d["teacher_forced_start_step_update_normalised"] = per_upd[0][1]
        res_d["teacher_forced_worst_other_step_update_normalised"] = max(r for _t, r, _u in per_upd[1:])
        res_d["teacher_forced_worst_eps_first_call"] = max(r for _t, r in per_eps)
        win = set(int(v) for v in in_win)
        fw = [(t, r, ru, uf, re) for (t, r), (_t2, ru, uf), (_t3, re) in zip(per, per_upd, per_eps) if t in win]
        if fw:
            # (the last scheduler step, t == 1, returns x0 with alpha_t ~ 1: eps weighs 3e-3 of it by construction -- kept out of the "visible" floor, not out of the maxima)
            res_d["fusion_window"] = {"steps": len(fw), "teacher_forced_min": min(r for t, r, *_x in fw if t != 1), "teacher_forced_max": max(r for _t, r, *_x in fw),
                                      "update_normalised_min": min(u for _t, _r, u, _f, _e in fw), "update_normalised_max": max(u for _t, _r, u, _f, _e in fw),
                                      "update_fraction_min": min(f for _t, _r, _u, f, _e in fw), "eps_first_call_max": max(e for *_x, e in fw)}
    return res_d


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--n", type=int, default=50)
    ap.add_argument("--kind", default="lora")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--masks", default="partition", choices=["partition", "overlap"])
    ap.add_argument("--no-teacher", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args(argv)
    from tweediemix_amd import unet as U, weights as Wt
    sd = Wt.synthetic_state_dict(U.SDXL, seed=1234, device="cuda", dtype=torch.float32)
    r = trajectory_parity(sd, a.kind, a.res, a.n, a.fp8, teacher_forced=not a.no_teacher, mask_kind=a.masks)
    print(json.dumps(r))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
