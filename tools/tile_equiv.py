"""which (launch shape, tiling) pairs of the tiny sampler's plans change the UNet output by more than rounding?  Every tunable launch of the single-seed and the
two-seed plans is switched to every candidate tiling, one shape at a time, and eps is compared with the plan's own baseline (all other launches untouched)."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_sampler_gpu import _tiny_setup
from tweediemix_amd import masks as M, sampler as S, unet as U, lib as L
K, n, h, w = 3, 10, 16, 16
bad = 0
for kind in ("lora", "custom"):
    _orc, W, te, ts = _tiny_setup(kind, K, n, h, w)
    cfg = S.make_config(guidance_scale=0.8, n_timesteps=n, t_cond=0.2, t_stop=0.8, resampling_steps=1, jumping_steps=1, resolution_h=h * 8, resolution_w=w * 8)
    for seeds in (1, 2):
        tw = S.Tweediemix(cfg, W, te, ts, lambda x0: None, concept_num=K, lora=(kind == "lora"), n_seeds=seeds)
        tw.init_fusion(2, 8) if kind == "lora" else tw.init_fusion(2)
        for call in ("fusion", "start", "plain", "fusion_base"):
            if call == "fusion_base" and kind != "lora": continue
            p = tw.plan(call)
            x = torch.randn(p.B, 4, h, w, generator=torch.Generator().manual_seed(1)).cuda()
            base = p(x, 501).clone()
            groups = {}
            for i, kd, d in p._tunable:
                groups.setdefault(p._tune_key(kd, d), []).append((kd, d))
            for key, mem in groups.items():
                orig = [d.tile_cfg for _k, d in mem]
                for c in L.TILE_CANDIDATES + (23, 24):
                    if mem[0][0] == "conv" and c in (16, 17, 18, 19, 21, 22, 23, 24): continue
                    for _k, d in mem: d.tile_cfg = c
                    p._link_ln()
                    try:
                        got = p(x, 501)
                        torch.cuda.synchronize()
                        dmax = (got - base).abs().max().item()
                    except Exception as e:
                        dmax = float("nan")
                    if not (dmax <= float(os.environ.get("TEQ_TOL", "2e-2")) * base.abs().max().item()):
                        bad += 1
                        print(f"{kind} seeds={seeds} {call} {key}: tiling {c} (was {orig[0]}) changes eps by {dmax:.4g} (max |eps| {base.abs().max().item():.3g})", flush=True)
                for (_k, d), o in zip(mem, orig): d.tile_cfg = o
                p._link_ln()
print("pairs beyond 2 % of max |eps|:", bad)
