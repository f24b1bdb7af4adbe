"""does the co-batched tiny sampler read memory it never wrote?  Fill the caching allocator's pool with NaN patterns, free it, then run the body of
tests/test_sampler_gpu.py::test_two_seeds_co_batched_equal_independent_runs and report where single and co-batched runs differ."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_sampler_gpu import _tiny_setup
from tweediemix_amd import masks as M, sampler as S, unet as U
if len(sys.argv) > 1 and sys.argv[1] == "poison":
    junk = [torch.full((1 << 28,), float("nan"), device="cuda", dtype=torch.bfloat16) for _ in range(16)]      # 8 GiB of NaN
    del junk
K, n, h, w = 3, 10, 16, 16
for kind in ("custom", "lora"):
    _orc, W, te, ts = _tiny_setup(kind, K, n, h, w)
    cfg = S.make_config(guidance_scale=0.8, n_timesteps=n, t_cond=0.2, t_stop=0.8, resampling_steps=1, jumping_steps=1, resolution_h=h * 8, resolution_w=w * 8)
    imgs = [M.random_rectangle_masks(K, h * 8, w * 8, seed=s) for s in (3, 4)]
    torch.manual_seed(7)
    xT = torch.randn(2, 4, h, w)
    singles = []
    for i in range(2):
        tw = S.Tweediemix(cfg, W, te, ts, lambda x0, i=i: M.build_masks(imgs[i], h, w), concept_num=K, lora=(kind == "lora"))
        singles.append(tw.run_fusion(xT[i:i + 1].clone()).cpu())
        if i == 0: t1 = {k: U.used_tilings(p) for k, p in tw.plans.items()}
    calls = {"n": 0}
    def provider(x0):
        i = calls["n"] % 2; calls["n"] += 1
        return M.build_masks(imgs[i], h, w)
    tw2 = S.Tweediemix(cfg, W, te, ts, provider, concept_num=K, lora=(kind == "lora"), n_seeds=2)
    both = tw2.run_fusion(xT.clone()).cpu()
    for i in range(2):
        d = (both[i] - singles[i][0]).abs()
        print(kind, "seed", i, "max diff", d.max().item(), "nan", torch.isnan(both[i]).any().item(), torch.isnan(singles[i]).any().item(), "mismatched", int((d > 1e-3).sum()), flush=True)
    print("  single tilings", t1, "\n  co-batched   ", {k: U.used_tilings(p) for k, p in tw2.plans.items()}, flush=True)
