"""debug helper: run a UNet plan op by op with a sync after each, printing the op before it runs."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import unet as U, weights as Wt, lib as L

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
cfg = U.SDXL
t0 = time.time()
sd = Wt.synthetic_state_dict(cfg, seed=1234, device=dev, dtype=torch.bfloat16)
con = Wt.synthetic_concepts(cfg, "custom", 3, device=dev)
W = U.UNetWeights(cfg, sd, dev, ("custom", con))
print("weights", time.time() - t0, W.nbytes() / 1e9, "GB", flush=True)
g = torch.Generator().manual_seed(0)
ehs = torch.randn(4, 77, 2048, generator=g)
kv = U.KVCache(W, ehs, [0, 1, 2, 3])
print("kv ok", flush=True)
h = w = res // 8
plan = U.UNetPlan(W, 4, h, w, kv, torch.randn(4, 1280, generator=g), torch.tensor([[res, res, 0, 0, res, res]] * 4.0 if False else [[float(res), float(res), 0, 0, float(res), float(res)]] * 4))
print("plan built: ops", len(plan.ops), "flops", plan.flops / 1e12, "TF; arena", plan.arena.total / 1e9, "GB", flush=True)
plan.latent.normal_()
plan.t_dev.fill_(781.0)
st = torch.cuda.current_stream().cuda_stream
for i, (fn, a) in enumerate(plan.ops):
    print(i, fn.__name__, flush=True)
    rc = fn(*a, st)
    assert rc == 0, L.load().tmix_last_error_string()
    torch.cuda.synchronize()
print("all ok", plan.eps.float().std().item())
