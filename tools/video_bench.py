#!/usr/bin/env python3
"""BASELINE config #5 timing: one denoising step of the I2VGen-XL loop = UNet forward on the CFG pair of 16-frame 768x448
clips (2 x 16 x 56 x 96 latents) + fused CFG / v-prediction / DDIM kernel, synthetic weights, hipGraph replay.
python tools/video_bench.py [--steps 10] [--res_w 768 --res_h 448] [--no-graphs] [--no-autotune]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tweediemix_amd.weights import synthetic_i2vgen_state_dict
from tweediemix_amd import i2vgen as I, ops, video as V

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--res_w", type=int, default=768); ap.add_argument("--res_h", type=int, default=448)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--no-graphs", action="store_true"); ap.add_argument("--no-autotune", action="store_true")
ap.add_argument("--streams", type=int, default=1, help="2: the unconditional and the text clip of the CFG pair run as two launch chains")
a = ap.parse_args()
h, w, Fr = a.res_h // 8, a.res_w // 8, a.frames
sd = {k: v.to(torch.bfloat16) for k, v in synthetic_i2vgen_state_dict(I.FULL).items()}
Wt = I.I2VWeights(I.FULL, sd)
g = torch.Generator().manual_seed(0)
il = torch.randn(2, 4, Fr, h, w, generator=g); emb = torch.randn(2, 1024, generator=g); ehs = torch.randn(2, 77, 1024, generator=g)
fe, ctx, ilf = I.conditioning(Wt, torch.tensor([8.0, 8.0]), il, emb, ehs)
t0 = time.time()
if a.streams == 2:
    plans = [I.I2VPlan(Wt, 1, Fr, h, w, fe[i:i + 1], ctx[i:i + 1], ilf[i:i + 1], autotune=not a.no_autotune) for i in range(2)]
    side = torch.cuda.Stream()
else:
    plans = [I.I2VPlan(Wt, 2, Fr, h, w, fe, ctx, ilf, autotune=not a.no_autotune)]
plan = plans[0]
torch.cuda.synchronize()
build_s = time.time() - t0
x = torch.randn(1, 4, Fr, h, w, generator=g).cuda()
acp = (np.cos((np.arange(1000) / 1000 + 0.008) / 1.008 * np.pi / 2) ** 2).astype(np.float32)
out = torch.empty_like(x)
v = torch.empty(2, 4, Fr, h, w, device="cuda")

def step(t, at, atn):
    for p in plans:
        p.x_in.view(p.clips, Fr, 8, h, w)[:, :, :4] = x.permute(0, 2, 1, 3, 4)          # both CFG rows see the same latent
        p.t_dev.fill_(float(t))
    if len(plans) == 2:
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            plans[1].run()
            ev1 = torch.cuda.Event(); ev1.record(side)
        plans[0].run()
        main.wait_event(ev1)
        for i, p in enumerate(plans):
            v[i] = p.eps.view(Fr, 4, h, w).permute(1, 0, 2, 3)
    else:
        plan.run()
        v.copy_(plan.eps.view(2, Fr, 4, h, w).permute(0, 2, 1, 3, 4))
    ops.vpred_step(x, v, 9.0, at, atn, out=out)
    x.copy_(out)

step(981, acp[981], acp[961]); torch.cuda.synchronize()
graph = None
if not a.no_graphs:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(981, acp[981], acp[961])
run = (lambda: graph.replay()) if graph else (lambda: step(981, acp[981], acp[961]))
for _ in range(a.warmup): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
print(json.dumps({"metric": "I2VGen-XL denoise steps/sec (16 frames %dx%d, CFG pair)" % (a.res_w, a.res_h), "value": 1e3 / ms, "unit": "steps/s",
                  "ms_per_step": ms, "unet_tflop_per_step": sum(p.flops for p in plans) / 1e12, "achieved_tflops": sum(p.flops for p in plans) / ms / 1e9,
                  "launches_per_step": sum(len(p.ops) for p in plans), "streams": a.streams, "plan_build_s": build_s, "hip_graph": graph is not None,
                  "seconds_per_50_step_video": 50 * ms / 1e3, "data": "synthetic", "dtype": "bf16"}))
