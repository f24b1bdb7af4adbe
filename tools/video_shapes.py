"""per-shape time table of one I2VGen-XL forward (eager, events around every GEMM / conv / attention launch)."""
import os, sys, collections, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import i2vgen as I, lib as L
from tweediemix_amd.weights import synthetic_i2vgen_state_dict
lib = L.load()
h, w, Fr = 56, 96, 16
Wt = I.I2VWeights(I.FULL, {k: v.to(torch.bfloat16) for k, v in synthetic_i2vgen_state_dict(I.FULL).items()})
g = torch.Generator().manual_seed(0)
fe, ctx, ilf = I.conditioning(Wt, torch.tensor([8.0, 8.0]), torch.randn(2, 4, Fr, h, w, generator=g), torch.randn(2, 1024, generator=g), torch.randn(2, 77, 1024, generator=g))
plan = I.I2VPlan(Wt, 2, Fr, h, w, fe, ctx, ilf)
st = torch.cuda.current_stream().cuda_stream
plan.run(); torch.cuda.synchronize()
tun = {i: (k, d) for i, k, d in plan._tunable}
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
evs = []
for i, (fn, a) in enumerate(plan.ops):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(*a, st); e1.record()
    evs.append((i, fn, e0, e1))
torch.cuda.synchronize()
tot = 0.0
for i, fn, e0, e1 in evs:
    ms = e0.elapsed_time(e1); tot += ms
    if i in tun:
        k, d = tun[i]
        key = ("gemm", d.batch, d.M, d.N, d.K, d.epilogue, d.tile_cfg) if k == "gemm" else ("conv", d.B, d.H, d.W, d.Cin, d.Cout, d.mode, d.tile_cfg)
        fl = 2 * d.batch * d.M * d.N * d.K if k == "gemm" else 2 * d.B * d.H * d.W * d.Cout * (3 if d.mode == 3 else 9) * d.Cin * (0.25 if d.mode in (1, 4) else 4 if d.mode == 2 else 1)
    else:
        key, fl = (getattr(fn, "__name__", "op"),), 0
    agg[key][0] += 1; agg[key][1] += ms; agg[key][2] += fl
print(f"total {tot:.1f} ms (eager, serialized)")
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {str(k):58s} n={n:4d} {ms:7.2f} ms  avg {1e3 * ms / n:7.1f} us  {fl / ms / 1e9 if fl else 0:6.0f} TF")
