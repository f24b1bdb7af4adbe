"""per-shape table of one I2VGen-XL step (BASELINE config #5, 2 clips x 16 frames x 56 x 96) from in-situ device-clock stamps inside the captured graph:
python tools/video_step_shapes.py [1|2]   (launch chains: one plan over both clips, or the two clips as two chains = bench.py's video leg)"""
import os, sys, time, collections, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import i2vgen as I, lib as L
from tweediemix_amd.weights import synthetic_i2vgen_state_dict
lib = L.load()
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h, w, Fr = 56, 96, 16
Wt = I.I2VWeights(I.FULL, synthetic_i2vgen_state_dict(I.FULL, dtype=torch.bfloat16, device="cuda"))
g = torch.Generator().manual_seed(0)
fe, ctx, ilf = I.conditioning(Wt, torch.tensor([8.0, 8.0]), torch.randn(2, 4, Fr, h, w, generator=g), torch.randn(2, 1024, generator=g), torch.randn(2, 77, 1024, generator=g))
plan = (I.I2VPlanGroup if streams == 2 else I.I2VPlan)(Wt, 2, Fr, h, w, fe, ctx, ilf)
plans = plan.plans[1:] + plan.plans[:1] if streams == 2 else [plan]
meta = [m for p in plans for m in p.issued_meta()]
n = len(meta)
plan.run(); torch.cuda.synchronize()
cap = n + 4096                                   # spare slots, so that a launch without a meta entry shows up as used != n
slots = torch.zeros(cap, 8, dtype=torch.int64, device="cuda")
init = torch.zeros(cap, 8, dtype=torch.int64); init[:, 0] = -1; init = init.cuda()
L.check(lib.tmix_prof_begin(slots.data_ptr(), cap, 0), "prof")
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    plan.run()
used = lib.tmix_prof_end()
assert used == n, (used, n)
runs = []
for _ in range(6):
    slots.copy_(init)
    torch.cuda.synchronize(); t0 = time.perf_counter()          # (host clock: see unet.refine_group on timing events and two-stream graphs)
    gr.replay(); torch.cuda.synchronize()
    runs.append((1e3 * (time.perf_counter() - t0), slots.cpu().numpy().astype("uint64")))
runs = sorted(runs[1:], key=lambda r: r[0])
ms, sl = runs[len(runs) // 2]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
cls_t = collections.defaultdict(float)
for (cls, fl, key), s in zip(meta, sl):
    d = (int(s[1]) - int(s[0])) * 1e-5
    k = key if isinstance(key, tuple) else ((cls, key.batch, key.M, key.N, key.K, key.epilogue, key.tile_cfg) if cls.startswith("gemm")
                                             else (cls, key.B, key.H, key.W, key.Cin, key.Cout, key.mode, key.tile_cfg))
    agg[k][0] += 1; agg[k][1] += d; agg[k][2] += fl
    cls_t[cls] += d
print(f"chains={streams} replay {ms:.2f} ms, {len(plan.ops)} launches, {n} instrumented; per class (sum of launch durations): " + ", ".join(f"{k} {v:.2f} ms" for k, v in cls_t.items()))
for k, (cnt, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {str(k):64s} n={cnt:4d} total={t:7.3f}ms avg={1e3 * t / cnt:7.1f}us {fl / t / 1e9 if t else 0:6.0f}TF")
names = collections.Counter(getattr(fn, "__name__", "op") for fn, _a in plan.ops)
print("launches by entry point:", dict(names))
