"""per-workgroup phase times of every GEMM / attention launch INSIDE the replayed fusion-step graph (tmix_prof_begin detail mode: exact start / end of a
launch, and the workgroups' mean prologue (entry -> first operands landed), main loop and epilogue), grouped by launch shape and tiling.
  TMIX_TUNE_FILE=<table> python tools/insitu_phases.py [lora|custom] [filter substring]"""
import os, sys, collections, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tweediemix_amd import lib as L
kind = sys.argv[1] if len(sys.argv) > 1 else "lora"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
args = argparse.Namespace(kind=kind, res=1024, tiny=False, no_graphs=False, streams=1, seeds_per_gpu=1, dtype="bf16", lora_mode="merged")
tw, _ = bench.build_sampler(args, kind, dev, seed=0)
lib = L.load()
plan = tw.plan("fusion")
meta = plan.issued_meta()
n = len(meta)
slots = torch.zeros(n + 64, 8, dtype=torch.int64, device=dev)
init = torch.zeros(n + 64, 8, dtype=torch.int64); init[:, 0] = -1; init = init.to(dev)
x = torch.randn(1, 4, tw.h, tw.w).to(dev)
tw.x_state.copy_(x)
t = bench.fusion_timesteps(tw)[3]
for _ in range(2):
    tw._run_step("fusion", L.STEP_FUSION, t, tw.alpha(t), tw.alpha(t - tw.skip))
torch.cuda.synchronize()
L.check(lib.tmix_prof_begin(slots.data_ptr(), n + 64, 1), "prof_begin")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tw._enqueue_step("fusion", L.STEP_FUSION)
assert lib.tmix_prof_end() == n
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0])
reps = 5
for r in range(reps + 1):
    slots.copy_(init); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    if r == 0: continue
    s = slots.cpu().numpy().astype("uint64")
    for i, ((cls, fl, key), v) in enumerate(zip(meta, s)):
        if isinstance(key, tuple): k = key
        elif cls.startswith("gemm"): k = (cls, key.batch, key.M, key.N, key.K, key.epilogue, bool(key.row_stats_out), bool(key.ln_stats), key.tile_cfg)
        else: k = (cls, key.B, key.H, key.W, key.Cin, key.Cout, key.mode, key.tile_cfg)
        wg = max(1, int(v[5]))
        a = acc[k]
        a[0] += 1; a[1] += (int(v[1]) - int(v[0])) * 0.01
        a[2] += int(v[2]) * 0.01 / wg; a[3] += (int(v[3]) - int(v[2])) * 0.01 / wg; a[4] += (int(v[4]) - int(v[3])) * 0.01 / wg
        if i + 1 < n: a[5] += (int(s[i + 1][0]) - int(v[1])) * 0.01
print(f"{'shape':70s} {'n':>4s} {'launch us':>9s} {'prologue':>8s} {'loop':>7s} {'epilogue':>8s} {'gap after':>9s}")
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if flt and flt not in str(k): continue
    c = a[0]
    print(f"{str(k):70s} {c // reps:4d} {a[1] / c:9.2f} {a[2] / c:8.2f} {a[3] / c:7.2f} {a[4] / c:8.2f} {a[5] / c:9.2f}")
