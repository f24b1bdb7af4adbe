"""how much of ms_per_step is BETWEEN graph replays?  (a) the bench's loop: 32-byte parameter upload + replay per step; (b) the same graph replayed back to back without the upload;
(c) one replay at a time behind a synchronize."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tweediemix_amd import lib as L
dev = torch.device("cuda:0")
args = argparse.Namespace(kind="lora", res=1024, tiny=False, no_graphs=False, streams=1, seeds_per_gpu=1, dtype="bf16", lora_mode="merged")
tw, _ = bench.build_sampler(args, "lora", dev, seed=0)
ts = bench.fusion_timesteps(tw)
tw.x_state.copy_(torch.randn(1, 4, tw.h, tw.w).to(dev))
def step(i):
    t = ts[i % len(ts)]
    tw._run_step("fusion", L.STEP_FUSION, t, tw.alpha(t), tw.alpha(t - tw.skip))
for i in range(10): step(i)
torch.cuda.synchronize()
g = tw.graphs[("fusion", L.STEP_FUSION)]
def timed(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for rep in range(3):
    a = timed(step); b = timed(lambda i: g.replay())
    c = 0.0
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); c += 1e3 * (time.perf_counter() - t0) / 10
    print(f"bench loop (upload + replay) {a:.3f} ms/step | replay only, back to back {b:.3f} | one replay behind a synchronize {c:.3f}", flush=True)
