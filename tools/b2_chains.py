"""the B = 2 (plain CFG pair) call of a single-image trajectory: ONE dependent chain at B = 2 against TWO chains of one row each on two streams
(PlanGroup; sampler.min_rows_per_stream = 1).  Prints ms per captured step for both.   python tools/b2_chains.py [kind]"""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tweediemix_amd import lib as L
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
mode = {"plain": L.STEP_PLAIN, "start": L.STEP_RESAMPLE, "fusion": L.STEP_FUSION, "fusion_base": L.STEP_FUSION}[kind]


def run(streams):
    args = argparse.Namespace(kind="lora", res=1024, tiny=False, no_graphs=False, streams=streams, seeds_per_gpu=1, dtype="bf16", lora_mode="merged")
    tw, _ = bench.build_sampler(args, "lora", dev, seed=0)
    tw.min_rows_per_stream = 1
    ts = bench.fusion_timesteps(tw)
    tw.x_state.copy_(torch.randn(1, 4, tw.h, tw.w).to(dev))

    def step(i):
        t = ts[i % len(ts)]
        tw._run_step(kind, mode, t, tw.alpha(t), tw.alpha(t - tw.skip))
    for i in range(8): step(i)
    torch.cuda.synchronize()
    p = tw.plan(kind)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(30): step(i)
        torch.cuda.synchronize(); best = min(best, 1e3 * (time.perf_counter() - t0) / 30)
    print(f"{kind}: B = {p.B}, {type(p).__name__}, streams {streams}: {best:.3f} ms per step", flush=True)
    x = tw.x_state.clone()
    del tw
    torch.cuda.empty_cache()
    return best


a = run(1)
b = run(2)
print(f"one chain {a:.3f} ms, two chains {b:.3f} ms")
