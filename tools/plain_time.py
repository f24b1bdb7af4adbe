"""time per whole step (one graph replay: prologue + UNet + fused update) of each call kind of the image sampler:
fusion / start (B=K+1) and plain (B=2), with the B=2 call as one chain or split into two B=1 chains."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
for mr in (1, 2):
    os.environ["TMIX_MIN_ROWS_PER_STREAM"] = str(mr)
    tw, parts = bench.build_sampler(args, "custom", dev, seed=0)
    from tweediemix_amd import lib as L
    tw.x_state.copy_(torch.randn(1, 4, tw.h, tw.w, device=dev))
    for kind in ("fusion", "start", "plain"):
        mode = {"fusion": L.STEP_FUSION, "start": L.STEP_RESAMPLE, "plain": L.STEP_PLAIN}[kind]
        step = lambda: tw._run_step(kind, mode, 501, tw.alpha(501), tw.alpha(481))
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        print(f"min_rows_per_stream={mr} {kind}: B={tw.plan(kind).B} {type(tw.plan(kind)).__name__} {(time.perf_counter() - t0) * 100:.2f} ms/call", flush=True)
    del tw, parts
    torch.cuda.empty_cache()
