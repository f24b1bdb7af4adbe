"""time per UNet call of each call kind of the image sampler (graph replay): fusion / start (B=K+1) and plain (B=2),
with the B=2 call as one chain or split into two B=1 chains."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
sys.argv = ["bench.py"]
args = bench.parse()
dev = torch.device("cuda", 0)
for mr in (1, 2):
    os.environ["TMIX_MIN_ROWS_PER_STREAM"] = str(mr)
    tw, parts = bench.build_sampler(args, dev, seed=0)
    x = torch.randn(1, 4, tw.h, tw.w, device=dev)
    for kind in ("fusion", "start", "plain"):
        for _ in range(3): tw._unet(kind, x, 500)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tw._unet(kind, x, 500)
        torch.cuda.synchronize()
        print(f"min_rows_per_stream={mr} {kind}: B={tw.plan(kind).B} {type(tw.plan(kind)).__name__} {(time.perf_counter() - t0) * 100:.2f} ms/call", flush=True)
    del tw, parts
    torch.cuda.empty_cache()
