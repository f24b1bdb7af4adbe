"""per-shape table of one call kind of the image sampler (in-situ device-clock stamps inside the captured step, summed per launch shape,
plus the kernel boundaries by class pair):  python tools/step_shapes.py [fusion|start|plain] [bench.py flags, e.g. --dtype fp8 --kind custom]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["TMIX_BENCH_SHAPES"] = "1"
import bench
kind = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "fusion"
sys.argv = ["bench.py"] + [a for a in sys.argv[1:] if a != kind]
args = bench.parse()
dev = torch.device("cuda", 0)
from tweediemix_amd import lib as L
tw, parts = bench.build_sampler(args, "lora" if args.kind == "both" else args.kind, dev, seed=0)
tw.x_state.copy_(torch.randn(*tw.x_state.shape, device=dev))
mode = {"fusion": L.STEP_FUSION, "start": L.STEP_RESAMPLE, "plain": L.STEP_PLAIN}[kind]
for _ in range(3):
    tw._run_step(kind, mode, 501, tw.alpha(501), tw.alpha(481))
torch.cuda.synchronize()
prof = bench.insitu_profile(tw, reps=5, kind=kind, mode=mode)
print(f"{kind}: B={tw.plan(kind).B} replay {prof['replay_ms']:.3f} ms (un-instrumented {prof['uninstrumented_replay_ms']}), boundaries {prof['boundaries_ms']:.3f} ms, "
      + ", ".join(f"{k} {v['sum_launch_ms']:.2f} ms / {v['launches']}" for k, v in prof.items() if isinstance(v, dict)))
