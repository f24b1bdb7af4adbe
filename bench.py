#!/usr/bin/env python3
"""bench.py -- denoise steps/sec of the Tweedie-mix fusion phase on MI355X.

A "step" = one fusion-phase iteration of `sample_loop` (fusion_sampling.py:493-494 with
t <= t_cond_cur): ONE SDXL UNet forward at batch K+1 (uncond + K concept rows, per-concept
weights routed) + the fused CFG/Tweedie/blend/DDIM kernel, latent resident in HBM.
Workload (BASELINE.json configs[1]): SDXL-base shapes, 1024x1024 (latent 128x128), K=3 concepts
(2 foreground + background), Custom-Diffusion K/V deltas (`--kind lora` = configs[2]), synthetic
random-init weights / prompt embeddings / rectangle masks (no checkpoints exist offline).

python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kind", default="custom", choices=["custom", "lora"])
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--tiny", action="store_true", help="tiny UNet (debug only; not a valid bench line)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--trajectory", action="store_true",
                    help="extra line: time whole 50-step trajectories (start/resampling, plain, jumping, fusion: 75 UNet calls)")
    ap.add_argument("--streams", type=int, default=2, help="independent launch chains per UNet call (rows split over HIP streams)")
    ap.add_argument("--seeds-per-gpu", type=int, default=1,
                    help="independent trajectories co-batched into every UNet launch (1 = the reference's one image per process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=64)
    ap.add_argument("--cpu-rows", type=int, default=1, help="batch rows of one fusion step timed on the CPU")
    return ap.parse_args()


def build_sampler(args, device, seed):
    from tweediemix_amd import masks as M, sampler as S, unet as U, weights as Wt
    cfg = U.TINY if args.tiny else U.SDXL
    K = 3
    sd = Wt.synthetic_state_dict(cfg, seed=1234, device=device, dtype=torch.bfloat16)
    con = Wt.synthetic_concepts(cfg, args.kind, K, device=device)
    W = U.UNetWeights(cfg, sd, device, (args.kind, con))
    g = torch.Generator(device="cpu").manual_seed(42)
    te = (torch.randn(K + 2, 77, cfg.cross_dim, generator=g), torch.randn(K + 2, cfg.pooled_dim, generator=g))
    ts = (torch.randn(K, 77, cfg.cross_dim, generator=g), torch.randn(K, cfg.pooled_dim, generator=g))
    h = w = args.res // 8
    imgs = M.random_rectangle_masks(K, args.res, args.res, seed=seed)
    conf = S.make_config(guidance_scale=0.8, n_timesteps=50, t_cond=0.2, t_stop=0.8, resampling_steps=10,
                         jumping_steps=5, resolution_h=args.res, resolution_w=args.res, seed=seed)
    tw = S.Tweediemix(conf, W, te, ts, lambda x0: M.build_masks(imgs, h, w, device), concept_num=K,
                      lora=(args.kind == "lora"), use_graphs=not args.no_graphs, n_seeds=args.seeds_per_gpu,
                      n_streams=args.streams)
    tw.min_rows_per_stream = int(os.environ.get("TMIX_MIN_ROWS_PER_STREAM", str(tw.min_rows_per_stream)))
    tw.init_fusion(int(50 * 0.2), int(50 * 0.8)) if args.kind == "lora" else tw.init_fusion(int(50 * 0.2))
    tw.masks = M.build_masks(imgs, h, w, device)
    if args.seeds_per_gpu > 1:
        tw.masks = torch.stack([M.build_masks(M.random_rectangle_masks(K, args.res, args.res, seed=1000 * seed + i), h, w, device)
                                for i in range(args.seeds_per_gpu)]).contiguous()
    return tw, (sd, con, te, ts, cfg)


def gemm_roofline(plan):
    """per-launch HIP-event timing of the dominant kernel (gemm_conv_kernel<..,0>, the bf16 MFMA GEMM) inside
    one eager single-stream forward of the SAME launches the timed region replays:
    achieved = sum(algorithmic flops) / sum(launch durations)."""
    import ctypes as C
    from tweediemix_amd import lib as L
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    gemm_fn = lib.tmix_gemm_bf16
    conv_fn = lib.tmix_conv3x3_nhwc
    attn_fn = lib.tmix_attn_fwd
    fl_by = {"gemm": [f for _d, f in plan.launches["gemm"]], "conv": [f for _d, f in plan.launches["conv"]],
             "attn": [f for _a, f in plan.launches["attn"]]}
    ev = {"gemm": [], "conv": [], "attn": []}
    plan.run()
    torch.cuda.synchronize()
    if hasattr(plan, "plans"):          # PlanGroup: the instrumented pass runs the sub-plans back to back on one stream
        pass
    for fn, a in plan.ops:
        key = "gemm" if fn is gemm_fn else "conv" if fn is conv_fn else "attn" if fn is attn_fn else None
        if key:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = fn(*a, st)
        assert rc == 0
        if key:
            e1.record()
            ev[key].append((e0, e1))
    torch.cuda.synchronize()
    if os.environ.get("TMIX_BENCH_SHAPES"):
        import collections
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for (d, fl), (a, b) in zip(plan.launches["gemm"], ev["gemm"]):
            k = ("gemm", d.batch, d.M, d.N, d.K, "geglu" if d.epilogue else "", "T" if d.n_trans_begin >= 0 else "", d.tile_cfg)
            agg[k][0] += 1; agg[k][1] += a.elapsed_time(b); agg[k][2] += fl
        for (d, fl), (a, b) in zip(plan.launches["conv"], ev["conv"]):
            k = ("conv", d.B, d.H, d.W, d.Cin, d.Cout, d.mode, d.tile_cfg)
            agg[k][0] += 1; agg[k][1] += a.elapsed_time(b); agg[k][2] += fl
        for (args, fl), (a, b) in zip(plan.launches["attn"], ev["attn"]):
            k = ("attn", args[12], args[13], args[14], args[15])
            agg[k][0] += 1; agg[k][1] += a.elapsed_time(b); agg[k][2] += fl
        for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"  {str(k):60s} n={n:4d} total={ms:7.3f}ms avg={1e3 * ms / n:7.1f}us {fl / ms / 1e9:6.0f}TF", file=sys.stderr)
    out = {}
    # algorithmic HBM bytes of a GEMM launch: A and W read once, C written once, residual read once (bf16), GEGLU halves C
    gb = 0
    for d, _f in plan.launches["gemm"]:
        n_out = d.N // 2 if d.epilogue == L.EPI_GEGLU else d.N
        wsets = d.batch if d.strideW else 1
        gb += 2 * (d.batch * d.M * d.K + wsets * d.N * d.K + d.batch * d.M * n_out + (d.batch * d.M * d.N if d.residual else 0))
    for key in ev:
        ms = [a.elapsed_time(b) for a, b in ev[key]]
        out[key] = dict(launches=len(ms), total_ms=float(sum(ms)), avg_us=float(1e3 * sum(ms) / max(1, len(ms))),
                        tflops=float(sum(fl_by[key]) / max(1e-9, sum(ms)) / 1e9), flops=float(sum(fl_by[key])))
    out["gemm"]["alg_bytes_per_launch"] = gb / max(1, len(plan.launches["gemm"]))
    return out


def gemm_concurrent(plan, reps=3):
    """GEMM launches only, replayed the way the timed region runs them (one chain per HIP stream, concurrently):
    aggregate TFLOP/s = sum(flops) / wall.  Complements `roofline.achieved`, which times each launch alone."""
    from tweediemix_amd import lib as L
    lib = L.load()
    subs = plan.plans if hasattr(plan, "plans") else [plan]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in subs[1:]]
    lists = [[(fn, a) for fn, a in p.ops if fn is lib.tmix_gemm_bf16] for p in subs]
    flops = sum(f for p in subs for _d, f in p.launches["gemm"])
    def enqueue():
        fork = torch.cuda.Event(); fork.record()
        joins = []
        for ops_, st in zip(lists, streams):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                for fn, a in ops_:
                    fn(*a, st.cuda_stream)
                ev = torch.cuda.Event(); ev.record(st); joins.append(ev)
        for ev in joins:
            torch.cuda.current_stream().wait_event(ev)

    # captured once and replayed, like the timed region: eager launches from Python (~20 us each) would serialise the chains
    enqueue()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        streams[0] = torch.cuda.current_stream()
        enqueue()
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return {"tflops": flops / best / 1e9, "ms": best, "streams": len(subs)}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
    separate runs, FETCH doubled per MI355X_MICROARCH.md section HBM); collected by tools/collect_profile.sh with this
    same workload, since counters cannot be read from inside the process.  None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["gemm"]["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None


def cpu_baseline(args, parts, K=3):
    """the oracle (fp32 torch-CPU restatement of the same UNet) on `cpu_rows` of the K+1 batch rows of one
    fusion step; steps/s extrapolated by (K+1)/rows (rows are independent inside the UNet)."""
    from oracle import unet_oracle as UO
    sd, con, te, ts, cfg = parts
    ocfg = UO.TINY if args.tiny else UO.SDXL
    t0 = time.time()
    sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    orc = UO.UNetOracle(ocfg, sd_cpu)        # row 0 (uncond) uses base weights only
    rows = args.cpu_rows
    h = w = args.res // 8
    torch.manual_seed(0)
    x = torch.randn(rows, 4, h, w)
    tid = torch.tensor([[args.res, args.res, 0, 0, args.res, args.res]] * rows, dtype=torch.float32)
    cores = min(os.cpu_count() or 1, args.cpu_threads)     # torch-CPU stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    prep = time.time() - t0
    t1 = time.time()
    orc.forward(x, 781, te[0][:rows], te[1][:rows], tid)
    dt = time.time() - t1
    return dict(value=rows / ((K + 1) * dt), unit="steps/s", cores=cores, kind="port",
                sample=f"{rows} of {K + 1} batch rows of one fusion-step UNet forward at {args.res}x{args.res} "
                       f"(fp32 torch-CPU oracle, {dt:.1f}s; weight copy {prep:.1f}s not counted); rows are independent, "
                       f"so steps/s = rows/((K+1)*t); fused epilogue negligible")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("TMIX_SINGLE_GPU_DIST_TEST"):     # debug only: all ranks share GPU 0 over gloo (control-flow test)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if os.environ.get("TMIX_SINGLE_GPU_DIST_TEST"):
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    tw, parts = build_sampler(args, device, seed=rank)       # each rank owns its own seeds (weak scaling)
    K = tw.concept_num
    plan = tw.plan("fusion")
    fusion_ts = [t for t in tw.scheduler.timesteps if t <= tw.t_cond_cur and t in tw._window]
    seed_gen = torch.Generator().manual_seed(1000 + rank)
    S = args.seeds_per_gpu
    x = torch.randn(S, 4, tw.h, tw.w, generator=seed_gen).to(device)

    def step(i, x):
        from tweediemix_amd import lib as L
        t = fusion_ts[i % len(fusion_ts)]
        eps = tw._unet("fusion", x, t)
        return tw._step(x, eps, L.STEP_FUSION, tw.alpha(t), tw.alpha(t - tw.skip))

    for i in range(args.warmup):
        x = step(i, x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x = step(args.warmup + i, x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    if world > 1:
        from tweediemix_amd import dist as D
        dt = D.max_over_ranks(dt, device)
        # result gather (the only collective on this path): final latents of every rank's seeds
        gathered = D.gather_latents(x[:1].contiguous(), world, rank, world)
        assert gathered.shape[0] == world and torch.isfinite(gathered).all()

    traj = None
    if args.trajectory:
        # whole sample_loop with the reference's default flags (n=50, t_cond=0.2, resampling 10, jumping 5): SURVEY 8d(ii)
        from tweediemix_amd import masks as M
        imgs = M.random_rectangle_masks(K, args.res, args.res, seed=7)
        if S > 1:                       # the sampler asks once per seed, in seed order: hand out that seed's mask set
            per_seed, turn = tw.masks.clone(), [0]

            def provider(x0):
                turn[0] += 1
                return per_seed[(turn[0] - 1) % S]
            tw.mask_provider = provider
        else:
            tw.mask_provider = lambda x0: M.build_masks(imgs, tw.h, tw.w, device)
        xT = torch.randn(S, 4, tw.h, tw.w, generator=seed_gen)
        from tweediemix_amd import vae as V
        tw.vae = (V.FULL, V.synthetic_state_dict(V.FULL, device=device))     # random-init decoder of the SDXL VAE shapes
        tw.unet_calls.clear()
        tw.run_fusion(xT.clone(), decode=True)          # builds/captures the start / plain / VAE plans too
        torch.cuda.synchronize()
        n_calls = len(tw.unet_calls)
        t1 = time.perf_counter()
        lat = tw.run_fusion(xT.clone())
        torch.cuda.synchronize()
        dtt = time.perf_counter() - t1
        img = tw.decode_final(lat)
        torch.cuda.synchronize()
        dti = time.perf_counter() - t1
        assert torch.isfinite(lat).all() and torch.isfinite(img).all()
        traj = {"trajectory_steps_per_s": 50 * S / dtt, "seconds_per_image": dtt / S, "unet_calls_per_image": n_calls,
                "images_per_s_incl_vae_decode": S / dti, "vae_decode_ms": 1e3 * (dti - dtt) / S,
                "calls_B4": sum(1 for c in tw.unet_calls[n_calls:] if c[1] == K + 1), "calls_B2": sum(1 for c in tw.unet_calls[n_calls:] if c[1] == 2)}
    if rank == 0:
        roof = gemm_roofline(plan)
        g = roof["gemm"]
        line = {
            "metric": "denoise steps/sec @ SDXL 1024^2 K=3 concepts (fusion phase)",
            "value": world * S * args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / (args.steps * S), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SDXL-base UNet shapes, {args.res}x{args.res}, K=3 concepts ({args.kind} deltas), "
                                   f"fusion-phase step = UNet B={K + 1} + fused Tweedie/CFG/blend/DDIM kernel"
                                   + (" [TINY DEBUG CONFIG]" if args.tiny else ""),
                       "seeds_per_gpu": S, "streams": args.streams, "hip_graph": not args.no_graphs, "parallelism": f"replicas x{world} (seed-sharded)"},
            "unet_tflop_per_step": plan.flops / 1e12 / S,
            "achieved_tflops_whole_step": plan.flops / 1e12 / (dt / args.steps),
            "roofline": {"bound": "mfma", "kernel": "gemm_conv_kernel<0> (tmix_gemm_bf16)", "achieved": g["tflops"],
                         "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": g["tflops"] / BF16_DENSE_PEAK_TFLOPS,
                         "traffic": pmc_traffic(), "algorithmic_bytes_per_launch": g["alg_bytes_per_launch"], "launches_per_step": g["launches"], "avg_launch_us": g["avg_us"],
                         "flops_per_step": g["flops"], "concurrent_replay": gemm_concurrent(plan),
                         "other_kernels": {k: {kk: v[kk] for kk in ("launches", "total_ms", "avg_us", "tflops")}
                                           for k, v in roof.items() if k != "gemm"}},
        }
        if traj is not None:
            line["trajectory"] = traj
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, parts)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
