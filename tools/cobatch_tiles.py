"""tilings for the launches of 4 co-batched seeds (B = 16 rows, the images/s regime): routed cubes / q|k|v with 4 periodic weight sets, FF1, FF2 --
bf16 and e4m3, each tiling timed as a graph of 16 launches over 8 rotating operand sets (weights and activations out of the L2 between repeats)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops
BF = torch.bfloat16
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(BF)
shapes = [("cube", 16, 1024, 1280, 1280, 4, False), ("qkv", 16, 1024, 3840, 1280, 4, False), ("ff2", 1, 16384, 1280, 5120, 1, False), ("ff1", 1, 16384, 10240, 1280, 1, True),
          ("cube64", 16, 4096, 640, 640, 4, False), ("cube8", 8, 1024, 1280, 1280, 4, False), ("cube64_4", 4, 4096, 640, 640, 4, False),
          ("cube64_8", 8, 4096, 640, 640, 4, False), ("cube4", 4, 1024, 1280, 1280, 4, False), ("ff2_8", 1, 8192, 1280, 5120, 1, False)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] in sys.argv[1:]]
NSET = 8
for name, Bz, M, N, K, P, geglu in shapes:
    As = [rnd(Bz, M, K) for _ in range(NSET)]
    Ws = [rnd(P, N, K, sc=K ** -0.5) for _ in range(NSET)]
    out = torch.empty(Bz, M, N // 2 if geglu else N, device=dev, dtype=BF)
    fl = 2.0 * Bz * M * N * K
    for f8 in (False, True):
        if f8:
            q = [(ops.quantize_fp8_rows(a), ops.quantize_fp8_rows(w)) for a, w in zip(As, Ws)]
        res = []
        for tile in ([0, 4, 7, 12, 14, 16, 17, 19, 20, 21] if not f8 else [0, 12, 16, 17, 19, 20, 21]):
            def run(i):
                a, w = As[i % NSET], Ws[i % NSET]
                kw = dict(tile_cfg=tile) if tile else {}
                if geglu:
                    kw["geglu"] = True
                aa = a if Bz > 1 else a[0]
                ww = w if P > 1 else w[0]
                oo = out if Bz > 1 else out[0]
                if f8:
                    (a8, sa), (w8, sw) = q[i % NSET]
                    ops.gemm_fp8(a8 if Bz > 1 else a8[0], sa if Bz > 1 else sa[0], w8 if P > 1 else w8[0], sw if P > 1 else sw[0], out=oo, **kw)
                else:
                    ops.gemm(aa, ww, out=oo, **kw)
            try:
                run(0); torch.cuda.synchronize()
            except Exception as e:
                res.append((tile, None)); continue
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(16):
                    run(i)
            gr.replay(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 16)
            res.append((tile, sorted(ts)[2]))
        print(f"{name:7s} {'fp8 ' if f8 else 'bf16'} B={Bz} M={M} N={N} K={K}: " + "  ".join(f"{t}:{'--' if ms is None else f'{ms * 1e3:.1f}us/{fl / ms / 1e9:.0f}TF'}" for t, ms in res), flush=True)
    del As, Ws
