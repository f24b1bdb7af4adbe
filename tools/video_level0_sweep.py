"""the video UNet's level-0 projections (16 frames x 56 x 96 = 86016 rows, C = 320: K = 320 / 1280, memory-bound) hot, every tiling, with the epilogues the plan uses:
python tools/video_level0_sweep.py"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
from tweediemix_amd.weights import interleave_geglu
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
def t(d, reps=10):
    for _ in range(2): L.check(lib.tmix_gemm_bf16(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.tmix_gemm_bf16(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M = 86016
for (N, K, kind) in ((2560, 320, "geglu"), (320, 320, "res"), (960, 320, "plain"), (320, 1280, "res"), (10240, 1280, "geglu5376"), (5120, 640, "geglu21504")):
    m = 5376 if kind.endswith("5376") else 21504 if kind.endswith("21504") else M
    a = torch.randn(m, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF); bias = torch.randn(N, device="cuda")
    row = []
    for cfg in (1, 2, 3, 4, 5, 7, 12, 13, 14, 16, 17, 20, 21, 22):
        try:
            if kind.startswith("geglu"):
                out = torch.empty(m, N // 2, device="cuda", dtype=BF)
                d = ops.make_gemm_desc(a, w, out, bias=bias, geglu=True, tile_cfg=cfg)
                by = 2 * (m * K + N * K + m * N // 2)
            elif kind == "res":
                out = torch.empty(m, N, device="cuda", dtype=BF); res = torch.randn(m, N, device="cuda").to(BF)
                d = ops.make_gemm_desc(a, w, out, bias=bias, residual=res, tile_cfg=cfg)
                by = 2 * (m * K + N * K + 2 * m * N)
            else:
                out = torch.empty(m, N, device="cuda", dtype=BF)
                d = ops.make_gemm_desc(a, w, out, bias=bias, tile_cfg=cfg)
                by = 2 * (m * K + N * K + m * N)
            us = t(d)
            row.append((us, cfg))
        except Exception as e:
            pass
    row.sort()
    print(f"{m}x{N}x{K} {kind}: algorithmic {by / 1e6:.0f} MB; " + " ".join(f"c{c}:{u:.0f}us({by / u / 1e6:.2f}TB/s)" for u, c in row[:6]), flush=True)
