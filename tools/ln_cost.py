"""time producer / consumer GEMMs with and without the fused-LayerNorm epilogues (per tile cfg)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream

def t(d, reps=20):
    for _ in range(3): lib.tmix_gemm_bf16(C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.tmix_gemm_bf16(C.byref(d), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for (M, Cc) in ((4096, 1280), (16384, 640), (2048, 1280), (8192, 640)):
    a = torch.randn(M, Cc, device="cuda").to(BF)
    w = (torch.randn(Cc, Cc, device="cuda") * Cc ** -0.5).to(BF)
    w8 = (torch.randn(8 * Cc, Cc, device="cuda") * Cc ** -0.5).to(BF)
    w3 = (torch.randn(3 * Cc, Cc, device="cuda") * Cc ** -0.5).to(BF)
    bias = torch.randn(Cc, device="cuda"); b8 = torch.randn(8 * Cc, device="cuda")
    res = torch.randn(M, Cc, device="cuda").to(BF)
    out = torch.empty(M, Cc, device="cuda", dtype=BF); f = torch.empty(M, 4 * Cc, device="cuda", dtype=BF)
    stats = torch.zeros((Cc + 127) // 128, M, 2, device="cuda")
    cs = torch.randn(8 * Cc, device="cuda")
    for cfg in (1, 2, 3, 7):
        p0 = t(ops.make_gemm_desc(a, w, out, bias=bias, residual=res, tile_cfg=cfg))
        p1 = t(ops.make_gemm_desc(a, w, out, bias=bias, residual=res, row_stats_out=stats, tile_cfg=cfg))
        c0 = t(ops.make_gemm_desc(a, w, out, bias=bias, tile_cfg=cfg))
        c1 = t(ops.make_gemm_desc(a, w, out, bias=bias, ln_stats=stats, ln_colsum=cs[:Cc], tile_cfg=cfg))
        g0 = t(ops.make_gemm_desc(a, w8, f, bias=b8, geglu=True, tile_cfg=cfg))
        g1 = t(ops.make_gemm_desc(a, w8, f, bias=b8, geglu=True, ln_stats=stats, ln_colsum=cs, tile_cfg=cfg))
        print(f"M={M} C={Cc} cfg{cfg}: producer {p0:.1f} -> {p1:.1f} us | consumer(q) {c0:.1f} -> {c1:.1f} | ff1 {g0:.1f} -> {g1:.1f}")
    x = torch.randn(M, Cc, device="cuda").to(BF)
    g = torch.ones(Cc, device="cuda"); b = torch.zeros(Cc, device="cuda")
    ops.layernorm(x, g, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.layernorm(x, g, b)
    e1.record(); e1.synchronize()
    print(f"M={M} C={Cc} layernorm kernel {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
