"""FF up-projection (GEGLU, folded LayerNorm) on persistent workgroups (tiling 24) against tilings 14 / 22, hot and inside a graph of 10 back-to-back launches"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
from tweediemix_amd.weights import fold_layernorm, interleave_geglu
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in ((4096, 10240, 1280), (16384, 5120, 640), (2048, 10240, 1280), (16384, 10240, 1280)):
    g = torch.Generator().manual_seed(0)
    h = torch.randn(M, K, generator=g).to(BF).cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF).cuda(); b = torch.randn(N, generator=g).cuda()
    wp, cs, t = fold_layernorm(w, torch.ones(K).cuda(), torch.zeros(K).cuda(), b)
    wi = interleave_geglu(wp, None)[0]; csi, ti = interleave_geglu(cs[:, None], t)
    hf = h.float(); stats = torch.stack([hf.sum(-1), (hf ** 2).sum(-1)], -1).view(1, M, 2).contiguous()
    out = torch.empty(M, N // 2, device="cuda", dtype=BF)
    row = []
    for cfg in (14, 22, 24):
        d = ops.make_gemm_desc(h, wi, out, bias=ti, geglu=True, ln_stats=stats, ln_colsum=csi[:, 0].contiguous(), tile_cfg=cfg)
        for _ in range(5): lib.tmix_gemm_bf16(C.byref(d), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): lib.tmix_gemm_bf16(C.byref(d), st)
        e1.record(); e1.synchronize()
        row.append(f"cfg{cfg}: {e0.elapsed_time(e1) * 1e3 / 30:6.1f} us")
    print(f"M={M} N={N} K={K}: " + " | ".join(row), flush=True)
