"""attn2 as one launch (tmix_gemm_q_cross_attn) against the two launches it replaces (tmix_gemm_bf16 tiling 21 -> tmix_attn_fwd), hot, inside a captured graph of
20 back-to-back pairs (so that launch gaps count)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops
from tweediemix_amd.weights import fold_layernorm
BF = torch.bfloat16
def bench(B, S, C, routed):
    g = torch.Generator().manual_seed(0)
    h = torch.randn(B, S, C, generator=g).to(BF).cuda()
    P = B if routed else 1
    wq = (torch.randn(P, C, C, generator=g) * C ** -0.5).to(BF).cuda()
    bq = torch.randn(P, C, generator=g).cuda()
    gamma, beta = torch.ones(C).cuda(), torch.zeros(C).cuda()
    fold = [fold_layernorm(wq[i], gamma, beta, bq[i]) for i in range(P)]
    wp, cs, t = [torch.stack([f[j] for f in fold]).contiguous() for j in range(3)]
    k = torch.randn(B, 77, C, generator=g).to(BF).cuda()
    vt = torch.zeros(B, C, 80, dtype=BF).cuda(); vt[:, :, :77] = torch.randn(B, C, 77, generator=g).to(BF).cuda()
    hf = h.float()
    stats = torch.stack([hf.sum(-1), (hf ** 2).sum(-1)], -1).view(1, B * S, 2).contiguous()
    a = h if routed else h.view(B * S, C)
    w_, b_, cs_ = (wp, t, cs) if routed else (wp[0], t[0], cs[0])
    out = torch.empty_like(a); q = torch.empty_like(a); out2 = torch.empty(B, S, C, dtype=BF, device="cuda")
    def fused(): ops.gemm_q_cross_attn(a, w_, k, vt, S, 0.125, out=out, bias=b_, ln_stats=stats, ln_colsum=cs_)
    def two():
        ops.gemm(a, w_, out=q, bias=b_, ln_stats=stats, ln_colsum=cs_, tile_cfg=21)
        ops.attention(q.view(B, S, C), k, vt, C // 64, 77, 0.125, out=out2)
    res = []
    for fn in (fused, two):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); e1.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 100)
    print(f"B={B} S={S} C={C} routed={routed}: fused {res[0]:6.1f} us | gemm(21) + attention {res[1]:6.1f} us", flush=True)
for cfg in ((4, 1024, 1280, True), (4, 1024, 1280, False), (4, 4096, 640, True), (2, 1024, 1280, False), (16, 1024, 1280, True)):
    bench(*cfg)
