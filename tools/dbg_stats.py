import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops
BF = torch.bfloat16
def rnd(*s, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).to(dtype).cuda()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
M, N, K = 520, 640, 256
a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5)
bias, res = rnd(N, seed=5, dtype=torch.float32), rnd(M, N, seed=6)
def run():
    st = torch.zeros(ops.stats_parts(N, cfg), M, 2, device="cuda")
    c2 = ops.gemm(a, w, bias=bias, residual=res, row_stats_out=st, tile_cfg=cfg)
    torch.cuda.synchronize()
    return c2, st
outs = []
for mode in ("wide", "wide", "narrow", "narrow"):
    if mode == "narrow": os.environ["TMIX_NARROW_EPILOGUE"] = "1"
    outs.append(run())
c = outs[0][0].float()
for p in range(outs[0][1].shape[0]):
    import ctypes as C
    from tweediemix_amd import lib as L
    bm_, bn_ = C.c_int(), C.c_int(); L.load().tmix_gemm_tile_shape(cfg, C.byref(bm_), C.byref(bn_)); bn = bn_.value
    ref1 = c[:, p * bn:(p + 1) * bn].sum(1); ref2 = (c[:, p * bn:(p + 1) * bn] ** 2).sum(1)
    for i, (cc, st) in enumerate(outs):
        d1 = (st[p, :, 0] - ref1).abs(); d2 = (st[p, :, 1] - ref2).abs()
        bad = ((d1 > 1e-2) | (d2 > 1e-1)).nonzero().flatten().tolist()
        print("part", p, ("wide", "wide", "narrow", "narrow")[i], "C equal:", torch.equal(cc, outs[0][0]), "bad rows:", bad[:40], len(bad))
