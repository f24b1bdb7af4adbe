"""one hot FF2 launch (4096 x 1280 x 5120, bias + residual + row statistics) per tiling, 20 times each, for the LDS counters of tools/jobs5/r5o_pmc.sh"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
M, N, K = 4096, 1280, 5120
a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(BF); out = torch.empty(M, N, device="cuda", dtype=BF)
for cfg in (21, 23):
    stats = torch.zeros(ops.stats_parts(N, cfg), M, 2, device="cuda")
    d = ops.make_gemm_desc(a, w, out, bias=bias, residual=res, row_stats_out=stats, tile_cfg=cfg)
    for _ in range(20): lib.tmix_gemm_bf16(C.byref(d), st)
    torch.cuda.synchronize()
