import os, sys, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load(); BF = torch.bfloat16
M, N, K, cfg = [int(v) for v in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
out = torch.empty(M, N, device="cuda", dtype=BF)
d = ops.make_gemm_desc(a, w, out, tile_cfg=cfg)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps): lib.tmix_gemm_bf16(C.byref(d), st)
torch.cuda.synchronize()
