import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops
B, H, S, Skv = [int(v) for v in sys.argv[1:5]]
BF = torch.bfloat16
C = H * 64
qk = torch.randn(B, S, 2 * C, device="cuda").to(BF)
k = torch.randn(B, Skv, C, device="cuda").to(BF)
vt = torch.randn(B, C, (Skv + 7) // 8 * 8, device="cuda").to(BF)
out = torch.empty(B, S, C, device="cuda", dtype=BF)
for _ in range(3): ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"attn B={B} H={H} S={S} Skv={Skv}: {us:.1f} us  {4*B*H*S*Skv*64/us/1e6:.0f} TF")
