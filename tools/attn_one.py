"""one self-attention launch shape, hot: tmix_attn_fwd against tmix_attn_fwd_ws (key-split tail).  python tools/attn_one.py B H S Skv"""
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
ws = ops.attention_split_ws(B, H, S, Skv, "cuda")


def timed(w):
    for _ in range(3): ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out, ws=w)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out, ws=w)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50)
    return best


us = timed(None)
print(f"attn B={B} H={H} S={S} Skv={Skv}: unsplit {us:.1f} us  {4*B*H*S*Skv*64/us/1e6:.0f} TF", end="")
if ws is not None:
    us2 = timed(ws)
    print(f"   key-split tail {us2:.1f} us  {4*B*H*S*Skv*64/us2/1e6:.0f} TF  ({ws.numel() / 2**20:.1f} MB workspace)")
else:
    print("   (shape does not split)")
