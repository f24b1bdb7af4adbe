"""per-workgroup phase times of the self-attention kernel (tmix_prof detail mode): prologue (entry -> first tiles landed), K/V loop, epilogue
python tools/attn_phases.py B H S Skv"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import ops, lib as L
lib = L.load()
B, H, S, Skv = [int(v) for v in sys.argv[1:5]]
BF = torch.bfloat16
C = H * 64
qk = torch.randn(B, S, 2 * C, device="cuda").to(BF)
k = torch.randn(B, Skv, C, device="cuda").to(BF)
vt = torch.randn(B, C, (Skv + 7) // 8 * 8, device="cuda").to(BF)
out = torch.empty(B, S, C, device="cuda", dtype=BF)
for _ in range(3): ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out)
torch.cuda.synchronize()
res = []
for rep in range(5):
    slots = torch.zeros(4, 8, dtype=torch.int64, device="cuda"); slots[:, 0] = -1
    L.check(lib.tmix_prof_begin(slots.data_ptr(), 4, 1), "prof")
    ops.attention(qk[:, :, :C], k, vt, H, Skv, 0.125, out=out)
    lib.tmix_prof_end(); torch.cuda.synchronize()
    s = slots[0].cpu().numpy().astype("uint64")
    n = int(s[5])
    res.append(((int(s[1]) - int(s[0])) / 100, int(s[2]) / n / 100, (int(s[3]) - int(s[2])) / n / 100, (int(s[4]) - int(s[3])) / n / 100, n))
res.sort()
span, pro, loop, epi, n = res[2]
print(f"attn B={B} H={H} S={S} Skv={Skv}: kernel span {span:.1f} us, {n} workgroups: prologue {pro:.2f} loop {loop:.2f} epilogue {epi:.2f} us per workgroup")
