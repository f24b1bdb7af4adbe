import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import tweedie_oracle as TO
from tweediemix_amd import masks as M, sampler as S
g = np.load("tests/golden/traj_custom_n50_f16.npz")
K, n, h, w = int(g["K"]), int(g["n"]), int(g["h"]), int(g["w"])
class NW: device=torch.device("cuda"); kind="custom"; K=3
cfg = S.make_config(guidance_scale=float(g["guidance_scale"]), n_timesteps=n, t_cond=0.2, resampling_steps=10, jumping_steps=5, resolution_h=h*8, resolution_w=w*8)
tw = S.Tweediemix(cfg, NW(), None, None, lambda x0: M.build_masks(list(g["mask_images"]), h, w), concept_num=K)
idx=[0]
def fake(kind, x, t):
    i = idx[0]; idx[0]+=1
    return torch.from_numpy(g[f"req{i}_eps"]).cuda().half()
tw._unet = fake
tw.init_fusion(10)
o = TO.TweedieOracle(K, n, g=float(g["guidance_scale"]), t_cond=0.2, lowp=np.float16, mask_fn=lambda: TO.build_masks(list(g["mask_images"]), h, w))
j=[0]
def ofn(xx, t, rows, kind, routed):
    j[0]+=1; return g[f"req{j[0]-1}_eps"]
x = torch.from_numpy(g["xs"][0]).cuda(); xo = g["xs"][0]
for k,t in enumerate(g["timesteps"]):
    x = tw.denoise_step(x, int(t)).clone()
    xo = o.denoise_step(xo, int(t), ofn)
    d = np.abs(x.cpu().numpy()-xo).max()
    if d>0: print(k, t, d)
print("---- isolate step 42")
from tweediemix_amd import ops, lib as L
x = torch.from_numpy(g["xs"][0]).cuda(); xo = g["xs"][0]
idx[0]=0; j[0]=0
tw.masks=None; o.masks=None
for k,t in enumerate(g["timesteps"]):
    if k==42:
        i = idx[0]
        eps = g[f"req{i}_eps"]
        at, an = tw.alpha(int(t)), tw.alpha(int(t)-20)
        ref,_ = TO.fused_fusion_step(xo, eps, o.masks, np.float32(0.8), at, an, False, np.float16)
        mk = torch.from_numpy(np.ascontiguousarray(o.masks)).cuda().contiguous(); print(mk.shape, mk.dtype, mk.is_contiguous())
        out = ops.fused_tweedie_step(torch.from_numpy(xo).cuda(), torch.from_numpy(eps).cuda().half(), mk, L.STEP_FUSION, 3, 0.8, at, an)
        outn = out.cpu().numpy()
        bad = np.argwhere(outn != ref)
        print("nbad", len(bad), "at", at, an)
        for b in bad[:3]:
            b = tuple(b); p = b[2:]
            print("idx", b, "hip", repr(outn[b]), "ref", repr(ref[b]), "x", repr(xo[b]), "eps", [repr(eps[(r,)+b[1:]]) for r in range(4)], "masks", [o.masks[(c,0)+p] for c in range(3)])
    x = tw.denoise_step(x, int(t)).clone()
    xo = o.denoise_step(xo, int(t), ofn)
