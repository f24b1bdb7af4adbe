#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE; runs ONLY in the build container).

Imports the reference's own hot-path modules from /root/reference/fusion_generation
(fusion_sampling.py, fusion_sampling_lora.py, utils_custom.py, utils_lora.py,
model_lora.py) with the third-party packages that are absent here (diffusers,
torchvision, xformers, sentence_transformers) replaced by attribute-mock stub
modules, drives the *real* `Tweediemix.init_fusion` / `denoise_step` /
`preprocess_mask` / `register_attention_control_efficient` code on CPU with a
recording fake UNet, and writes the resulting tensors as small .npz fixtures under
tests/golden/.  Nothing of the reference (source, bytecode) is written anywhere;
only tensors are.  The GPU box never runs this file (there is no /root/reference
there); tests read the committed fixtures.

Usage:  python oracle/gen_golden.py            (rewrites tests/golden/*.npz)
"""
import importlib.abc
import importlib.machinery
import os
import shutil
import sys
import tempfile
import types
from types import SimpleNamespace
from unittest import mock

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# ----------------------------------------------------------------------------
# stub loader for missing third-party roots
# ----------------------------------------------------------------------------
def install_stubs():
    # resolve transformers' lazy attributes BEFORE torchvision looks installed
    from transformers import CLIPTextModel, CLIPTokenizer, CLIPTextModelWithProjection  # noqa: F401
    sys.modules["transformers"].CLIPFeatureExtractor = mock.MagicMock()
    import accelerate.logging  # noqa: F401

    missing = {"torchvision", "diffusers", "sentence_transformers", "xformers"}

    class _Loader(importlib.abc.Loader):
        def create_module(self, spec):
            m = types.ModuleType(spec.name)
            m.__path__ = []

            def _ga(n, _p=spec.name):
                if n.startswith("__"):
                    raise AttributeError(n)
                return mock.MagicMock(name=f"{_p}.{n}")
            m.__getattr__ = _ga
            return m

        def exec_module(self, m):
            pass

    class _Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in missing:
                return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
            return None

    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, os.path.join(REF, "fusion_generation"))


# ----------------------------------------------------------------------------
# fake UNet with the SDXL attention topology the hooks walk
# ----------------------------------------------------------------------------
class FakeAttention(nn.Module):
    """Minimal stand-in for diffusers' Attention: exactly the attributes the
    reference hooks touch (to_q,to_k,to_v,to_out,heads,scale,head_to_batch_dim,
    batch_to_head_dim, .processor)."""

    def __init__(self, C, cross_dim, heads, gen):
        super().__init__()
        self.heads = heads
        self.scale = (C // heads) ** -0.5
        kd = cross_dim if cross_dim is not None else C
        self.to_q = nn.Linear(C, C, bias=False)
        self.to_k = nn.Linear(kd, C, bias=False)
        self.to_v = nn.Linear(kd, C, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(C, C, bias=True), nn.Dropout(0.0)])
        for p in self.parameters():
            p.data = torch.randn(p.shape, generator=gen) * (0.5 / np.sqrt(p.shape[-1]))
        self.processor = None

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        h = self.heads
        return t.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)


class _TB(nn.Module):
    def __init__(self, C, cross, heads, gen):
        super().__init__()
        self.attn1 = FakeAttention(C, None, heads, gen)
        self.attn2 = FakeAttention(C, cross, heads, gen)


class _T2D(nn.Module):
    def __init__(self, n, C, cross, heads, gen):
        super().__init__()
        self.transformer_blocks = nn.ModuleList([_TB(C, cross, heads, gen) for _ in range(n)])


class _Blk(nn.Module):
    def __init__(self, layers, C, cross, heads, gen):
        super().__init__()
        self.attentions = nn.ModuleList([_T2D(n, C, cross, heads, gen) for n in layers])


class FakeUNet(nn.Module):
    """SDXL block topology (down[1]:2x2, down[2]:2x10, mid:10, up[0]:3x10, up[1]:3x2) with
    tiny widths; forward() returns a recorded/deterministic eps and logs the request."""

    def __init__(self, C=16, cross=8, heads=2, seed=0, eps_dtype=torch.float32, with_lora=False):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.down_blocks = nn.ModuleList([
            nn.Module(), _Blk([2, 2], C, cross, heads, gen), _Blk([10, 10], C, cross, heads, gen)])
        self.mid_block = _Blk([10], C, cross, heads, gen)
        self.up_blocks = nn.ModuleList([
            _Blk([10, 10, 10], C, cross, heads, gen), _Blk([2, 2, 2], C, cross, heads, gen), nn.Module()])
        self._dev = torch.device("cpu")
        self.eps_dtype = eps_dtype
        self.log = []           # list of dict(B, t, rows, pooled_rows, x, eps)
        self.acp = sdxl_alphas_cumprod()
        if with_lora:
            from model_lora import LoRAAttnProcessor_base
            for m in self.modules():
                if isinstance(m, _TB):
                    for a, cd in ((m.attn1, None), (m.attn2, cross)):
                        a.processor = LoRAAttnProcessor_base(hidden_size=C, cross_attention_dim=cd)
                        for p in a.processor.parameters():
                            p.data = torch.randn(p.shape, generator=gen) * 0.3

    @property
    def device(self):
        return self._dev

    def forward(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None):
        B = x.shape[0]
        rows = encoder_hidden_states[:, 0, 0].clone()
        prow = added_cond_kwargs["text_embeds"][:, 0].clone()
        tt = int(t)
        # deterministic toy eps-model: depends on latent, timestep and the text row id
        g = torch.Generator().manual_seed(1000 * len(self.log) + 7)
        base = torch.randn(x.shape[1:], generator=g)
        # (optimal denoiser of unit-variance gaussian data keeps the trajectory bounded)
        a_t = float(self.acp[tt - 1]) if tt >= 1 else 1.0
        s1 = (1.0 - a_t) ** 0.5
        eps = torch.stack([s1 * x[b].float() + 0.3 * base + 0.1 * torch.sin(rows[b] + 0.001 * tt + x[b].float())
                           for b in range(B)])
        eps = eps.to(self.eps_dtype)
        self.log.append(dict(B=B, t=tt, rows=rows.numpy().copy(), prow=prow.numpy().copy(),
                             time_ids=added_cond_kwargs["time_ids"].numpy().copy(),
                             x=x.detach().float().numpy().copy(), eps=eps.detach().float().numpy().copy()))
        return {"sample": eps}


def sdxl_alphas_cumprod():
    """diffusers DDIMScheduler(beta_schedule='scaled_linear', 0.00085..0.012, 1000 steps)."""
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def leading_timesteps(n, steps_offset=1):
    step_ratio = 1000 // n
    ts = (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64) + steps_offset
    return torch.from_numpy(ts)


def make_tweedie(mod, K, n, h, w, cfg_over, eps_dtype, lora, outdir, mask_arrays):
    Tw = mod.Tweediemix
    tw = Tw.__new__(Tw)
    nn.Module.__init__(tw)
    seg = "+".join(f"fg{i}" for i in range(K - 1))
    cfg = dict(guidance_scale=0.8, resampling_steps=10, jumping_steps=5, output_path=outdir,
               output_path_all=outdir, seg_gpu=0, seg_concepts=seg, resolution_h=h * 8, resolution_w=w * 8,
               n_timesteps=n, t_cond=0.2, t_stop=0.8, prompt_orig="p", seed=0)
    cfg.update(cfg_over)
    tw.config = SimpleNamespace(**cfg)
    tw.unet = FakeUNet(eps_dtype=eps_dtype, with_lora=lora, seed=1)
    for i in range(K):
        setattr(tw, f"unet_{i}", FakeUNet(eps_dtype=eps_dtype, with_lora=lora, seed=10 + i))
    tw.concept_num = K
    tw.masks = None
    # text rows carry their identity in every element: rows 0..K+1 -> 0,1,2..; singles -> 100+
    te = torch.stack([torch.full((77, 8), float(r)) for r in range(K + 2)])
    tp = torch.stack([torch.full((12,), float(r)) for r in range(K + 2)])
    ts_ = torch.stack([torch.full((77, 8), 100.0 + r) for r in range(K)])
    tps = torch.stack([torch.full((12,), 100.0 + r) for r in range(K)])
    tw.text_embeds = (te, tp)
    tw.text_embeds_single = (ts_, tps)
    ac = sdxl_alphas_cumprod()
    tw.scheduler = SimpleNamespace(alphas_cumprod=torch.cat([torch.tensor([1.0]), ac]),
                                   timesteps=leading_timesteps(n), init_noise_sigma=1.0)
    tw.final_alpha_cumprod = ac[0]
    tw.skip = 1000 // n
    tw.add_time_ids = torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]])
    tw.decoded = []
    tw.decode_latent = lambda lat: (tw.decoded.append(lat.detach().float().numpy().copy()),
                                    torch.zeros(1, 3, 8, 8))[1]
    # pre-place the segmentation masks the side-car would have written
    from PIL import Image
    for i, arr in enumerate(mask_arrays):
        Image.fromarray(arr, mode="L").save(os.path.join(outdir, f"fg{i}.png"))
    return tw


def run_trajectory(mod, name, K, n, h, w, cfg_over, eps_dtype, lora, seed):
    outdir = tempfile.mkdtemp(prefix="tmix_gold_")
    rng = np.random.RandomState(seed)
    H, W = h * 8, w * 8
    mask_arrays = []
    overlap = bool(cfg_over.pop("overlap_masks", False))
    for i in range(K - 1):
        m = np.zeros((H, W), np.uint8)
        if overlap:      # un-normalised blend weights (sum>1 where fg rectangles overlap), as :467-469 allows
            y0, x0 = rng.randint(0, H // 2), rng.randint(0, W // 2)
            hh, ww = rng.randint(H // 4, H // 2), rng.randint(W // 4, W // 2)
        else:            # disjoint vertical strips, like run_expand.py's overlap resolution produces
            sw = W // (K - 1)
            y0, hh = rng.randint(0, H // 3), rng.randint(H // 3, H // 2)
            x0 = i * sw + rng.randint(0, sw // 4)
            ww = rng.randint(sw // 3, sw // 2)
        m[y0:y0 + hh, x0:x0 + ww] = 255
        # soft edge values to exercise the 0.5 threshold
        m[y0:y0 + hh, x0] = 127
        m[y0:y0 + hh, min(x0 + ww, W - 1)] = 128
        mask_arrays.append(m)
    tw = make_tweedie(mod, K, n, h, w, cfg_over, eps_dtype, lora, outdir, mask_arrays)

    # lossless masks: the reference reads "<concept>.jpg"; patch the path join inside preprocess
    # by saving PNG bytes under the .jpg name (PIL sniffs the content, not the extension).
    for i in range(K - 1):
        shutil.copy(os.path.join(outdir, f"fg{i}.png"), os.path.join(outdir, f"fg{i}.jpg"))

    sys_calls = []
    with mock.patch("os.system", lambda cmd: sys_calls.append(cmd) or 0):
        t_cond = int(n * tw.config.t_cond)
        if lora:
            tw.init_fusion(t_cond, int(n * tw.config.t_stop))
        else:
            tw.init_fusion(t_cond)
        torch.manual_seed(seed)
        x = torch.randn(1, 4, h, w)
        xs = [x.numpy().copy()]
        with torch.no_grad():
            for t in tw.scheduler.timesteps:
                x = tw.denoise_step(x, t)
                xs.append(x.detach().float().numpy().copy())
    log = tw.unet.log
    out = dict(
        K=K, n=n, h=h, w=w, lora=int(lora), seed=seed,
        guidance_scale=tw.config.guidance_scale, resampling_steps=tw.config.resampling_steps,
        jumping_steps=tw.config.jumping_steps, t_cond=tw.config.t_cond, t_stop=tw.config.t_stop,
        eps_is_fp16=int(eps_dtype == torch.float16),
        timesteps=tw.scheduler.timesteps.numpy(), xs=np.stack(xs),
        masks=tw.masks.numpy(), mask_images=np.stack(mask_arrays),
        req_B=np.array([r["B"] for r in log]), req_t=np.array([r["t"] for r in log]),
        preview_x0=np.stack(tw.decoded) if tw.decoded else np.zeros((0,)),
        n_sys_calls=len(sys_calls),
        t_cond_list=np.array([int(v) for v in tw.t_cond]),
    )
    for i, r in enumerate(log):
        out[f"req{i}_rows"] = r["rows"]
        out[f"req{i}_prow"] = r["prow"]
        out[f"req{i}_x"] = r["x"]
        out[f"req{i}_eps"] = r["eps"]
        out[f"req{i}_time_ids"] = r["time_ids"]
    np.savez_compressed(os.path.join(OUT, name), **out)
    shutil.rmtree(outdir)
    print(f"{name}: {len(log)} unet requests, B4={int((out['req_B'] == K + 1).sum())} B2={int((out['req_B'] == 2).sum())}")


def gen_schedule(mod):
    ac = sdxl_alphas_cumprod()
    out = dict(alphas_cumprod=ac.numpy())
    Tw = mod.Tweediemix
    for n in (20, 50):
        tw = Tw.__new__(Tw)
        nn.Module.__init__(tw)
        tw.scheduler = SimpleNamespace(alphas_cumprod=torch.cat([torch.tensor([1.0]), ac]),
                                       timesteps=leading_timesteps(n))
        tw.final_alpha_cumprod = ac[0]
        ts = tw.scheduler.timesteps
        skip = 1000 // n
        out[f"timesteps_{n}"] = ts.numpy()
        out[f"alpha_{n}"] = np.array([float(tw.alpha(t)) for t in ts], np.float32)
        out[f"alpha_next_{n}"] = np.array([float(tw.alpha(t - skip)) for t in ts], np.float32)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), **out)
    print("schedule.npz")


def gen_masks(mod):
    from PIL import Image
    out = {}
    fgs = {}
    for nm in ("a cat", "a dog"):
        p = os.path.join(REF, "example_results", "test_out", nm + ".jpg")
        arr = np.array(Image.open(p).convert("L"))
        # store the decoded 8-bit image bit-packed after the reference's own 0.5 threshold can't be
        # applied yet (we need the raw levels), so keep a compressed copy of the raw array
        out[f"img_{nm.replace(' ', '_')}"] = arr
        for hw in (128, 64):
            m = mod.preprocess_mask(p, hw, hw, "cpu")
            out[f"mask_{nm.replace(' ', '_')}_{hw}"] = m.numpy()
            fgs.setdefault(hw, []).append(m)
    for hw, lst in fgs.items():
        fg = torch.cat(lst)
        bg = 1 - torch.sum(fg, dim=0, keepdim=True)       # fusion_sampling.py:467-469
        bg[bg < 0] = 0
        out[f"masks_all_{hw}"] = torch.cat([fg, bg]).numpy()
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)
    print("masks.npz", {k: v.shape for k, v in out.items()})


def gen_attention(custom_mod, lora_mod):
    """sa_forward of utils_custom.py / utils_lora.py on one patched module, in/out of window."""
    K = 3
    out = {}
    for kind, umod in (("custom", custom_mod), ("lora", lora_mod)):
        lora = kind == "lora"
        holder = nn.Module()
        C, cross, heads = 128, 48, 2
        holder.unet = FakeUNet(C=C, cross=cross, heads=heads, seed=3, with_lora=lora)
        for i in range(K):
            setattr(holder, f"unet_{i}", FakeUNet(C=C, cross=cross, heads=heads, seed=20 + i, with_lora=lora))
        t_window = torch.tensor([781, 761, 741])
        umod.register_attention_control_efficient(holder, t_window, K)
        blk = holder.unet.down_blocks[1].attentions[0].transformer_blocks[1]
        gen = torch.Generator().manual_seed(5)
        S, L = 24, 7
        for which in ("attn1", "attn2"):
            mod = getattr(blk, which)
            patched = lora or which == "attn2"
            out[f"{kind}_{which}_wq"] = mod.to_q.weight.detach().numpy()
            out[f"{kind}_{which}_wk"] = mod.to_k.weight.detach().numpy()
            out[f"{kind}_{which}_wv"] = mod.to_v.weight.detach().numpy()
            out[f"{kind}_{which}_wo"] = mod.to_out[0].weight.detach().numpy()
            out[f"{kind}_{which}_bo"] = mod.to_out[0].bias.detach().numpy()
            if not patched:
                continue
            for i in range(K):
                if lora:
                    for nm in ("q", "k", "v", "out"):
                        l = getattr(mod, f"to_{nm}_{i}_lora")
                        out[f"{kind}_{which}_{nm}{i}_down"] = l.down.weight.detach().numpy()
                        out[f"{kind}_{which}_{nm}{i}_up"] = l.up.weight.detach().numpy()
                else:
                    out[f"{kind}_{which}_wk{i}"] = getattr(mod, f"to_k_{i}").weight.detach().numpy()
                    out[f"{kind}_{which}_wv{i}"] = getattr(mod, f"to_v_{i}").weight.detach().numpy()
            for B in (4, 2):
                x = torch.randn(B, S, C, generator=gen)
                ehs = torch.randn(B, L, cross, generator=gen) if which == "attn2" else None
                out[f"{kind}_{which}_B{B}_x"] = x.numpy()
                if ehs is not None:
                    out[f"{kind}_{which}_B{B}_ehs"] = ehs.numpy()
                for tag, t in (("in", 761), ("out", 801)):
                    umod.register_time(holder, t)
                    with torch.no_grad():
                        y = mod.forward(x, encoder_hidden_states=ehs)
                    out[f"{kind}_{which}_B{B}_{tag}_y"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "attention.npz"), **out)
    print("attention.npz", len(out), "arrays")


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    import fusion_sampling
    import fusion_sampling_lora
    import utils_custom
    import utils_lora

    gen_schedule(fusion_sampling)
    gen_masks(fusion_sampling)
    gen_attention(utils_custom, utils_lora)
    # full trajectories: every denoise_step branch (start+resampling / plain / jumping+mask / fusion / t==1)
    run_trajectory(fusion_sampling, "traj_custom_n50_f32.npz", 3, 50, 16, 16, {}, torch.float32, False, 11)
    run_trajectory(fusion_sampling, "traj_custom_n50_f16.npz", 3, 50, 16, 16, {}, torch.float16, False, 12)
    run_trajectory(fusion_sampling, "traj_custom_n20_f32.npz", 3, 20, 8, 12,
                   dict(resampling_steps=2, jumping_steps=2, guidance_scale=0.65, overlap_masks=True), torch.float32, False, 13)
    run_trajectory(fusion_sampling_lora, "traj_lora_n50_f32.npz", 3, 50, 16, 16, {}, torch.float32, True, 14)
    run_trajectory(fusion_sampling_lora, "traj_lora_n50_f16.npz", 3, 50, 8, 8,
                   dict(t_stop=0.9), torch.float16, True, 15)
    # K != 3 (two fg + ... ) : K=2 and K=4 trajectories (routing is hard-wired to batch==4 in the
    # reference hooks, but the sampler arithmetic is general in K)
    run_trajectory(fusion_sampling, "traj_custom_K2_n20_f32.npz", 2, 20, 8, 8,
                   dict(resampling_steps=1, jumping_steps=1), torch.float32, False, 16)
    run_trajectory(fusion_sampling, "traj_custom_K4_n20_f32.npz", 4, 20, 8, 8,
                   dict(resampling_steps=1, jumping_steps=0), torch.float32, False, 17)


if __name__ == "__main__":
    main()
