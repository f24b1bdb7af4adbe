"""CPU/fp32 oracle of the SDXL UNet forward (what `self.unet(...)` executes at
fusion_sampling.py:340/374/406/414/440) -- TEST INFRASTRUCTURE ONLY, parity UNPINNED.

The arithmetic lives in diffusers==0.29.2 (requirements.txt:4), which is neither vendored in the
reference nor installed here, and the reference holds no test or golden vector at this boundary.
This file restates the published SDXL-base architecture (UNet2DConditionModel config of
stabilityai/stable-diffusion-xl-base-1.0) with stock torch fp32 ops
(F.conv2d, F.group_norm, F.layer_norm, F.scaled_dot_product_attention) and diffusers-compatible
state-dict keys.  Anchors: (i) the parameter count of the full config is exactly 2,567,463,684
(tests/test_host_cpu.py::test_parameter_inventories), (ii) the attention hooks follow utils_custom.py:53-108 and
utils_lora.py:55-123, which ARE pinned by tests/golden/attention.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_layers: tuple = (0, 2, 10)       # 0 = block without attention
    head_dim: int = 64
    cross_dim: int = 2048
    pooled_dim: int = 1280
    addition_time_embed_dim: int = 256
    norm_groups: int = 32

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    @property
    def add_in_dim(self):
        return self.pooled_dim + 6 * self.addition_time_embed_dim


SDXL = UNetConfig()
TINY = UNetConfig(block_out_channels=(64, 128, 256), transformer_layers=(0, 1, 2), cross_dim=128,
                  pooled_dim=64, addition_time_embed_dim=32)


def param_shapes(cfg: UNetConfig) -> dict:
    """name -> shape, diffusers key scheme."""
    P = {}
    C0 = cfg.block_out_channels[0]
    T = cfg.time_embed_dim

    def lin(name, i, o, bias=True):
        P[name + ".weight"] = (o, i)
        if bias:
            P[name + ".bias"] = (o,)

    def conv(name, i, o, k=3):
        P[name + ".weight"] = (o, i, k, k)
        P[name + ".bias"] = (o,)

    def norm(name, c):
        P[name + ".weight"] = (c,)
        P[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", ci, co)
        lin(name + ".time_emb_proj", T, co)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co)
        if ci != co:
            conv(name + ".conv_shortcut", ci, co, 1)

    def t2d(name, c, n):
        norm(name + ".norm", c)
        lin(name + ".proj_in", c, c)
        for i in range(n):
            b = f"{name}.transformer_blocks.{i}"
            norm(b + ".norm1", c)
            for a, kd in (("attn1", c), ("attn2", cfg.cross_dim)):
                lin(f"{b}.{a}.to_q", c, c, False)
                lin(f"{b}.{a}.to_k", kd, c, False)
                lin(f"{b}.{a}.to_v", kd, c, False)
                lin(f"{b}.{a}.to_out.0", c, c)
            norm(b + ".norm2", c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        lin(name + ".proj_out", c, c)

    conv("conv_in", cfg.in_channels, C0)
    lin("time_embedding.linear_1", C0, T)
    lin("time_embedding.linear_2", T, T)
    lin("add_embedding.linear_1", cfg.add_in_dim, T)
    lin("add_embedding.linear_2", T, T)
    chans = cfg.block_out_channels
    nb = len(chans)
    skip = [C0]
    ci = C0
    for bi, co in enumerate(chans):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{bi}.resnets.{j}", ci, co)
            if cfg.transformer_layers[bi]:
                t2d(f"down_blocks.{bi}.attentions.{j}", co, cfg.transformer_layers[bi])
            ci = co
            skip.append(co)
        if bi < nb - 1:
            conv(f"down_blocks.{bi}.downsamplers.0.conv", co, co)
            skip.append(co)
    cm = chans[-1]
    resnet("mid_block.resnets.0", cm, cm)
    t2d("mid_block.attentions.0", cm, cfg.transformer_layers[-1])
    resnet("mid_block.resnets.1", cm, cm)
    for ui, co in enumerate(reversed(chans)):
        bi = nb - 1 - ui
        for j in range(cfg.layers_per_block + 1):
            resnet(f"up_blocks.{ui}.resnets.{j}", ci + skip.pop(), co)
            if cfg.transformer_layers[bi]:
                t2d(f"up_blocks.{ui}.attentions.{j}", co, cfg.transformer_layers[bi])
            ci = co
        if ui < nb - 1:
            conv(f"up_blocks.{ui}.upsamplers.0.conv", co, co)
    norm("conv_norm_out", C0)
    conv("conv_out", C0, cfg.out_channels)
    return P


def attention_prefixes(cfg: UNetConfig):
    """every transformer block prefix '<...>.transformer_blocks.i' in forward order."""
    return sorted({k.rsplit(".attn1", 1)[0] for k in param_shapes(cfg) if ".attn1.to_q" in k})


def timestep_embedding(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    arg = t[:, None].float() * torch.exp(exponent)[None]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


@dataclass
class Concepts:
    """per-concept weights borrowed by the attention hooks.
    kind 'custom': kv[prefix] = [(to_k_i, to_v_i)]            (utils_custom.py:120-128)
    kind 'lora'  : lora[prefix.attnN] = [{q,k,v,out: (down, up)}]   (utils_lora.py:135-170)"""
    kind: str = "none"
    kv: dict = field(default_factory=dict)
    lora: dict = field(default_factory=dict)


class UNetOracle:
    def __init__(self, cfg: UNetConfig, sd: dict, concepts: Concepts | None = None):
        self.cfg, self.sd = cfg, sd
        self.con = concepts or Concepts()

    # -------------------------------------------------------------- pieces
    def _lin(self, x, name):
        return F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"))

    def _gn(self, x, name, eps):
        return F.group_norm(x, self.cfg.norm_groups, self.sd[name + ".weight"], self.sd[name + ".bias"], eps)

    def _resnet(self, x, emb, name):
        sd = self.sd
        h = F.silu(self._gn(x, name + ".norm1", 1e-5))
        h = F.conv2d(h, sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], padding=1)
        h = h + self._lin(F.silu(emb), name + ".time_emb_proj")[:, :, None, None]
        h = F.silu(self._gn(h, name + ".norm2", 1e-5))
        h = F.conv2d(h, sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1)
        if name + ".conv_shortcut.weight" in sd:
            x = F.conv2d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
        return x + h

    def _attn(self, x, ehs, name, routed):
        """the patched forward of utils_custom.py:53-108 / utils_lora.py:55-123 (unpatched modules are
        plain attention, which both reduce to when not routed)."""
        sd, con = self.sd, self.con
        src = x if ehs is None else ehs
        B = x.shape[0]
        q, k, v = self._lin(x, name + ".to_q"), self._lin(src, name + ".to_k"), self._lin(src, name + ".to_v")
        on = routed and src.shape[0] == 4
        tb = name.rsplit(".", 1)[0]
        if on and con.kind == "custom" and ehs is not None:
            kv = con.kv[tb]
            k = torch.cat([k[:1]] + [F.linear(src[i + 1:i + 2], kv[i][0]) for i in range(len(kv))])
            v = torch.cat([v[:1]] + [F.linear(src[i + 1:i + 2], kv[i][1]) for i in range(len(kv))])
        lo = con.lora.get(name) if (on and con.kind == "lora") else None
        if lo is not None:
            q, k, v = q.clone(), k.clone(), v.clone()
            for i, l in enumerate(lo):
                q[i + 1] += F.linear(F.linear(x[i + 1], l["q"][0]), l["q"][1])
                k[i + 1] += F.linear(F.linear(src[i + 1], l["k"][0]), l["k"][1])
                v[i + 1] += F.linear(F.linear(src[i + 1], l["v"][0]), l["v"][1])
        H = x.shape[-1] // self.cfg.head_dim

        def heads(t):
            return t.reshape(B, -1, H, self.cfg.head_dim).transpose(1, 2)
        o = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), scale=self.cfg.head_dim ** -0.5)
        o = o.transpose(1, 2).reshape(B, -1, x.shape[-1])
        out = self._lin(o, name + ".to_out.0")
        if lo is not None:
            out = out.clone()
            for i, l in enumerate(lo):
                out[i + 1] += F.linear(F.linear(o[i + 1], l["out"][0]), l["out"][1])
        return out

    def _t2d(self, x, ehs, name, n, routed):
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, name + ".norm", 1e-6)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self._lin(h, name + ".proj_in")
        for i in range(n):
            b = f"{name}.transformer_blocks.{i}"
            sd = self.sd

            def ln(t, nm):
                return F.layer_norm(t, (C,), sd[nm + ".weight"], sd[nm + ".bias"], 1e-5)
            h = h + self._attn(ln(h, b + ".norm1"), None, b + ".attn1", routed)
            h = h + self._attn(ln(h, b + ".norm2"), ehs, b + ".attn2", routed)
            y = self._lin(ln(h, b + ".norm3"), b + ".ff.net.0.proj")
            a, g = y.chunk(2, dim=-1)
            h = h + self._lin(a * F.gelu(g), b + ".ff.net.2")
        h = self._lin(h, name + ".proj_out")
        return h.reshape(B, H, W, C).permute(0, 3, 1, 2) + res

    # -------------------------------------------------------------- forward
    @torch.no_grad()
    def forward(self, sample, t, ehs, pooled, time_ids, routed=False):
        """sample [B,4,h,w], t scalar, ehs [B,77,cross], pooled [B,P], time_ids [B,6] -> eps [B,4,h,w]."""
        cfg, sd = self.cfg, self.sd
        B = sample.shape[0]
        C0 = cfg.block_out_channels[0]
        tt = torch.full((B,), float(t), dtype=torch.float32, device=sample.device)
        emb = self._lin(F.silu(self._lin(timestep_embedding(tt, C0), "time_embedding.linear_1")), "time_embedding.linear_2")
        tid = timestep_embedding(time_ids.reshape(-1).float(), cfg.addition_time_embed_dim).reshape(B, -1)
        add = torch.cat([pooled.float(), tid], dim=-1)
        emb = emb + self._lin(F.silu(self._lin(add, "add_embedding.linear_1")), "add_embedding.linear_2")

        x = F.conv2d(sample.float(), sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
        skips = [x]
        nb = len(cfg.block_out_channels)
        for bi in range(nb):
            for j in range(cfg.layers_per_block):
                x = self._resnet(x, emb, f"down_blocks.{bi}.resnets.{j}")
                if cfg.transformer_layers[bi]:
                    x = self._t2d(x, ehs, f"down_blocks.{bi}.attentions.{j}", cfg.transformer_layers[bi], routed)
                skips.append(x)
            if bi < nb - 1:
                n = f"down_blocks.{bi}.downsamplers.0.conv"
                x = F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], stride=2, padding=1)
                skips.append(x)
        x = self._resnet(x, emb, "mid_block.resnets.0")
        x = self._t2d(x, ehs, "mid_block.attentions.0", cfg.transformer_layers[-1], routed)
        x = self._resnet(x, emb, "mid_block.resnets.1")
        for ui in range(nb):
            bi = nb - 1 - ui
            for j in range(cfg.layers_per_block + 1):
                x = torch.cat([x, skips.pop()], dim=1)
                x = self._resnet(x, emb, f"up_blocks.{ui}.resnets.{j}")
                if cfg.transformer_layers[bi]:
                    x = self._t2d(x, ehs, f"up_blocks.{ui}.attentions.{j}", cfg.transformer_layers[bi], routed)
            if ui < nb - 1:
                n = f"up_blocks.{ui}.upsamplers.0.conv"
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = F.conv2d(x, sd[n + ".weight"], sd[n + ".bias"], padding=1)
        x = F.silu(self._gn(x, "conv_norm_out", 1e-5))
        return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
