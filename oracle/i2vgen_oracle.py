"""TEST INFRASTRUCTURE ONLY: fp32 torch restatement of diffusers' `I2VGenXLUNet` forward (the network the reference's video
pipeline calls at video_gen/pipeline_i2vgen_xl.py:688-697) and of its parameter inventory.

PARITY UNPINNED: the class lives in the third-party dependency diffusers (pinned diffusers==0.29.2, requirements.txt:4:
models/unets/unet_i2vgen_xl.py, unet_3d_blocks.py, models/transformers/transformer_temporal.py, models/resnet.py
TemporalConvLayer), which is neither vendored in /root/reference nor installed here, and no checkpoint exists offline.  This
file restates the published architecture from its documented structure: channels (320, 640, 1280, 1280), 2 layers per block,
CrossAttnDownBlock3D x3 + DownBlock3D / UpBlock3D + CrossAttnUpBlock3D x3, every ResnetBlock2D followed by a
TemporalConvLayer (4 x [GroupNorm, SiLU, Conv3d (3,1,1)], residual), every Transformer2DModel (1 layer, heads = C/64, cross
dim 1024, linear projections) followed by a TransformerTemporalModel (1 layer over the frame axis, both attentions self),
`transformer_in` (8 heads x 64 over 320 channels), image-latent / context / fps embeddings.  Only tests/ may import it."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class I2VConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280, 1280)
    attn_levels: tuple = (True, True, True, False)          # CrossAttn*Block3D vs plain
    layers_per_block: int = 2
    groups: int = 32
    cross_dim: int = 1024
    head_dim: int = 64
    transformer_in_heads: int = 8
    ctx_pool: int = 32                                      # AdaptiveAvgPool2d target of the image-latent context branch

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


FULL = I2VConfig()
TINY = I2VConfig(block_out_channels=(64, 128, 128, 128), cross_dim=128, transformer_in_heads=2, ctx_pool=8)


# ------------------------------------------------------------------------------------------- parameter inventory
def param_shapes(cfg: I2VConfig) -> dict:
    s = {}
    ch, T, ic, cd = cfg.block_out_channels, cfg.time_embed_dim, cfg.in_channels, cfg.cross_dim

    def conv(n, i, o, k=3):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k, k), (o,)

    def lin(n, i, o, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def vec2(n, c):
        s[n + ".weight"], s[n + ".bias"] = (c,), (c,)

    def resnet(n, ci, co):
        vec2(n + ".norm1", ci); conv(n + ".conv1", ci, co); lin(n + ".time_emb_proj", T, co)
        vec2(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    def temp_conv(n, c):
        for k, idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
            vec2(f"{n}.conv{k}.0", c)
            s[f"{n}.conv{k}.{idx}.weight"], s[f"{n}.conv{k}.{idx}.bias"] = (c, c, 3, 1, 1), (c,)

    def block(n, dim, cross):
        vec2(n + ".norm1", dim); vec2(n + ".norm2", dim); vec2(n + ".norm3", dim)
        for a, kd in (("attn1", dim), ("attn2", cross)):
            lin(f"{n}.{a}.to_q", dim, dim, False); lin(f"{n}.{a}.to_k", kd, dim, False); lin(f"{n}.{a}.to_v", kd, dim, False)
            lin(f"{n}.{a}.to_out.0", dim, dim)
        lin(n + ".ff.net.0.proj", dim, 8 * dim); lin(n + ".ff.net.2", 4 * dim, dim)

    def t2d(n, c):
        vec2(n + ".norm", c); lin(n + ".proj_in", c, c); block(n + ".transformer_blocks.0", c, cd); lin(n + ".proj_out", c, c)

    def ttemp(n, c, inner):
        vec2(n + ".norm", c); lin(n + ".proj_in", c, inner); block(n + ".transformer_blocks.0", inner, inner); lin(n + ".proj_out", inner, c)

    conv("conv_in", 2 * ic, ch[0])
    ttemp("transformer_in", ch[0], cfg.transformer_in_heads * cfg.head_dim)
    conv("image_latents_proj_in.0", 4, ic * 4); conv("image_latents_proj_in.2", ic * 4, ic * 4); conv("image_latents_proj_in.4", ic * 4, ic)
    e = "image_latents_temporal_encoder"
    vec2(e + ".norm1", ic)
    lin(e + ".attn1.to_q", ic, 2 * ic, False); lin(e + ".attn1.to_k", ic, 2 * ic, False); lin(e + ".attn1.to_v", ic, 2 * ic, False)
    lin(e + ".attn1.to_out.0", 2 * ic, ic)
    lin(e + ".ff.net.0.proj", ic, ic * 4); lin(e + ".ff.net.2", ic * 4, ic)
    conv("image_latents_context_embedding.0", 4, ic * 8); conv("image_latents_context_embedding.3", ic * 8, ic * 16)
    conv("image_latents_context_embedding.5", ic * 16, cd)
    lin("time_embedding.linear_1", ch[0], T); lin("time_embedding.linear_2", T, T)
    lin("context_embedding.0", cd, T); lin("context_embedding.2", T, cd * ic)
    lin("fps_embedding.0", ch[0], T); lin("fps_embedding.2", T, T)
    nb = len(ch)
    ci = ch[0]
    skips = [ch[0]]
    for bi, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{bi}.resnets.{j}", ci, co); temp_conv(f"down_blocks.{bi}.temp_convs.{j}", co)
            if cfg.attn_levels[bi]:
                t2d(f"down_blocks.{bi}.attentions.{j}", co); ttemp(f"down_blocks.{bi}.temp_attentions.{j}", co, co)
            ci = co
            skips.append(co)
        if bi < nb - 1:
            conv(f"down_blocks.{bi}.downsamplers.0.conv", co, co)
            skips.append(co)
    cm = ch[-1]
    resnet("mid_block.resnets.0", cm, cm); temp_conv("mid_block.temp_convs.0", cm)
    t2d("mid_block.attentions.0", cm); ttemp("mid_block.temp_attentions.0", cm, cm)
    resnet("mid_block.resnets.1", cm, cm); temp_conv("mid_block.temp_convs.1", cm)
    for ui in range(nb):
        bi = nb - 1 - ui
        co = ch[bi]
        for j in range(cfg.layers_per_block + 1):
            cs = skips.pop()
            resnet(f"up_blocks.{ui}.resnets.{j}", ci + cs, co); temp_conv(f"up_blocks.{ui}.temp_convs.{j}", co)
            if cfg.attn_levels[bi]:
                t2d(f"up_blocks.{ui}.attentions.{j}", co); ttemp(f"up_blocks.{ui}.temp_attentions.{j}", co, co)
            ci = co
        if ui < nb - 1:
            conv(f"up_blocks.{ui}.upsamplers.0.conv", co, co)
    vec2("conv_norm_out", ch[0]); conv("conv_out", ch[0], cfg.out_channels)
    return s


def synthetic_state_dict(cfg: I2VConfig, seed=99, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in param_shapes(cfg).items():
        if name.endswith(".bias"):
            v = torch.randn(shp, generator=g) * 0.02
        elif len(shp) == 1:
            v = 1 + torch.randn(shp, generator=g) * 0.05
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * fan_in ** -0.5
            if name.endswith("conv4.3.weight"):
                v = v * 0.3                                   # (zero-init in diffusers; small but non-zero so tests exercise it)
        sd[name] = v.to(dtype)
    return sd


# ------------------------------------------------------------------------------------------- forward
def _timesteps(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    freqs = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _attn(x, ctx, p, n, heads):
    B, S, C = x.shape
    q = F.linear(x, p[n + ".to_q.weight"]).view(B, S, heads, -1).transpose(1, 2)
    k = F.linear(ctx, p[n + ".to_k.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    v = F.linear(ctx, p[n + ".to_v.weight"]).view(B, ctx.shape[1], heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, -1)
    return F.linear(o, p[n + ".to_out.0.weight"], p[n + ".to_out.0.bias"])


def _block(x, ctx, p, n, heads, double_self):
    C = x.shape[-1]
    ln = lambda v, k: F.layer_norm(v, (C,), p[f"{n}.{k}.weight"], p[f"{n}.{k}.bias"], 1e-5)
    x = x + _attn(ln(x, "norm1"), ln(x, "norm1"), p, n + ".attn1", heads)
    h = ln(x, "norm2")
    x = x + _attn(h, h if double_self else ctx, p, n + ".attn2", heads)
    h = F.linear(ln(x, "norm3"), p[n + ".ff.net.0.proj.weight"], p[n + ".ff.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    return x + F.linear(a * F.gelu(gate), p[n + ".ff.net.2.weight"], p[n + ".ff.net.2.bias"])


def _resnet(x, temb, p, n, groups):
    h = F.silu(F.group_norm(x, groups, p[n + ".norm1.weight"], p[n + ".norm1.bias"], 1e-5))
    h = F.conv2d(h, p[n + ".conv1.weight"], p[n + ".conv1.bias"], padding=1)
    h = h + F.linear(F.silu(temb), p[n + ".time_emb_proj.weight"], p[n + ".time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, p[n + ".norm2.weight"], p[n + ".norm2.bias"], 1e-5))
    h = F.conv2d(h, p[n + ".conv2.weight"], p[n + ".conv2.bias"], padding=1)
    if n + ".conv_shortcut.weight" in p:
        x = F.conv2d(x, p[n + ".conv_shortcut.weight"], p[n + ".conv_shortcut.bias"])
    return x + h


def _temp_conv(x, p, n, frames, groups):
    bf, c, h, w = x.shape
    v = x.view(bf // frames, frames, c, h, w).permute(0, 2, 1, 3, 4)
    idn = v
    for k, idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        v = F.silu(F.group_norm(v, groups, p[f"{n}.conv{k}.0.weight"], p[f"{n}.conv{k}.0.bias"], 1e-5))
        v = F.conv3d(v, p[f"{n}.conv{k}.{idx}.weight"], p[f"{n}.conv{k}.{idx}.bias"], padding=(1, 0, 0))
    return (idn + v).permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


def _t2d(x, ctx, p, n, cfg):
    bf, c, h, w = x.shape
    v = F.group_norm(x, cfg.groups, p[n + ".norm.weight"], p[n + ".norm.bias"], 1e-6)
    v = v.permute(0, 2, 3, 1).reshape(bf, h * w, c)
    v = F.linear(v, p[n + ".proj_in.weight"], p[n + ".proj_in.bias"])
    v = _block(v, ctx, p, n + ".transformer_blocks.0", c // cfg.head_dim, False)
    v = F.linear(v, p[n + ".proj_out.weight"], p[n + ".proj_out.bias"])
    return v.view(bf, h, w, c).permute(0, 3, 1, 2) + x


def _ttemp(x, p, n, frames, heads, cfg):
    bf, c, h, w = x.shape
    b = bf // frames
    v = x.view(b, frames, c, h, w).permute(0, 2, 1, 3, 4)
    v = F.group_norm(v, cfg.groups, p[n + ".norm.weight"], p[n + ".norm.bias"], 1e-6)
    v = v.permute(0, 3, 4, 2, 1).reshape(b * h * w, frames, c)
    v = F.linear(v, p[n + ".proj_in.weight"], p[n + ".proj_in.bias"])
    v = _block(v, None, p, n + ".transformer_blocks.0", heads, True)
    v = F.linear(v, p[n + ".proj_out.weight"], p[n + ".proj_out.bias"])
    v = v.view(b, h, w, frames, c).permute(0, 3, 4, 1, 2).reshape(bf, c, h, w)
    return v + x


def conditioning(p, cfg, fps, image_latents, image_embeddings, encoder_hidden_states):
    """everything of I2VGenXLUNet.forward that does not depend on the sample or the timestep (constant over the loop):
    returns (fps_emb [B,T], context_emb [B, 77 + pool^2/16 + in_channels, cross], image-latent features [B,C,F,h,w])."""
    B, C, Fr, H, W = image_latents.shape
    fps_emb = F.linear(F.silu(F.linear(_timesteps(fps, cfg.block_out_channels[0]), p["fps_embedding.0.weight"], p["fps_embedding.0.bias"])),
                       p["fps_embedding.2.weight"], p["fps_embedding.2.bias"])
    n = "image_latents_context_embedding"
    v = image_latents[:, :, 0]
    v = F.silu(F.conv2d(v, p[n + ".0.weight"], p[n + ".0.bias"], padding=1))
    v = F.adaptive_avg_pool2d(v, (cfg.ctx_pool, cfg.ctx_pool))
    v = F.silu(F.conv2d(v, p[n + ".3.weight"], p[n + ".3.bias"], stride=2, padding=1))
    v = F.conv2d(v, p[n + ".5.weight"], p[n + ".5.bias"], stride=2, padding=1)
    ctx_img = v.permute(0, 2, 3, 1).reshape(B, -1, cfg.cross_dim)
    e = F.linear(F.silu(F.linear(image_embeddings, p["context_embedding.0.weight"], p["context_embedding.0.bias"])),
                 p["context_embedding.2.weight"], p["context_embedding.2.bias"]).view(B, cfg.in_channels, cfg.cross_dim)
    context = torch.cat([encoder_hidden_states, ctx_img, e], dim=1)
    n = "image_latents_proj_in"
    il = image_latents.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    il = F.silu(F.conv2d(il, p[n + ".0.weight"], p[n + ".0.bias"], padding=1))
    il = F.silu(F.conv2d(il, p[n + ".2.weight"], p[n + ".2.bias"], padding=1))
    il = F.conv2d(il, p[n + ".4.weight"], p[n + ".4.bias"], padding=1)
    il = il.view(B, Fr, C, H, W).permute(0, 3, 4, 1, 2).reshape(B * H * W, Fr, C)
    n = "image_latents_temporal_encoder"
    h = F.layer_norm(il, (C,), p[n + ".norm1.weight"], p[n + ".norm1.bias"], 1e-5)
    il = il + _attn(h, h, p, n + ".attn1", 2)
    ff = F.linear(F.gelu(F.linear(il, p[n + ".ff.net.0.proj.weight"], p[n + ".ff.net.0.proj.bias"])), p[n + ".ff.net.2.weight"], p[n + ".ff.net.2.bias"])
    il = (il + ff).view(B, H, W, Fr, C).permute(0, 4, 3, 1, 2)
    return fps_emb, context, il


def forward(p, cfg: I2VConfig, sample, t, fps_emb, context, il_feat):
    """sample [B,C,F,h,w], t scalar -> prediction [B,C,F,h,w]; (fps_emb, context, il_feat) from conditioning()."""
    p = {k: v.float() for k, v in p.items()}
    B, C, Fr, H, W = sample.shape
    ch, nb = cfg.block_out_channels, len(cfg.block_out_channels)
    te = _timesteps(torch.full((B,), float(t), device=sample.device), ch[0])
    emb = F.linear(F.silu(F.linear(te, p["time_embedding.linear_1.weight"], p["time_embedding.linear_1.bias"])),
                   p["time_embedding.linear_2.weight"], p["time_embedding.linear_2.bias"]) + fps_emb
    emb = emb.repeat_interleave(Fr, dim=0)
    ctx = context.repeat_interleave(Fr, dim=0)
    x = torch.cat([sample, il_feat], dim=1).permute(0, 2, 1, 3, 4).reshape(B * Fr, 2 * C, H, W)
    x = F.conv2d(x, p["conv_in.weight"], p["conv_in.bias"], padding=1)
    x = _ttemp(x, p, "transformer_in", Fr, cfg.transformer_in_heads, cfg)
    skips = [x]
    for bi, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            x = _resnet(x, emb, p, f"down_blocks.{bi}.resnets.{j}", cfg.groups)
            x = _temp_conv(x, p, f"down_blocks.{bi}.temp_convs.{j}", Fr, cfg.groups)
            if cfg.attn_levels[bi]:
                x = _t2d(x, ctx, p, f"down_blocks.{bi}.attentions.{j}", cfg)
                x = _ttemp(x, p, f"down_blocks.{bi}.temp_attentions.{j}", Fr, co // cfg.head_dim, cfg)
            skips.append(x)
        if bi < nb - 1:
            n = f"down_blocks.{bi}.downsamplers.0.conv"
            x = F.conv2d(x, p[n + ".weight"], p[n + ".bias"], stride=2, padding=1)
            skips.append(x)
    cm = ch[-1]
    x = _resnet(x, emb, p, "mid_block.resnets.0", cfg.groups)
    x = _temp_conv(x, p, "mid_block.temp_convs.0", Fr, cfg.groups)
    x = _t2d(x, ctx, p, "mid_block.attentions.0", cfg)
    x = _ttemp(x, p, "mid_block.temp_attentions.0", Fr, cm // cfg.head_dim, cfg)
    x = _resnet(x, emb, p, "mid_block.resnets.1", cfg.groups)
    x = _temp_conv(x, p, "mid_block.temp_convs.1", Fr, cfg.groups)
    for ui in range(nb):
        bi = nb - 1 - ui
        co = ch[bi]
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = _resnet(x, emb, p, f"up_blocks.{ui}.resnets.{j}", cfg.groups)
            x = _temp_conv(x, p, f"up_blocks.{ui}.temp_convs.{j}", Fr, cfg.groups)
            if cfg.attn_levels[bi]:
                x = _t2d(x, ctx, p, f"up_blocks.{ui}.attentions.{j}", cfg)
                x = _ttemp(x, p, f"up_blocks.{ui}.temp_attentions.{j}", Fr, co // cfg.head_dim, cfg)
        if ui < nb - 1:
            n = f"up_blocks.{ui}.upsamplers.0.conv"
            x = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), p[n + ".weight"], p[n + ".bias"], padding=1)
    x = F.silu(F.group_norm(x, cfg.groups, p["conv_norm_out.weight"], p["conv_norm_out.bias"], 1e-5))
    x = F.conv2d(x, p["conv_out.weight"], p["conv_out.bias"], padding=1)
    return x.view(B, Fr, cfg.out_channels, H, W).permute(0, 2, 1, 3, 4)
