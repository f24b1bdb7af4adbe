"""fp32 torch oracle of the SDXL VAE decoder (AutoencoderKL.decode as called at fusion_sampling.py:297-303 and
:496-528) -- TEST INFRASTRUCTURE ONLY, parity UNPINNED (diffusers==0.29.2 is not vendored / installed).
Restates the published architecture (madebyollin/sdxl-vae-fp16-fix config: block_out_channels (128,256,512,512),
layers_per_block 2, norm groups 32, eps 1e-6, one single-head mid-block attention) with stock torch ops and
diffusers parameter names."""
from __future__ import annotations

import torch
import torch.nn.functional as F

FULL = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3, groups=32)
TINY = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=4, out_channels=3, groups=32)


def param_shapes(cfg) -> dict:
    P = {}
    ch = list(reversed(cfg["block_out_channels"]))
    lc = cfg["latent_channels"]

    def conv(n, i, o, k=3):
        P[n + ".weight"] = (o, i, k, k)
        P[n + ".bias"] = (o,)

    def norm(n, c):
        P[n + ".weight"] = (c,)
        P[n + ".bias"] = (c,)

    def lin(n, i, o):
        P[n + ".weight"] = (o, i)
        P[n + ".bias"] = (o,)

    def resnet(n, ci, co):
        norm(n + ".norm1", ci); conv(n + ".conv1", ci, co); norm(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    conv("post_quant_conv", lc, lc, 1)
    conv("decoder.conv_in", lc, ch[0])
    resnet("decoder.mid_block.resnets.0", ch[0], ch[0])
    norm("decoder.mid_block.attentions.0.group_norm", ch[0])
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin("decoder.mid_block.attentions.0." + nm, ch[0], ch[0])
    resnet("decoder.mid_block.resnets.1", ch[0], ch[0])
    ci = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i < len(ch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
    norm("decoder.conv_norm_out", ch[-1])
    conv("decoder.conv_out", ch[-1], cfg["out_channels"])
    return P


class VAEDecoderOracle:
    def __init__(self, cfg, sd):
        self.cfg, self.sd = cfg, sd

    def _gn(self, x, n):
        return F.group_norm(x, self.cfg["groups"], self.sd[n + ".weight"], self.sd[n + ".bias"], 1e-6)

    def _conv(self, x, n, **kw):
        return F.conv2d(x, self.sd[n + ".weight"], self.sd[n + ".bias"], **kw)

    def _resnet(self, x, n):
        h = self._conv(F.silu(self._gn(x, n + ".norm1")), n + ".conv1", padding=1)
        h = self._conv(F.silu(self._gn(h, n + ".norm2")), n + ".conv2", padding=1)
        if n + ".conv_shortcut.weight" in self.sd:
            x = self._conv(x, n + ".conv_shortcut")
        return x + h

    def _attn(self, x, n):
        B, C, H, W = x.shape
        h = self._gn(x, n + ".group_norm").reshape(B, C, H * W).transpose(1, 2)
        lin = lambda t, m: F.linear(t, self.sd[f"{n}.{m}.weight"], self.sd[f"{n}.{m}.bias"])
        q, k, v = lin(h, "to_q"), lin(h, "to_k"), lin(h, "to_v")
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None], scale=C ** -0.5)[:, 0]
        o = lin(o, "to_out.0")
        return x + o.transpose(1, 2).reshape(B, C, H, W)

    @torch.no_grad()
    def decode(self, latent, inv_scale):
        """latent [B,4,h,w]; returns (img/2+0.5).clamp(0,1) as [B,3,8h,8w] (decode_latent / final decode)."""
        cfg = self.cfg
        x = self._conv(latent.float() * inv_scale, "post_quant_conv")
        x = self._conv(x, "decoder.conv_in", padding=1)
        x = self._resnet(x, "decoder.mid_block.resnets.0")
        x = self._attn(x, "decoder.mid_block.attentions.0")
        x = self._resnet(x, "decoder.mid_block.resnets.1")
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"] + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
            if i < nb - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", padding=1)
        x = F.silu(self._gn(x, "decoder.conv_norm_out"))
        img = self._conv(x, "decoder.conv_out", padding=1)
        return (img / 2 + 0.5).clamp(0, 1)

    @torch.no_grad()
    def encode(self, image):
        """AutoencoderKL.encode up to the diagonal Gaussian's parameters: image [B,3,H,W] in [-1,1] -> (mean, logvar) [B,4,H/8,W/8]
        (diffusers Encoder: conv_in, DownEncoderBlock2D x4 with Downsample2D(padding=0) = F.pad (0,1,0,1) + stride-2 conv, mid block,
        conv_norm_out + SiLU + conv_out, then quant_conv; DiagonalGaussianDistribution clamps logvar to [-30, 20]).  Unpinned like
        decode (no diffusers here); the inventory has the published 34,163,592 encoder parameters."""
        cfg = self.cfg
        x = self._conv(image.float(), "encoder.conv_in", padding=1)
        nb = len(cfg["block_out_channels"])
        for i in range(nb):
            for j in range(cfg["layers_per_block"]):
                x = self._resnet(x, f"encoder.down_blocks.{i}.resnets.{j}")
            if i < nb - 1:
                x = self._conv(F.pad(x, (0, 1, 0, 1)), f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2)
        x = self._resnet(x, "encoder.mid_block.resnets.0")
        x = self._attn(x, "encoder.mid_block.attentions.0")
        x = self._resnet(x, "encoder.mid_block.resnets.1")
        x = self._conv(F.silu(self._gn(x, "encoder.conv_norm_out")), "encoder.conv_out", padding=1)
        m = self._conv(x, "quant_conv")
        lc = cfg["latent_channels"]
        return m[:, :lc], m[:, lc:].clamp(-30.0, 20.0)
