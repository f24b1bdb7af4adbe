"""SDXL VAE decoder as a launch plan over the C ABI ("next" row 2 of SURVEY 8f).

What `self.vae.decode(latent / scaling_factor)` + `(img/2+0.5).clamp(0,1)` compute at fusion_sampling.py:297-303
(preview, with the reference's 1/0.18215 quirk) and :496-528 (final image, 1/0.13025): diffusers AutoencoderKL
decoder (block_out_channels (128,256,512,512), 3 resnets per up block, one single-head mid-block attention of
dim 512, GroupNorm(32, eps 1e-6)).  Reuses the UNet's kernels: tmix_conv3x3_nhwc (incl. the fused nearest-x2
upsample), tmix_groupnorm_nhwc, tmix_gemm_bf16 (1x1 shortcuts, q/k/v/out projections, and the attention itself as
QK^T -> fp32 scores -> tmix_softmax_rows -> PV, since the head dimension is 512), tmix_conv_out.
The 1/scaling_factor and post_quant_conv are folded into a per-pixel 4x4 map inside tmix_conv_in_pre.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from . import ops
from .unet import _Arena

BF16, F32 = torch.bfloat16, torch.float32

FULL = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3, groups=32)
TINY = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, latent_channels=4, out_channels=3, groups=32)


def param_shapes(cfg) -> dict:
    P = {}
    ch = list(reversed(cfg["block_out_channels"]))
    lc = cfg["latent_channels"]

    def conv(n, i, o, k=3):
        P[n + ".weight"] = (o, i, k, k)
        P[n + ".bias"] = (o,)

    def vec2(n, c):
        P[n + ".weight"] = (c,)
        P[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        vec2(n + ".norm1", ci); conv(n + ".conv1", ci, co); vec2(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    conv("post_quant_conv", lc, lc, 1)
    conv("decoder.conv_in", lc, ch[0])
    resnet("decoder.mid_block.resnets.0", ch[0], ch[0])
    vec2("decoder.mid_block.attentions.0.group_norm", ch[0])
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        P[f"decoder.mid_block.attentions.0.{nm}.weight"] = (ch[0], ch[0])
        P[f"decoder.mid_block.attentions.0.{nm}.bias"] = (ch[0],)
    resnet("decoder.mid_block.resnets.1", ch[0], ch[0])
    ci = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i < len(ch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co)
    vec2("decoder.conv_norm_out", ch[-1])
    conv("decoder.conv_out", ch[-1], cfg["out_channels"])
    return P


def encoder_param_shapes(cfg) -> dict:
    """AutoencoderKL encoder + quant_conv (diffusers Encoder / DownEncoderBlock2D; FULL: 34,163,592 + 72 parameters)."""
    P = {}
    ch = list(cfg["block_out_channels"])
    lc = cfg["latent_channels"]

    def conv(n, i, o, k=3):
        P[n + ".weight"] = (o, i, k, k)
        P[n + ".bias"] = (o,)

    def vec2(n, c):
        P[n + ".weight"] = (c,)
        P[n + ".bias"] = (c,)

    def resnet(n, ci, co):
        vec2(n + ".norm1", ci); conv(n + ".conv1", ci, co); vec2(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    conv("encoder.conv_in", cfg["out_channels"], ch[0])
    ci = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", ci, co)
            ci = co
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co)
    resnet("encoder.mid_block.resnets.0", ci, ci)
    vec2("encoder.mid_block.attentions.0.group_norm", ci)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        P[f"encoder.mid_block.attentions.0.{nm}.weight"] = (ci, ci)
        P[f"encoder.mid_block.attentions.0.{nm}.bias"] = (ci,)
    resnet("encoder.mid_block.resnets.1", ci, ci)
    vec2("encoder.conv_norm_out", ci)
    conv("encoder.conv_out", ci, 2 * lc)
    conv("quant_conv", 2 * lc, 2 * lc, 1)
    return P


def synthetic_state_dict(cfg, seed=4321, device="cpu", nontrivial=False, encoder=False):
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in (encoder_param_shapes(cfg) if encoder else param_shapes(cfg)).items():
        is_norm = "norm" in name
        if name.endswith(".bias"):
            v = torch.randn(shape, generator=gen, device=device) * 0.05 if nontrivial else torch.zeros(shape, device=device)
        elif is_norm:
            v = torch.ones(shape, device=device) + (0.1 * torch.randn(shape, generator=gen, device=device) if nontrivial else 0)
        else:
            fan = 1
            for s_ in shape[1:]:
                fan *= s_
            v = torch.randn(shape, generator=gen, device=device) * fan ** -0.5
            if ".conv2." in name or ".to_out.0." in name:
                v = v * 0.3
        sd[name] = v.to(BF16).float() if not name.startswith(("post_quant_conv", "decoder.conv_in", "quant_conv", "encoder.conv_in")) else v.float()
    return sd


class VAEDecoderPlan:
    """decode(latent[B,4,h,w] fp32) -> image [B,3,8h,8w] fp32 in [0,1]."""

    def __init__(self, cfg, sd, B, h, w, inv_scale, device="cuda"):
        self.cfg, self.B, self.h, self.w = cfg, B, h, w
        self.dev = torch.device(device)
        self.lib = L.load()
        self.ops, self.keep = [], []
        self.arena = _Arena(self.dev)
        self.flops = 0
        dev = self.dev
        t = {}
        for k, v in sd.items():
            v = v.to(dev)
            if k.endswith(".bias") or "norm" in k:
                t[k] = v.to(F32).contiguous()
            elif v.dim() == 4 and v.shape[-1] == 3 and not k.startswith("decoder.conv_in"):
                t[k] = v.permute(0, 2, 3, 1).to(BF16).contiguous()
            elif v.dim() == 4 and v.shape[-1] == 1 and not k.startswith("post_quant"):
                t[k] = v.reshape(v.shape[0], v.shape[1]).to(BF16).contiguous()
            elif v.dim() == 2:
                t[k] = v.to(BF16).contiguous()
        t["decoder.conv_in.weight"] = sd["decoder.conv_in.weight"].to(dev, F32).permute(0, 2, 3, 1).contiguous()
        a = "decoder.mid_block.attentions.0"
        t[a + ".qkv"] = torch.cat([t[a + ".to_q.weight"], t[a + ".to_k.weight"], t[a + ".to_v.weight"]]).contiguous()
        t[a + ".qkv.bias"] = torch.cat([t[a + ".to_q.bias"], t[a + ".to_k.bias"], t[a + ".to_v.bias"]]).contiguous()
        self.t = t
        # z -> post_quant_conv(z * inv_scale): per-pixel 4x4 map (host floats, passed by value to the kernel)
        pq = sd["post_quant_conv.weight"].float().reshape(4, 4) * float(inv_scale)
        self._pre_w = (C.c_float * 16)(*[float(x) for x in pq.reshape(-1)])
        self._pre_b = (C.c_float * 4)(*[float(x) for x in sd["post_quant_conv.bias"].float()])
        self.latent = torch.zeros(B, 4, h, w, device=dev, dtype=F32)
        self.image = torch.zeros(B, cfg["out_channels"], 8 * h, 8 * w, device=dev, dtype=F32)
        self._gn_ws = ops.groupnorm_ws(B, 4096, cfg["groups"], dev)
        self._build()

    def _emit(self, fn, *a):
        self.ops.append((fn, a))

    def _gn(self, x, Cc, HW, name, silu):
        out = self.arena.get(self.B, HW, Cc)
        self._emit(self.lib.tmix_groupnorm_nhwc, x.data_ptr(), Cc, None, 0, out.data_ptr(), self.t[name + ".weight"].data_ptr(),
                   self.t[name + ".bias"].data_ptr(), self._gn_ws.data_ptr(), self.B, HW, self.cfg["groups"], 1e-6, int(silu))
        return out

    def _conv(self, x, name, Hh, Ww, Ci, Co, mode=L.CONV_S1, residual=None):
        Ho, Wo = ops.conv_out_hw(Hh, Ww, mode)
        out = self.arena.get(self.B, Ho * Wo, Co)
        d = ops.make_conv_desc(x.view(self.B, Hh, Ww, Ci), self.t[name + ".weight"], out.view(self.B, Ho, Wo, Co),
                               self.t[name + ".bias"], None, residual, mode)
        self.keep.append(d)
        self._emit(self.lib.tmix_conv3x3_nhwc, C.byref(d))
        self.flops += 2 * self.B * Ho * Wo * Co * 9 * Ci
        return out

    def _gemm(self, a, w, out, **kw):
        d = ops.make_gemm_desc(a, w, out, **kw)
        self.keep.append(d)
        self._emit(self.lib.tmix_gemm_bf16, C.byref(d))
        self.flops += 2 * d.M * d.N * d.K * d.batch
        return d

    def _resnet(self, x, Ci, Co, Hh, Ww, name):
        A, B = self.arena, self.B
        HW = Hh * Ww
        h1 = self._gn(x, Ci, HW, name + ".norm1", True)
        h2 = self._conv(h1, name + ".conv1", Hh, Ww, Ci, Co)
        A.put(h1)
        h3 = self._gn(h2, Co, HW, name + ".norm2", True)
        A.put(h2)
        if Ci != Co:
            sc = A.get(B, HW, Co)
            self._gemm(x.view(B * HW, Ci), self.t[name + ".conv_shortcut.weight"], sc.view(B * HW, Co), bias=self.t[name + ".conv_shortcut.bias"])
        else:
            sc = x
        out = self._conv(h3, name + ".conv2", Hh, Ww, Co, Co, residual=sc)
        A.put(h3)
        if Ci != Co:
            A.put(sc)
        return out

    def _attn(self, x, Cc, Hh, Ww, name):
        """single head of dim Cc: QK^T as a GEMM with fp32 scores, row softmax, PV as a GEMM against V^T."""
        A, B, t = self.arena, self.B, self.t
        S = Hh * Ww
        g = self._gn(x, Cc, S, name + ".group_norm", False)
        qk = A.get(B, S, 2 * Cc)
        vt = torch.zeros(B, Cc, S, device=self.dev, dtype=BF16)
        self.keep.append(vt)
        self._gemm(g.view(B, S, Cc), t[name + ".qkv"], qk, bias=t[name + ".qkv.bias"], out_t=vt, n_trans_begin=2 * Cc)
        A.put(g)
        scores = torch.empty(B, S, S, device=self.dev, dtype=F32)
        probs = torch.empty(B, S, S, device=self.dev, dtype=BF16)
        self.keep += [scores, probs]
        d = ops.make_gemm_desc(qk[:, :, :Cc], qk[:, :, Cc:], None)
        d.C, d.ldc, d.strideC, d.epilogue = scores.data_ptr(), S, S * S, L.EPI_F32OUT
        self.keep.append(d)
        self._emit(self.lib.tmix_gemm_bf16, C.byref(d))
        self.flops += 2 * B * S * S * Cc
        self._emit(self.lib.tmix_softmax_rows, scores.data_ptr(), S, probs.data_ptr(), S, B * S, S, Cc ** -0.5)
        ao = A.get(B, S, Cc)
        self._gemm(probs, vt, ao)
        A.put(qk)
        out = A.get(B, S, Cc)
        self._gemm(ao.view(B * S, Cc), t[name + ".to_out.0.weight"], out.view(B * S, Cc), bias=t[name + ".to_out.0.bias"],
                   residual=x.view(B * S, Cc))
        A.put(ao)
        return out

    def _build(self):
        cfg, A, B, lib, t = self.cfg, self.arena, self.B, self.lib, self.t
        ch = list(reversed(cfg["block_out_channels"]))
        Hh, Ww = self.h, self.w
        x = A.get(B, Hh * Ww, ch[0])
        self._emit(lib.tmix_conv_in_pre, self.latent.data_ptr(), t["decoder.conv_in.weight"].data_ptr(),
                   t["decoder.conv_in.bias"].data_ptr(), x.data_ptr(), B, 4, Hh, Ww, ch[0], self._pre_w, self._pre_b)
        x2 = self._resnet(x, ch[0], ch[0], Hh, Ww, "decoder.mid_block.resnets.0"); A.put(x)
        x3 = self._attn(x2, ch[0], Hh, Ww, "decoder.mid_block.attentions.0"); A.put(x2)
        x = self._resnet(x3, ch[0], ch[0], Hh, Ww, "decoder.mid_block.resnets.1"); A.put(x3)
        ci = ch[0]
        for i, co in enumerate(ch):
            for j in range(cfg["layers_per_block"] + 1):
                x2 = self._resnet(x, ci, co, Hh, Ww, f"decoder.up_blocks.{i}.resnets.{j}")
                A.put(x)
                x, ci = x2, co
            if i < len(ch) - 1:
                x2 = self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", Hh, Ww, co, co, mode=L.CONV_UP2)
                A.put(x)
                x = x2
                Hh, Ww = Hh * 2, Ww * 2
        y = self._gn(x, ch[-1], Hh * Ww, "decoder.conv_norm_out", True)
        A.put(x)
        raw = torch.empty_like(self.image)
        self.keep.append(raw)
        self._emit(lib.tmix_conv_out, y.data_ptr(), t["decoder.conv_out.weight"].data_ptr(), t["decoder.conv_out.bias"].data_ptr(),
                   raw.data_ptr(), B, ch[-1], Hh, Ww, cfg["out_channels"])
        self._emit(lib.tmix_affine_clamp, raw.data_ptr(), self.image.data_ptr(), raw.numel(), 0.5, 0.5, 0.0, 1.0)

    def run(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        for fn, a in self.ops:
            rc = fn(*a, st)
            if rc:
                L.check(rc, fn.__name__)

    def __call__(self, latent):
        self.latent.copy_(latent)
        self.run()
        return self.image


def decode_in_groups(vae, latent, inv_scale, plans, device="cuda"):
    """decode `latent` [n,4,h,w] with VAEDecoderPlan(s) cached in `plans`.  The decoder's largest activation is
    [n, 8h, 8w, 256] bf16 and the conv kernel's element offsets are 32-bit (< 2^30), so co-batched latents are decoded in
    groups (three 1024 x 1024 images per plan) -- fusion_sampling.py:496-528 decodes its single latent in one call."""
    cfg, sd = vae
    h, w = latent.shape[-2:]
    per = max(1, min(latent.shape[0], (1 << 30) // max(1, 64 * h * w * 256 + 1)))
    outs = []
    for i in range(0, latent.shape[0], per):
        part = latent[i:i + per]
        key = (round(inv_scale, 6), part.shape[0], h, w)
        if key not in plans:
            plans[key] = VAEDecoderPlan(cfg, sd, part.shape[0], h, w, inv_scale, device)
        y = plans[key](part)
        outs.append(y if latent.shape[0] <= per else y.clone())
    return outs[0] if len(outs) == 1 else torch.cat(outs)


class VAEEncoderPlan(VAEDecoderPlan):
    """encode(image [B,3,H,W] fp32 in [-1,1]) -> (mean, logvar) [B,4,H/8,W/8] fp32 of AutoencoderKL.encode (the I2VGen-XL pipeline's
    `prepare_image_latents`, video_gen/pipeline_i2vgen_xl.py:421-451, samples from it and scales by 0.18215).  Same emitters as the
    decoder; the stride-2 convs pad right / bottom only (`TMIX_CONV_S2A`), conv_in takes the 3 RGB planes in fp32."""

    def __init__(self, cfg, sd, B, H, W, device="cuda"):
        self.cfg, self.B, self.H, self.Wd = cfg, B, H, W
        self.dev = torch.device(device)
        self.lib = L.load()
        self.ops, self.keep = [], []
        self.arena = _Arena(self.dev)
        self.flops = 0
        dev = self.dev
        t = {}
        for k, v in sd.items():
            if not k.startswith(("encoder.", "quant_conv")):
                continue
            v = v.to(dev)
            if k.endswith(".bias") or "norm" in k:
                t[k] = v.to(F32).contiguous()
            elif v.dim() == 4 and v.shape[-1] == 3 and not k.startswith("encoder.conv_in"):
                t[k] = v.permute(0, 2, 3, 1).to(BF16).contiguous()
            elif v.dim() == 4 and v.shape[-1] == 1 and not k.startswith("quant_conv"):
                t[k] = v.reshape(v.shape[0], v.shape[1]).to(BF16).contiguous()
            elif v.dim() == 2:
                t[k] = v.to(BF16).contiguous()
        t["encoder.conv_in.weight"] = sd["encoder.conv_in.weight"].to(dev, F32).permute(0, 2, 3, 1).contiguous()
        a = "encoder.mid_block.attentions.0"
        t[a + ".qkv"] = torch.cat([t[a + ".to_q.weight"], t[a + ".to_k.weight"], t[a + ".to_v.weight"]]).contiguous()
        t[a + ".qkv.bias"] = torch.cat([t[a + ".to_q.bias"], t[a + ".to_k.bias"], t[a + ".to_v.bias"]]).contiguous()
        self.t = t
        lc = cfg["latent_channels"]
        self.qw = sd["quant_conv.weight"].to(dev, F32).reshape(2 * lc, 2 * lc)
        self.qb = sd["quant_conv.bias"].to(dev, F32)
        self.image = torch.zeros(B, cfg["out_channels"], H, W, device=dev, dtype=F32)
        nl = len(cfg["block_out_channels"]) - 1
        self.moments = torch.zeros(B, 2 * lc, H >> nl, W >> nl, device=dev, dtype=F32)
        self._gn_ws = ops.groupnorm_ws(B, 4096, cfg["groups"], dev)
        self._build()

    def _build(self):
        cfg, A, B, lib, t = self.cfg, self.arena, self.B, self.lib, self.t
        ch = list(cfg["block_out_channels"])
        Hh, Ww = self.H, self.Wd
        x = A.get(B, Hh * Ww, ch[0])
        self._emit(lib.tmix_conv_in, self.image.data_ptr(), t["encoder.conv_in.weight"].data_ptr(), t["encoder.conv_in.bias"].data_ptr(),
                   x.data_ptr(), B, cfg["out_channels"], Hh, Ww, ch[0])
        ci = ch[0]
        for i, co in enumerate(ch):
            for j in range(cfg["layers_per_block"]):
                x2 = self._resnet(x, ci, co, Hh, Ww, f"encoder.down_blocks.{i}.resnets.{j}")
                A.put(x)
                x, ci = x2, co
            if i < len(ch) - 1:
                x2 = self._conv(x, f"encoder.down_blocks.{i}.downsamplers.0.conv", Hh, Ww, co, co, mode=L.CONV_S2A)
                A.put(x)
                x = x2
                Hh, Ww = Hh // 2, Ww // 2
        x2 = self._resnet(x, ci, ci, Hh, Ww, "encoder.mid_block.resnets.0"); A.put(x)
        x3 = self._attn(x2, ci, Hh, Ww, "encoder.mid_block.attentions.0"); A.put(x2)
        x = self._resnet(x3, ci, ci, Hh, Ww, "encoder.mid_block.resnets.1"); A.put(x3)
        y = self._gn(x, ci, Hh * Ww, "encoder.conv_norm_out", True)
        A.put(x)
        self._emit(lib.tmix_conv_out, y.data_ptr(), t["encoder.conv_out.weight"].data_ptr(), t["encoder.conv_out.bias"].data_ptr(),
                   self.moments.data_ptr(), B, ci, Hh, Ww, 2 * cfg["latent_channels"])

    def __call__(self, image):
        """-> (mean, logvar) after quant_conv (an 8x8 per-pixel map on the 1/8-resolution moments, once per video: tmix_linear_f32 over 256-pixel row blocks --
        the product reaches no torch matmul / hipBLASLt anywhere)."""
        self.image.copy_(image)
        self.run()
        B, C2, h, w = self.moments.shape
        rows = self.moments.permute(0, 2, 3, 1).reshape(B * h * w, C2).contiguous()
        out = torch.cat([ops.linear_f32(rows[i:i + 256], self.qw, self.qb) for i in range(0, rows.shape[0], 256)])
        m = out.view(B, h, w, C2).permute(0, 3, 1, 2)
        lc = self.cfg["latent_channels"]
        return m[:, :lc].contiguous(), m[:, lc:].clamp(-30.0, 20.0).contiguous()
