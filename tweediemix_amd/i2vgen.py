"""The I2VGen-XL UNet (BASELINE config #5; the network behind video_gen/pipeline_i2vgen_xl.py:688-697) on this package's HIP
kernels: the spatial skeleton reuses the SDXL plan's emitters (ResnetBlock2D, Transformer2DModel with cached cross-attention
K/V, GroupNorm, implicit-GEMM convs) with the frames folded into the batch; the temporal layers run on the same GEMM /
GroupNorm kernels over the frame axis plus `TMIX_CONV_T3` (TemporalConvLayer) and `tmix_temporal_attn`
(TransformerTemporalModel).  Everything that does not depend on the sample or the timestep (fps / context / image-latent
embeddings: a handful of 4..64-channel convolutions and one 4-wide temporal encoder) is evaluated ONCE per video by
`conditioning()` below on the library's fp32 kernels (csrc/conditioning.hip) -- init-time, outside the per-step path.

PARITY UNPINNED (diffusers is neither vendored in the reference nor installed here; no checkpoint offline): structure and
key names follow the published `I2VGenXLUNet`; the restated inventory has 1,420,469,224 parameters = 2.84 GB in fp16, the
size of the published fp16 checkpoint.  Tests compare this plan with oracle/i2vgen_oracle.py (the fp32 restatement)."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import lib as L
from . import ops
from .unet import UNetPlan, _Arena, BF16, F32, refine_group, SHARED as U_SHARED
from .weights import fold_layernorm, interleave_geglu


@dataclass
class I2VConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280, 1280)
    attn_levels: tuple = (True, True, True, False)
    layers_per_block: int = 2
    norm_groups: int = 32
    cross_dim: int = 1024
    head_dim: int = 64
    transformer_in_heads: int = 8
    ctx_pool: int = 32

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4


FULL = I2VConfig()
TINY = I2VConfig(block_out_channels=(64, 128, 128, 128), cross_dim=128, transformer_in_heads=2, ctx_pool=8)


def _sites(cfg):
    """(spatial t2d prefixes, temporal transformer prefixes with (channels, inner)) in forward order."""
    ch, nb = cfg.block_out_channels, len(cfg.block_out_channels)
    t2d, tt = [], [("transformer_in", ch[0], cfg.transformer_in_heads * cfg.head_dim)]
    for bi, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            if cfg.attn_levels[bi]:
                t2d.append((f"down_blocks.{bi}.attentions.{j}", co)); tt.append((f"down_blocks.{bi}.temp_attentions.{j}", co, co))
    t2d.append(("mid_block.attentions.0", ch[-1])); tt.append(("mid_block.temp_attentions.0", ch[-1], ch[-1]))
    for ui in range(nb):
        bi = nb - 1 - ui
        for j in range(cfg.layers_per_block + 1):
            if cfg.attn_levels[bi]:
                t2d.append((f"up_blocks.{ui}.attentions.{j}", ch[bi])); tt.append((f"up_blocks.{ui}.temp_attentions.{j}", ch[bi], ch[bi]))
    return t2d, tt


class I2VWeights:
    """device-resident weights in kernel layouts from a diffusers-keyed I2VGenXLUNet state dict."""

    def __init__(self, cfg: I2VConfig, sd: dict, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        self.kind, self.K = "none", 0
        dev = self.device
        self.raw = {k: v for k, v in sd.items() if k.startswith(("image_latents_", "context_embedding", "fps_embedding"))}
        t = {}
        g = lambda n: sd[n].to(dev)
        bf = lambda x: x.to(dev, BF16).contiguous()
        f32 = lambda x: x.to(dev, F32).contiguous()
        for name, v in sd.items():
            if name in self.raw or name == "conv_in.weight":
                continue
            if name.endswith(".bias") or v.dim() == 1:
                t[name] = f32(v)
            elif v.dim() == 5:                                   # Conv3d (3,1,1): [Co,Ci,3,1,1] -> [Co,3,Ci]
                t[name] = bf(v.to(dev)[:, :, :, 0, 0].permute(0, 2, 1))
            elif v.dim() == 4 and v.shape[-1] == 3:
                t[name] = bf(v.to(dev).permute(0, 2, 3, 1))
            elif v.dim() == 4:                                   # 1x1 conv_shortcut -> Linear
                t[name] = bf(v.to(dev).reshape(v.shape[0], v.shape[1]))
            elif v.dim() == 2 and ".attn" not in name and ".ff.net.0.proj" not in name:
                t[name] = bf(v)
        t["conv_in.weight"] = f32(g("conv_in.weight").permute(0, 2, 3, 1))

        def fold(key, w, norm, bias=None):
            wp, cs, tt = fold_layernorm(w.to(dev, F32), g(norm + ".weight"), g(norm + ".bias"), bias)
            t[key], t[key + ".colsum"], t[key + ".bias"] = wp, cs, tt

        def ff(tb):
            wp, cs, tt = fold_layernorm(g(tb + ".ff.net.0.proj.weight").to(F32), g(tb + ".norm3.weight"), g(tb + ".norm3.bias"),
                                        g(tb + ".ff.net.0.proj.bias"))
            t[tb + ".ff1"] = interleave_geglu(wp, None)[0].contiguous()
            csi, bi = interleave_geglu(cs[:, None], tt)
            t[tb + ".ff1.colsum"], t[tb + ".ff1.bias"] = csi[:, 0].contiguous(), bi.contiguous()

        t2d, tt = _sites(cfg)
        for pfx, _c in t2d:
            tb = pfx + ".transformer_blocks.0"
            a1, a2 = tb + ".attn1", tb + ".attn2"
            fold(a1 + ".qkv", torch.cat([g(a1 + ".to_q.weight"), g(a1 + ".to_k.weight"), g(a1 + ".to_v.weight")]), tb + ".norm1")
            t[a1 + ".out"] = bf(g(a1 + ".to_out.0.weight"))
            fold(a2 + ".q", g(a2 + ".to_q.weight"), tb + ".norm2")
            t[a2 + ".out"] = bf(g(a2 + ".to_out.0.weight"))
            t[a2 + ".kv_rows"] = bf(torch.cat([g(a2 + ".to_k.weight"), g(a2 + ".to_v.weight")])[None])
            ff(tb)
        for pfx, _c, _inner in tt:
            tb = pfx + ".transformer_blocks.0"
            for a, norm in ((tb + ".attn1", tb + ".norm1"), (tb + ".attn2", tb + ".norm2")):
                fold(a + ".qkv", torch.cat([g(a + ".to_q.weight"), g(a + ".to_k.weight"), g(a + ".to_v.weight")]), norm)
                t[a + ".out"] = bf(g(a + ".to_out.0.weight"))
            ff(tb)
        self.t = t

    def __getitem__(self, k):
        return self.t[k]

    def stacked_time_proj(self):
        """as UNetWeights.stacked_time_proj: every ResnetBlock2D's time_emb_proj stacked along N (weight [sum Co, T] bf16, bias fp32, section starts on the
        device, names), built once -- one tmix_linear_small_sections launch per step instead of one tmix_linear_small per resnet"""
        if getattr(self, "_tproj", None) is None:
            names = sorted(k[:-len(".time_emb_proj.weight")] for k in self.t if k.endswith(".time_emb_proj.weight"))
            w = torch.cat([self.t[n + ".time_emb_proj.weight"] for n in names], 0).contiguous()
            b = torch.cat([self.t[n + ".time_emb_proj.bias"].float() for n in names], 0).contiguous()
            st = [0]
            for n in names:
                st.append(st[-1] + self.t[n + ".time_emb_proj.weight"].shape[0])
            self._tproj = (w, b, torch.tensor(st, device=w.device, dtype=torch.int32), names)
        return self._tproj

    def conv2_with_shortcut(self, name):
        """as UNetWeights.conv2_with_shortcut: ([Co, 9*Co + Ci] rows [conv2 taps | conv_shortcut], the two biases' sum), built once"""
        k = name + ".conv2+shortcut"
        if k + ".weight" not in self.t:
            self.t[k + ".weight"] = ops.shortcut_weight(self.t[name + ".conv2.weight"], self.t[name + ".conv_shortcut.weight"])
            self.t[k + ".bias"] = (self.t[name + ".conv2.bias"] + self.t[name + ".conv_shortcut.bias"]).contiguous()
        return self.t[k + ".weight"], self.t[k + ".bias"]


@torch.no_grad()
def conditioning(W: I2VWeights, fps, image_latents, image_embeddings, encoder_hidden_states):
    """the sample- and timestep-independent part of I2VGenXLUNet.forward (fps embedding, context tokens, image-latent features: what
    video_gen/pipeline_i2vgen_xl.py:604-639 prepares), once per video, on the library's fp32 kernels (csrc/conditioning.hip: layers of 4 .. 64
    channels, nothing an MFMA tile could be filled with).  Shapes as the pipeline passes them:
    fps [B], image_latents [B,4,F,h,w], image_embeddings [B,cross], encoder_hidden_states [B,77,cross]."""
    cfg, dev = W.cfg, W.device
    p = {k: v.to(dev, F32).contiguous() for k, v in W.raw.items()}
    lin = lambda x, n, **kw: ops.linear_f32(x, p[n + ".weight"], p[n + ".bias"], **kw)
    conv = lambda x, n, **kw: ops.conv3x3_f32(x, p[n + ".weight"], p[n + ".bias"], **kw)
    B, Cc, Fr, H, Wd = image_latents.shape
    il = image_latents.to(dev, F32)
    sin = torch.empty(B, cfg.block_out_channels[0], device=dev, dtype=F32)
    L.check(L.load().tmix_timestep_embedding(fps.to(dev, F32).contiguous().data_ptr(), sin.data_ptr(), B, cfg.block_out_channels[0],
                                             torch.cuda.current_stream().cuda_stream), "tmix_timestep_embedding")
    fps_emb = lin(lin(sin, "fps_embedding.0", act_out=True), "fps_embedding.2")
    n = "image_latents_context_embedding"
    v = conv(il[:, :, 0].contiguous(), n + ".0", silu=True)
    v = ops.adaptive_avgpool_f32(v, cfg.ctx_pool, cfg.ctx_pool)
    v = conv(conv(v, n + ".3", stride=2, silu=True), n + ".5", stride=2)
    ctx_img = v.permute(0, 2, 3, 1).reshape(B, -1, cfg.cross_dim)
    e = lin(lin(image_embeddings.to(dev, F32).contiguous(), "context_embedding.0", act_out=True), "context_embedding.2").view(B, cfg.in_channels, cfg.cross_dim)
    context = torch.cat([encoder_hidden_states.to(dev, F32), ctx_img, e], dim=1)
    n = "image_latents_proj_in"
    x = il.permute(0, 2, 1, 3, 4).reshape(B * Fr, Cc, H, Wd).contiguous()
    x = conv(conv(conv(x, n + ".0", silu=True), n + ".2", silu=True), n + ".4")
    il_feat = ops.i2v_temporal_encoder(x, B, Fr, p, "image_latents_temporal_encoder")
    return fps_emb, context, il_feat


class _KV:
    """cross-attention K / V^T of every spatial transformer for the (constant) context, one row per image (= clip x frame)."""

    def __init__(self, W: I2VWeights, context, frames):
        t2d, _tt = _sites(W.cfg)
        ctx = context.to(W.device, BF16).contiguous()
        B, Lk, _ = ctx.shape
        self.B, self.Lk = B * frames, Lk
        self.ld = (Lk + 7) // 8 * 8
        self.k, self.vt = {}, {}
        for pfx, Cc in t2d:
            a2 = pfx + ".transformer_blocks.0.attn2"
            k = torch.empty(B, Lk, Cc, device=W.device, dtype=BF16)
            vt = torch.zeros(B, Cc, self.ld, device=W.device, dtype=BF16)
            wkv = W[a2 + ".kv_rows"][0]                            # [2C, cross]; C = 320 is not a multiple of the 128-column
            ops.gemm(ctx, wkv[:Cc].contiguous(), out=k)            # transposed-region granularity, so K and V^T are two launches
            ops.gemm(ctx, wkv[Cc:].contiguous(), out_t=vt, n_trans_begin=0)
            self.k[a2] = k.repeat_interleave(frames, dim=0).contiguous()
            self.vt[a2] = vt.repeat_interleave(frames, dim=0).contiguous()
        torch.cuda.synchronize()


class _Inject:
    """first-frame feature injection of the reference's patched ResnetBlock2D.forward (video_gen/utils_attn.py:433-455) as an op
    of the recorded forward: a no-op unless the plan's `inject` flag is up (the loop raises it for the scheduled timesteps)."""
    __name__ = "tmix_frame_inject"

    def __init__(self, plan, buf, per_frame, hard):
        self.plan, self.ptr, self.per_frame, self.hard = plan, buf.data_ptr(), per_frame, hard

    def __call__(self, st):
        p = self.plan
        if not p.inject:
            return 0
        a = 0.0 if self.hard else float(torch.tensor(p.interp, dtype=torch.float32))
        b = 0.0 if self.hard else float(torch.tensor(1.0 - float(p.interp), dtype=torch.float32))
        return p.lib.tmix_frame_inject(self.ptr, L.BF16, p.clips, p.frames, self.per_frame, int(self.hard), a, b, st)


class I2VPlan(UNetPlan):
    """pre-recorded forward of the I2VGen-XL UNet for `clips` videos of `frames` latent frames of h x w (CFG: clips = 2).
    __call__(sample [clips,4,F,h,w], t) -> prediction [clips,4,F,h,w] fp32."""

    INJECT_SITES = {"mid_block.resnets.0": True, "mid_block.resnets.1": True, "up_blocks.1.resnets.0": False}   # site -> hard copy?

    def __init__(self, W: I2VWeights, clips: int, frames: int, h: int, w: int, fps_emb, context, il_feat, autotune: bool = True,
                 interp: float = 0.7, shared: bool = False):
        cfg = W.cfg
        self.tune_ctx = U_SHARED if shared else ""
        self.inject, self.interp = False, interp         # raised per step by the sampling loop (FeatureInjector schedule)
        self.W, self.cfg, self.clips, self.frames, self.h, self.w = W, cfg, clips, frames, h, w
        self.B = B = clips * frames                     # spatial layers see every frame as one image
        self.row_sets, self.routed, self._rows_cache = list(range(B)), False, {}
        self.lowrank, self._sets_dev = False, None       # (no LoRA routing in the video UNet; UNetPlan's emitters ask -- tests/test_plan_attrs_cpu.py keeps this list honest)
        self.fp8_chain_ff = False
        self.fp8_attn_out = False
        self.fp8_tile = 0
        self.fp8_conv, self.fp8_conv_tile = False, 0
        # GroupNorm statistics from the producers' column partials (UNetPlan._colstats) where 32-row blocks fit the normalised image: the per-frame norms of the
        # first two levels (5376 / 1344 pixels) and the clip-wide norms of TemporalConvLayer / TransformerTemporalModel up to CLIP_COLSTATS_MAX rows (16 frames x HW is
        # always a multiple of 32: every level since round 5); per-frame norms of 336 / 84 pixels and the injection sites keep the statistics kernel
        self._gn_fused = not os.environ.get("TMIX_GN_STATS_KERNEL")
        self._sc_fused = not os.environ.get("TMIX_SHORTCUT_GEMM")      # conv_shortcut in conv2's launch, no concat launches (UNetPlan._resnet)
        self.lib, self.dev = L.load(), W.device
        dev = self.dev
        self.ops, self.keep, self.arena = [], [], _Arena(dev)
        self.flops = self.gemm_flops = 0
        self.launches = {"gemm": [], "conv": [], "attn": []}
        self.op_meta = {}
        self.fp8 = False                                 # (the fp8 projections are wired for the image UNet only)
        self._tunable, self._ln_links, self._vt = [], [], {}
        self._pf_prev, self._pf_on = None, not os.environ.get("TMIX_NO_PREFETCH")      # next-launch weight prefetch hints (UNetPlan._hint_weights)
        self._pf_cap = int(float(os.environ.get("TMIX_PF_CAP_MB", "0")) * (1 << 20))    # (whole tensors: a clip's launches last 50 - 800 us, UNetPlan.__init__)
        self._pf_cap_over = int(float(os.environ.get("TMIX_PF_CAP_OVER_MB", "20")) * (1 << 20))
        self.kv = _KV(W, context, frames)
        self.x_in = torch.zeros(B, 2 * cfg.in_channels, h, w, device=dev, dtype=F32)
        self.x_in.view(clips, frames, 2 * cfg.in_channels, h, w)[:, :, cfg.in_channels:] = il_feat.to(dev, F32).permute(0, 2, 1, 3, 4)
        self.latent = self.x_in                          # (UNetPlan interface name)
        self.t_dev = torch.zeros(clips, device=dev, dtype=F32)
        self.eps = torch.zeros(B, cfg.out_channels, h, w, device=dev, dtype=F32)
        self.fps_emb = fps_emb.to(dev, F32).contiguous()
        cmax = max(max(cfg.block_out_channels) * 2, cfg.transformer_in_heads * cfg.head_dim)
        self._gn_ws = ops.groupnorm_ws(B, cmax, cfg.norm_groups, dev)
        t2d, tt = _sites(cfg)
        need = max(((inner + 127) // 128) for _p, _c, inner in tt + [(p_, c_, c_) for p_, c_ in t2d]) * B * h * w * 2
        self._ln_buf = torch.zeros(need, device=dev, dtype=F32)
        self._build()
        self._link_ln()
        if autotune:
            self.autotune()

    # ------------------------------------------------------------------ emitters the image UNet did not need
    def _time_bias(self, name, Co, emb):
        """ResnetBlock2D as in the image UNet (UNetPlan._resnet), except that the time embedding exists once per CLIP ([clips, T]): its
        projection (this block's section of the one stacked launch in _build) is added to every frame of the clip through the conv's batch_bias_images."""
        temb = self._temb[name]
        assert temb.shape == (self.clips, Co)
        return temb, self.frames

    # rows of a clip-wide norm whose combine launch walks the producers' partials: all levels (first level at 16 x 768 x 448: 86,016 rows = 2688 blocks per channel,
    # where the walk still beats the statistics kernel's pass over 110 MB: 74.94 -> 74.69 ms per step, 50 launches fewer; TMIX_CLIP_COLSTATS_MAX=32768 = round 4's limit)
    CLIP_COLSTATS_MAX = int(os.environ.get("TMIX_CLIP_COLSTATS_MAX", "131072"))

    def _colstats(self, owner, rows, HW, Cc):
        """as UNetPlan._colstats, for two kinds of readers: per-frame norms (HW pixels per image) and clip-wide norms (frames x HW rows per image)"""
        R = ops.COLSTATS_ROWS
        chw = self.frames * HW
        ok = (HW % R == 0 and HW <= ops.COLSTATS_MAX_HW) or (chw % R == 0 and chw <= self.CLIP_COLSTATS_MAX)
        if not self._gn_fused or not ok or rows % R or Cc % 8:
            return None
        cs = self.arena.get(rows // R, 2, Cc, dtype=F32)
        owner._cs = ((cs, Cc),)
        return cs

    def _gn_b(self, x, Bn, Cc, HW, name, eps, silu):
        """GroupNorm over whole clips (Bn = clips images of HW = frames x hw rows)"""
        out = self.arena.get(*x.shape)
        W = self.W
        parts = getattr(x, "_cs", None)
        if parts and len(parts) == 1 and parts[0][1] == Cc and HW % ops.COLSTATS_ROWS == 0 and HW <= self.CLIP_COLSTATS_MAX:
            self._emit(self.lib.tmix_groupnorm_nhwc_pre, x.data_ptr(), Cc, None, 0, out.data_ptr(), W[name + ".weight"].data_ptr(), W[name + ".bias"].data_ptr(),
                       self._gn_ws.data_ptr(), Bn, HW, self.cfg.norm_groups, eps, int(silu), parts[0][0].data_ptr(), Cc, None, 0)
        else:
            self._emit(self.lib.tmix_groupnorm_nhwc, x.data_ptr(), Cc, None, 0, out.data_ptr(), W[name + ".weight"].data_ptr(),
                       W[name + ".bias"].data_ptr(), self._gn_ws.data_ptr(), Bn, HW, self.cfg.norm_groups, eps, int(silu))
        self.op_meta[len(self.ops) - 1] = ("norm", 0, ("norm", Bn, HW, Cc))      # (every instrumented launch needs its entry: the slots are dealt out in issue order)
        return out

    def _inject_site(self, buf, site, per_frame):
        if site in self.INJECT_SITES:
            self.ops.append((_Inject(self, buf, per_frame, self.INJECT_SITES[site]), ()))
            for cs, _c in getattr(buf, "_cs", None) or ():      # the injection rewrites the tensor behind its producer: the partials no longer describe it
                self.arena.put(cs)
            buf._cs = None

    def _conv_t3(self, x, wname, HW, Cc, residual=None):
        """Conv3d (3,1,1) over the frame axis: x [(clips frames), hw, C] seen as [clips, frames, hw, C]."""
        out = self.arena.get(self.B, HW, Cc)
        shp = (self.clips, self.frames, HW, Cc)
        d = ops.make_conv_desc(x.view(*shp), self.W[wname + ".weight"], out.view(*shp), self.W[wname + ".bias"], None,
                               None if residual is None else residual, L.CONV_T3, col_stats_out=self._colstats(out, self.B * HW, HW, Cc))
        self.keep.append(d)
        self._emit(self.lib.tmix_conv3x3_nhwc, C.byref(d))
        fl = 2 * self.B * HW * Cc * 3 * Cc
        self.flops += fl
        self.launches["conv"].append((d, fl))
        self._tunable.append((len(self.ops) - 1, "conv", d))
        self.op_meta[len(self.ops) - 1] = ("conv", fl, d)
        return out

    def _temp_conv(self, x, Cc, HW, name):
        """diffusers TemporalConvLayer: x + conv4(conv3(conv2(conv1(x)))), each conv = GroupNorm (over the whole clip) + SiLU + Conv3d."""
        A = self.arena
        v = x
        for k, idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
            gq = self._gn_b(v, self.clips, Cc, self.frames * HW, f"{name}.conv{k}.0", 1e-5, True)
            if v is not x:
                A.put(v)
            v = self._conv_t3(gq, f"{name}.conv{k}.{idx}", HW, Cc, residual=x if k == 4 else None)
            A.put(gq)
        return v

    def _tattn(self, h, a, inner, heads, HW, st, st_next):
        """one self-attention over the frame axis (LayerNorm folded into the fused q/k/v projection)."""
        A, W, M = self.arena, self.W, self.B * HW
        qkv = A.get(self.B, HW, 3 * inner)
        self._gemm(h.view(M, inner), W[a + ".qkv"], qkv.view(M, 3 * inner), bias=W[a + ".qkv.bias"], ln_stats=st, ln_colsum=W[a + ".qkv.colsum"])
        ao = A.get(self.B, HW, inner)
        self._emit(self.lib.tmix_temporal_attn, qkv.data_ptr(), 3 * inner, ao.data_ptr(), inner, self.clips, self.frames, HW, heads,
                   self.cfg.head_dim ** -0.5)
        self.flops += 4 * self.clips * HW * heads * self.frames * self.frames * 64
        A.put(qkv)
        self._gemm(ao.view(M, inner), W[a + ".out"], h.view(M, inner), bias=W[a + ".to_out.0.bias"], residual=h.view(M, inner), row_stats_out=st_next)
        A.put(ao)

    def _ttemp(self, x, Cc, inner, HW, name):
        """diffusers TransformerTemporalModel (1 layer, both attentions self over the frames, GEGLU feed-forward)."""
        A, W, M = self.arena, self.W, self.B * HW
        heads = inner // self.cfg.head_dim
        gq = self._gn_b(x, self.clips, Cc, self.frames * HW, name + ".norm", 1e-6, False)
        h = A.get(self.B, HW, inner)
        pm = (inner + 127) // 128
        st = self._ln_buf[:pm * M * 2].view(pm, M, 2)
        self._gemm(gq.view(M, Cc), W[name + ".proj_in.weight"], h.view(M, inner), bias=W[name + ".proj_in.bias"], row_stats_out=st)
        A.put(gq)
        tb = name + ".transformer_blocks.0"
        self._tattn(h, tb + ".attn1", inner, heads, HW, st, st)
        self._tattn(h, tb + ".attn2", inner, heads, HW, st, st)
        f = A.get(M, 4 * inner)
        self._gemm(h.view(M, inner), W[tb + ".ff1"], f, bias=W[tb + ".ff1.bias"], geglu=True, ln_stats=st, ln_colsum=W[tb + ".ff1.colsum"])
        self._gemm(f, W[tb + ".ff.net.2.weight"], h.view(M, inner), bias=W[tb + ".ff.net.2.bias"], residual=h.view(M, inner))
        A.put(f)
        out = A.get(self.B, HW, Cc)
        self._gemm(h.view(M, inner), W[name + ".proj_out.weight"], out.view(M, Cc), bias=W[name + ".proj_out.bias"], residual=x.view(M, Cc), cs_owner=out)
        A.put(h)
        return out

    # ------------------------------------------------------------------ whole network (order of I2VGenXLUNet.forward)
    def _build(self):
        cfg, W, B, A, lib = self.cfg, self.W, self.B, self.arena, self.lib
        ch, nb, T = cfg.block_out_channels, len(cfg.block_out_channels), cfg.time_embed_dim
        C0 = ch[0]
        nc = self.clips                               # time / fps embeddings exist once per clip
        tsin = torch.empty(nc, C0, device=self.dev, dtype=F32)
        thid = torch.empty(nc, T, device=self.dev, dtype=F32)
        emb = torch.empty(nc, T, device=self.dev, dtype=F32)
        self.keep += [tsin, thid, emb]
        self._emit(lib.tmix_timestep_embedding, self.t_dev.data_ptr(), tsin.data_ptr(), nc, C0)
        self._emit(lib.tmix_linear_small, tsin.data_ptr(), W["time_embedding.linear_1.weight"].data_ptr(),
                   W["time_embedding.linear_1.bias"].data_ptr(), None, thid.data_ptr(), nc, T, C0, 0, 1)
        self._emit(lib.tmix_linear_small, thid.data_ptr(), W["time_embedding.linear_2.weight"].data_ptr(),
                   W["time_embedding.linear_2.bias"].data_ptr(), self.fps_emb.data_ptr(), emb.data_ptr(), nc, T, T, 0, 0)
        tw, tb_, starts, names = W.stacked_time_proj()        # every ResnetBlock2D's time_emb_proj(SiLU(emb)) in ONE launch
        tall = torch.empty(nc * tw.shape[0], device=self.dev, dtype=F32)
        self.keep.append(tall)
        self._emit(lib.tmix_linear_small_sections, emb.data_ptr(), tw.data_ptr(), tb_.data_ptr(), tall.data_ptr(), nc, tw.shape[0], T, 1,
                   starts.data_ptr(), len(names))
        hs = starts.tolist()
        self._temb = {n: tall[hs[i] * nc:hs[i + 1] * nc].view(nc, hs[i + 1] - hs[i]) for i, n in enumerate(names)}
        Hh, Ww = self.h, self.w
        x = A.get(B, Hh * Ww, C0)
        self._emit(lib.tmix_conv_in, self.x_in.data_ptr(), W["conv_in.weight"].data_ptr(), W["conv_in.bias"].data_ptr(),
                   x.data_ptr(), B, 2 * cfg.in_channels, Hh, Ww, C0)
        x2 = self._ttemp(x, C0, cfg.transformer_in_heads * cfg.head_dim, Hh * Ww, "transformer_in")
        A.put(x)
        x = x2

        def layer(x, ci, co, pfx, j, attn, x2=None):
            y = self._resnet(x, ci, co, Hh, Ww, f"{pfx}.resnets.{j}", emb, x2=x2)
            self._inject_site(y, f"{pfx}.resnets.{j}", Hh * Ww * co)
            y2 = self._temp_conv(y, co, Hh * Ww, f"{pfx}.temp_convs.{j}")
            A.put(y)
            if attn:
                y = self._t2d(y2, co, Hh, Ww, f"{pfx}.attentions.{j}", 1)
                A.put(y2)
                y2 = self._ttemp(y, co, co, Hh * Ww, f"{pfx}.temp_attentions.{j}")
                A.put(y)
            return y2

        skips = [(x, C0)]
        ci = C0
        for bi, co in enumerate(ch):
            for j in range(cfg.layers_per_block):
                x = layer(x, ci, co, f"down_blocks.{bi}", j, cfg.attn_levels[bi])
                ci = co
                skips.append((x, co))
            if bi < nb - 1:
                x = self._conv(x, f"down_blocks.{bi}.downsamplers.0.conv", Hh, Ww, co, co, mode=L.CONV_S2)
                Hh, Ww = Hh // 2, Ww // 2
                skips.append((x, co))
        cm = ch[-1]
        y = self._resnet(x, cm, cm, Hh, Ww, "mid_block.resnets.0", emb)
        self._inject_site(y, "mid_block.resnets.0", Hh * Ww * cm)
        y2 = self._temp_conv(y, cm, Hh * Ww, "mid_block.temp_convs.0"); A.put(y)
        y = self._t2d(y2, cm, Hh, Ww, "mid_block.attentions.0", 1); A.put(y2)
        y2 = self._ttemp(y, cm, cm, Hh * Ww, "mid_block.temp_attentions.0"); A.put(y)
        y = self._resnet(y2, cm, cm, Hh, Ww, "mid_block.resnets.1", emb); A.put(y2)
        self._inject_site(y, "mid_block.resnets.1", Hh * Ww * cm)
        x = self._temp_conv(y, cm, Hh * Ww, "mid_block.temp_convs.1"); A.put(y)
        for ui in range(nb):
            bi = nb - 1 - ui
            co = ch[bi]
            for j in range(cfg.layers_per_block + 1):
                sk, cs = skips.pop()
                if self._sc_ok(ci + cs, co, ci, cs):             # no concatenation: norm1 and conv2's shortcut taps read the two tensors
                    xn = layer(x, ci + cs, co, f"up_blocks.{ui}", j, cfg.attn_levels[bi], x2=sk)
                    A.put(x, sk)
                    x = xn
                else:
                    xc = self._cat(x, ci, sk, cs, Hh * Ww)
                    A.put(x, sk)
                    x = layer(xc, ci + cs, co, f"up_blocks.{ui}", j, cfg.attn_levels[bi])
                    A.put(xc)
                ci = co
            if ui < nb - 1:
                x2 = self._conv(x, f"up_blocks.{ui}.upsamplers.0.conv", Hh, Ww, co, co, mode=L.CONV_UP2)
                A.put(x)
                x = x2
                Hh, Ww = Hh * 2, Ww * 2
        y = self._gn(x, C0, Hh * Ww, "conv_norm_out", 1e-5, True)
        A.put(x)
        self._emit(lib.tmix_conv_out, y.data_ptr(), W["conv_out.weight"].data_ptr(), W["conv_out.bias"].data_ptr(),
                   self.eps.data_ptr(), B, C0, Hh, Ww, cfg.out_channels)
        self.ops = [(fn, tuple(a)) for fn, a in self.ops]

    def __call__(self, sample, t):
        cfg = self.cfg
        xin = self.x_in.view(self.clips, self.frames, 2 * cfg.in_channels, self.h, self.w)
        xin[:, :, :cfg.in_channels] = sample.to(self.dev, F32).permute(0, 2, 1, 3, 4)
        self.t_dev.fill_(float(t))
        self.run()
        return self.eps.view(self.clips, self.frames, cfg.out_channels, self.h, self.w).permute(0, 2, 1, 3, 4)


class I2VPlanGroup:
    """the clips of one UNet call (the unconditional and the text row of the CFG pair) as independent launch chains on their own
    HIP streams -- the same trick as the image sampler's PlanGroup: a dependent chain leaves the chip idle at every kernel
    boundary, the other clip's chain fills the holes (measured 103 -> 97 ms per step at 16 x 768 x 448).  Interface of I2VPlan."""

    def __init__(self, W: I2VWeights, clips: int, frames: int, h: int, w: int, fps_emb, context, il_feat, autotune: bool = True,
                 interp: float = 0.7):
        self.cfg, self.clips, self.frames, self.h, self.w = W.cfg, clips, frames, h, w
        self.plans = [I2VPlan(W, 1, frames, h, w, fps_emb[i:i + 1], context[i:i + 1], il_feat[i:i + 1], autotune=autotune, interp=interp, shared=clips > 1)
                      for i in range(clips)]
        self.streams = [None] + [torch.cuda.Stream(device=W.device) for _ in range(clips - 1)]
        self.eps = torch.zeros(clips * frames, W.cfg.out_channels, h, w, device=W.device, dtype=F32)
        self.flops = sum(p.flops for p in self.plans)
        self.ops = [op for p in self.plans for op in p.ops]

    inject = property(lambda self: self.plans[0].inject, lambda self, v: [setattr(p, "inject", v) for p in self.plans])
    interp = property(lambda self: self.plans[0].interp, lambda self, v: [setattr(p, "interp", v) for p in self.plans])

    def refine(self, **kw):
        return refine_group(self, **kw)

    def set_input(self, sample, t):
        """sample [clips,4,F,h,w] (or [1,...] broadcast to every clip)."""
        c = self.cfg.in_channels
        for i, p in enumerate(self.plans):
            s = sample[i if sample.shape[0] > 1 else 0]
            p.x_in.view(self.frames, 2 * c, self.h, self.w)[:, :c] = s.to(p.dev, F32).permute(1, 0, 2, 3)
            p.t_dev.fill_(float(t))

    def run(self):
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for p, st in zip(self.plans[1:], self.streams[1:]):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                p.run()
                ev = torch.cuda.Event()
                ev.record(st)
                joins.append(ev)
        self.plans[0].run()
        for ev in joins:
            main.wait_event(ev)
        n = self.frames
        for i, p in enumerate(self.plans):
            self.eps[i * n:(i + 1) * n].copy_(p.eps)

    def __call__(self, sample, t):
        self.set_input(sample, t)
        self.run()
        return self.eps.view(self.clips, self.frames, self.cfg.out_channels, self.h, self.w).permute(0, 2, 1, 3, 4)
