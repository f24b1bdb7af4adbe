"""SDXL UNet forward as a static launch plan over the C ABI (include/tmix.h).

What `self.unet(latents, t, encoder_hidden_states, added_cond_kwargs)['sample']` computes at
fusion_sampling.py:340/374/406/414/440 (diffusers UNet2DConditionModel, SDXL-base config), with the
reference's attention hooks (utils_custom.py:53-108, utils_lora.py:55-123) folded in:

* activations are NHWC bf16 ([B, H*W, C] is both the conv and the token layout: no permutes),
* every Linear/conv is tmix_gemm_bf16 / tmix_conv3x3_nhwc with bias, time-embedding, residual and
  GEGLU fused into the epilogue; QKV is one GEMM whose V third is stored transposed for
  tmix_attn_fwd,
* cross-attention K/V depend only on the prompt rows and the (per-concept) weights, so they are
  computed ONCE per call kind (`KVCache`) instead of 75x per image; the per-row concept routing of
  the hooks becomes "which weight set produced row b of the cache",
* LoRA rows use merged weights W + up@down as one weight set per batch row (batched GEMM).

A plan is a fixed list of (C function, argument tuple): running it is a tight loop with no tensor
allocation, so it can be captured into a hipGraph (torch.cuda.graphs) and replayed.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass

import torch

from . import lib as L
from . import ops
from .weights import fold_layernorm, interleave_geglu

BF16 = torch.bfloat16
F32 = torch.float32


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_layers: tuple = (0, 2, 10)
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


    @classmethod
    def from_diffusers(cls, cfg: dict):
        """UNet2DConditionModel config.json of a diffusers-layout SDXL checkpoint.  `attention_head_dim` there holds the
        HEAD COUNT per level (a diffusers naming quirk): SDXL's [5,10,20] over (320,640,1280) channels = 64 per head."""
        ch = tuple(cfg["block_out_channels"])
        tl = cfg.get("transformer_layers_per_block", 1)
        tl = tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * len(ch)
        down = cfg.get("down_block_types", ["CrossAttnDownBlock2D"] * len(ch))
        tl = tuple(n if "CrossAttn" in t else 0 for n, t in zip(tl, down))
        heads = cfg.get("attention_head_dim", 8)
        heads = tuple(heads) if isinstance(heads, (list, tuple)) else (heads,) * len(ch)
        hd = {c // h for c, h, n in zip(ch, heads, tl) if n}
        if hd != {64}:
            raise ValueError(f"attention head size {sorted(hd)}: the attention kernel is built for 64 (SDXL)")
        add = cfg.get("addition_time_embed_dim", 256)
        return cls(in_channels=cfg.get("in_channels", 4), out_channels=cfg.get("out_channels", 4), block_out_channels=ch,
                   layers_per_block=cfg.get("layers_per_block", 2), transformer_layers=tl, head_dim=64,
                   cross_dim=cfg.get("cross_attention_dim", 2048),
                   pooled_dim=cfg.get("projection_class_embeddings_input_dim", 2816) - 6 * add,
                   addition_time_embed_dim=add, norm_groups=cfg.get("norm_num_groups", 32))


SDXL = UNetConfig()
TINY = UNetConfig(block_out_channels=(64, 128, 256), transformer_layers=(0, 1, 2), cross_dim=128,
                  pooled_dim=64, addition_time_embed_dim=32)


def transformer_sites(cfg: UNetConfig):
    """[(t2d prefix, channels, n_layers)] in forward order (down, mid, up)."""
    out = []
    ch = cfg.block_out_channels
    nb = len(ch)
    for bi in range(nb):
        for j in range(cfg.layers_per_block):
            if cfg.transformer_layers[bi]:
                out.append((f"down_blocks.{bi}.attentions.{j}", ch[bi], cfg.transformer_layers[bi]))
    out.append(("mid_block.attentions.0", ch[-1], cfg.transformer_layers[-1]))
    for ui in range(nb):
        bi = nb - 1 - ui
        for j in range(cfg.layers_per_block + 1):
            if cfg.transformer_layers[bi]:
                out.append((f"up_blocks.{ui}.attentions.{j}", ch[bi], cfg.transformer_layers[bi]))
    return out


def attention_blocks(cfg: UNetConfig):
    """every '<t2d>.transformer_blocks.i' prefix with its width, forward order."""
    return [(f"{p}.transformer_blocks.{i}", c) for p, c, n in transformer_sites(cfg) for i in range(n)]


# =====================================================================================
class UNetWeights:
    """Device-resident weights in kernel layouts, built from a diffusers-keyed state dict.

    concepts: None | ('custom', [sd_i]) with keys '<tb>.attn2.to_k.weight' / '.to_v.weight'
              (fusion_sampling.py:203-210) | ('lora', [sd_i]) with keys
              '<tb>.attnN.processor.to_{q,k,v,out}_lora.{down,up}.weight' (fusion_sampling_lora.py:207-210).
    """

    def __init__(self, cfg: UNetConfig, sd: dict, device="cuda", concepts=None, lora_mode="merged"):
        """lora_mode (kind 'lora' only): "merged" -- one bf16 weight set W + up @ down per concept and projection, picked per batch row
        by the batched GEMM (the default: fastest; +4.9 GB at K = 3, and a delta below ulp_bf16(W) is only dithered into W) -- or
        "lowrank" -- the reference's own form up(down(x)) (utils_lora.py:68,76-77,118): shared weights [W | U | 0] over K + 64 input
        columns, the rows' down-projections written into the 64 pad columns by tmix_lora_down in front of every routed projection."""
        self.cfg, self.device = cfg, torch.device(device)
        self.kind = concepts[0] if concepts else "none"
        self.K = len(concepts[1]) if concepts else 0
        assert lora_mode in ("merged", "lowrank")
        self.lora_mode = lora_mode if self.kind == "lora" else "merged"
        self._fp8 = {}
        self._tproj = None
        dev = self.device
        t = {}

        def g(name):
            return sd[name].to(dev)

        def bf(x):
            return x.to(dev, BF16).contiguous()

        def f32(x):
            return x.to(dev, F32).contiguous()

        for name, v in sd.items():
            if name.endswith(".bias") or ".norm" in name and name.endswith(".weight") or name.startswith("conv_norm_out"):
                t[name] = f32(v)
        t["conv_in.weight"] = f32(g("conv_in.weight").permute(0, 2, 3, 1))
        t["conv_out.weight"] = bf(g("conv_out.weight").permute(0, 2, 3, 1))
        for name, v in sd.items():
            if not name.endswith(".weight") or name in ("conv_in.weight", "conv_out.weight"):
                continue
            if v.dim() == 4 and v.shape[-1] == 3:
                t[name] = bf(v.to(dev).permute(0, 2, 3, 1))
            elif v.dim() == 4:                                   # 1x1 conv_shortcut -> Linear
                t[name] = bf(v.to(dev).reshape(v.shape[0], v.shape[1]))
            elif v.dim() == 2 and ".attn" not in name and ".ff.net.0.proj" not in name:
                t[name] = bf(v)
        lora = concepts[1] if self.kind == "lora" else None
        custom = concepts[1] if self.kind == "custom" else None

        def merged(base, key, which):
            """[1+K, N, Kin]: row 0 base, row 1+i = W + up_i @ down_i (utils_lora.py:65-79,113-119)."""
            rows = [base.to(dev, F32)]
            for csd in lora:
                kd, ku = f"{key}.processor.to_{which}_lora.down.weight", f"{key}.processor.to_{which}_lora.up.weight"
                if kd in csd and ku in csd:
                    rows.append(rows[0] + csd[ku].to(dev, F32) @ csd[kd].to(dev, F32))
                else:       # the trainer's freeze_model='lora' checkpoints carry attn2 pairs only: a missing pair is a zero delta
                    rows.append(rows[0])
            return torch.stack(rows)

        def lowrank(key, base, a, whiches, norm=None):
            """the low-rank form of projection `key` (base weight [N, Kin], already LayerNorm-folded when norm is given): t[key + ".lr"] =
            [W | U | 0] bf16 [N, Kin + 64] and the stacked down matrices D [(1 + K) * P, Kin] (set 0 = base: zeros), P = 4 per fused
            projection in `whiches`; column Kin + (set * len(whiches) + c) * 4 + r pairs rank r of concept `set`'s projection c."""
            nw, nsets = len(whiches), 1 + len(lora)
            P = 4 * nw
            assert nsets * P <= 64, "low-rank LoRA: (1 + concepts) x 4 x fused projections must fit the 64 pad columns"
            N, Kin = base.shape
            n1 = N // nw
            U = torch.zeros(N, 64, device=dev, dtype=F32)
            Dm = torch.zeros(nsets * P, Kin, device=dev, dtype=F32)
            for si, csd in enumerate(lora):
                for c, which in enumerate(whiches):
                    kd, ku = f"{a}.processor.to_{which}_lora.down.weight", f"{a}.processor.to_{which}_lora.up.weight"
                    if kd in csd and ku in csd:
                        col = ((si + 1) * nw + c) * 4
                        U[c * n1:(c + 1) * n1, col:col + 4] = csd[ku].to(dev, F32)
                        Dm[(si + 1) * P + c * 4:(si + 1) * P + c * 4 + 4] = csd[kd].to(dev, F32)
            t[key + ".lr"] = torch.cat([base.to(dev, BF16), U.to(BF16)], dim=1).contiguous()
            if norm is not None:                      # T = LN(x) D^T through the GEMM's folded LayerNorm: see tmix_lora_down
                gm, bt = g(norm + ".weight").to(F32), g(norm + ".bias").to(F32)
                Dp = (Dm * gm).to(BF16)
                t[key + ".lr.D"], t[key + ".lr.dcolsum"], t[key + ".lr.dbias"] = Dp.contiguous(), Dp.float().sum(1).contiguous(), (Dm @ bt).contiguous()
            else:
                t[key + ".lr.D"] = Dm.to(BF16).contiguous()
            t[key + ".lr.P"] = P

        def fold(key, w, norm, bias=None):
            """store Linear(LayerNorm(.)) folded for tmix_gemm_desc.ln_*: W*gamma, its column sums and W@beta (+bias)."""
            wp, cs, tt = fold_layernorm(w.to(dev, F32), g(norm + ".weight"), g(norm + ".bias"), bias)
            t[key], t[key + ".colsum"], t[key + ".bias"] = wp, cs, tt

        for tb, _c in attention_blocks(cfg):
            a1, a2 = tb + ".attn1", tb + ".attn2"
            n1, n2, n3 = tb + ".norm1", tb + ".norm2", tb + ".norm3"
            fold(a1 + ".qkv", torch.cat([g(a1 + ".to_q.weight"), g(a1 + ".to_k.weight"), g(a1 + ".to_v.weight")]), n1)
            t[a1 + ".out"] = bf(g(a1 + ".to_out.0.weight"))
            fold(a2 + ".q", g(a2 + ".to_q.weight"), n2)
            t[a2 + ".out"] = bf(g(a2 + ".to_out.0.weight"))
            kv_rows = [torch.cat([g(a2 + ".to_k.weight"), g(a2 + ".to_v.weight")]).to(F32)]
            if custom is not None:                      # row 1+i: concept i's to_k / to_v (utils_custom.py:66-80)
                for csd in custom:      # a block the checkpoint does not cover keeps the base projection (`if name in single_st['unet']`, fusion_sampling.py:206-209)
                    ck = csd[a2 + ".to_k.weight"].to(dev, F32) if a2 + ".to_k.weight" in csd else kv_rows[0][:kv_rows[0].shape[0] // 2]
                    cv = csd[a2 + ".to_v.weight"].to(dev, F32) if a2 + ".to_v.weight" in csd else kv_rows[0][kv_rows[0].shape[0] // 2:]
                    kv_rows.append(torch.cat([ck, cv]))
            if lora is not None:
                mk = merged(g(a2 + ".to_k.weight"), a2, "k")
                mv = merged(g(a2 + ".to_v.weight"), a2, "v")
                kv_rows = [torch.cat([mk[i], mv[i]]) for i in range(mk.shape[0])]     # attn2 K / V: projected once per call kind (KVCache), merged in fp32
                if self.lora_mode == "lowrank":
                    lowrank(a1 + ".qkv", t[a1 + ".qkv"], a1, ("q", "k", "v"), n1)
                    lowrank(a1 + ".out", t[a1 + ".out"], a1, ("out",))
                    lowrank(a2 + ".q", t[a2 + ".q"], a2, ("q",), n2)
                    lowrank(a2 + ".out", t[a2 + ".out"], a2, ("out",))
                else:
                    fold(a1 + ".qkv_rows", torch.cat([merged(g(a1 + ".to_q.weight"), a1, "q"),
                                                      merged(g(a1 + ".to_k.weight"), a1, "k"),
                                                      merged(g(a1 + ".to_v.weight"), a1, "v")], dim=1), n1)
                    t[a1 + ".out_rows"] = bf(merged(g(a1 + ".to_out.0.weight"), a1, "out"))
                    fold(a2 + ".q_rows", merged(g(a2 + ".to_q.weight"), a2, "q"), n2)
                    t[a2 + ".out_rows"] = bf(merged(g(a2 + ".to_out.0.weight"), a2, "out"))
            t[a2 + ".kv_rows"] = bf(torch.stack(kv_rows))       # [1 or 1+K, 2C, cross]
            wp, cs, tt = fold_layernorm(g(tb + ".ff.net.0.proj.weight").to(F32), g(n3 + ".weight"), g(n3 + ".bias"),
                                        g(tb + ".ff.net.0.proj.bias"))
            t[tb + ".ff1"] = interleave_geglu(wp, None)[0].contiguous()
            csi, bi = interleave_geglu(cs[:, None], tt)
            t[tb + ".ff1.colsum"], t[tb + ".ff1.bias"] = csi[:, 0].contiguous(), bi.contiguous()
        self.t = t

    def __getitem__(self, k):
        return self.t[k]

    def stacked_time_proj(self):
        """(weight [sum Co, T] bf16, bias [sum Co] fp32, section starts int32 [n+1] on the device, resnet names) of all
        `<resnet>.time_emb_proj` layers (diffusers ResnetBlock2D.time_emb_proj), stacked once."""
        if self._tproj is None:
            names = sorted(k[:-len(".time_emb_proj.weight")] for k in self.t if k.endswith(".time_emb_proj.weight"))
            w = torch.cat([self[n + ".time_emb_proj.weight"] for n in names], 0).contiguous()
            b = torch.cat([self[n + ".time_emb_proj.bias"].float() for n in names], 0).contiguous()
            st = [0]
            for n in names:
                st.append(st[-1] + self[n + ".time_emb_proj.weight"].shape[0])
            self._tproj = (w, b, torch.tensor(st, device=w.device, dtype=torch.int32), names)
        return self._tproj

    def conv2_with_shortcut(self, name):
        """(weight [Co, 9*Co + Ci] bf16, bias [Co] fp32) of resnet `name`: conv2's taps followed by conv_shortcut's 1x1 weights, and the sum of the two
        biases -- the operands of a tmix_conv3x3_nhwc launch with shortcut taps (ResnetBlock2D.forward: conv2(h) + conv_shortcut(x)); built once."""
        k = name + ".conv2+shortcut"
        if k + ".weight" not in self.t:
            self.t[k + ".weight"] = ops.shortcut_weight(self.t[name + ".conv2.weight"], self.t[name + ".conv_shortcut.weight"])
            self.t[k + ".bias"] = (self.t[name + ".conv2.bias"] + self.t[name + ".conv_shortcut.bias"]).contiguous()
        return self.t[k + ".weight"], self.t[k + ".bias"]

    def fp8(self, key, rows=None):
        """(e4m3 bytes, E8M0 row scales) of weight `key` ([N,K], or the [R,N,K] stack gathered by `rows`), quantised once per
        tensor by tmix_quantize_fp8_rows -- the operands of tmix_gemm_fp8 (`--dtype fp8`)."""
        ck = (key, None if rows is None else tuple(rows))
        if ck not in self._fp8:
            w = self.t[key]
            if w.dim() == 4 and rows is None:         # conv weight [Cout, 3, 3, Cin]: one scale per output channel over its 9 * Cin values (tmix_conv3x3_nhwc_fp8)
                w = w.reshape(w.shape[0], -1)
            if rows is not None and list(rows) != list(range(w.shape[0])):
                w = w[torch.tensor(list(rows), device=w.device)]
            self._fp8[ck] = ops.quantize_fp8_rows(w.contiguous())
        return self._fp8[ck]

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values() if torch.is_tensor(v))      # (lowrank keeps the Python int '.lr.P' beside the tensors)


# =====================================================================================
class KVCache:
    """Cross-attention K and V^T for one call kind: for every attn2 module, K [B,77,C] and
    V^T [B,C,80], where batch row b was projected with weight-set wsel[b] (0 = base UNet, 1+i =
    concept i) -- the time-invariant part of utils_custom.py:64-83 hoisted out of the loop."""

    def __init__(self, W: UNetWeights, ehs: torch.Tensor, wsel):
        cfg = W.cfg
        B, Lk, cross = ehs.shape
        assert cross == cfg.cross_dim and len(wsel) == B
        ehs = ehs.to(W.device, BF16).contiguous()
        self.B, self.Lk = B, Lk
        self.ld = (Lk + 7) // 8 * 8
        self.k, self.vt = {}, {}
        idx = torch.tensor(list(wsel), device=W.device)
        for tb, Cc in attention_blocks(cfg):
            a2 = tb + ".attn2"
            rows = W[a2 + ".kv_rows"]
            wsel_w = rows[idx] if rows.shape[0] > 1 else rows[:1].expand(B, -1, -1)
            wsel_w = wsel_w.contiguous()
            k = torch.empty(B, Lk, Cc, device=W.device, dtype=BF16)
            vt = torch.zeros(B, Cc, self.ld, device=W.device, dtype=BF16)
            ops.gemm(ehs, wsel_w, out=k, out_t=vt, n_trans_begin=Cc)
            self.k[a2], self.vt[a2] = k, vt
        torch.cuda.synchronize()


# =====================================================================================
class _Arena:
    """Size-keyed free list: the plan is a static, stream-ordered launch sequence, so a buffer released
    at plan position i can be handed to any op planned after i."""

    def __init__(self, device):
        self.device, self.free, self.total = device, {}, 0
        self.bufs = []          # owns every buffer for the plan's lifetime (views handed out are not owners)

    def get(self, *shape, dtype=BF16):
        n = 1
        for s in shape:
            n *= s
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
        lst = self.free.get(nbytes)
        if lst:
            buf = lst.pop()
        else:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.bufs.append(buf)
            self.total += nbytes
        tns = buf.view(dtype)[:n].view(*shape)
        tns._arena_buf = buf
        return tns

    def put(self, *ts):
        for tns in ts:
            buf = tns._arena_buf
            self.free.setdefault(buf.numel(), []).append(buf)
            for cs, _c in getattr(tns, "_cs", None) or ():  # GroupNorm column partials of this tensor (UNetPlan._colstats): every reader came before
                self.put(cs)
            tns._cs = None


_TUNE_CACHE = {}      # repr((kind, shape key)) -> best TMIX_TILE_* id, shared by every plan in the process
# The same launch wants a different tiling when its chain has the chip to itself than when a sibling chain runs beside it
# (PlanGroup): alone, the tilings that put exactly one workgroup on each of the 256 CUs win by 15-35 % (L.TILE_EXCLUSIVE:
# 64x160 / 32x160 over five waves, 256x320); next to another chain those own every CU while they run and whole steps come
# out 3 % slower than with 128x128 / 128x160 grids that leave CUs to the sibling.  Entries measured for a chain that shares
# the chip carry this prefix; a shape without one falls back to the plain entry unless that is an exclusive tiling.
SHARED = "shared|" 
# A third context since round 6: the launches of a plan whose projections read one weight set PER BATCH ROW (the LoRA-routed fusion step).  The same launch shape
# can want another tiling there: FF2 on the 2 x 2-wave tiling 23 is 0.46 ms per step FASTER in the Custom-Diffusion step and 0.30 ms SLOWER in the routed LoRA step
# (three interleaved rounds on one box, profiles/r6_experiments/table_variants_routed_context.txt) -- behind it sits a q/k/v launch with four 9.8 MB weight sets, and
# tiling 23's four loader waves queue that launch's weight touches in front of their first K-tile.  Entries with this prefix win for routed plans; without one the plain entry applies.
ROUTED = "routed|"
_TUNE_FILE = os.environ.get("TMIX_TUNE_FILE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json"))


def load_tune_table(path=None):
    """seed the tiling cache from a JSON table {repr(shape key): TMIX_TILE_* id} measured on an MI355X by
    tools/make_tune_table.py (the shipped tweediemix_amd/tuned_gfx950.json; TMIX_TUNE_FILE overrides, '' disables).
    Shapes missing from the table are still timed in situ when a plan is built."""
    path = _TUNE_FILE if path is None else path
    if path and os.path.exists(path):
        with open(path) as f:
            _TUNE_CACHE.update({k: int(v) for k, v in json.load(f).items()})
    return len(_TUNE_CACHE)


def save_tune_table(path):
    with open(path, "w") as f:
        json.dump(dict(sorted(_TUNE_CACHE.items())), f, indent=0)


def tune_lookup(ctx, key):
    """tiling for launch shape `key` in context `ctx` ("" = the chain has the chip to itself, SHARED = a sibling chain runs
    beside it), or None when it must be measured: a shared chain falls back to the plain entry (tables written before the
    contexts existed) unless that entry is a one-workgroup-per-CU tiling, which must not be inherited."""
    c = _TUNE_CACHE.get(ctx + key)
    if c is None and ctx:
        c = _TUNE_CACHE.get(key)
        if ctx == SHARED and c in L.TILE_EXCLUSIVE:
            c = None
    return c


load_tune_table()


def hint_policy(B, h, w):
    """(cap, over) in bytes for the weight hints of a UNet call of B latents of h x w: tensors larger than `over` are named by their first `cap`
    bytes (cap 0: whole tensors).  See UNetPlan.__init__ for the measurements behind the defaults; TMIX_PF_CAP_MB / TMIX_PF_CAP_OVER_MB override."""
    small_call = B * h * w <= 4 * 128 * 128
    cap = int(float(os.environ.get("TMIX_PF_CAP_MB", "8" if small_call else "0")) * (1 << 20))
    over = int(float(os.environ.get("TMIX_PF_CAP_OVER_MB", "20")) * (1 << 20))
    return cap, over


def hint_bytes(nbytes, cap, over):
    """bytes of an `nbytes` weight tensor that the launch in front of its consumer touches (tmix_gemm_prefetch_next)"""
    return min(nbytes, cap) if cap and nbytes > over else nbytes


class UNetPlan:
    """One UNet call shape: batch B, latent h x w, a KV cache (prompt rows) and a routing flag.

    inputs  : self.latent [B,4,h,w] fp32, self.t_dev [B] fp32 (timestep, same value in every row)
    output  : self.eps [B,4,h,w] fp32
    """

    def __init__(self, W: UNetWeights, B: int, h: int, w: int, kv: KVCache, pooled: torch.Tensor,
                 time_ids: torch.Tensor, routed: bool = False, autotune: bool = True, row_sets=None,
                 latent=None, eps=None, shared: bool = False, t_dev=None, fp8: bool = False):
        self.W, self.cfg, self.B, self.h, self.w = W, W.cfg, B, h, w
        # fp8: the attn1 q/k/v, FF up- and down-projections run as tmix_gemm_fp8 (e4m3 operands, per-row power-of-two scales);
        # their A operand is quantised by one tmix_quantize_fp8_rows launch in front of the GEMM
        self.fp8 = bool(fp8)
        self.fp8_chain_ff = not os.environ.get("TMIX_FP8_FF_ROWS")    # =1: quantise the FF intermediate per row with a separate launch
        self.fp8_attn_out = not os.environ.get("TMIX_FP8_NO_ATTN_OUT")  # =1: attention output in bf16, out-projections on bf16 operands (round 3)
        self.fp8_conv = not os.environ.get("TMIX_FP8_NO_CONV")         # =1: every convolution on bf16 operands (before round 4's tmix_conv3x3_nhwc_fp8)
        self.fp8_conv_tile = int(os.environ.get("TMIX_FP8_CONV_TILE", "20"))   # 128 x 160 with two loader waves (12: without)
        self.fp8_tile = int(os.environ.get("TMIX_FP8_TILE", "21"))      # 128 x 160 e4m3 tiling of the N = 1280 / 640 launches (21: loader waves, 12: none, 0: phase-offset only)
        self.kv = kv
        # LoRA routing: batch row b uses merged weight set row_sets[b] (default: row b of a single seed)
        self.row_sets = list(row_sets) if row_sets is not None else list(range(B))
        self.routed = bool(routed) and W.kind == "lora" and len(self.row_sets) == B and max(self.row_sets) <= W.K
        # tile-table context: this chain runs beside a sibling chain (PlanGroup member) / its projections are concept-routed / neither
        self.tune_ctx = SHARED if shared else (ROUTED if (self.routed and getattr(W, "lora_mode", "merged") != "lowrank") else "")
        # low-rank LoRA (UNetWeights.lora_mode): the routed projections run on SHARED weights over K + 64 input columns; the concept of a
        # batch row only decides what tmix_lora_down writes into the pad columns of its rows
        self.lowrank = self.routed and getattr(W, "lora_mode", "merged") == "lowrank"
        if self.lowrank:
            assert not self.fp8, "low-rank LoRA runs the bf16 projections only"
            self.routed = False
            self._sets_dev = torch.tensor(self.row_sets, device=W.device, dtype=torch.int32)
        self._rows_cache = {}
        self.lib = L.load()
        self.dev = W.device
        self.ops = []
        self.keep = []                      # descriptors / tensors that must outlive the plan
        self.arena = _Arena(self.dev)
        self.flops = 0
        self.gemm_flops = 0
        self.launches = {"gemm": [], "conv": [], "attn": []}
        self.op_meta = {}                   # index into self.ops -> (class, flops, shape key) of the instrumented launches
        cfg = self.cfg
        assert kv.B == B
        dev = self.dev
        # I/O buffers may be views into a larger batch owned by a PlanGroup (row-split execution on several streams)
        self.latent = latent if latent is not None else torch.zeros(B, cfg.in_channels, h, w, device=dev, dtype=F32)
        self.t_dev = t_dev if t_dev is not None else torch.zeros(B, device=dev, dtype=F32)
        self.eps = eps if eps is not None else torch.zeros(B, cfg.out_channels, h, w, device=dev, dtype=F32)
        assert self.latent.is_contiguous() and self.eps.is_contiguous() and self.latent.shape[0] == B
        # static conditioning: aug_emb = add_embedding(cat[pooled, sinusoid(time_ids)])  (depends on rows only)
        tid = ops.timestep_embedding(time_ids.to(dev, F32).reshape(-1).contiguous(), cfg.addition_time_embed_dim)
        add_in = torch.cat([pooled.to(dev, F32), tid.reshape(B, -1)], dim=-1).contiguous()
        hid = ops.linear_small(add_in, W["add_embedding.linear_1.weight"], W["add_embedding.linear_1.bias"], act_out=True)
        self.aug = ops.linear_small(hid, W["add_embedding.linear_2.weight"], W["add_embedding.linear_2.bias"])
        torch.cuda.synchronize()
        self._gn_ws = ops.groupnorm_ws(B, 4096, cfg.norm_groups, dev)
        self._vt = {}
        # LayerNorm row statistics travel from the GEMM that writes the hidden state to the GEMM behind the norm as
        # per-column-tile partial sums [parts][B*S][2]; launches on one stream are ordered, so all sites share one buffer
        nb = len(cfg.block_out_channels)
        need = 1
        for pfx, cc, _n in transformer_sites(cfg):
            lvl = nb - 1 if pfx.startswith("mid") else int(pfx.split(".")[1]) if pfx.startswith("down") else nb - 1 - int(pfx.split(".")[1])
            need = max(need, ((cc + 127) // 128) * B * (h >> lvl) * (w >> lvl) * 2)
        self._ln_buf = torch.zeros(need, device=dev, dtype=F32)
        # every GEMM / conv launch is preceded by a tmix_gemm_prefetch_next hint naming the weights of the launch AFTER it (patched in
        # when that launch is planned): the chain otherwise meets every weight cold from HBM (TMIX_NO_PREFETCH=1 switches it off)
        self._pf_prev = None
        self._pf_on = not os.environ.get("TMIX_NO_PREFETCH")
        # How much of a LARGE weight tensor a hint names (round 6, profiles/r6_experiments/weight_hint_cap_*.txt): a launch of a single-seed call lasts 20 - 90 us, and
        # touching the 39 MB of a routed q/k/v weight (or FF1's 26 MB) inside it costs the hinting launch more than the hinted one gains -- tensors over 20 MB
        # are named by their first 8 MB there (1 MB ... 12 MB measure the same; 13 MB tensors want all of themselves: capping those loses 0.5 - 1 ms).  The co-batched
        # calls (launches of 0.2 - 1 ms) keep whole-tensor hints.  TMIX_PF_CAP_MB / TMIX_PF_CAP_OVER_MB override (cap 0: whole tensors everywhere).
        self._pf_cap, self._pf_cap_over = hint_policy(B, h, w)
        # GroupNorm statistics come from the launch that WRITES the normalised tensor (col_stats_out of the conv / proj_out epilogue), so a
        # norm is two launches (combine partials, apply) and one pass over x instead of three and two (TMIX_GN_STATS_KERNEL=1: the old form)
        self._gn_fused = not os.environ.get("TMIX_GN_STATS_KERNEL")
        # conv_shortcut rides in conv2's K loop (tmix_conv_desc.S1 / S2): no shortcut GEMM, and the up-blocks' concatenations are never written
        # (TMIX_SHORTCUT_GEMM=1: the separate GEMM + concat launches)
        self._sc_fused = not os.environ.get("TMIX_SHORTCUT_GEMM")
        # attn2.to_q and the 77-key cross-attention behind it as ONE launch (tmix_gemm_q_cross_attn: no q round trip, 70 launches fewer per call); bf16 plans
        # with the merged / shared weights, five-head tiles (C % 320 == 0), <= 80 cached keys (TMIX_NO_QATTN=1: the two-launch form)
        self._qattn = not os.environ.get("TMIX_NO_QATTN") and not self.fp8 and not self.lowrank and kv.Lk <= 80 and kv.ld == 80
        self._tunable = []                  # (index into self.ops, kind, descriptor) of every GEMM / conv launch
        self._ln_links = []                 # (producer desc, [consumer descs]): ln_parts follows the producer's tiling
        self._build()
        self._link_ln()
        if autotune:
            self.autotune()

    @staticmethod
    def _tune_key(kind, d):
        if kind == "gemm":
            return repr((kind, d.M, d.N, d.K, d.batch, d.epilogue, d.n_trans_begin >= 0, bool(d.residual), d.strideW != 0,
                         bool(d.row_stats_out), bool(d.ln_stats)))
        csc = d.S1_channels + d.S2_channels          # shortcut taps ride in the K loop (K = 9 Cin + csc): a shape of its own
        return repr((kind, d.B, d.H, d.W, d.Cin, d.Cout, d.mode) + ((csc,) if csc else ()))

    def autotune(self, reps=None):
        """pick the fastest workgroup tiling (TMIX_TILE_*) per distinct GEMM / conv shape (the shapes of this path are
        small and awkward -- M=4096, N=1280 -- so tile quantisation over 256 CUs, not peak MFMA rate, decides).
        Candidates are timed IN SITU: the whole forward runs once per candidate with every tunable launch bracketed by
        events, so each launch sees the cache state it meets in the real sequence (weights cold from HBM, activations
        fresh from the previous kernel); back-to-back replays of one launch rank the tilings differently and picked a
        mix that lost 4 % to the best single tiling.  Descriptors are patched in place; choices are cached per shape."""
        reps = reps or int(os.environ.get("TMIX_TUNE_REPS", "3"))
        tun = [(i, kind, d) for i, kind, d in self._tunable]
        force = int(os.environ.get("TMIX_FORCE_TILE", "0"))          # debugging / sensitivity studies
        if force:
            for _i, _k, d in tun:
                d.tile_cfg = force
            self._link_ln()
            return
        keys = [self._tune_key(kind, d) for _i, kind, d in tun]
        ctx = getattr(self, "tune_ctx", "")
        lookup = lambda k: tune_lookup(ctx, k)
        if any(lookup(k) is None for k in keys):
            idx = {i: n for n, (i, _k, _d) in enumerate(tun)}
            st = torch.cuda.current_stream().cuda_stream
            best_t = {}                                             # (key, cfg) -> min over reps of the summed launch times
            cands = list(L.TILE_CANDIDATES)                         # (16 / 17 and the loader-wave tilings 19 / 20 exist for the plain GEMM only)
            conv_alias = {16: 4, 17: 2, 18: 12, 19: 12, 21: 12, 22: 14, 23: 12, 24: 14, 25: 12}    # what gemm_conv.hip runs for a convolution: timed once, under the live id
            for _rep in range(reps):
                for cfg in cands:
                    for _i, kind, d in tun:
                        d.tile_cfg = cfg if kind == "gemm" else conv_alias.get(cfg, cfg)
                    self._link_ln()
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in tun]
                    for i, (fn, args) in enumerate(self.ops):
                        n = idx.get(i)
                        if n is not None:
                            ev[n][0].record()
                        rc = fn(*args, st)
                        if rc:
                            L.check(rc, fn.__name__)
                        if n is not None:
                            ev[n][1].record()
                    torch.cuda.synchronize()
                    tot = {}
                    for n, k in enumerate(keys):
                        tot[k] = tot.get(k, 0.0) + ev[n][0].elapsed_time(ev[n][1])
                    for k, t in tot.items():
                        best_t[(k, cfg)] = min(best_t.get((k, cfg), float("inf")), t)
            for k in set(keys):
                ok = [c for c in cands if k.startswith("('gemm'") or c not in conv_alias]
                if k not in _TUNE_CACHE:
                    _TUNE_CACHE[k] = min(ok, key=lambda c: best_t[(k, c)])
                if SHARED + k not in _TUNE_CACHE:     # starting point for chains that share the chip (refine_group re-ranks under load)
                    _TUNE_CACHE[SHARED + k] = min([c for c in ok if c not in L.TILE_EXCLUSIVE], key=lambda c: best_t[(k, c)])
        for (_i, _kind, d), k in zip(tun, keys):
            d.tile_cfg = lookup(k)
        self._link_ln()
        torch.cuda.synchronize()

    # ------------------------------------------------------------------ op emitters
    def _emit(self, fn, *args):
        self.ops.append((fn, args))

    def _hint_weights(self, w):
        """called when a GEMM / conv launch over weight tensor `w` is planned, BEFORE its op is emitted: names `w` in the previous
        launch's prefetch hint and opens this launch's own hint slot."""
        if not self._pf_on:
            return
        if self._pf_prev is not None:
            self._pf_prev[0], self._pf_prev[1] = w.data_ptr(), hint_bytes(w.numel() * w.element_size(), self._pf_cap, self._pf_cap_over)
        self._pf_prev = [None, 0]
        self.keep.append(w)
        self.ops.append((self.lib.tmix_gemm_prefetch_next, self._pf_prev))

    def _colstats(self, owner, rows, HW, Cc):
        """column-partials buffer [rows/32, 2, Cc] for the launch that writes `owner` ([B, HW, Cc]); it travels with the tensor (owner._cs, released
        with it) and _gn picks it up.  None when the geometry does not allow it (32-row blocks must not straddle images)."""
        if not getattr(self, "_gn_fused", False) or HW % ops.COLSTATS_ROWS or HW > ops.COLSTATS_MAX_HW or rows % ops.COLSTATS_ROWS or Cc % 8:
            return None
        cs = self.arena.get(rows // ops.COLSTATS_ROWS, 2, Cc, dtype=F32)
        owner._cs = ((cs, Cc),)
        return cs

    def _gn_f8_ok(self, x, Cc, HW, x2=None):
        """can the GroupNorm over x (| x2) leave its result as e4m3 + row-major MX scales for tmix_conv3x3_nhwc_fp8?  fp8 plans, Cc % 128 == 0, and the
        statistics must come from the producers (the e4m3 output exists for tmix_groupnorm_nhwc_pre only)"""
        if not (self.fp8 and getattr(self, "fp8_conv", False)) or Cc % 128 or HW % 32:
            return False
        parts = getattr(x, "_cs", None)
        if x2 is not None:
            p2 = getattr(x2, "_cs", None)
            parts = (parts[0], p2[0]) if parts and p2 and len(parts) == 1 and len(p2) == 1 else None
        return bool(parts) and sum(c for _t, c in parts) == Cc

    def _gn(self, x, Cc, HW, name, eps, silu, out=None, x2=None, f8=False):
        """x2: a second tensor normalised as the channel-concatenation [x | x2] (Cc counts both).
        f8 (only when _gn_f8_ok): returns (e4m3 bytes [B, HW, Cc], scales [B * HW, Cc / 32]) instead of the bf16 tensor."""
        if f8:
            out = (self.arena.get(self.B, HW, Cc, dtype=torch.uint8), self.arena.get(self.B * HW, Cc // 32, dtype=torch.uint8))
        else:
            out = out if out is not None else self.arena.get(self.B, HW, Cc)
        W = self.W
        C2 = 0 if x2 is None else x2.shape[-1]
        parts = getattr(x, "_cs", None)
        if x2 is not None:
            p2 = getattr(x2, "_cs", None)
            parts = (parts[0], p2[0]) if parts and p2 and len(parts) == 1 and len(p2) == 1 else None
        if parts and sum(c for _t, c in parts) == Cc and HW % ops.COLSTATS_ROWS == 0:     # (video plans keep partials whose 32-row blocks fit the CLIP but not the frame)
            (cs1, c1), (cs2, c2) = (parts[0], parts[1]) if len(parts) == 2 else (parts[0], (None, 0))
            if f8:
                self._emit(self.lib.tmix_groupnorm_nhwc_pre_f8, x.data_ptr(), Cc - C2, x2.data_ptr() if C2 else None, C2, out[0].data_ptr(), out[1].data_ptr(),
                           W[name + ".weight"].data_ptr(), W[name + ".bias"].data_ptr(), self._gn_ws.data_ptr(), self.B, HW, self.cfg.norm_groups, eps, int(silu),
                           cs1.data_ptr(), c1, cs2.data_ptr() if cs2 is not None else None, c2)
            else:
              self._emit(self.lib.tmix_groupnorm_nhwc_pre, x.data_ptr(), Cc - C2, x2.data_ptr() if C2 else None, C2, out.data_ptr(), W[name + ".weight"].data_ptr(),
                       W[name + ".bias"].data_ptr(), self._gn_ws.data_ptr(), self.B, HW, self.cfg.norm_groups, eps, int(silu),
                       cs1.data_ptr(), c1, cs2.data_ptr() if cs2 is not None else None, c2)
        else:
            assert not f8
            self._emit(self.lib.tmix_groupnorm_nhwc, x.data_ptr(), Cc - C2, x2.data_ptr() if C2 else None, C2, out.data_ptr(), W[name + ".weight"].data_ptr(),
                       W[name + ".bias"].data_ptr(), self._gn_ws.data_ptr(), self.B, HW, self.cfg.norm_groups, eps, int(silu))
        self.op_meta[len(self.ops) - 1] = ("norm", 0, ("norm", self.B, HW, Cc))
        return out

    def _gemm(self, a, w, out, fp8_key=None, f8_copy=None, a8=None, **kw):
        """f8_copy: an ops.F8Copy that also receives the e4m3 + MX-block copy of the stored rows (fp8 plans: the next projection's A
        operand); a8: such a copy of `a`, used instead of a quantiser launch when this GEMM runs on fp8 operands."""
        if fp8_key is not None and self.fp8:
            if a8 is not None:
                a = (a8.q.view(*a.shape), a8.scales)
            return self._gemm_fp8(a, fp8_key, out, f8_copy=f8_copy, **kw)
        if kw.get("row_stats_out") is not None:
            kw.setdefault("tile_cfg", 1)            # the partial count depends on the tiling: never TMIX_TILE_AUTO
        cs_owner = kw.pop("cs_owner", None)         # the [B, HW, N] tensor `out` is a view of: a GroupNorm reads it next
        if cs_owner is not None and out.dim() == 2:
            kw["col_stats_out"] = self._colstats(cs_owner, out.shape[0], cs_owner.shape[1], out.shape[1])
        d = ops.make_gemm_desc(a, w, out, **kw)
        if f8_copy is not None:
            f8_copy.attach(d)
            self.keep.append(f8_copy)
        if kw.get("row_stats_out") is not None:
            self._ln_links.append((d, []))
        elif kw.get("ln_stats") is not None:
            self._ln_links[-1][1].append(d)
        self.keep.append(d)
        self._hint_weights(w)
        self._emit(self.lib.tmix_gemm_bf16, C.byref(d))
        fl = 2 * d.M * d.N * d.K * d.batch
        self.flops += fl
        self.gemm_flops += fl
        self.launches["gemm"].append((d, fl))
        self._tunable.append((len(self.ops) - 1, "gemm", d))
        self.op_meta[len(self.ops) - 1] = ("gemm", fl, d)
        return out

    def _gemm_fp8(self, a, key, out, f8_out=None, f8_copy=None, **kw):
        """tmix_gemm_fp8 against the cached e4m3 copy of weight `key` (key = (name, row_sets) for per-row LoRA weight sets).
        a: bf16 rows -- quantised by one tmix_quantize_fp8_rows launch in front of the GEMM -- or a pair (e4m3 bytes, MX block
        scales [K/32][rows]) as left by a GEMM planned with f8_out=(bytes [M][N/2], scales [N/64][M]) (GEGLU epilogue only):
        FF1 -> FF2 without a quantiser pass and with half the bytes of the bf16 intermediate."""
        name, rows = key if isinstance(key, tuple) else (key, None)
        w8, sw = self.W.fp8(name, rows)
        flags = 0
        if isinstance(a, tuple):
            a8, sa = a
            flags |= L.F8_A_BLOCK_SCALES
            owned = ()
        else:
            K = a.shape[-1]
            a8 = self.arena.get(*a.shape, dtype=torch.uint8)
            sa = self.arena.get(*a.shape[:-1], dtype=torch.uint8)
            a2 = a.reshape(-1, K)
            assert a2.data_ptr() == a.data_ptr() and a2.stride(1) == 1          # a view, rows contiguous in K
            self._emit(self.lib.tmix_quantize_fp8_rows, a2.data_ptr(), a2.stride(0), a8.data_ptr(), K, sa.data_ptr(), a2.shape[0], K)
            owned = (a8, sa)
        # the N = 1280 / 640 launches (attention out-projections, attn2 to_q, FF2): 128 x 160 tiles fill the chip where 256 x 128 leaves 96 CUs idle, and
        # on e4m3 operands the lock-step loop with loader waves runs FF2 at 1.7 PFLOP/s (31 us against 45.5 for the phase-offset 256 x 128, 51.5 in bf16)
        Ng, Kg = w8.shape[-2], w8.shape[-1]
        # ... where the launch is one round of them or K is long; with more rows (co-batched seeds: B = 8 / 16, the 64 x 64 level) the phase-offset 256 x 256
        # tiling wins the K <= 1280 launches again (tools/cobatch_tiles.py: B = 16 cube 50.8 vs 55.3 us, 640-cube 55.7 vs 69.0; B = 4 cube 22.3 vs 16.9)
        rows_g = a8.shape[-2] * (a8.shape[0] if a8.dim() == 3 else 1)
        lock = self.fp8_tile and f8_out is None and kw.get("out_t") is None and Kg % 128 == 0 and Ng % 160 == 0 and Ng <= 1280
        many = not (Kg >= 2560 or rows_g * Ng <= 4096 * 1280)
        if lock and (not many or kw.get("row_stats_out") is not None):       # (with the row statistics + e4m3 copy epilogue the 128 x 160 tiles stay ahead: 64 vs 72.5 us at B = 16)
            kw.setdefault("tile_cfg", self.fp8_tile)
        if kw.get("row_stats_out") is not None:
            kw.setdefault("tile_cfg", 17)           # explicit (the partial count depends on it)
        d = ops.make_gemm_desc(a8, w8, None if f8_out is not None else out, **kw)
        if f8_out is not None:
            c8, cs = f8_out
            assert kw.get("geglu") and c8.dim() == 2 and c8.stride(1) == 1 and cs.is_contiguous()
            d.C, d.ldc, d.strideC = c8.data_ptr(), c8.stride(0), 0
            d.Ct, d.ldct = cs.data_ptr(), cs.shape[1]
            flags |= L.F8_GEGLU_OUT
        d.reserved0 = flags
        if f8_copy is not None:
            f8_copy.attach(d)
            self.keep.append(f8_copy)
        if kw.get("row_stats_out") is not None:
            self._ln_links.append((d, []))
        elif kw.get("ln_stats") is not None:
            self._ln_links[-1][1].append(d)
        self.keep += [d, a8, sa]
        self._hint_weights(w8)
        self._emit(self.lib.tmix_gemm_fp8, C.byref(d), sa.data_ptr(), sw.data_ptr())
        fl = 2 * d.M * d.N * d.K * d.batch
        self.flops += fl
        self.gemm_flops += fl
        self.launches["gemm"].append((d, fl))
        self.op_meta[len(self.ops) - 1] = ("gemm_fp8", fl, d)       # its own class: priced against the fp8 MFMA peak (bench.py)
        if owned:
            self.arena.put(*owned)                  # stream-ordered: free for ops planned after this GEMM
        return out

    def _conv(self, x, wname, Hh, Ww, Cin, Cout, mode=L.CONV_S1, batch_bias=None, residual=None, bias_images=1, shortcut=None):
        """shortcut: (resnet name, x1, x2 or None) -- the block's input(s), whose 1x1 conv_shortcut rides in this launch's K loop"""
        Ho, Wo = ops.conv_out_hw(Hh, Ww, mode)
        out = self.arena.get(self.B, Ho * Wo, Cout)
        if isinstance(x, tuple):                        # (e4m3 bytes, row-major MX scales) from _gn(f8=True): tmix_conv3x3_nhwc_fp8
            assert shortcut is None
            x8, sx = x
            w8, sw = self.W.fp8(wname + ".weight")
            d = ops.make_conv_desc(x8.view(self.B, Hh, Ww, Cin), w8.view(Cout, 3, 3, Cin), out.view(self.B, Ho, Wo, Cout), self.W[wname + ".bias"], batch_bias,
                                   residual, mode, tile_cfg=self.fp8_conv_tile, bias_images=bias_images,
                                   col_stats_out=self._colstats(out, self.B * Ho * Wo, Ho * Wo, Cout), _fp8=True)
            self.keep += [d, x8, sx, w8, sw]
            self._hint_weights(w8)
            self._emit(self.lib.tmix_conv3x3_nhwc_fp8, C.byref(d), sx.data_ptr(), sw.data_ptr())
            fl = 2 * self.B * Ho * Wo * Cout * 9 * Cin
            self.flops += fl
            self.launches["conv"].append((d, fl))
            self.op_meta[len(self.ops) - 1] = ("conv_fp8", fl, d)
            return out
        w, bias, sc, csc = self.W[wname + ".weight"], self.W[wname + ".bias"], None, 0
        if shortcut is not None:
            w, bias = self.W.conv2_with_shortcut(shortcut[0])
            sc = (shortcut[1], shortcut[2])
            csc = w.shape[1] - 9 * Cin
        d = ops.make_conv_desc(x.view(self.B, Hh, Ww, Cin), w, out.view(self.B, Ho, Wo, Cout),
                               bias, batch_bias, residual, mode, bias_images=bias_images,
                               col_stats_out=self._colstats(out, self.B * Ho * Wo, Ho * Wo, Cout),      # every conv output of this network feeds a GroupNorm
                               shortcut=sc)
        self.keep.append(d)
        self._hint_weights(w)
        self._emit(self.lib.tmix_conv3x3_nhwc, C.byref(d))
        fl = 2 * self.B * Ho * Wo * Cout * (9 * Cin + csc)
        self.flops += fl
        self.launches["conv"].append((d, fl))
        self._tunable.append((len(self.ops) - 1, "conv", d))
        self.op_meta[len(self.ops) - 1] = ("conv", fl, d)
        return out

    def _ln_stats(self, S, Cc):
        """fp32 [parts_max, B*S, 2] view of the shared row-statistics buffer for one LayerNorm site: written by the GEMM
        that stores the hidden state (row_stats_out), read by the GEMM behind the norm (ln_stats)."""
        pm = (Cc + 127) // 128
        return self._ln_buf[:pm * self.B * S * 2].view(pm, self.B * S, 2)

    def _link_ln(self):
        for _i, kind, d in self._tunable:       # the tiling the library would substitute for a GEMM that leaves an e4m3 copy
            if kind == "gemm" and (d.reserved0 & L.F8_COPY_OUT):
                d.tile_cfg = L.F8COPY_TILE_ALT.get(d.tile_cfg, d.tile_cfg)
        for prod, cons in self._ln_links:
            parts = ops.stats_parts(prod.N, prod.tile_cfg)
            for c in cons:
                c.ln_parts = parts

    def _q_attn(self, h, key, k, vt, ao, S, Cc, ln):
        """attn2: to_q (LayerNorm folded, per-row merged weights when LoRA-routed) + cross-attention against the cached K / V^T in one launch"""
        W = self.W
        shp = (self.B, S) if self.routed else (self.B * S,)
        kw = {"ln_stats": ln, "ln_colsum": self._rows(key, ".colsum") if self.routed else W[key + ".colsum"],
              "bias": self._rows(key, ".bias") if self.routed else W[key + ".bias"]}
        w = self._rows(key) if self.routed else W[key]
        d = ops.make_gemm_desc(h.view(*shp, Cc), w, None, **kw)
        self._ln_links[-1][1].append(d)
        args = ops.q_cross_attn_args(k, vt, ao, S, self.cfg.head_dim ** -0.5)
        self.keep += [d, k, vt, ao]
        self._hint_weights(w)
        self._emit(self.lib.tmix_gemm_q_cross_attn, C.byref(d), *args)
        fl = 2 * d.M * d.N * d.K * d.batch
        fla = 4 * self.B * (Cc // 64) * S * k.shape[1] * 64
        self.flops += fl + fla
        self.gemm_flops += fl
        self.launches["gemm"].append((d, fl))
        self.op_meta[len(self.ops) - 1] = ("gemm", fl + fla, d)
        return ao

    def _attn(self, q, k, vt, out, H, Sq, Skv, f8_out=None):
        """f8_out: an ops.F8Copy that receives the output as e4m3 + MX block scales instead of the bf16 tensor `out` (fp8 plans: the out-projection's
        A operand without a quantiser launch and with half the bytes)"""
        # key-split tail (tmix_attn_fwd_ws), by request only (TMIX_ATTN_SPLIT=1): hot it takes 5 % off the S = 4096 launches and 2 % off S = 1024, in the
        # captured step nothing -- and a co-batched seed would no longer equal its single run bit for bit.  One zeroed workspace per
        # launch shape of this chain (launches of a chain are serial on its stream)
        ws = None
        if os.environ.get("TMIX_ATTN_SPLIT"):
            wss = getattr(self, "_attn_ws", None)
            if wss is None:
                wss = {}
                setattr(self, "_attn_ws", wss)
            wkey = (self.B, H, Sq, Skv)
            if wkey not in wss:
                wss[wkey] = ops.attention_split_ws(*wkey, q.device)
            ws = wss[wkey]
        wsa = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
        if f8_out is not None:
            args = (q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0),
                    vt.data_ptr(), vt.stride(1), vt.stride(0), f8_out.q.data_ptr(), f8_out.N, f8_out.scales.data_ptr(), f8_out.rows,
                    self.B, H, Sq, Skv, self.cfg.head_dim ** -0.5, *wsa)
            self._emit(self.lib.tmix_attn_fwd_f8_ws, *args)
            self.keep.append(f8_out)
        else:
            args = (q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0),
                    vt.data_ptr(), vt.stride(1), vt.stride(0), out.data_ptr(), out.stride(1), out.stride(0),
                    self.B, H, Sq, Skv, self.cfg.head_dim ** -0.5, *wsa)
            self._emit(self.lib.tmix_attn_fwd_ws, *args)
        fl = 4 * self.B * H * Sq * Skv * 64
        self.flops += fl
        self.launches["attn"].append((args, fl))
        self.op_meta[len(self.ops) - 1] = ("attn", fl, ("attn", self.B, H, Sq, Skv))
        return out

    def _vt_buf(self, Cc, S):
        ld = (S + 7) // 8 * 8
        key = (Cc, ld)
        if key not in self._vt:               # zero-filled once: the GEMM only writes columns < S
            self._vt[key] = torch.zeros(self.B, Cc, ld, device=self.dev, dtype=BF16)
        return self._vt[key]

    # ------------------------------------------------------------------ blocks
    def _time_bias(self, name, Co, emb):
        """(time_emb_proj(SiLU(emb)) rows for resnet `name` as conv1's batch_bias, consecutive images that share a row)"""
        temb = self._temb[name]                     # [B, Co] fp32: this block's section of the one stacked time_emb_proj launch
        assert temb.shape == (self.B, Co)
        return temb, 1

    def _sc_ok(self, Ci, Co, c1, c2=0):
        return getattr(self, "_sc_fused", False) and Ci != Co and c1 % 64 == 0 and c2 % 64 == 0

    def _resnet(self, x, Ci, Co, Hh, Ww, name, emb, x2=None):
        """x2: the block's input is the channel-concatenation [x | x2] (up-blocks; Ci counts both), which is never written: norm1 reads the two
        tensors, and conv_shortcut's two halves ride in conv2's K loop (only when _sc_ok)"""
        B, W, A = self.B, self.W, self.arena
        HW = Hh * Ww
        # fp8 plans: norm1 / norm2 leave e4m3 + MX scales and conv1 / conv2 run on e4m3 operands wherever a K-tile can be 128 channels of one tap (Cin % 128 == 0)
        # and the launch carries no shortcut taps (their sources are raw bf16 tensors)
        f81 = self._gn_f8_ok(x, Ci, HW, x2)
        h1 = self._gn(x, Ci, HW, name + ".norm1", 1e-5, True, x2=x2, f8=f81)
        temb, per = self._time_bias(name, Co, emb)
        h2 = self._conv(h1, name + ".conv1", Hh, Ww, Ci, Co, batch_bias=temb, bias_images=per)
        A.put(*h1) if f81 else A.put(h1)
        sc_fused = self._sc_ok(Ci, Co, x.shape[-1], 0 if x2 is None else x2.shape[-1])
        f82 = (not sc_fused) and self._gn_f8_ok(h2, Co, HW)
        h3 = self._gn(h2, Co, HW, name + ".norm2", 1e-5, True, f8=f82)
        A.put(h2)
        if f82:
            assert x2 is None
            if Ci != Co:
                sc = A.get(B, HW, Co)
                self._gemm(x.view(B * HW, Ci), W[name + ".conv_shortcut.weight"], sc.view(B * HW, Co), bias=W[name + ".conv_shortcut.bias"])
            else:
                sc = x
            out = self._conv(h3, name + ".conv2", Hh, Ww, Co, Co, residual=sc)
            A.put(*h3)
            if Ci != Co:
                A.put(sc)
            return out
        if sc_fused:
            out = self._conv(h3, name + ".conv2", Hh, Ww, Co, Co, shortcut=(name, x, x2))
            A.put(h3)
            return out
        assert x2 is None
        if Ci != Co:
            sc = A.get(B, HW, Co)
            self._gemm(x.view(B * HW, Ci), W[name + ".conv_shortcut.weight"], sc.view(B * HW, Co), bias=W[name + ".conv_shortcut.bias"])
        else:
            sc = x
        out = self._conv(h3, name + ".conv2", Hh, Ww, Co, Co, residual=sc)
        A.put(h3)
        if Ci != Co:
            A.put(sc)
        return out

    def _proj(self, a, key, out, S, Cin, ln=None, stats_out=None, fp8=False, a8=None, a_full=None, **kw):
        """Linear over [B,S,Cin] tokens: per-row merged weights when LoRA-routed, else one shared GEMM.
        ln: statistics of a LayerNorm folded into this projection (weights stored folded, see UNetWeights.fold);
        stats_out: accumulate the statistics of the rows this projection writes."""
        W = self.W
        batched = self.routed or kw.get("out_t") is not None    # transposed V is per batch row -> keep the batch dimension
        shp = (self.B, S) if batched else (self.B * S,)
        if self.lowrank and (key + ".lr") in W.t:
            # up(down(x)) as the projection's last K-tile: the pad columns behind the rows of `a` (a_full = the [B, S, Cin + 64] buffer `a`
            # is a view of) receive the rows' down-projections, then ONE shared GEMM over Cin + 64 columns
            assert a_full is not None and a_full.shape[-1] == Cin + 64 and a_full.data_ptr() == a.data_ptr()
            a2 = a_full.view(self.B * S, Cin + 64)
            lnk = ln is not None
            self._emit(self.lib.tmix_lora_down, a2.data_ptr(), a2.stride(0), Cin, self.B * S, W[key + ".lr.D"].data_ptr(), W[key + ".lr.P"],
                       1 + W.K, W[key + ".lr.dcolsum"].data_ptr() if lnk else None, W[key + ".lr.dbias"].data_ptr() if lnk else None,
                       1e-5, self._sets_dev.data_ptr(), S)
            if lnk:
                kw["ln_stats"], kw["ln_colsum"], kw["bias"] = ln, W[key + ".colsum"], W[key + ".bias"]
            if stats_out is not None:
                kw["row_stats_out"] = stats_out
            if kw.get("residual") is not None:
                kw["residual"] = kw["residual"].view(*shp, kw["residual"].shape[-1])
            kw["ln_k"] = Cin if lnk else None
            return self._gemm(a_full.view(*shp, Cin + 64), W[key + ".lr"], out.view(*shp, out.shape[-1]), **kw)
        if ln is not None:
            kw["ln_stats"] = ln
            kw["ln_colsum"] = self._rows(key, ".colsum") if self.routed else W[key + ".colsum"]
            kw["bias"] = self._rows(key, ".bias") if self.routed else W[key + ".bias"]
        if stats_out is not None:
            kw["row_stats_out"] = stats_out
        if kw.get("residual") is not None:
            kw["residual"] = kw["residual"].view(*shp, kw["residual"].shape[-1])
        w = self._rows(key) if self.routed else W[key]
        if fp8 and self.fp8:
            kw["fp8_key"] = (key + "_rows", tuple(range(w.shape[0])) if self._periodic(w.shape[0]) else tuple(self.row_sets)) if self.routed else key
            kw["a8"] = a8
        return self._gemm(a.view(*shp, Cin), w, out.view(*shp, out.shape[-1]), **kw)

    def _periodic(self, P):
        """are the batch rows' weight sets 0 .. P-1 repeated (co-batched seeds, each with its 1 + K prompt rows in order)?"""
        rs = list(self.row_sets)
        return len(rs) % P == 0 and rs == list(range(P)) * (len(rs) // P)

    def _rows(self, key, suffix=""):
        """[B, N, K] per-row weight sets for a routed projection (a view when rows are 0..K in order)."""
        w = self.W[key + "_rows" + suffix]
        if self._periodic(w.shape[0]):                # several seeds x (1 + K) rows: the GEMM reads set b % (1 + K) (tmix_gemm_desc.w_period), no gathered copies
            return w
        if key + suffix not in self._rows_cache:
            self._rows_cache[key + suffix] = w[torch.tensor(self.row_sets, device=w.device)].contiguous()
        return self._rows_cache[key + suffix]

    def _t2d(self, x, Cc, Hh, Ww, name, n):
        B, W, A = self.B, self.W, self.arena
        S = Hh * Ww
        H = Cc // self.cfg.head_dim
        g = self._gn(x, Cc, S, name + ".norm", 1e-6, False)
        pad = 64 if self.lowrank else 0              # low-rank LoRA: 64 pad columns behind every row that feeds a routed projection
        h_full = A.get(B, S, Cc + pad)
        h = h_full[:, :, :Cc] if pad else h_full
        st = self._ln_stats(S, Cc)
        # fp8 plans: every GEMM that writes the residual stream h also leaves its e4m3 + MX-block copy (TMIX_F8_COPY_OUT), which the
        # next projection (attn1 q/k/v, attn2 to_q, FF1) reads as its A operand -- no quantiser launches inside a block
        h8 = None
        if self.fp8 and self.fp8_chain_ff and n and S % 32 == 0 and Cc % 32 == 0:
            h8buf = A.get(ops.F8Copy.bytes_for(B * S, Cc), dtype=torch.uint8)
            h8 = ops.F8Copy(B * S, Cc, self.dev, buf=h8buf)
        ao8 = ao8buf = None
        if h8 is not None and self.fp8_attn_out:
            ao8buf = A.get(ops.F8Copy.bytes_for(B * S, Cc), dtype=torch.uint8)
            ao8 = ops.F8Copy(B * S, Cc, self.dev, buf=ao8buf)
        self._gemm(g.view(B * S, Cc), W[name + ".proj_in.weight"], h.view(B * S, Cc), bias=W[name + ".proj_in.bias"],
                   row_stats_out=st if n else None, f8_copy=h8)
        A.put(g)
        vt = self._vt_buf(Cc, S)
        for i in range(n):
            tb = f"{name}.transformer_blocks.{i}"
            a1, a2 = tb + ".attn1", tb + ".attn2"
            # --- self attention; norm1 is folded into the q/k/v projection
            qk = A.get(B, S, 2 * Cc)
            self._proj(h, a1 + ".qkv", qk, S, Cc, ln=st, out_t=vt, n_trans_begin=2 * Cc, fp8=True, a8=h8, a_full=h_full)
            if ao8 is not None:                        # fp8 plans: attention output as e4m3 + MX blocks, out-projection on e4m3 operands
                self._attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, None, H, S, S, f8_out=ao8)
                A.put(qk)
                self._proj(ao8.q.view(B, S, Cc), a1 + ".out", h, S, Cc, bias=W[a1 + ".to_out.0.bias"], residual=h, stats_out=st, f8_copy=h8, fp8=True, a8=ao8)
            else:
                ao_full = A.get(B, S, Cc + pad)
                ao = ao_full[:, :, :Cc] if pad else ao_full
                self._attn(qk[:, :, :Cc], qk[:, :, Cc:], vt, ao, H, S, S)
                A.put(qk)
                self._proj(ao, a1 + ".out", h, S, Cc, bias=W[a1 + ".to_out.0.bias"], residual=h, stats_out=st, f8_copy=h8, a_full=ao_full)
                A.put(ao_full)
            # --- cross attention against the cached K / V^T; norm2 folded into to_q
            # The choice is a function of the IMAGE's shape only, never of the batch (like gn_small_gpw): the one-launch form rounds q and runs its softmax
            # differently from the pair, so a batch-dependent choice would make a co-batched seed differ from its single-seed run.  64 tiles of
            # 64 x 320 per image = the 1024^2 latent's 32 x 32 level; the B = 2 calls there fill half the chip (128 tiles: 24.2 us against 23.1 + a kernel
            # boundary for the pair), smaller images (512^2: 32 / 16 tiles per image) keep the pair at every batch.
            if getattr(self, "_qattn", False) and Cc % 320 == 0 and S % 64 == 0 and (S // 64) * (Cc // 320) >= 64:
                ao = A.get(B, S, Cc)
                self._q_attn(h, a2 + ".q", self.kv.k[a2], self.kv.vt[a2], ao, S, Cc, st)
                self._proj(ao, a2 + ".out", h, S, Cc, bias=W[a2 + ".to_out.0.bias"], residual=h, stats_out=st, f8_copy=h8, a_full=ao)
                A.put(ao)
                q = None
            else:
                q = A.get(B, S, Cc)
                self._proj(h, a2 + ".q", q, S, Cc, ln=st, fp8=h8 is not None, a8=h8, a_full=h_full)
            if q is None:
                pass
            elif ao8 is not None:
                self._attn(q, self.kv.k[a2], self.kv.vt[a2], None, H, S, self.kv.Lk, f8_out=ao8)
                A.put(q)
                self._proj(ao8.q.view(B, S, Cc), a2 + ".out", h, S, Cc, bias=W[a2 + ".to_out.0.bias"], residual=h, stats_out=st, f8_copy=h8, fp8=True, a8=ao8)
            else:
                ao_full = A.get(B, S, Cc + pad)
                ao = ao_full[:, :, :Cc] if pad else ao_full
                self._attn(q, self.kv.k[a2], self.kv.vt[a2], ao, H, S, self.kv.Lk)
                A.put(q)
                self._proj(ao, a2 + ".out", h, S, Cc, bias=W[a2 + ".to_out.0.bias"], residual=h, stats_out=st, f8_copy=h8, a_full=ao_full)
                A.put(ao_full)
            # --- feed forward: norm3 folded into the first GEMM, GEGLU fused in its epilogue
            if self.fp8 and self.fp8_chain_ff:
                # the intermediate leaves FF1 as e4m3 with one E8M0 scale per 32 columns and FF2 reads it as block-scaled A
                f = A.get(B * S, 4 * Cc, dtype=torch.uint8)
                fs = A.get(4 * Cc // 32, B * S, dtype=torch.uint8)
                self._gemm_fp8((h8.q, h8.scales) if h8 is not None else h.view(B * S, Cc), tb + ".ff1", None, f8_out=(f, fs),
                               bias=W[tb + ".ff1.bias"], geglu=True, ln_stats=st, ln_colsum=W[tb + ".ff1.colsum"])
                self._gemm_fp8((f, fs), tb + ".ff.net.2.weight", h.view(B * S, Cc), bias=W[tb + ".ff.net.2.bias"],
                               residual=h.view(B * S, Cc), row_stats_out=st if i + 1 < n else None,
                               f8_copy=h8 if i + 1 < n else None)
                A.put(f, fs)
                continue
            f = A.get(B * S, 4 * Cc)
            self._gemm(h.view(B * S, Cc), W[tb + ".ff1"], f, bias=W[tb + ".ff1.bias"], geglu=True,
                       ln_stats=st, ln_colsum=W[tb + ".ff1.colsum"], fp8_key=tb + ".ff1")
            self._gemm(f, W[tb + ".ff.net.2.weight"], h.view(B * S, Cc), bias=W[tb + ".ff.net.2.bias"], residual=h.view(B * S, Cc),
                       row_stats_out=st if i + 1 < n else None, fp8_key=tb + ".ff.net.2.weight")
            A.put(f)
        out = A.get(B, S, Cc)
        self._gemm(h.view(B * S, Cc), W[name + ".proj_out.weight"], out.view(B * S, Cc), bias=W[name + ".proj_out.bias"],
                   residual=x.view(B * S, Cc), cs_owner=out)
        A.put(h_full)
        if h8 is not None:
            A.put(h8buf)
        if ao8buf is not None:
            A.put(ao8buf)
        return out

    def _cat(self, x1, C1, x2, C2, HW):
        out = self.arena.get(self.B, HW, C1 + C2)
        self._emit(self.lib.tmix_concat_channels, x1.data_ptr(), C1, x2.data_ptr(), C2, out.data_ptr(), self.B * HW)
        # the concatenation's column partials are those of its two sources: they move to it (and are released with it, not with x1 / x2)
        p1, p2 = getattr(x1, "_cs", None), getattr(x2, "_cs", None)
        if getattr(self, "_gn_fused", False) and p1 and p2 and len(p1) == 1 and len(p2) == 1:
            out._cs = (p1[0], p2[0])
            x1._cs = x2._cs = None
        return out

    # ------------------------------------------------------------------ whole network
    def _build(self):
        cfg, W, B, A = self.cfg, self.W, self.B, self.arena
        lib = self.lib
        ch = cfg.block_out_channels
        C0, T = ch[0], cfg.time_embed_dim
        nb = len(ch)
        # time embedding: sinusoid(t) -> linear_1 -> SiLU -> linear_2 (+ static aug_emb)
        tsin = torch.empty(B, C0, device=self.dev, dtype=F32)
        thid = torch.empty(B, T, device=self.dev, dtype=F32)
        emb = torch.empty(B, T, device=self.dev, dtype=F32)
        self.keep += [tsin, thid, emb]
        self._emit(lib.tmix_timestep_embedding, self.t_dev.data_ptr(), tsin.data_ptr(), B, C0)
        self._emit(lib.tmix_linear_small, tsin.data_ptr(), W["time_embedding.linear_1.weight"].data_ptr(),
                   W["time_embedding.linear_1.bias"].data_ptr(), None, thid.data_ptr(), B, T, C0, 0, 1)
        # (emb is only ever used as SiLU(emb) -- every ResnetBlock2D applies its nonlinearity first -- so linear_2 stores SiLU(emb) and the stacked
        # projection below reads it as is: the activation is evaluated 4 x 1280 times per step instead of once per output column, 20,480 x that)
        self._emit(lib.tmix_linear_small, thid.data_ptr(), W["time_embedding.linear_2.weight"].data_ptr(),
                   W["time_embedding.linear_2.bias"].data_ptr(), self.aug.data_ptr(), emb.data_ptr(), B, T, T, 0, 1)
        # every ResnetBlock2D's time_emb_proj(SiLU(emb)) in ONE launch: the weights are stacked along N once per checkpoint
        tw, tb_, starts, names = W.stacked_time_proj()
        tall = torch.empty(B * tw.shape[0], device=self.dev, dtype=F32)
        self.keep.append(tall)
        self._emit(lib.tmix_linear_small_sections, emb.data_ptr(), tw.data_ptr(), tb_.data_ptr(), tall.data_ptr(), B, tw.shape[0], T, 0,
                   starts.data_ptr(), len(names))
        hs = starts.tolist()
        self._temb = {n: tall[hs[i] * B:hs[i + 1] * B].view(B, hs[i + 1] - hs[i]) for i, n in enumerate(names)}
        Hh, Ww = self.h, self.w
        x = A.get(B, Hh * Ww, C0)
        self._emit(lib.tmix_conv_in, self.latent.data_ptr(), W["conv_in.weight"].data_ptr(), W["conv_in.bias"].data_ptr(),
                   x.data_ptr(), B, cfg.in_channels, Hh, Ww, C0)
        skips = [(x, C0)]
        ci = C0
        for bi, co in enumerate(ch):
            for j in range(cfg.layers_per_block):
                xin = x
                x = self._resnet(xin, ci, co, Hh, Ww, f"down_blocks.{bi}.resnets.{j}", emb)
                if cfg.transformer_layers[bi]:
                    x2 = self._t2d(x, co, Hh, Ww, f"down_blocks.{bi}.attentions.{j}", cfg.transformer_layers[bi])
                    A.put(x)
                    x = x2
                ci = co
                skips.append((x, co))
            if bi < nb - 1:
                x = self._conv(x, f"down_blocks.{bi}.downsamplers.0.conv", Hh, Ww, co, co, mode=L.CONV_S2)
                Hh, Ww = Hh // 2, Ww // 2
                skips.append((x, co))
        cm = ch[-1]
        x2 = self._resnet(x, cm, cm, Hh, Ww, "mid_block.resnets.0", emb)
        x3 = self._t2d(x2, cm, Hh, Ww, "mid_block.attentions.0", cfg.transformer_layers[-1])
        A.put(x2)
        x = self._resnet(x3, cm, cm, Hh, Ww, "mid_block.resnets.1", emb)
        A.put(x3)
        for ui, co in enumerate(reversed(ch)):
            bi = nb - 1 - ui
            for j in range(cfg.layers_per_block + 1):
                sk, cs = skips.pop()
                if self._sc_ok(ci + cs, co, ci, cs):                 # no concatenation: norm1 and conv2's shortcut taps read the two tensors
                    xn = self._resnet(x, ci + cs, co, Hh, Ww, f"up_blocks.{ui}.resnets.{j}", emb, x2=sk)
                    A.put(x, sk)
                    x = xn
                else:
                    xc = self._cat(x, ci, sk, cs, Hh * Ww)
                    A.put(x, sk)
                    x = self._resnet(xc, ci + cs, co, Hh, Ww, f"up_blocks.{ui}.resnets.{j}", emb)
                    A.put(xc)
                if cfg.transformer_layers[bi]:
                    x2 = self._t2d(x, co, Hh, Ww, f"up_blocks.{ui}.attentions.{j}", cfg.transformer_layers[bi])
                    A.put(x)
                    x = x2
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
        self.flops += 2 * B * Hh * Ww * 9 * (cfg.in_channels * C0 + C0 * cfg.out_channels)
        self.ops = [(fn, tuple(a)) for fn, a in self.ops]

    # ------------------------------------------------------------------ execution
    def issued_meta(self):
        """(class, flops, key) of the instrumented launches (tmix_prof_begin) in the order run() issues them."""
        return [self.op_meta[i] for i in range(len(self.ops)) if i in self.op_meta]

    def run(self, stream=None):
        """enqueue the whole forward on `stream` (default: torch's current stream). No sync, no alloc."""
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        for fn, args in self.ops:
            rc = fn(*args, st)
            if rc:
                L.check(rc, fn.__name__)

    def __call__(self, latent, t):
        """eps = unet(latent[B,4,h,w], t) with this plan's prompt rows.  Returns the plan's eps buffer."""
        self.latent.copy_(latent)
        self.t_dev.fill_(float(t))
        self.run()
        return self.eps

    def norm_condition(self, latent, t):
        """calibration pass: run the forward launch by launch and read, behind every GEMM that leaves LayerNorm row statistics, how far the rows'
        means sit from zero in units of their standard deviation: [(op index, rows, width, max |mean| / std, median)] per LayerNorm site.
        What the number means for this plan (DESIGN "hostile statistics"): a bf16 hidden state with |mean| / std = r carries rounding noise of
        ~ r * 2^-9 of a row's std per element whatever normalises it, and the single-pass fp32 variance of the folded LayerNorm
        (gemm_kernel.h ln_reduce) a relative error of ~ 2 r^2 * 2^-24 -- smaller than the first for every r a bf16 stream can carry.
        LN_COND_WARN marks the sites where the STREAM's noise passes 2 % of a row's std."""
        self.latent.copy_(latent)
        self.t_dev.fill_(float(t))
        st = torch.cuda.current_stream().cuda_stream
        prods = {id(d) for d, _c in self._ln_links}
        out = []
        for i, (fn, args) in enumerate(self.ops):
            rc = fn(*args, st)
            if rc:
                L.check(rc, fn.__name__)
            m = self.op_meta.get(i)
            if not m or m[0] not in ("gemm", "gemm_fp8") or id(m[2]) not in prods:
                continue
            d = m[2]
            torch.cuda.synchronize()
            parts = ops.stats_parts(d.N, d.tile_cfg)
            base = (d.row_stats_out - self._ln_buf.data_ptr()) // 4
            rows = torch.arange(d.M, device=self.dev)
            s = torch.zeros(d.batch, d.M, 2, device=self.dev, dtype=torch.float64)
            for bz in range(d.batch):
                for q in range(parts):
                    off = base + bz * d.strideStatsOut + (q * d.ldStatsOut + rows) * 2
                    s[bz, :, 0] += self._ln_buf[off].double()
                    s[bz, :, 1] += self._ln_buf[off + 1].double()
            mean = s[..., 0] / d.N
            var = (s[..., 1] / d.N - mean * mean).clamp_min(1e-30)
            r = (mean.abs() / var.sqrt()).flatten()
            out.append((i, d.M * d.batch, d.N, r.max().item(), r.median().item()))
        torch.cuda.synchronize()
        return out


LN_COND_WARN = 16.0       # |mean| / std of a LayerNorm row beyond which a bf16 hidden state is itself the problem (its rounding noise > 2 % of the row's std)


def refine_group(self, top=14, reps=9, verbose=False, cands=None, only_kind=None):
    """second tuning pass, for the chains of a group or for ONE plan (the event-timed eager ranking of UNetPlan.autotune is noisy
    at the +-1 % level, and a captured graph schedules launches differently from eager issue): UNetPlan.autotune ranks tilings with one chain running alone, but a
    tiling that owns its CU (one workgroup, deep ring) can lose once the other chain competes for the same CUs.  For the
    `top` heaviest launch shapes every candidate is tried in place and the WHOLE group is timed as a captured graph
    (median of `reps` replays); the winner replaces the cache entry.  Used offline by tools/make_tune_table.py."""
    def timed():
        self.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run()
        for _ in range(2):
            g.replay()
        # host clock around a device-wide synchronize, not HIP events: timing events recorded around replays of a graph captured over TWO streams (the video
        # step's two 978-launch chains) corrupt the process heap within a few dozen replays on ROCm 7.0 ("corrupted size vs. prev_size in fastbins" /
        # "double free or corruption"; every run of tools/jobs4/r4zv_refine_video.sh), the same loop without them ran 400 re-captures clean
        import time as _time
        # (each sample = `burst` back-to-back replays: the host launch + synchronize latency of a sample, of the order of the 20 us acceptance threshold
        # below, is paid once per burst instead of once per replay)
        ts = []
        burst = 3
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = _time.perf_counter()
            for _b in range(burst):
                g.replay()
            torch.cuda.synchronize()
            ts.append(1e3 * (_time.perf_counter() - t0) / burst)
        return sorted(ts)[len(ts) // 2]

    plans = _plans_of(self)
    weight, members = {}, {}
    for p in plans:
        for _i, kind, d in p._tunable:
            k = p._tune_key(kind, d)
            fl = 2.0 * d.M * d.N * d.K * d.batch if kind == "gemm" else 2.0 * d.B * d.H * d.W * d.Cout * (9 * d.Cin + d.S1_channels + d.S2_channels)
            weight[k] = weight.get(k, 0.0) + fl
            members.setdefault(k, []).append((p, kind, d))
    base = timed()
    if only_kind is not None:                          # e.g. "conv": re-rank the convolutions only (a kernel change that touches one class)
        weight = {k: v for k, v in weight.items() if members[k][0][1] == only_kind}
    for k in sorted(weight, key=weight.get, reverse=True)[:top]:
        cur = members[k][0][2].tile_cfg
        best, best_t = cur, base
        for cfg in (cands or L.TILE_CANDIDATES):
            if cfg == cur or (cfg in L.TILE_EXCLUSIVE and len(plans) > 1) or (members[k][0][1] == "conv" and cfg in (16, 17, 18, 19, 21, 22, 23, 24, 25)):
                continue
            if cfg == L.TILE_CONV_HALO:               # the halo-patch kernel: convolutions it can run only (anything else would be timed under an alias of 20 / 21)
                d0 = members[k][0][2]
                if members[k][0][1] != "conv" or d0.mode != L.CONV_S1 or d0.S1_channels or d0.W % 32 or d0.H % 4 or d0.Cout % 160:
                    continue
            for p, _kind, d in members[k]:
                d.tile_cfg = cfg
            for p in plans:
                p._link_ln()
            if os.environ.get("TMIX_REFINE_TRACE"):
                print(f"    try {k} cfg {cfg}", flush=True)
            t = timed()
            if t < best_t - 0.02:                       # 20 us: above the replay-to-replay noise of the median
                best, best_t = cfg, t
        for p, _kind, d in members[k]:
            d.tile_cfg = best
        for p in plans:
            p._link_ln()
        if verbose:
            print(f"  refine {k}: {cur} -> {best}  ({base:.3f} -> {best_t:.3f} ms)", flush=True)
        _TUNE_CACHE[getattr(members[k][0][0], "tune_ctx", "") + k] = best
        base = best_t
    return base



def _plans_of(plan):
    return list(plan.plans) if hasattr(plan, "plans") else [plan]


def used_tilings(plan):
    """{"gemm:<TMIX_TILE id>": launches, "conv:<id>": launches} of a UNetPlan / PlanGroup as it will be launched (descriptor
    values; fp8 launches and the F8-copy substitutions of gemm_conv.hip:launch are resolved by the library, the descriptor
    holds the request).  bench.py prints it; the parity tests assert it for the plan they check against the oracle."""
    out = {}
    for p in _plans_of(plan):
        for _i, kind, d in p._tunable:
            k = f"{kind}:{d.tile_cfg}"
            out[k] = out.get(k, 0) + 1
    return dict(sorted(out.items()))


def tilings_follow_table(plan):
    """True when every tunable launch of the plan carries the tiling the SHIPPED table (tuned_gfx950.json) prescribes for its
    shape and context -- i.e. nothing was re-tuned on this box, so a plan built here and one built by bench.py for the same
    call run the same kernels.  Returns (ok, [(shape key, descriptor tiling, table tiling)] of the mismatches)."""
    bad = []
    with open(_TUNE_FILE) as f:                 # the FILE, not the process cache (which also holds shapes timed on this box)
        table = {k: int(v) for k, v in json.load(f).items()}

    def shipped(ctx, key):
        c = table.get(ctx + key)
        if c is None and ctx:
            c = table.get(key)
            if ctx == SHARED and c in L.TILE_EXCLUSIVE:
                c = None
        return c

    for p in _plans_of(plan):
        for _i, kind, d in p._tunable:
            k = p._tune_key(kind, d)
            want = shipped(getattr(p, "tune_ctx", ""), k)
            if kind == "gemm" and (d.reserved0 & L.F8_COPY_OUT):
                want = L.F8COPY_TILE_ALT.get(want, want)
            if want != d.tile_cfg:
                bad.append((k, d.tile_cfg, want))
    return not bad, bad


UNetPlan.refine = lambda self, **kw: refine_group(self, **kw)


class PlanGroup:
    """The batch rows of one UNet call split into `n_groups` independent sub-plans, each enqueued on its own HIP
    stream (rows never interact inside the UNet).  Dependent launches of one chain leave the chip partly idle at
    every kernel boundary (tail of one GEMM, launch gap, cold first loads of the next: ~10 us per launch against
    30-150 us kernels); two or more independent chains in flight let the hardware fill those holes with the other
    chain's workgroups.  Presents the same interface as a UNetPlan (latent, t_dev.fill_, eps, run, B, flops)."""

    def __init__(self, W: UNetWeights, h: int, w: int, ehs: torch.Tensor, wsel, pooled: torch.Tensor,
                 time_ids: torch.Tensor, routed: bool, n_groups: int, fp8: bool = False):
        B = ehs.shape[0]
        assert B % n_groups == 0
        self.B, self.n_groups = B, n_groups
        dev = W.device
        cfg = W.cfg
        self.latent = torch.zeros(B, cfg.in_channels, h, w, device=dev, dtype=F32)
        self.eps = torch.zeros(B, cfg.out_channels, h, w, device=dev, dtype=F32)
        self.t_dev = torch.zeros(B, device=dev, dtype=F32)      # one timestep vector; the chains hold views of their rows
        per = B // n_groups
        self.plans, self.streams = [], []
        for g in range(n_groups):
            sl = slice(g * per, (g + 1) * per)
            kv = KVCache(W, ehs[sl], list(wsel)[sl])
            self.plans.append(UNetPlan(W, per, h, w, kv, pooled[sl], time_ids[sl], routed=routed,
                                       row_sets=list(wsel)[sl] if routed else None,
                                       latent=self.latent[sl], eps=self.eps[sl], shared=n_groups > 1, t_dev=self.t_dev[sl], fp8=fp8))
            self.streams.append(torch.cuda.Stream(device=dev) if g > 0 else None)
        self.flops = sum(p.flops for p in self.plans)
        self.gemm_flops = sum(p.gemm_flops for p in self.plans)
        self.launches = {k: [x for p in self.plans for x in p.launches[k]] for k in ("gemm", "conv", "attn")}
        self.ops = [op for p in self.plans for op in p.ops]

    def refine(self, **kw):
        return refine_group(self, **kw)

    def issued_meta(self):
        """run() enqueues the side-stream chains first, the main-stream chain last."""
        return [m for p in self.plans[1:] + self.plans[:1] for m in p.issued_meta()]

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
