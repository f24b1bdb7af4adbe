"""Thin torch-tensor wrappers over the C ABI (include/tmix.h).

PyTorch is plumbing only: device memory (data_ptr) and the current HIP stream.  Every wrapper
validates on the C side and raises TmixError on failure; there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as L

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.TmixError("tweediemix_amd ops need device tensors (no CPU fallback exists)")


_EPS_DT = {torch.float32: L.F32, torch.float16: L.F16, torch.bfloat16: L.BF16}


def step_coeffs(at, at_next):
    """fp32 sqrt(at), sqrt(1-at), sqrt(at'), sqrt(1-at') exactly as the fp32 reference computes them."""
    at, an = np.float32(at), np.float32(at_next)
    one = np.float32(1)
    return (float(np.sqrt(at)), float(np.sqrt(one - at)), float(np.sqrt(an)), float(np.sqrt(one - an)))


def fused_tweedie_step(x, eps, masks, mode, K, g, at, at_next, is_last=False, out_x=None, out_x0=None):
    """x [1,C,h,w] fp32; eps [rows,C,h,w] f32|f16|bf16; masks [K,1,h,w] fp32 (FUSION).  Returns out_x."""
    _need_cuda(x, eps, masks)
    lib = L.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and eps.is_contiguous()
    Cc, h, w = x.shape[-3], x.shape[-2], x.shape[-1]
    rows = eps.shape[0]
    need = {L.STEP_FUSION: K + 1, L.STEP_PLAIN: 2, L.STEP_RESAMPLE: K + 1}[mode]
    if rows < need:
        raise L.TmixError(f"tweedie step mode {mode} needs {need} eps rows, got {rows}")
    if mode == L.STEP_FUSION:
        assert masks is not None and masks.dtype == torch.float32 and masks.is_contiguous() and masks.shape[0] == K
        assert masks.numel() == K * h * w
    if out_x is None:
        out_x = torch.empty_like(x)
    sa, s1, san, s1n = step_coeffs(at, at_next)
    L.check(lib.tmix_fused_tweedie_step(_p(x), _p(eps), _EPS_DT[eps.dtype], _p(masks), _p(out_x), _p(out_x0),
                                        K, Cc, h * w, mode, float(g), sa, s1, san, s1n, int(bool(is_last)),
                                        _stream()), "tmix_fused_tweedie_step")
    return out_x


def stats_parts(N, tile_cfg):
    """number of row-statistics partials a GEMM of width N writes with tiling tile_cfg (tmix_gemm_stats_parts)."""
    n = L.load().tmix_gemm_stats_parts(int(N), int(tile_cfg))
    assert n > 0, (N, tile_cfg)
    return n


def make_gemm_desc(a, w, out, bias=None, residual=None, rowgroup_bias=None, rows_per_group=0, geglu=False,
                   out_t=None, n_trans_begin=-1, tile_cfg=0, row_stats_out=None, ln_stats=None, ln_colsum=None,
                   ln_eps=1e-5, ln_parts=0, act=None, out_f32=None, ln_k=None, col_stats_out=None):
    """a [batch?,M,K] bf16 (last dim contiguous), w [batch?,N,K] bf16, out [batch?,M,N'] bf16."""
    a3 = a if a.dim() == 3 else a.unsqueeze(0)
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    batch, M, K = a3.shape
    N = w3.shape[1]
    assert a3.dtype == w3.dtype and a3.dtype in (BF16, torch.uint8) and a3.stride(2) == 1 and w3.stride(2) == 1 and w3.shape[2] == K   # uint8 = e4m3 bytes (gemm_fp8)
    d = L.GemmDesc()
    d.A, d.lda, d.strideA = a3.data_ptr(), a3.stride(1), (a3.stride(0) if batch > 1 else 0)
    d.W, d.ldw, d.strideW = w3.data_ptr(), w3.stride(1), (w3.stride(0) if w3.shape[0] > 1 else 0)
    P = w3.shape[0]
    assert P in (1, batch) or batch % P == 0          # P weight sets for a batch of several x P rows: slice b reads set b % P (tmix_gemm_desc.w_period)
    d.w_period = P if 1 < P < batch else 0
    if out is not None:
        o3 = out if out.dim() == 3 else out.unsqueeze(0)
        assert o3.dtype == BF16 and o3.stride(2) == 1
        d.C, d.ldc, d.strideC = o3.data_ptr(), o3.stride(1), (o3.stride(0) if batch > 1 else 0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.stride(-1) == 1
        d.bias = bias.data_ptr()
        d.strideBias = bias.stride(0) if (bias.dim() == 2 and bias.shape[0] > 1) else 0
        assert bias.dim() == 1 or bias.shape[0] in (1, P)
    if residual is not None:
        r3 = residual if residual.dim() == 3 else residual.unsqueeze(0)
        assert r3.dtype == BF16 and r3.stride(2) == 1
        d.residual, d.ldr, d.strideR = r3.data_ptr(), r3.stride(1), (r3.stride(0) if batch > 1 else 0)
    if rowgroup_bias is not None:
        assert rowgroup_bias.dtype == torch.float32 and rowgroup_bias.is_contiguous()
        d.rowgroup_bias, d.rows_per_group = rowgroup_bias.data_ptr(), rows_per_group
    d.n_trans_begin = -1
    if out_t is not None:
        t3 = out_t if out_t.dim() == 3 else out_t.unsqueeze(0)
        assert t3.dtype == BF16 and t3.stride(2) == 1
        d.Ct, d.ldct, d.strideCt = t3.data_ptr(), t3.stride(1), (t3.stride(0) if batch > 1 else 0)
        d.n_trans_begin = n_trans_begin
    d.M, d.N, d.K, d.batch = M, N, K, batch
    d.epilogue = L.EPI_GEGLU if geglu else {None: L.EPI_NONE, "gelu": L.EPI_GELU, "quick_gelu": L.EPI_QUICKGELU}[act]
    if out_f32 is not None:                # fp32 C (attention scores): [batch?, M, ld >= N]
        f3 = out_f32 if out_f32.dim() == 3 else out_f32.unsqueeze(0)
        assert f3.dtype == torch.float32 and f3.stride(2) == 1 and out is None and act is None and not geglu
        d.C, d.ldc, d.strideC, d.epilogue = f3.data_ptr(), f3.stride(1), (f3.stride(0) if batch > 1 else 0), L.EPI_F32OUT
    d.tile_cfg = tile_cfg
    if row_stats_out is not None:          # fp32 [parts, rows, 2] per-column-tile partial sums, rows = batch*M
        st = row_stats_out
        assert st.dtype == torch.float32 and st.is_contiguous() and st.dim() == 3 and st.shape[1:] == (batch * M, 2)
        assert tile_cfg > 0 and st.shape[0] >= stats_parts(N, tile_cfg)
        d.row_stats_out, d.strideStatsOut, d.ldStatsOut = st.data_ptr(), 2 * M, st.shape[1]
    if ln_stats is not None:               # fused LayerNorm on the A rows (see include/tmix.h)
        st = ln_stats
        assert st.dtype == torch.float32 and st.is_contiguous() and st.dim() == 3 and st.shape[1:] == (batch * M, 2)
        assert ln_colsum is not None and ln_colsum.dtype == torch.float32 and ln_colsum.is_contiguous()
        assert ln_colsum.shape in ((N,), (P, N))
        d.ln_stats, d.strideLnStats, d.ldLnStats = st.data_ptr(), 2 * M, st.shape[1]
        d.ln_parts = int(ln_parts) if ln_parts else st.shape[0]
        d.ln_colsum, d.strideLnColsum = ln_colsum.data_ptr(), (N if ln_colsum.dim() == 2 and batch > 1 else 0)
        d.ln_inv_c, d.ln_eps = 1.0 / (ln_k or K), ln_eps       # ln_k: the LayerNorm's width when A carries extra columns behind it (low-rank LoRA pad)
    if col_stats_out is not None:          # GroupNorm partials of the stored rows: fp32 [M/32, 2, N] (colstats_buf)
        cs = col_stats_out
        assert batch == 1 and cs.dtype == torch.float32 and cs.is_contiguous() and tuple(cs.shape) == (M // COLSTATS_ROWS, 2, N) and M % COLSTATS_ROWS == 0
        d.col_stats_out = cs.data_ptr()
    return d


COLSTATS_ROWS = 32          # TMIX_COLSTATS_ROWS
# largest image (pixels) whose GroupNorms take their statistics from the producers: the combine launch walks HW / 32 partials per channel, and beyond
# 128 x 128 that pass loses to the statistics kernel's pass over x itself (VAE decode at 1024^2, all levels on the partials: 15.4 vs 14.7 ms)
COLSTATS_MAX_HW = 16384


def colstats_buf(rows, C, device):
    """receiver of tmix_gemm_desc.col_stats_out / tmix_conv_desc.col_stats_out for an output of `rows` x C: fp32 [rows/32, 2, C],
    plane 0 = column sums, plane 1 = column sums of squares over each 32-row block of the stored bf16 values."""
    assert rows % COLSTATS_ROWS == 0 and C % 8 == 0
    return torch.empty(rows // COLSTATS_ROWS, 2, C, device=device, dtype=torch.float32)


class F8Copy:
    """Receiver of TMIX_F8_COPY_OUT: one byte buffer holding the e4m3 copy [rows, N] of a GEMM's bf16 output and, behind it,
    the MX block scales [N/32, rows] -- the pair tmix_gemm_fp8 takes as a block-scaled A operand."""

    @staticmethod
    def bytes_for(rows, N):
        return (rows * N + 255) // 256 * 256 + (N // 32) * rows

    def __init__(self, rows, N, device, buf=None):
        self.rows, self.N = rows, N
        self.off = (rows * N + 255) // 256 * 256
        self.nbytes = self.off + (N // 32) * rows
        self.buf = buf if buf is not None else torch.empty(self.nbytes, device=device, dtype=torch.uint8)
        assert self.buf.numel() >= self.nbytes and self.buf.dtype == torch.uint8
        self.q = self.buf[:rows * N].view(rows, N)
        self.scales = self.buf[self.off:self.off + (N // 32) * rows].view(N // 32, rows)

    def attach(self, d):
        assert d.batch * d.M == self.rows and d.N == self.N
        d.Ct, d.ldct, d.strideCt = self.buf.data_ptr(), self.N, self.off
        d.reserved0 |= L.F8_COPY_OUT


def gemm(a, w, out=None, f8_copy=None, **kw):
    """out = epi(a @ w^T).  Allocates out ([.., M, N] or [.., M, N/2] for GEGLU) when not given.
    f8_copy: an F8Copy that also receives the e4m3 + MX-block-scale copy of the stored rows."""
    _need_cuda(a, w)
    lib = L.load()
    if out is None and kw.get("out_f32") is None and not (kw.get("out_t") is not None and kw.get("n_trans_begin", -1) == 0):
        N = w.shape[-2]
        No = N // 2 if kw.get("geglu") else N
        if kw.get("out_t") is not None:
            No = kw["n_trans_begin"]
        out = torch.empty(*a.shape[:-1], No, device=a.device, dtype=BF16)
    d = make_gemm_desc(a, w, out, **kw)
    if f8_copy is not None:
        f8_copy.attach(d)
    L.check(lib.tmix_gemm_bf16(C.byref(d), _stream()), "tmix_gemm_bf16")
    return out if out is not None else kw.get("out_f32")


def q_cross_attn_args(k, vt, out, rows_per_image, scale):
    """the arguments of tmix_gemm_q_cross_attn behind its descriptor: cached K [images, Skv, C], V^T [images, C, 80], output [rows, C]"""
    assert k.dtype == BF16 and vt.dtype == BF16 and out.dtype == BF16 and k.stride(2) == 1 and vt.stride(2) == 1 and out.stride(-1) == 1
    o2 = out.reshape(-1, out.shape[-1])
    assert o2.data_ptr() == out.data_ptr()
    return (k.data_ptr(), k.stride(1), k.stride(0), vt.data_ptr(), vt.stride(1), vt.stride(0), o2.data_ptr(), o2.stride(0),
            int(rows_per_image), int(k.shape[1]), float(scale))


def gemm_q_cross_attn(a, w, k, vt, rows_per_image, scale, out=None, **kw):
    """attn2 in one launch (tmix_gemm_q_cross_attn): softmax((a @ w^T [+ bias, folded LayerNorm]) K^T * scale) V per 64-wide head.
    a [batch?, M, C], w [batch?, C, C]; k [images, Skv <= 80, C], vt [images, C, 80]; returns the attention output [batch?, M, C]."""
    _need_cuda(a, w, k, vt)
    if out is None:
        out = torch.empty(*a.shape[:-1], w.shape[-2], device=a.device, dtype=BF16)
    d = make_gemm_desc(a, w, None, **kw)
    L.check(L.load().tmix_gemm_q_cross_attn(C.byref(d), *q_cross_attn_args(k, vt, out, rows_per_image, scale), _stream()), "tmix_gemm_q_cross_attn")
    return out


def quantize_fp8_rows(x, q=None, scale=None):
    """x bf16 [..., K] (rows contiguous in K) -> (q uint8 [..., K] holding OCP e4m3 bytes, scale uint8 [...] E8M0 exponents):
    x[r] ~= e4m3(q[r]) * 2^(scale[r] - 127)."""
    _need_cuda(x)
    assert x.dtype == BF16 and x.stride(-1) == 1
    K = x.shape[-1]
    if K > 8192:                       # rows longer than the kernel keeps in registers (conv weight rows: 9 * Cin): the same arithmetic in torch, once per checkpoint
        xf = x.float()
        amax = xf.abs().amax(dim=-1)
        # e8m0_for_amax (csrc/common.h), bit for bit: the fp32 product amax * fl(1/448), its exponent, + 1 when the mantissa is not zero
        bits = (amax.float() * torch.tensor(1.0 / 448.0, dtype=torch.float32, device=amax.device)).view(torch.int32)
        e = (((bits >> 23) & 0xff) - 127 + ((bits & 0x7fffff) != 0).to(torch.int32)).clamp(-127, 127)
        e = torch.where(amax > 0, e, torch.zeros_like(e)).float()
        qt = (xf * torch.exp2(-e).unsqueeze(-1)).to(torch.float8_e4m3fn).view(torch.uint8)
        st = (e + 127).to(torch.uint8)
        if q is not None:
            q.copy_(qt); qt = q
        if scale is not None:
            scale.copy_(st); st = scale
        return qt.contiguous(), st.contiguous()
    x2 = x.reshape(-1, K)
    if q is None:
        q = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    if scale is None:
        scale = torch.empty(x.shape[:-1], device=x.device, dtype=torch.uint8)
    q2 = q.view(-1, K)
    L.check(L.load().tmix_quantize_fp8_rows(x2.data_ptr(), x2.stride(0), q2.data_ptr(), q2.stride(0), scale.data_ptr(), x2.shape[0], K,
                                            _stream()), "tmix_quantize_fp8_rows")
    return q, scale


def dequantize_fp8_rows(q, scale):
    """fp32 values of (q, scale) as tmix_gemm_fp8 reads them (tests / error measurements)."""
    return q.view(torch.float8_e4m3fn).float() * torch.exp2(scale.float() - 127.0).unsqueeze(-1)


def gemm_fp8(a8, sa, w8, sw, out=None, a_block_scales=False, f8_out=None, f8_copy=None, **kw):
    """gemm() on e4m3 operands with per-row E8M0 scales (tmix_gemm_fp8): a8 [.., M, K] uint8 + sa [.., M]; w8 [.., N, K] + sw [.., N].
    a_block_scales: sa is the MX block form [K/32, rows] (k-block major) instead.
    f8_out=(c8 uint8 [.., M, N/2], scales uint8 [N/64, rows]) with geglu=True: the GEGLU result leaves as e4m3 + block scales."""
    _need_cuda(a8, w8, sa, sw)
    lib = L.load()
    if f8_out is not None:
        c8, cs = f8_out
        assert kw.get("geglu") and c8.dtype == torch.uint8 and cs.dtype == torch.uint8 and cs.is_contiguous() and c8.stride(-1) == 1
        d = make_gemm_desc(a8, w8, None, **kw)
        c3 = c8 if c8.dim() == 3 else c8.unsqueeze(0)
        d.C, d.ldc, d.strideC = c3.data_ptr(), c3.stride(1), (c3.stride(0) if c3.shape[0] > 1 else 0)
        d.Ct, d.ldct = cs.data_ptr(), cs.shape[1]
        d.reserved0 = L.F8_GEGLU_OUT | (L.F8_A_BLOCK_SCALES if a_block_scales else 0)
        L.check(lib.tmix_gemm_fp8(C.byref(d), sa.data_ptr(), sw.data_ptr(), _stream()), "tmix_gemm_fp8")
        return c8, cs
    if out is None and kw.get("out_f32") is None:
        N = w8.shape[-2]
        No = N // 2 if kw.get("geglu") else (kw["n_trans_begin"] if kw.get("out_t") is not None else N)
        out = torch.empty(*a8.shape[:-1], No, device=a8.device, dtype=BF16)
    assert sa.dtype == torch.uint8 and sw.dtype == torch.uint8 and sa.is_contiguous() and sw.is_contiguous()
    d = make_gemm_desc(a8, w8, out, **kw)
    if a_block_scales:
        assert sa.dim() == 2 and sa.shape[0] == a8.shape[-1] // 32
        d.reserved0 = L.F8_A_BLOCK_SCALES
    if f8_copy is not None:
        f8_copy.attach(d)
    L.check(lib.tmix_gemm_fp8(C.byref(d), sa.data_ptr(), sw.data_ptr(), _stream()), "tmix_gemm_fp8")
    return out if out is not None else kw.get("out_f32")


def make_conv_desc(x, w, out, bias=None, batch_bias=None, residual=None, mode=L.CONV_S1, tile_cfg=0, bias_images=1, col_stats_out=None,
                   shortcut=None, _fp8=False):
    """x [B,H,W,Cin] bf16 NHWC contiguous; w [Cout,3,3,Cin] bf16 contiguous ([Cout,3,Cin] for the temporal CONV_T3,
    where x is [clips, frames, h*w, Cin]).
    shortcut: (s1, s2 or None) -- NHWC tensors whose 1x1 conv_shortcut rides in the same K loop; w is then the 2-d [Cout, 9*Cin + C(s1) + C(s2)]
    matrix [conv taps | shortcut weights] (shortcut_weight) and bias the sum of the two biases."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    dt = torch.uint8 if _fp8 else BF16                  # (e4m3 bytes: tmix_conv3x3_nhwc_fp8)
    assert x.dtype == dt and w.dtype == dt and x.is_contiguous() and w.is_contiguous() and out.is_contiguous()
    if shortcut is not None:
        s1, s2 = shortcut
        c1, c2 = s1.shape[-1], (0 if s2 is None else s2.shape[-1])
        assert mode == L.CONV_S1 and tuple(w.shape) == (Cout, 9 * Cin + c1 + c2) and residual is None
        assert s1.dtype == BF16 and s1.is_contiguous() and s1.numel() == B * H * W * c1 and (s2 is None or (s2.dtype == BF16 and s2.is_contiguous() and s2.numel() == B * H * W * c2))
    else:
        assert tuple(w.shape) == ((Cout, 3, Cin) if mode == L.CONV_T3 else (Cout, 3, 3, Cin))
    d = L.ConvDesc()
    d.X, d.Wt, d.Y = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.batch_bias, d.residual = _p(bias), _p(batch_bias), _p(residual)
    d.B, d.H, d.W, d.Cin, d.Cout, d.mode = B, H, W, Cin, Cout, mode
    d.tile_cfg = tile_cfg
    d.batch_bias_images = bias_images      # consecutive images sharing one batch_bias row (video: frames of a clip)
    if col_stats_out is not None:
        cs = col_stats_out
        assert cs.dtype == torch.float32 and cs.is_contiguous() and tuple(cs.shape) == (out.numel() // Cout // COLSTATS_ROWS, 2, Cout)
        d.col_stats_out = cs.data_ptr()
    if shortcut is not None:
        d.S1, d.S1_channels = s1.data_ptr(), c1
        if s2 is not None:
            d.S2, d.S2_channels = s2.data_ptr(), c2
    return d


def shortcut_weight(w_conv, w_sc):
    """[Cout, 9*Cin + Csc]: the OHWI taps of a 3x3 conv followed by the [Cout, Csc] weights of the 1x1 shortcut that shares its launch"""
    Cout = w_conv.shape[0]
    return torch.cat([w_conv.reshape(Cout, -1), w_sc.reshape(Cout, -1)], dim=1).contiguous()


def conv_out_hw(H, W, mode):
    return (H // 2, W // 2) if mode in (L.CONV_S2, L.CONV_S2A) else ((2 * H, 2 * W) if mode == L.CONV_UP2 else (H, W))


def conv3x3_fp8(x8, sx, w8, sw, bias=None, batch_bias=None, residual=None, mode=L.CONV_S1, out=None, tile_cfg=12, col_stats_out=None, bias_images=1):
    """conv3x3 on e4m3 operands (tmix_conv3x3_nhwc_fp8): x8 uint8 [B,H,W,Cin] + sx uint8 [B*H*W, Cin/32] (row-major MX blocks, as groupnorm(f8_out=) writes);
    w8 uint8 [Cout,3,3,Cin] (or [Cout,3,Cin]) + sw uint8 [Cout] (quantize_fp8_rows over the flattened weight rows)."""
    _need_cuda(x8, w8, sx, sw)
    lib = L.load()
    B, H, W, Cin = x8.shape
    Ho, Wo = conv_out_hw(H, W, mode)
    if out is None:
        out = torch.empty(B, Ho, Wo, w8.shape[0], device=x8.device, dtype=BF16)
    assert x8.dtype == torch.uint8 and w8.dtype == torch.uint8 and sx.dtype == torch.uint8 and sw.dtype == torch.uint8
    assert x8.is_contiguous() and w8.is_contiguous() and sx.is_contiguous() and sw.is_contiguous() and sx.numel() == B * H * W * Cin // 32 and sw.numel() == w8.shape[0]
    d = make_conv_desc(x8, w8, out, bias, batch_bias, residual, mode, tile_cfg, bias_images=bias_images, col_stats_out=col_stats_out, _fp8=True)
    L.check(lib.tmix_conv3x3_nhwc_fp8(C.byref(d), sx.data_ptr(), sw.data_ptr(), _stream()), "tmix_conv3x3_nhwc_fp8")
    return out


def conv3x3(x, w, bias=None, batch_bias=None, residual=None, mode=L.CONV_S1, out=None, tile_cfg=0, col_stats_out=None, shortcut=None):
    _need_cuda(x, w)
    lib = L.load()
    B, H, W, _ = x.shape
    Ho, Wo = conv_out_hw(H, W, mode)
    if out is None:
        out = torch.empty(B, Ho, Wo, w.shape[0], device=x.device, dtype=BF16)
    d = make_conv_desc(x, w, out, bias, batch_bias, residual, mode, tile_cfg, col_stats_out=col_stats_out, shortcut=shortcut)
    L.check(lib.tmix_conv3x3_nhwc(C.byref(d), _stream()), "tmix_conv3x3_nhwc")
    return out


def conv_in(x_nchw, w_ohwi, bias, out=None):
    """fp32 NCHW [B,Cin,H,W] -> bf16 NHWC [B,H,W,Cout]; w fp32 [Cout,3,3,Cin]."""
    _need_cuda(x_nchw, w_ohwi)
    lib = L.load()
    B, Cin, H, W = x_nchw.shape
    Cout = w_ohwi.shape[0]
    assert x_nchw.dtype == torch.float32 and w_ohwi.dtype == torch.float32 and x_nchw.is_contiguous() and w_ohwi.is_contiguous()
    if out is None:
        out = torch.empty(B, H, W, Cout, device=x_nchw.device, dtype=BF16)
    L.check(lib.tmix_conv_in(_p(x_nchw), _p(w_ohwi), _p(bias), _p(out), B, Cin, H, W, Cout, _stream()), "tmix_conv_in")
    return out


def conv_out(x_nhwc, w_ohwi, bias, out=None):
    """bf16 NHWC [B,H,W,Cin] -> fp32 NCHW [B,Cout,H,W]; w bf16 [Cout,3,3,Cin]."""
    _need_cuda(x_nhwc, w_ohwi)
    lib = L.load()
    B, H, W, Cin = x_nhwc.shape
    Cout = w_ohwi.shape[0]
    assert x_nhwc.dtype == BF16 and w_ohwi.dtype == BF16 and x_nhwc.is_contiguous() and w_ohwi.is_contiguous()
    if out is None:
        out = torch.empty(B, Cout, H, W, device=x_nhwc.device, dtype=torch.float32)
    L.check(lib.tmix_conv_out(_p(x_nhwc), _p(w_ohwi), _p(bias), _p(out), B, Cin, H, W, Cout, _stream()), "tmix_conv_out")
    return out


def attention_split_ws(B, H, Sq, Skv, device):
    """zero-filled workspace that lets tmix_attn_fwd_ws cut the items of a partly filled last round into key ranges (None: this shape does not split).
    One per stream that runs attention; the launches leave its ticket counters at zero."""
    n = L.load().tmix_attn_split_ws_bytes(B, H, Sq, Skv)
    return torch.zeros(n, device=device, dtype=torch.uint8) if n > 0 else None


def attention(q, k, vt, H, Skv, scale, out=None, f8_out=None, ws=None):
    """q [B,Sq,>=H*64] bf16 (row stride free), k [B,>=Skv,..] bf16, vt [B,H*64,ldvt] bf16 (V transposed).
    f8_out: an F8Copy(B * Sq, H * 64) that receives the output as e4m3 + MX block scales INSTEAD of the bf16 tensor (tmix_attn_fwd_f8).
    ws: attention_split_ws(...) buffer (at least as large as this shape needs) -> the key-split tail (tmix_attn_fwd_ws / tmix_attn_fwd_f8_ws)."""
    _need_cuda(q, k, vt)
    lib = L.load()
    B, Sq = q.shape[0], q.shape[1]
    assert q.dtype == BF16 and k.dtype == BF16 and vt.dtype == BF16
    assert q.stride(2) == 1 and k.stride(2) == 1 and vt.stride(2) == 1
    wsa = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
    if f8_out is not None:
        assert f8_out.rows == B * Sq and f8_out.N == H * 64
        L.check(lib.tmix_attn_fwd_f8_ws(_p(q), q.stride(1), q.stride(0), _p(k), k.stride(1), k.stride(0), _p(vt), vt.stride(1), vt.stride(0),
                                        f8_out.q.data_ptr(), f8_out.N, f8_out.scales.data_ptr(), f8_out.rows,
                                        B, H, Sq, Skv, float(scale), *wsa, _stream()), "tmix_attn_fwd_f8_ws")
        return f8_out
    if out is None:
        out = torch.empty(B, Sq, H * 64, device=q.device, dtype=BF16)
    L.check(lib.tmix_attn_fwd_ws(_p(q), q.stride(1), q.stride(0), _p(k), k.stride(1), k.stride(0),
                                 _p(vt), vt.stride(1), vt.stride(0), _p(out), out.stride(1), out.stride(0),
                                 B, H, Sq, Skv, float(scale), *wsa, _stream()), "tmix_attn_fwd_ws")
    return out


def groupnorm_ws(B, C, groups, device):
    return torch.empty(L.load().tmix_groupnorm_ws_floats(B, C, groups), device=device, dtype=torch.float32)


def groupnorm(x1, gamma, beta, groups=32, eps=1e-5, silu=False, x2=None, out=None, ws=None, colstats=None, f8_out=None):
    """x1 [B,HW,C1] (+ optional x2 [B,HW,C2], normalised as channel-concat) bf16 NHWC.
    colstats: (cs1, cs2 or None) -- the column partials the tensor's producers left (colstats_buf; their channel counts add up to C1 + C2 but
    need not be C1 and C2): tmix_groupnorm_nhwc_pre, no statistics pass."""
    _need_cuda(x1, gamma, beta)
    lib = L.load()
    B, HW, C1 = x1.shape[0], x1.numel() // (x1.shape[0] * x1.shape[-1]), x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    assert x1.is_contiguous() and (x2 is None or x2.is_contiguous()) and gamma.dtype == torch.float32
    assert f8_out is None or colstats is not None, "the e4m3 output exists for the producer-statistics form (tmix_groupnorm_nhwc_pre_f8)"
    if out is None and f8_out is None:
        out = torch.empty(*x1.shape[:-1], C1 + C2, device=x1.device, dtype=BF16)
    if ws is None:
        ws = groupnorm_ws(B, C1 + C2, groups, x1.device)
    if colstats is not None:
        cs1, cs2 = colstats
        ca, cb = cs1.shape[-1], (0 if cs2 is None else cs2.shape[-1])
        assert cs1.is_contiguous() and cs1.shape[1] == 2 and (cs2 is None or (cs2.is_contiguous() and cs2.shape[:2] == cs1.shape[:2]))
        if HW % COLSTATS_ROWS == 0:
            assert cs1.shape[0] == B * HW // COLSTATS_ROWS
        if f8_out is not None:                             # (y8 uint8 [.., C], s8 uint8 [pixels, C / 32]): e4m3 + row-major MX scales instead of bf16
            y8, s8 = f8_out
            assert y8.dtype == torch.uint8 and s8.dtype == torch.uint8 and y8.is_contiguous() and s8.is_contiguous()
            assert y8.numel() == B * HW * (C1 + C2) and s8.numel() == B * HW * (C1 + C2) // 32
            L.check(lib.tmix_groupnorm_nhwc_pre_f8(_p(x1), C1, _p(x2), C2, _p(y8), _p(s8), _p(gamma), _p(beta), _p(ws), B, HW, groups,
                                                   float(eps), int(bool(silu)), _p(cs1), ca, _p(cs2), cb, _stream()), "tmix_groupnorm_nhwc_pre_f8")
            return f8_out
        L.check(lib.tmix_groupnorm_nhwc_pre(_p(x1), C1, _p(x2), C2, _p(out), _p(gamma), _p(beta), _p(ws), B, HW, groups,
                                            float(eps), int(bool(silu)), _p(cs1), ca, _p(cs2), cb, _stream()), "tmix_groupnorm_nhwc_pre")
        return out
    L.check(lib.tmix_groupnorm_nhwc(_p(x1), C1, _p(x2), C2, _p(out), _p(gamma), _p(beta), _p(ws), B, HW, groups,
                                    float(eps), int(bool(silu)), _stream()), "tmix_groupnorm_nhwc")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    _need_cuda(x, gamma, beta)
    lib = L.load()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    assert x.is_contiguous() and x.dtype == BF16 and gamma.dtype == torch.float32
    if out is None:
        out = torch.empty_like(x)
    L.check(lib.tmix_layernorm(_p(x), _p(out), _p(gamma), _p(beta), rows, Cc, float(eps), _stream()), "tmix_layernorm")
    return out


def concat_channels(x1, x2, out=None):
    _need_cuda(x1, x2)
    lib = L.load()
    C1, C2 = x1.shape[-1], x2.shape[-1]
    rows = x1.numel() // C1
    assert x1.is_contiguous() and x2.is_contiguous() and x1.dtype == BF16 and x2.dtype == BF16
    if out is None:
        out = torch.empty(*x1.shape[:-1], C1 + C2, device=x1.device, dtype=BF16)
    L.check(lib.tmix_concat_channels(_p(x1), C1, _p(x2), C2, _p(out), rows, _stream()), "tmix_concat_channels")
    return out


def timestep_embedding(values, dim, out=None):
    _need_cuda(values)
    lib = L.load()
    assert values.dtype == torch.float32 and values.is_contiguous()
    n = values.numel()
    if out is None:
        out = torch.empty(n, dim, device=values.device, dtype=torch.float32)
    L.check(lib.tmix_timestep_embedding(_p(values), _p(out), n, dim, _stream()), "tmix_timestep_embedding")
    return out


def linear_small_sections(x, w, bias, sec_starts, act_in=False, out=None):
    """x [M<=256,K] fp32 against weight matrices stacked along N (w [N,K] bf16, sec_starts int32 device [nsec+1]):
    returns the flat fp32 buffer whose slice [sec_starts[s]*M, sec_starts[s+1]*M) is section s as a dense [M, width_s] matrix."""
    _need_cuda(x, w, sec_starts)
    lib = L.load()
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == BF16 and w.is_contiguous() and w.shape[1] == K
    assert sec_starts.dtype == torch.int32 and sec_starts.is_contiguous()
    if out is None:
        out = torch.empty(M * N, device=x.device, dtype=torch.float32)
    L.check(lib.tmix_linear_small_sections(_p(x), _p(w), _p(bias), _p(out), M, N, K, int(bool(act_in)), _p(sec_starts),
                                           sec_starts.numel() - 1, _stream()), "tmix_linear_small_sections")
    return out


def linear_small(x, w, bias=None, add=None, act_in=False, act_out=False, out=None):
    """x [M<=256,K] fp32, w [N,K] bf16 -> [M,N] fp32."""
    _need_cuda(x, w)
    lib = L.load()
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == BF16 and w.is_contiguous() and w.shape[1] == K
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    L.check(lib.tmix_linear_small(_p(x), _p(w), _p(bias), _p(add), _p(out), M, N, K, int(bool(act_in)),
                                  int(bool(act_out)), _stream()), "tmix_linear_small")
    return out


def _f32c(*ts):
    for t in ts:
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous()), "fp32 contiguous tensors"


def conv3x3_f32(x, w, bias=None, stride=1, silu=False):
    """x [B,Cin,H,W] fp32 NCHW, w [Cout,Cin,3,3] fp32 (checkpoint layout), padding 1 -> [B,Cout,OH,OW] fp32 (tmix_conv3x3_f32)."""
    _need_cuda(x, w)
    _f32c(x, w, bias)
    B, Ci, H, W_ = x.shape
    Co = w.shape[0]
    assert tuple(w.shape) == (Co, Ci, 3, 3)
    y = torch.empty(B, Co, (H - 1) // stride + 1, (W_ - 1) // stride + 1, device=x.device, dtype=torch.float32)
    L.check(L.load().tmix_conv3x3_f32(_p(x), _p(w), _p(bias), _p(y), B, Ci, H, W_, Co, int(stride), int(bool(silu)), _stream()), "tmix_conv3x3_f32")
    return y


def adaptive_avgpool_f32(x, oh, ow):
    """torch.nn.AdaptiveAvgPool2d((oh, ow)) on fp32 [..., H, W] (tmix_adaptive_avgpool_f32)."""
    _need_cuda(x)
    _f32c(x)
    H, W_ = x.shape[-2:]
    y = torch.empty(*x.shape[:-2], oh, ow, device=x.device, dtype=torch.float32)
    L.check(L.load().tmix_adaptive_avgpool_f32(_p(x), _p(y), x.numel() // (H * W_), H, W_, int(oh), int(ow), _stream()), "tmix_adaptive_avgpool_f32")
    return y


def linear_f32(x, w, bias=None, act_in=False, act_out=False):
    """x [M<=256,K] fp32, w [N,K] fp32 -> act_out(act_in(x) w^T + bias) [M,N] fp32 (tmix_linear_f32; act = SiLU)."""
    _need_cuda(x, w)
    _f32c(x, w, bias)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    L.check(L.load().tmix_linear_f32(_p(x), _p(w), _p(bias), _p(out), M, N, K, int(bool(act_in)), int(bool(act_out)), _stream()), "tmix_linear_f32")
    return out


def i2v_temporal_encoder(x, clips, frames, p, name):
    """x [clips*frames, 4, H, W] fp32 -> [clips, 4, frames, H, W] fp32: the whole image_latents_temporal_encoder block (tmix_i2v_temporal_encoder);
    p: fp32 state-dict entries under `name` (norm1, attn1.to_q/k/v, attn1.to_out.0, ff.net.0.proj, ff.net.2)."""
    _need_cuda(x)
    BF, Cc, H, W_ = x.shape
    assert BF == clips * frames
    ws = [p[name + k] for k in (".norm1.weight", ".norm1.bias", ".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_v.weight", ".attn1.to_out.0.weight",
                                ".attn1.to_out.0.bias", ".ff.net.0.proj.weight", ".ff.net.0.proj.bias", ".ff.net.2.weight", ".ff.net.2.bias")]
    _f32c(x, *ws)
    want = [(Cc,), (Cc,), (2 * Cc, Cc), (2 * Cc, Cc), (2 * Cc, Cc), (Cc, 2 * Cc), (Cc,), (4 * Cc, Cc), (4 * Cc,), (Cc, 4 * Cc), (Cc,)]
    assert [tuple(w.shape) for w in ws] == want, [tuple(w.shape) for w in ws]
    y = torch.empty(clips, Cc, frames, H, W_, device=x.device, dtype=torch.float32)
    L.check(L.load().tmix_i2v_temporal_encoder(_p(x), _p(y), clips, frames, Cc, H * W_, *[_p(w) for w in ws], _stream()), "tmix_i2v_temporal_encoder")
    return y


def softmax_rows_causal(scores, probs, seq, scale):
    """probs[r, :] = softmax(scale * scores[r, :c<=r%seq]) (bf16), zeros elsewhere; scores fp32 [rows, cols]."""
    _need_cuda(scores, probs)
    rows, cols = scores.shape
    assert scores.dtype == torch.float32 and probs.dtype == BF16 and probs.shape == scores.shape
    L.check(L.load().tmix_softmax_rows_causal(_p(scores), scores.stride(0), _p(probs), probs.stride(0), rows, cols, float(scale),
                                              int(seq), _stream()), "tmix_softmax_rows_causal")
    return probs


def vpred_step(x, v, g, at, at_next, out=None):
    """I2VGen-XL loop update (video_gen/pipeline_i2vgen_xl.py:699-719): x [B,...], v [2B,...] (uncond rows first) of one
    dtype; at / at_next from the un-shifted alpha table.  Returns the next latents (same dtype)."""
    _need_cuda(x, v)
    assert x.is_contiguous() and v.is_contiguous() and v.numel() == 2 * x.numel() and v.dtype == x.dtype
    out = torch.empty_like(x) if out is None else out
    f = np.float32
    sa, s1 = np.sqrt(f(at)), np.sqrt(f(1) - f(at))
    san, s1n = np.sqrt(f(at_next)), np.sqrt(f(1) - f(at_next))
    L.check(L.load().tmix_vpred_step(_p(x), _p(v), _p(out), _EPS_DT[x.dtype], x.numel(), float(g), float(sa), float(s1),
                                     float(san), float(s1n), _stream()), "tmix_vpred_step")
    return out


def frame_inject(x, clips, frames, interp=None):
    """first-frame feature injection (video_gen/utils_attn.py:433-455), in place on x [(clips*frames), ...] contiguous."""
    _need_cuda(x)
    assert x.is_contiguous() and x.shape[0] == clips * frames
    per = x.numel() // (clips * frames)
    hard = interp is None
    a = 0.0 if hard else float(np.float32(interp))
    b = 0.0 if hard else float(np.float32(1.0 - float(interp)))
    L.check(L.load().tmix_frame_inject(_p(x), _EPS_DT[x.dtype], clips, frames, per, int(hard), a, b, _stream()), "tmix_frame_inject")
    return x


def temporal_attention(qkv, clips, frames, heads, scale=None, out=None):
    """self-attention over the frame axis: qkv [(clips*frames), hw, 3*heads*64] bf16 -> [(clips*frames), hw, heads*64]."""
    _need_cuda(qkv)
    n, hw, ld = qkv.shape
    C3 = 3 * heads * 64
    assert n == clips * frames and qkv.dtype == BF16 and qkv.stride(2) == 1 and qkv.stride(0) == hw * qkv.stride(1) and ld >= C3
    out = torch.empty(n, hw, heads * 64, device=qkv.device, dtype=BF16) if out is None else out
    L.check(L.load().tmix_temporal_attn(_p(qkv), qkv.stride(1), _p(out), out.stride(1), clips, frames, hw, heads,
                                        float(scale if scale is not None else 64 ** -0.5), _stream()), "tmix_temporal_attn")
    return out


def softmax_rows_masked(scores, probs, valid, scale):
    """probs[r, c] = softmax over the first `valid` columns of scale * scores[r] (bf16), zeros in the padding columns."""
    _need_cuda(scores, probs)
    rows, cols = scores.shape
    assert scores.dtype == torch.float32 and probs.dtype == BF16 and probs.shape == scores.shape
    L.check(L.load().tmix_softmax_rows_masked(_p(scores), scores.stride(0), _p(probs), probs.stride(0), rows, cols, int(valid),
                                              float(scale), _stream()), "tmix_softmax_rows_masked")
    return probs


def lora_down(a_pad, K, D, P, nsets, sets, rows_per_set, dcolsum=None, dbias=None, eps=1e-5):
    """fill the 64 pad columns behind the K values of every row of a_pad [rows, K + 64] (bf16, in place) with the rows' LoRA
    down-projections (tmix_lora_down); sets: int32 device tensor, one concept set per block of rows_per_set rows."""
    _need_cuda(a_pad, D, sets)
    assert a_pad.dtype == BF16 and a_pad.dim() == 2 and a_pad.stride(1) == 1 and a_pad.shape[1] >= K + 64 and sets.dtype == torch.int32
    assert D.dtype == BF16 and D.is_contiguous() and D.shape == (nsets * P, K)
    lib = L.load()
    L.check(lib.tmix_lora_down(a_pad.data_ptr(), a_pad.stride(0), K, a_pad.shape[0], D.data_ptr(), P, nsets,
                               None if dcolsum is None else dcolsum.data_ptr(), None if dbias is None else dbias.data_ptr(), eps,
                               sets.data_ptr(), rows_per_set, _stream()), "tmix_lora_down")
    return a_pad

