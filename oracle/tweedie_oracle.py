"""CPU oracle for the Tweedie-mix denoising loop -- TEST INFRASTRUCTURE ONLY.

This is a plain numpy/torch-fp32 restatement of the reference's own arithmetic for the hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product package (tweediemix_amd/) never does.

Pinned against: tests/golden/*.npz, produced by oracle/gen_golden.py from the reference's own
modules (fusion_sampling.py, fusion_sampling_lora.py, utils_custom.py, utils_lora.py) running
on CPU in the build container.  The SDXL UNet itself lives in diffusers==0.29.2 (not vendored
in the reference, not installed): its restatement (oracle/unet_oracle.py) is "parity unpinned".

Reference lines followed (relative to /root/reference/fusion_generation/):
  schedule / alpha(t)            fusion_sampling.py:212-218, 305-307
  denoise_step                   fusion_sampling.py:309-474   (lora window: fusion_sampling_lora.py:324,378)
  init_fusion / run_fusion       fusion_sampling.py:476-489   (lora: fusion_sampling_lora.py:476-492)
  preprocess_mask + bg mask      fusion_sampling.py:81-89, 466-469
  sa_forward (Custom Diffusion)  utils_custom.py:53-108
  sa_forward (LoRA)              utils_lora.py:55-123 ; LoRALinearLayer model_lora.py:28-48
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------ schedule
def alphas_cumprod_sdxl() -> np.ndarray:
    """DDIMScheduler(scaled_linear 0.00085..0.012, 1000 steps) cumulative alphas, fp32
    (diffusers computes the table in torch.float32; linspace/cumprod are reproduced in fp32)."""
    import torch
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).numpy()


def timesteps_leading(n: int, steps_offset: int = 1) -> np.ndarray:
    """'leading' spacing with steps_offset=1: arange(n)[::-1]*(1000//n)+1."""
    return (np.arange(n)[::-1] * (1000 // n) + steps_offset).astype(np.int64)


class Schedule:
    """alpha(t) of fusion_sampling.py:305-307 with the 1.0-prepended table of :218."""

    def __init__(self, n: int):
        ac = alphas_cumprod_sdxl()
        self.table = np.concatenate([np.ones(1, F32), ac]).astype(F32)   # alpha(t) = acp[t-1]
        self.final = F32(ac[0])
        self.n = n
        self.skip = 1000 // n
        self.timesteps = timesteps_leading(n)

    def alpha(self, t: int) -> np.float32:
        return self.table[t] if t >= 0 else self.final


# --------------------------------------------------------------------- masks
def preprocess_mask(img_u8: np.ndarray, h: int, w: int) -> np.ndarray:
    """fusion_sampling.py:81-89: /255, threshold .5, nearest resize (src = floor(i*H/h))."""
    m = img_u8.astype(F32) / F32(255.0)
    m = np.where(m < 0.5, F32(0), F32(1)).astype(F32)
    H, W = m.shape
    # torch 'nearest': src index = floor(dst * scale), scale = in/out (float32 arithmetic)
    ys = np.minimum(np.floor(np.arange(h, dtype=F32) * F32(H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=F32) * F32(W / w)).astype(np.int64), W - 1)
    return m[ys][:, xs][None, None]


def build_masks(fg_imgs, h, w) -> np.ndarray:
    """fusion_sampling.py:466-469: masks = cat(fg..., clamp(1 - sum fg, 0))."""
    fg = np.concatenate([preprocess_mask(a, h, w) for a in fg_imgs], axis=0)
    bg = 1 - fg.sum(axis=0, keepdims=True)
    bg[bg < 0] = 0
    return np.concatenate([fg, bg], axis=0).astype(F32)


def expand_masks(masks):
    """The side-car's post-processing of the two SAM masks (text_segment/run_expand.py:35-87): every mask becomes its
    bounding rectangle; where the two rectangles overlap, the bounding box of the overlap is re-filled with the
    ORIGINAL masks restricted to the overlap, and mask 1 loses it entirely when more than 80 % of original mask 0
    lies inside the overlap.  Like the reference this handles exactly two foreground masks (:62); for any other
    count only the rectangles are produced.  masks: list of bool arrays [H,W]; returns list of bool arrays."""
    import numpy as np
    orig = [np.asarray(m).astype(bool) for m in masks]
    rect = []
    for m in orig:
        ys, xs = np.nonzero(m)
        r = np.zeros_like(m)
        r[ys.min():ys.max() + 1, xs.min():xs.max() + 1] = True
        rect.append(r)
    if len(rect) == 2:
        ov = rect[0] & rect[1]
        if ov.any():
            ys, xs = np.nonzero(ov)
            y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
            o1 = ov & orig[0]
            o2 = ov & orig[1]
            if o1.sum() / orig[0].sum() > 0.8:
                o2 = np.zeros_like(o2)
            rect[0][y0:y1, x0:x1] = o1[y0:y1, x0:x1]
            rect[1][y0:y1, x0:x1] = o2[y0:y1, x0:x1]
    return rect


# ------------------------------------------------------- fused step arithmetic
def _h(a, lowp):
    """round to the low-precision eps dtype when emulating the reference's autocast dtypes."""
    return a.astype(lowp).astype(F32) if lowp is not None else a


# How a 0-dim fp32 *tensor* (sqrt(1-at)) multiplies an fp16 tensor differs by device in torch:
# CUDA keeps the scalar in fp32 (opmath) and rounds the product once -- what the reference really
# executes; torch-CPU rounds the scalar to fp16 first -- what oracle/gen_golden.py executed.  The
# golden-vector tests flip this switch to pin the restatement bit-for-bit against the CPU fixtures.
CPU_SCALAR_TENSOR_SEMANTICS = False


def _s(scalar, lowp):
    if lowp is not None and CPU_SCALAR_TENSOR_SEMANTICS:
        return F32(np.asarray(scalar, F32).astype(lowp))
    return F32(scalar)


def cfg_combine(eps_u, eps_c, g, lowp=None):
    """eps_u + g*(eps_c-eps_u); in the reference this is evaluated in the eps dtype (fp16)."""
    d = _h(eps_c - eps_u, lowp)
    gd = _h(F32(g) * d, lowp)
    return _h(eps_u + gd, lowp)


def tweedie_x0(x, eps_hat, at, lowp=None):
    """(x - sqrt(1-at)*eps_hat)/sqrt(at): the product is in the eps dtype, the rest fp32."""
    s1 = np.sqrt(F32(1) - F32(at)).astype(F32)
    sa = np.sqrt(F32(at)).astype(F32)
    return ((x - _h(_s(s1, lowp) * eps_hat, lowp)) / sa).astype(F32)


def ddim_move(x0, eps_u, at_next, lowp=None):
    """sqrt(a')*x0 + sqrt(1-a')*eps_u  (fusion_sampling.py:430), second product in eps dtype."""
    sa = np.sqrt(F32(at_next)).astype(F32)
    s1 = np.sqrt(F32(1) - F32(at_next)).astype(F32)
    return (sa * x0 + _h(_s(s1, lowp) * eps_u, lowp)).astype(F32)


def fused_fusion_step(x, eps, masks, g, at, at_next, is_last, lowp=None):
    """fusion branch :376-385, 430, 471-472.  eps [K+1,C,h,w], masks [K,1,h,w], x [1,C,h,w]."""
    K = masks.shape[0]
    eps_u = eps[:1]
    x0 = np.zeros_like(x, dtype=F32)
    for c in range(K):
        e = cfg_combine(eps_u, eps[1 + c:2 + c], g, lowp)
        x0 = (x0 + masks[c][None] * tweedie_x0(x, e, at, lowp)).astype(F32)
    xn = ddim_move(x0, eps_u, at_next, lowp)
    return (x0 if is_last else xn), x0


def fused_plain_step(x, eps, g, at, at_next, is_last, lowp=None):
    """plain CFG branch :424-430."""
    e = cfg_combine(eps[:1], eps[1:2], g, lowp)
    x0 = tweedie_x0(x, e, at, lowp)
    xn = ddim_move(x0, eps[:1], at_next, lowp)
    return (x0 if is_last else xn), x0


def fused_resample_down(x, eps, K, g, at, at_next, lowp=None):
    """resampling, first half :392-403: x0=(K-1)*x0_multi - sum x0_single ; move to next_t."""
    eps_u = eps[:1]
    e_m = cfg_combine(eps_u, eps[1:2], g, lowp)
    x0 = (F32(K - 1) * tweedie_x0(x, e_m, at, lowp)).astype(F32)
    for c in range(K - 1):
        e_s = cfg_combine(eps_u, eps[2 + c:3 + c], g, lowp)
        x0 = (x0 - tweedie_x0(x, e_s, at, lowp)).astype(F32)
    return ddim_move(x0, eps_u, at_next, lowp)


def fused_resample_up(x_down, eps_next, g, at, at_next, lowp=None):
    """resampling, second half :406-412: Tweedie at next_t, re-noise to t with predicted uncond eps."""
    e = cfg_combine(eps_next[:1], eps_next[1:2], g, lowp)
    x0n = tweedie_x0(x_down, e, at_next, lowp)
    return ddim_move(x0n, eps_next[:1], at, lowp)


# ------------------------------------------------------------------- sampler
class TweedieOracle:
    """Restatement of Tweediemix.{init_fusion,denoise_step} driving a user-supplied unet_fn.

    unet_fn(x[B,C,h,w], t:int, rows:list[(str,int)], kind:str) -> eps[B,C,h,w]
      rows identify text rows: ('e', r) = text_embeds[r], ('s', r) = text_embeds_single[r].
      kind in {'fusion','start','plain'} names the batch layout; 'routed' tells whether the
      attention hooks are inside the t_cond window for this call.
    """

    def __init__(self, K, n, g=0.8, t_cond=0.2, t_stop=None, resampling_steps=10, jumping_steps=5,
                 lowp=None, mask_fn=None, preview_fn=None):
        self.K, self.n, self.g = K, n, F32(g)
        self.sch = Schedule(n)
        self.resampling_steps, self.jumping_steps = resampling_steps, jumping_steps
        self.lowp = lowp
        self.mask_fn = mask_fn          # () -> masks [K,1,h,w]  (stands in for the side-car)
        self.preview_fn = preview_fn    # (x0) -> None
        self.masks = None
        ts = self.sch.timesteps
        ic = int(n * t_cond)
        self.lora = t_stop is not None
        if self.lora:                   # fusion_sampling_lora.py:476-479
            istop = int(n * t_stop)
            self.window = set(int(v) for v in ts[ic:istop])
            self.t_stop_cur = int(ts[istop])
        else:                           # fusion_sampling.py:477
            self.window = set(int(v) for v in ts[ic:])
            self.t_stop_cur = None
        self.t_cond_prev = int(ts[ic - 1])
        self.t_cond_cur = int(ts[ic])
        self.start_t = int(ts[0])
        self.requests = []

    def _in_fusion(self, t):
        if self.lora:
            return t <= self.t_cond_cur and t >= self.t_stop_cur
        return t <= self.t_cond_cur

    def _unet(self, unet_fn, x, B, t, rows, kind):
        routed = (t in self.window) and B == 4     # utils_custom.py:61-62 hard-codes batch 4
        self.requests.append((B, t, tuple(rows), kind, routed))
        xin = np.concatenate([x] * B, axis=0)
        return np.asarray(unet_fn(xin, t, rows, kind, routed), dtype=F32)

    def denoise_step(self, x, t, unet_fn):
        K, g, lp, sch = self.K, self.g, self.lowp, self.sch
        t = int(t)
        next_t = t - sch.skip
        at, at_next = sch.alpha(t), sch.alpha(next_t)
        rows_plain = [("e", 0), ("e", 1)]
        if self._in_fusion(t):
            rows = [("e", 0)] + [("e", 2 + c) for c in range(K)]
            eps = self._unet(unet_fn, x, K + 1, t, rows, "fusion")
            out, x0 = fused_fusion_step(x, eps, self.masks, g, at, at_next, False, lp)
            eps_u = eps[:1]
        elif t == self.start_t:
            rows = rows_plain + [("s", 1 + c) for c in range(K - 1)]
            eps = self._unet(unet_fn, x, K + 1, t, rows, "start")
            for _ in range(self.resampling_steps):
                xd = fused_resample_down(x, eps, K, g, at, at_next, lp)
                eps_n = self._unet(unet_fn, xd, 2, next_t, rows_plain, "plain")
                x = fused_resample_up(xd, eps_n, g, at, at_next, lp)
                eps = self._unet(unet_fn, x, K + 1, t, rows, "start")
            out, x0 = fused_plain_step(x, eps[:2], g, at, at_next, False, lp)
            eps_u = eps[:1]
        else:
            eps = self._unet(unet_fn, x, 2, t, rows_plain, "plain")
            out, x0 = fused_plain_step(x, eps, g, at, at_next, False, lp)
            eps_u = eps[:1]

        if t == self.t_cond_prev:                   # :431-469 jumping + mask acquisition
            xt, tt = out, next_t
            x0j = x0
            for _ in range(self.jumping_steps):
                a_t = sch.alpha(tt)
                eps_j = self._unet(unet_fn, xt, 2, tt, rows_plain, "plain")
                tt = tt - 150
                a_n = sch.alpha(tt)
                xt, x0j = fused_plain_step(xt, eps_j, g, a_t, a_n, False, lp)
            if self.preview_fn is not None:
                self.preview_fn(x0j)
            self.masks = np.asarray(self.mask_fn(), dtype=F32)
        if t == 1:
            out = x0
        return out


# --------------------------------------------------------------- attention hooks
# ------------------------------------------------------- video sampler (config #5) step arithmetic
def video_alpha(alphas_cumprod, final_alpha_cumprod, t):
    """video_gen/pipeline_i2vgen_xl.py:480-482: the UN-shifted table (alpha(t) = acp[t]), final_alpha_cumprod below 0."""
    return F32(alphas_cumprod[int(t)]) if t >= 0 else F32(final_alpha_cumprod)


def video_vpred_step(x, v, g, at, at_next, lowp=None):
    """pipeline_i2vgen_xl.py:699-719: CFG on the v-prediction, eps = sqrt(at) v + sqrt(1-at) x,
    x0 = sqrt(at) x - sqrt(1-at) v, x' = sqrt(at') x0 + sqrt(1-at') eps.  x [B,...], v [2B,...] (uncond rows first).
    Latents and predictions share the model dtype there, so with lowp every binary op rounds to it."""
    B = x.shape[0]
    vu, vt = v[:B], v[B:]
    vv = _h(vu + _h(F32(g) * _h(vt - vu, lowp), lowp), lowp)
    sa, s1 = np.sqrt(F32(at)).astype(F32), np.sqrt(F32(1) - F32(at)).astype(F32)
    san, s1n = np.sqrt(F32(at_next)).astype(F32), np.sqrt(F32(1) - F32(at_next)).astype(F32)
    eps = _h(_h(_s(sa, lowp) * vv, lowp) + _h(_s(s1, lowp) * x, lowp), lowp)
    x0 = _h(_h(_s(sa, lowp) * x, lowp) - _h(_s(s1, lowp) * vv, lowp), lowp)
    return _h(_h(_s(san, lowp) * x0, lowp) + _h(_s(s1n, lowp) * eps, lowp), lowp)


def inject_first_frame(x, batch, frames, interp=None, lowp=None):
    """video_gen/utils_attn.py:433-455 on a [(b t), ...] tensor: frames 1.. of every clip become the clip's first frame
    (interp None: hard copy, mid_block resnets) or interp*first + (1-interp)*frame (up_blocks[1].resnets[0])."""
    y = x.reshape(batch, frames, *x.shape[1:]).copy()
    first = y[:, :1]
    if interp is None:
        y[:, 1:] = first
    else:
        # interp is a Python float there: a wrapped number multiplies in fp32 on every device (1-interp formed in double)
        a, b = _h(F32(interp) * np.broadcast_to(first, y[:, 1:].shape), lowp), _h(F32(1.0 - float(interp)) * y[:, 1:], lowp)
        y[:, 1:] = _h(a + b, lowp)
    return y.reshape(x.shape)


def _heads(t, h):
    b, s, c = t.shape
    return t.reshape(b, s, h, c // h).transpose(0, 2, 1, 3)


def attention_core(q, k, v, heads, scale):
    """explicit softmax(QK^T*scale)V per head (utils_custom.py:93-103)."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    sim = np.einsum("bhid,bhjd->bhij", qh, kh).astype(F32) * F32(scale)
    sim = sim - sim.max(axis=-1, keepdims=True)
    p = np.exp(sim)
    p = p / p.sum(axis=-1, keepdims=True)
    o = np.einsum("bhij,bhjd->bhid", p, vh).astype(F32)
    b, h, s, d = o.shape
    return o.transpose(0, 2, 1, 3).reshape(b, s, h * d)


def sa_forward_custom(x, ehs, wq, wk, wv, wo, bo, wk_c, wv_c, heads, scale, routed):
    """utils_custom.py:53-108.  routed <=> is_cross and t in t_cond and ehs.shape[0]==4."""
    q = x @ wq.T
    src = ehs if ehs is not None else x
    if routed and ehs is not None and src.shape[0] == 4:
        k = np.concatenate([src[0:1] @ wk.T] + [src[i + 1:i + 2] @ wk_c[i].T for i in range(len(wk_c))])
        v = np.concatenate([src[0:1] @ wv.T] + [src[i + 1:i + 2] @ wv_c[i].T for i in range(len(wv_c))])
    else:
        k, v = src @ wk.T, src @ wv.T
    o = attention_core(q.astype(F32), k.astype(F32), v.astype(F32), heads, scale)
    return (o @ wo.T + bo).astype(F32)


def sa_forward_lora(x, ehs, wq, wk, wv, wo, bo, lora, heads, scale, routed):
    """utils_lora.py:55-123. lora[i] = dict(q=(down,up), k=..., v=..., out=...) for concept i."""
    src = ehs if ehs is not None else x
    q, k, v = x @ wq.T, src @ wk.T, src @ wv.T
    on = routed and src.shape[0] == 4
    if on:
        q, k, v = q.copy(), k.copy(), v.copy()
        for i, l in enumerate(lora):
            q[i + 1] += (x[i + 1] @ l["q"][0].T) @ l["q"][1].T
            k[i + 1] += (src[i + 1] @ l["k"][0].T) @ l["k"][1].T
            v[i + 1] += (src[i + 1] @ l["v"][0].T) @ l["v"][1].T
    o = attention_core(q.astype(F32), k.astype(F32), v.astype(F32), heads, scale)
    out = o @ wo.T + bo
    if on:
        for i, l in enumerate(lora):
            out[i + 1] += (o[i + 1] @ l["out"][0].T) @ l["out"][1].T
    return out.astype(F32)


def lora_merged_weight(w, down, up):
    """row i of the routed batch behaves as Linear(W + up@down) (SURVEY 3.2 [probe])."""
    return (w + up @ down).astype(F32)
