"""Tweedie-mix sampler: host-side mirror of the reference's `Tweediemix` hot path.

Same method names, argument meaning and step semantics as fusion_generation/fusion_sampling.py
(`alpha` :305-307, `denoise_step` :309-474, `init_fusion` :476-483, `run_fusion` :485-489,
`sample_loop` :490-530) and fusion_sampling_lora.py (`--t_stop` window :324,378,476-492), with

* phase decisions and alpha tables on the host as plain ints/floats (the reference syncs the device
  several times per step through `.item()` and CPU-tensor indexing),
* one UNet launch plan per call kind (fusion / start / plain), each with its own cross-attention K/V
  cache,
* CFG + Tweedie + blend + DDIM done by ONE kernel (tmix_fused_tweedie_step_dev) that updates the latent
  state in place,
* ONE hipGraph per (call kind, step mode) holding the whole step -- latent broadcast + timestep
  (tmix_step_prologue), the UNet launch chains, the fused step for all co-batched seeds: a timestep is one 32-byte
  parameter upload and one graph replay.

The constructor takes prompt embeddings, masks and weights as tensors; `tweediemix_amd/text.py` (tokenizer + text
towers), `vae.py` (final / preview decode) and `masks.py` (side-car contract) produce them from the reference's inputs
-- the CLI `fusion_generation/fusion_sampling.py` wires them together.  The segmentation process itself
(GroundingDINO + SAM) stays an external command.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import lib as L
from . import ops
from .schedule import Schedule
from .unet import KVCache, PlanGroup, UNetPlan, UNetWeights

F32 = torch.float32


def seed_everything(seed: int):
    """utils_custom.py:10-14"""
    import random

    import numpy as np
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


DEFAULTS = dict(seed=182, guidance_scale=9.0, n_timesteps=50, t_cond=0.4, t_stop=0.9, resampling_steps=10,
                jumping_steps=5, resolution_h=1024, resolution_w=1024, crops_coords_top_left_h=0,
                crops_coords_top_left_w=0)      # argparse defaults of fusion_sampling.py:534-585


def make_config(**kw):
    d = dict(DEFAULTS)
    d.update(kw)
    return SimpleNamespace(**d)


class Tweediemix:
    """config: namespace with the reference's flag names (guidance_scale, n_timesteps, t_cond,
    [t_stop], resampling_steps, jumping_steps, resolution_h/w, crops_coords_top_left_h/w, seed).

    weights           UNetWeights (base UNet + K concept weight sets, kind 'custom' | 'lora' | 'none')
    text_embeds       ([K+2,77,D], [K+2,P])  rows: 0 uncond, 1 multi-concept prompt, 2.. per-concept prompts
    text_embeds_single([K,77,D],   [K,P])    rows: 0 uncond, 1.. single-concept prompts without modifier tokens
    mask_provider     callable(x0_preview [1,4,h,w]) -> masks [K,1,h,w] fp32 (stands in for decode +
                      run_expand.py + preprocess_mask at fusion_sampling.py:453-469)
    lora              True selects the fusion_sampling_lora.py window semantics (needs config.t_stop)
    strict_reference  keep the hooks' hard-coded `batch == 4` routing test (utils_custom.py:62)
    """

    def __init__(self, config, weights: UNetWeights, text_embeds, text_embeds_single, mask_provider,
                 concept_num: int, lora: bool = False, strict_reference: bool = True, use_graphs: bool = False,
                 n_seeds: int = 1, n_streams: int = 1, vae=None, fp8: bool = False):
        self.config = config
        self.fp8 = bool(fp8)          # optional: FF / QKV projections on e4m3 operands (tmix_gemm_fp8); default bf16 like the reference's fp16
        self.W = weights
        self.device = weights.device
        self.concept_num = int(concept_num)
        self.lora = bool(lora)
        self.strict_reference = strict_reference
        self.use_graphs = use_graphs
        # n_seeds independent trajectories share every UNet launch (rows [seed][uncond, concepts...]); the
        # reference runs one seed per process -- co-batching only raises the GEMM M dimension.
        self.n_seeds = int(n_seeds)
        # n_streams > 1 splits the rows of every UNet call into that many independent launch chains (PlanGroup)
        self.n_streams = int(n_streams)
        # optional VAE decoder: (config, state_dict) of tweediemix_amd.vae -- enables decode_latent / decoded outputs
        self.vae = vae
        self.vae_scaling_factor = 0.13025       # SDXL VAE config value; the CLI overrides it from the checkpoint's vae/config.json
        self._vae_plans = {}
        # a call is split into chains only when each keeps >= 2 batch rows: the B = 2 CFG-pair calls run faster as ONE chain
        # with the one-workgroup-per-CU tilings (25.3 ms) than as two single-row chains (27.5 ms)
        self.min_rows_per_stream = 2
        self.text_embeds = text_embeds
        self.text_embeds_single = text_embeds_single
        self.mask_provider = mask_provider
        self._mask_buf = None
        self.scheduler = Schedule(config.n_timesteps)
        self.skip = self.scheduler.skip
        self.final_alpha_cumprod = self.scheduler.final_alpha_cumprod
        self.h, self.w = config.resolution_h // 8, config.resolution_w // 8
        # compute_time_ids, fusion_sampling.py:70-78
        self.add_time_ids = torch.tensor([[config.resolution_h, config.resolution_w, config.crops_coords_top_left_h,
                                           config.crops_coords_top_left_w, config.resolution_h, config.resolution_w]],
                                         dtype=F32)
        self.plans = {}
        self.graphs = {}
        self.unet_calls = []          # (kind, B, t) trace, for tests / accounting
        self.preview_x0 = None
        S = self.n_seeds
        # latent state of the running trajectories (updated in place by every step), its Tweedie estimate, a backup for
        # the jumping look-ahead (fusion_sampling.py:431-447 does not move the trajectory), and the step parameters
        self.x_state = torch.zeros(S, 4, self.h, self.w, device=self.device, dtype=F32)
        self.x0_state = torch.zeros_like(self.x_state)
        self._x_backup = torch.zeros_like(self.x_state)
        self.step_params = torch.zeros(8, device=self.device, dtype=F32)      # {t, sa, s1, sa_next, s1_next, is_last, g, -}
        # pinned staging ring for the asynchronous parameter upload: a slot is rewritten only after the copy that read it
        # has completed (the host runs ahead of the device by whole steps)
        self._hp = torch.zeros(64, 8, dtype=F32)
        if self.device.type == "cuda":
            self._hp = self._hp.pin_memory()
        self._hp_ev = [None] * self._hp.shape[0]
        self._hp_i = 0
        self._mask_buf = None            # fixed-address copy of self.masks that the captured fusion step reads

    # ------------------------------------------------------------------ schedule
    def alpha(self, t):
        return self.scheduler.alpha(int(t))

    # ------------------------------------------------------------------ plans
    def _routes(self, B):
        return B == 4 if self.strict_reference else B == self.concept_num + 1

    def _build_plan(self, kind):
        K = self.concept_num
        te, tp = self.text_embeds
        if kind in ("fusion", "fusion_base"):
            ehs = torch.cat([te[0:1], te[2:2 + K]])
            pooled = torch.cat([tp[0:1], tp[2:2 + K]])
            routed = kind == "fusion" and self._routes(K + 1) and self.W.kind != "none"
            wsel = list(range(K + 1)) if routed else [0] * (K + 1)
        elif kind == "start":
            ts_, tps = self.text_embeds_single
            ehs = torch.cat([te[0:1], te[1:2], ts_[1:K]])
            pooled = torch.cat([tp[0:1], tp[1:2], tps[1:K]])
            routed, wsel = False, [0] * (K + 1)
        elif kind == "plain":
            ehs, pooled, routed, wsel = te[0:2], tp[0:2], False, [0, 0]
        else:
            raise ValueError(kind)
        S = self.n_seeds
        if S > 1:                                     # seed-major rows: b = seed * rows_per_seed + row
            ehs, pooled, wsel = ehs.repeat(S, 1, 1), pooled.repeat(S, 1), list(wsel) * S
        B = ehs.shape[0]
        if self.n_streams > 1 and B % self.n_streams == 0 and B // self.n_streams >= self.min_rows_per_stream:
            return PlanGroup(self.W, self.h, self.w, ehs, wsel, pooled, self.add_time_ids.repeat(B, 1), routed,
                             self.n_streams, fp8=self.fp8)
        kv = KVCache(self.W, ehs, wsel)
        return UNetPlan(self.W, B, self.h, self.w, kv, pooled, self.add_time_ids.repeat(B, 1), routed=routed,
                        row_sets=wsel if routed else None, fp8=self.fp8)

    def plan(self, kind):
        if kind not in self.plans:
            self.plans[kind] = self._build_plan(kind)
        return self.plans[kind]

    def _unet(self, kind, x, t):
        """eps [n_seeds*rows,4,h,w] fp32 for the call kind's prompt rows; each seed's latent is broadcast over
        its rows.  (UNet call alone, eager: used by tests and tools; the sampler itself runs `_run_step`.)"""
        p = self.plan(kind)
        S = self.n_seeds
        self.unet_calls.append((kind, p.B // S, int(t)))
        p.latent.view(S, p.B // S, *p.latent.shape[1:]).copy_(x.unsqueeze(1))
        p.t_dev.fill_(float(t))
        p.run()
        return p.eps

    # ------------------------------------------------------------------ one whole step = one graph
    def _set_masks(self, masks):
        """masks [K,1,h,w] (one seed) or [n_seeds,K,1,h,w]: kept at a fixed address, because captured steps read it."""
        masks = masks.to(self.device, F32).contiguous()
        if self._mask_buf is None or self._mask_buf.shape != masks.shape:
            assert not any(k[1] == L.STEP_FUSION for k in self.graphs), "mask shape changed after the fusion step was captured"
            self._mask_buf = torch.empty_like(masks)
        self._mask_buf.copy_(masks)

    @property
    def masks(self):
        return self._mask_buf

    @masks.setter
    def masks(self, m):
        if m is None:
            self._mask_buf = None
        else:
            self._set_masks(m)

    def _enqueue_step(self, kind, mode):
        """the launches of one denoising step on the current stream: prologue, UNet chains, fused step (in place)."""
        p = self.plan(kind)
        S = self.n_seeds
        rows = p.B // S
        n = 4 * self.h * self.w
        lib = L.load()
        st = torch.cuda.current_stream().cuda_stream
        L.check(lib.tmix_step_prologue(self.x_state.data_ptr(), p.latent.data_ptr(), p.t_dev.data_ptr(),
                                       self.step_params.data_ptr(), S, rows, n, st), "tmix_step_prologue")
        p.run()
        m = self._mask_buf if mode == L.STEP_FUSION else None
        mss = 0 if (m is None or m.dim() == 4) else self.concept_num * self.h * self.w
        L.check(lib.tmix_fused_tweedie_step_dev(self.x_state.data_ptr(), p.eps.data_ptr(), L.F32, None if m is None else m.data_ptr(),
                                                mss, self.x_state.data_ptr(), self.x0_state.data_ptr(), self.concept_num, 4,
                                                self.h * self.w, mode, rows, S, self.step_params.data_ptr(), st),
                "tmix_fused_tweedie_step_dev")

    def _run_step(self, kind, mode, t, at, at_next, is_last=False):
        """x_state <- step(x_state) for every seed; x0_state <- the Tweedie estimate.  Host work per step: eight floats."""
        if "_unet" not in vars(self):
            self.unet_calls.append((kind, self.plan(kind).B // self.n_seeds, int(t)))
        sa, s1, san, s1n = ops.step_coeffs(at, at_next)
        i = self._hp_i
        self._hp_i = (i + 1) % self._hp.shape[0]
        if self._hp_ev[i] is not None:
            self._hp_ev[i].synchronize()
        hp = self._hp[i]
        hp[0], hp[1], hp[2], hp[3], hp[4] = float(t), sa, s1, san, s1n
        hp[5], hp[6] = (1.0 if is_last else 0.0), float(self.config.guidance_scale)
        self.step_params.copy_(hp, non_blocking=True)
        if self.device.type == "cuda":
            self._hp_ev[i] = torch.cuda.Event()
            self._hp_ev[i].record()
        if mode == L.STEP_FUSION:
            assert self._mask_buf is not None, "fusion step before the masks were acquired"
        if "_unet" in vars(self):
            # a stand-in UNet was attached to this instance (tests replay recorded eps; a caller may plug the reference's own
            # module in): the same fused step, eagerly, on whatever eps dtype the stand-in returns
            eps = self._unet(kind, self.x_state, t).contiguous()
            S = self.n_seeds
            m = self._mask_buf if mode == L.STEP_FUSION else None
            mss = 0 if (m is None or m.dim() == 4) else self.concept_num * self.h * self.w
            L.check(L.load().tmix_fused_tweedie_step_dev(
                self.x_state.data_ptr(), eps.data_ptr(), ops._EPS_DT[eps.dtype], None if m is None else m.data_ptr(), mss,
                self.x_state.data_ptr(), self.x0_state.data_ptr(), self.concept_num, 4, self.h * self.w, mode, eps.shape[0] // S, S,
                self.step_params.data_ptr(), torch.cuda.current_stream().cuda_stream), "tmix_fused_tweedie_step_dev")
            return
        if not self.use_graphs:
            self._enqueue_step(kind, mode)
            return
        g = self.graphs.get((kind, mode))
        if g is None:
            self._enqueue_step(kind, mode)                # warm-up outside capture (kernel attributes, lazy module load);
            torch.cuda.synchronize()                      # it has already performed this step, so no replay now
            g = torch.cuda.CUDAGraph()
            keep = self.x_state.clone()
            with torch.cuda.graph(g):
                self._enqueue_step(kind, mode)
            self.x_state.copy_(keep)                      # capture does not execute, but keep the state explicit
            self.graphs[(kind, mode)] = g
            return
        g.replay()

    # ------------------------------------------------------------------ VAE
    def _decode(self, latent, inv_scale):
        if self.vae is None:
            raise L.TmixError("no VAE weights were given to Tweediemix(vae=(config, state_dict))")
        from .vae import decode_in_groups
        return decode_in_groups(self.vae, latent, inv_scale, self._vae_plans, self.device)

    @torch.no_grad()
    def decode_latent(self, latent):
        """fusion_sampling.py:297-303: the PREVIEW decode, with the reference's 1/0.18215 scale (not SDXL's 0.13025)."""
        return self._decode(latent, 1 / 0.18215)

    @torch.no_grad()
    def decode_final(self, latent):
        """fusion_sampling.py:496-524: x / vae.config.scaling_factor (0.13025) -> decoder -> (img/2+0.5).clamp(0,1)."""
        return self._decode(latent, 1 / self.vae_scaling_factor)

    # ------------------------------------------------------------------ phases
    def init_fusion(self, t_cond, t_stop=None):
        ts = self.scheduler.timesteps
        if self.lora:
            assert t_stop is not None
            self.t_cond = ts[t_cond:t_stop] if t_cond >= 0 else []       # fusion_sampling_lora.py:477
            self.t_stop_cur = ts[t_stop]
        else:
            self.t_cond = ts[t_cond:] if t_cond >= 0 else []             # fusion_sampling.py:477
            self.t_stop_cur = None
        self._window = set(self.t_cond)
        self.t_cond_prev = ts[t_cond - 1]
        self.t_cond_cur = ts[t_cond]
        self.start_t = ts[0]

    def _in_fusion(self, t):
        if self.lora:
            return t <= self.t_cond_cur and t >= self.t_stop_cur
        return t <= self.t_cond_cur

    def _step(self, x, eps, mode, at, at_next, is_last=False, out=None, out_x0=None):
        """fused CFG/Tweedie/blend/DDIM with host-side coefficients (scalar ABI form; tests and tools)."""
        S = self.n_seeds
        if out is None:
            out = torch.empty_like(x)
        rows = eps.shape[0] // S
        for sd in range(S):
            m = None
            if mode == L.STEP_FUSION:
                m = self.masks if self.masks.dim() == 4 else self.masks[sd]
            ops.fused_tweedie_step(x[sd:sd + 1], eps[sd * rows:(sd + 1) * rows], m, mode, self.concept_num,
                                   self.config.guidance_scale, at, at_next, is_last, out_x=out[sd:sd + 1],
                                   out_x0=None if out_x0 is None else out_x0[sd:sd + 1])
        return out

    def _denoise_inplace(self, t):
        """one scheduler timestep on x_state (fusion_sampling.py:309-474)."""
        t = int(t)
        cfg = self.config
        next_t = t - self.skip
        at, at_next = self.alpha(t), self.alpha(next_t)
        last = t == 1
        if self._in_fusion(t):
            kind = "fusion" if (t in self._window) else "fusion_base"
            self._run_step(kind, L.STEP_FUSION, t, at, at_next, last)
        elif t == self.start_t:
            for _ in range(cfg.resampling_steps):
                self._run_step("start", L.STEP_RESAMPLE, t, at, at_next)
                self._run_step("plain", L.STEP_PLAIN, next_t, at_next, at)        # Tweedie at next_t, re-noise to t
            self._run_step("start", L.STEP_PLAIN, t, at, at_next, last)
        else:
            self._run_step("plain", L.STEP_PLAIN, t, at, at_next, last)

        if t == self.t_cond_prev:                       # fusion_sampling.py:431-469
            self._x_backup.copy_(self.x_state)          # the look-ahead does not move the trajectory
            tt = next_t
            for _ in range(cfg.jumping_steps):
                a_t = self.alpha(tt)
                self._run_step("plain", L.STEP_PLAIN, tt, a_t, self.alpha(tt - 150))
                tt = tt - 150
            self.preview_x0 = (self.x0_state if cfg.jumping_steps else self._x_backup_x0()).clone()
            self.x_state.copy_(self._x_backup)
            if self.n_seeds == 1:
                m = self.mask_provider(self.preview_x0).to(self.device, F32).contiguous()
                assert m.shape[0] == self.concept_num
            else:                                     # one mask set per seed: [n_seeds, K, 1, h, w]
                m = torch.stack([self.mask_provider(self.preview_x0[i:i + 1]).to(self.device, F32)
                                 for i in range(self.n_seeds)]).contiguous()
            self._set_masks(m)

    def _x_backup_x0(self):
        return self.x0_state            # jumping_steps == 0: the preview is the Tweedie estimate of the step just taken

    @torch.no_grad()
    def denoise_step(self, x, t):
        """x [n_seeds,4,h,w] fp32 on the device, t python int (or 0-dim tensor). Returns the next latent(s)."""
        self.x_state.copy_(x)
        self._denoise_inplace(t)
        return self.x_state.clone()

    def run_fusion(self, x=None, decode=False):
        cfg = self.config
        t_cond = int(cfg.n_timesteps * cfg.t_cond)
        if self.lora:
            self.init_fusion(t_cond, int(cfg.n_timesteps * cfg.t_stop))
        else:
            self.init_fusion(t_cond)
        if x is None:      # drawn on the CPU like the reference (fusion_sampling.py:488): device-independent seeds
            x = torch.randn(self.n_seeds, 4, self.h, self.w)
        x = x * self.scheduler.init_noise_sigma          # :488 (1.0 for this scheduler), however x arrived
        return self.sample_loop(x.to(self.device, F32), decode=decode)

    @torch.no_grad()
    def sample_loop(self, x, decode=False):
        """runs every scheduler timestep; returns the final latent, or the decoded image [n,3,H,W] in [0,1] when
        decode=True and VAE weights were given (fusion_sampling.py:496-528)."""
        self.x_state.copy_(x)
        for t in self.scheduler.timesteps:
            self._denoise_inplace(t)
        x = self.x_state.clone()
        return self.decode_final(x) if decode else x
