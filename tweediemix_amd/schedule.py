"""DDIM tables of the image sampler, kept on the HOST as plain floats/ints (no device syncs).

Mirrors fusion_generation/fusion_sampling.py:212-218 (scheduler set-up) and :305-307 (alpha):
SDXL's DDIMScheduler is scaled-linear beta 0.00085..0.012 over 1000 steps, 'leading' spacing,
steps_offset=1, set_alpha_to_one=False; the reference prepends 1.0 to alphas_cumprod, so
alpha(t) = alphas_cumprod[t-1], and alpha(t<0) = final_alpha_cumprod = alphas_cumprod[0].
"""
from __future__ import annotations

import numpy as np
import torch


class Schedule:
    def __init__(self, n_timesteps: int, num_train_timesteps: int = 1000, beta_start: float = 0.00085,
                 beta_end: float = 0.012, steps_offset: int = 1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0).numpy()
        self.alphas_cumprod = np.concatenate([np.ones(1, np.float32), acp]).astype(np.float32)   # :218
        self.final_alpha_cumprod = np.float32(acp[0])
        self.n = int(n_timesteps)
        self.skip = num_train_timesteps // self.n                                                # :216
        ratio = num_train_timesteps // self.n
        self.timesteps = [int(v) for v in (np.arange(self.n)[::-1] * ratio + steps_offset)]
        self.init_noise_sigma = 1.0

    def alpha(self, t: int) -> np.float32:
        return self.alphas_cumprod[t] if t >= 0 else self.final_alpha_cumprod
