"""Blend-mask preparation (host side, once per image).

preprocess_mask / build_masks follow fusion_generation/fusion_sampling.py:81-89 and :461-469: the
segmentation side-car's '<concept>.jpg' (8-bit grey) -> /255 -> threshold 0.5 -> nearest resize to the
latent grid -> masks = [fg_1..fg_{K-1}, clamp(1 - sum fg, 0)].  random_rectangle_masks is the synthetic
stand-in for the side-car used by the benchmark (run_expand.py:50-51 emits bounding rectangles).
"""
from __future__ import annotations

import numpy as np
import torch


def preprocess_mask(mask, h: int, w: int, device="cuda") -> torch.Tensor:
    """mask: path or uint8 array [H,W]. Returns [1,1,h,w] fp32 {0,1} on `device`."""
    if isinstance(mask, (str, bytes)):
        from PIL import Image
        mask = np.array(Image.open(mask).convert("L"))
    m = np.asarray(mask).astype(np.float32) / np.float32(255.0)
    m = (m >= 0.5).astype(np.float32)
    H, W = m.shape
    ys = np.minimum(np.floor(np.arange(h, dtype=np.float32) * np.float32(H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=np.float32) * np.float32(W / w)).astype(np.int64), W - 1)
    return torch.from_numpy(np.ascontiguousarray(m[ys][:, xs]))[None, None].to(device)


def build_masks(fg_sources, h: int, w: int, device="cuda") -> torch.Tensor:
    """[K,1,h,w]: foreground masks then background = clamp(1 - sum fg, min 0) (NOT renormalised)."""
    fg = torch.cat([preprocess_mask(s, h, w, device) for s in fg_sources])
    bg = 1 - torch.sum(fg, dim=0, keepdim=True)
    bg[bg < 0] = 0
    return torch.cat([fg, bg]).contiguous()


def random_rectangle_masks(K: int, H: int, W: int, seed: int = 0):
    """K-1 seeded axis-aligned rectangles (10-30 % area each) on the H x W image grid, uint8 {0,255}."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(K - 1):
        area = rng.uniform(0.10, 0.30) * H * W
        ar = rng.uniform(0.6, 1.6)
        hh = int(min(H, max(8, round((area * ar) ** 0.5))))
        ww = int(min(W, max(8, round(area / hh))))
        y0 = rng.randint(0, H - hh + 1)
        x0 = rng.randint(0, W - ww + 1)
        m = np.zeros((H, W), np.uint8)
        m[y0:y0 + hh, x0:x0 + ww] = 255
        out.append(m)
    return out
