"""Weight preparation for the HIP kernels (layout transforms done once at load time)."""
from __future__ import annotations

import torch


def interleave_geglu(w: torch.Tensor, b: torch.Tensor | None):
    """diffusers GEGLU: proj = Linear(C, 8C); value = out[:, :4C], gate = out[:, 4C:].
    The GEMM epilogue fuses value*gelu(gate) when the weight rows alternate in groups of 16:
    [value 0..15 | gate 0..15 | value 16..31 | gate 16..31 | ...]."""
    n2 = w.shape[0]
    n = n2 // 2
    assert n % 16 == 0
    idx = torch.arange(n2, device=w.device).reshape(2, n // 16, 16).permute(1, 0, 2).reshape(-1)
    return w[idx].contiguous(), (None if b is None else b[idx].contiguous())
