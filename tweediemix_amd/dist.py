"""Multi-GPU: independent seeds shard over ranks (one process per GPU); the only collective on this
path is the final gather of latents (RCCL over xGMI via torch.distributed backend 'nccl'; the same
code runs over 'gloo' on CPU in the tests).  There is no exchange step inside a trajectory: the
Tweedie blend is per-pixel within one sample and all K+1 batch rows of a UNet call live on one GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def seed_shard(seeds, rank: int, world: int):
    """round-robin: seed i -> rank i % world."""
    return [s for i, s in enumerate(seeds) if i % world == rank]


def gather_latents(local: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """local [n_local, C, h, w] (this rank's seeds, round-robin order) -> [n_total, C, h, w] in global
    seed order on every rank.  Ragged shards (n_total % world != 0) are padded to the largest shard so a
    single fixed-size all_gather_into_tensor suffices (one ring pass over xGMI, <= 16 MiB for 64 seeds)."""
    if world == 1:
        return local
    per = (n_total + world - 1) // world
    shape = local.shape[1:]
    pad = torch.zeros(per, *shape, dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty(world * per, *shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    out = out.view(world, per, *shape)
    res = torch.empty(n_total, *shape, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = list(range(r, n_total, world))
        res[idx] = out[r, :len(idx)]
    return res


def max_over_ranks(seconds: float, device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
