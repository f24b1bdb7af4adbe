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
    if not dist.is_initialized():
        assert world <= 1, "gather_latents: world > 1 but no process group (dist.init was not called): the other ranks' seeds would be dropped silently"
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


def init(device, world: int, backend: str | None = None):
    """join the job's process group on `device`: backend 'nccl' (= RCCL on ROCm) for GPUs, also at world == 1 -- a one-rank
    group costs nothing per step and makes every run load librccl, bind the communicator to the device (`device_id`) and push
    the result-gather / timing collectives through it, so the N-rank path is the one-rank path with a larger world.  The
    rendezvous address defaults to 127.0.0.1 (the container hostname may not resolve).  A rendezvous bind that fails with
    EADDRINUSE exits with launch.EADDRINUSE_RC so that launch.self_launch can retry on another port."""
    import errno
    import os
    import sys
    if dist.is_initialized():
        return dist.get_backend()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        from .launch import free_port
        os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = backend or ("nccl" if device.type == "cuda" else "gloo")
    rank = int(os.environ.get("RANK", "0"))
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    except (OSError, RuntimeError) as e:
        if getattr(e, "errno", None) == errno.EADDRINUSE or "EADDRINUSE" in str(e) or "address already in use" in str(e).lower():
            from .launch import EADDRINUSE_RC
            sys.exit(EADDRINUSE_RC)
        raise
    return backend


def ranks_seen(device) -> int:
    """all-reduce of a one: how many ranks the collective really spanned (printed in the bench line)."""
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t)
    return int(t.item())
