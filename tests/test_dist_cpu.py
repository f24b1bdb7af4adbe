"""CPU, world_size 2, gloo: the N>1 path of the sampler (seed sharding + the final latent gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tweediemix_amd import dist as D
    seeds = list(range(100, 100 + n_total))
    mine = D.seed_shard(seeds, rank, world)
    # a stand-in "trajectory": the final latent of seed s is a deterministic function of s only
    local = torch.stack([torch.full((4, 3, 5), float(s)) + torch.arange(5.0) for s in mine]) if mine else torch.zeros(0, 4, 3, 5)
    allx = D.gather_latents(local, n_total, rank, world)
    ok = all(torch.equal(allx[i], torch.full((4, 3, 5), float(s)) + torch.arange(5.0)) for i, s in enumerate(seeds))
    tmax = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, ok, tmax, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])
def test_seed_shard_and_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _r, ok, _t, _n in res)
    assert all(t == 2.0 for _r, _ok, t, _n in res)            # MAX over ranks
    assert sum(n for *_x, n in res) == n_total


def test_seed_shard_round_robin():
    from tweediemix_amd import dist as D
    assert D.seed_shard(list(range(10)), 1, 4) == [1, 5, 9]
    assert sum(len(D.seed_shard(list(range(64)), r, 8)) for r in range(8)) == 64
