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


def _json_lines(out):
    import json
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("num_seeds", [0, 5])
def test_bench_gpus2_launches_its_own_ranks(num_seeds):
    """`python bench.py --gpus 2` with NO outer launcher starts two ranks itself (tweediemix_amd/launch.py), rendezvous on
    127.0.0.1, barrier-bracketed timing, MAX over ranks, seed sharding + latent gather, ONE line from rank 0 with n_gpus == 2.
    CPU stand-in step (--host-dry-run): the control flow is the GPU run's, the kernels are not involved."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--host-dry-run", "--steps", "3", "--warmup", "1"]
    if num_seeds:
        cmd += ["--num-seeds", str(num_seeds)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["gather_ok"] is True
    assert d["config"]["seeds_total"] == (num_seeds or 2) and "DRY RUN" in d["data"]


def test_self_launch_propagates_failure(tmp_path):
    """a rank that dies takes the job down with a non-zero exit code (no hung rendezvous)."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "job.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {root!r})
        from tweediemix_amd import launch as LA
        if not LA.launched():
            sys.exit(LA.self_launch(2))
        rank, local, world = LA.rank_env()
        assert world == 2 and os.environ["MASTER_ADDR"] == "127.0.0.1"
        if rank == 1:
            sys.exit(7)
        time.sleep(30)
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(script)], timeout=25, env=env)
    assert r.returncode == 7
