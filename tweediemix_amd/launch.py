"""One process per GPU without an outer launcher.

The reference is single-GPU (`sample_catdog.sh:3` pins CUDA_VISIBLE_DEVICES=0, one seed per process:
fusion_sampling.py:485-489); sharding independent seeds over the GPUs of a node is this build's own component
(SURVEY 8e).  `python bench.py --gpus N` / `python fusion_generation/fusion_sampling.py --gpus N` call
`self_launch(N)`: when no launcher has set WORLD_SIZE, the script re-executes itself N times with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment (the contract
`torch.distributed.run` provides, so the same script also runs under it unchanged) and waits for the ranks.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched() -> bool:
    """True inside a rank (torch.distributed.run or self_launch set the rendezvous environment)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch(n: int, argv=None, env_extra=None) -> int:
    """re-exec `sys.argv` as n ranks; returns the worst exit code.  Rank r's stdout/stderr are inherited, so the one
    JSON line rank 0 prints is the parent's output.  A rank that dies takes the others down (no hung rendezvous)."""
    argv = list(sys.argv if argv is None else argv)
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL across processes)
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    worst = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            try:
                rc = procs[r].wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            alive.discard(r)
            if rc != 0:
                worst = worst or rc
                for o in alive:                  # exactly the processes started above
                    procs[o].terminate()
    return worst


def rank_env():
    """(rank, local_rank, world) of this process; (0, 0, 1) when not launched."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
