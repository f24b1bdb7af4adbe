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


def self_launch(n: int, argv=None, env_extra=None, retries: int = 2) -> int:
    """re-exec `sys.argv` as n ranks; returns the worst exit code.  Rank r's stdout/stderr are inherited, so the one
    JSON line rank 0 prints is the parent's output.  A rank that dies takes the others down (no hung rendezvous); so does
    SIGINT / SIGTERM to the parent or any exception in the wait loop (no orphaned GPU processes).  The rendezvous port is
    found by bind-and-release, so another job can take it before rank 0 binds it: a start that dies on EADDRINUSE within
    the first seconds is retried on a fresh port."""
    import signal
    import time
    argv = list(sys.argv if argv is None else argv)
    for attempt in range(retries + 1):
        port = free_port()
        procs = []
        t0 = time.time()

        def stop_all(sig=signal.SIGTERM):
            for pr in procs:                      # exactly the processes started below
                if pr.poll() is None:
                    try:
                        pr.send_signal(sig)
                    except OSError:
                        pass

        old = {}

        def forward(signum, _frame):
            stop_all(signum)
            raise KeyboardInterrupt

        for sg in (signal.SIGINT, signal.SIGTERM):
            try:
                old[sg] = signal.signal(sg, forward)
            except ValueError:                    # not the main thread: nothing to forward from
                pass
        worst = 0
        try:
            for r in range(n):
                env = dict(os.environ)
                env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
                env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL across processes)
                if env_extra:
                    env.update(env_extra)
                procs.append(subprocess.Popen([sys.executable] + argv, env=env))
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
                        stop_all()
        finally:
            stop_all()
            deadline = time.time() + 10
            for pr in procs:
                try:
                    pr.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    pr.kill()
            for sg, h in old.items():
                signal.signal(sg, h)
        if worst == EADDRINUSE_RC and time.time() - t0 < 30 and attempt < retries:
            continue                                  # the port was taken between free_port() and rank 0's bind
        return worst
    return worst


EADDRINUSE_RC = 98          # exit code a rank uses when its rendezvous bind fails with EADDRINUSE (dist.init_process_group)


def rank_env():
    """(rank, local_rank, world) of this process; (0, 0, 1) when not launched."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
