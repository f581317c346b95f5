"""Launching the ranks of a multi-process test: `world` children of this process running tests/dist_worker.py, a free rendezvous port, ONE
overall deadline, and — whatever happens (a rank that fails, a time-out, an exception in between) — every child that is still alive is
killed and reaped before the helper returns (exact PIDs; a surviving rank would keep spinning in a shared-memory exchange, hold the GPU
and a core, and outlive the temporary directory)."""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world: int, worker_args, extra_env=None, deadline_s: float = 600.0):
    """starts rank 0 .. world - 1 of tests/dist_worker.py with `worker_args`; returns when all have exited with status 0, raises AssertionError
    with the first failing rank otherwise (the others are killed)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world))
    env.setdefault("CENO_DIST_SHM_TIMEOUT_S", "120")   # a peer that died is noticed by the others within this (host/dist.cpp)
    env.update(extra_env or {})
    procs = []
    try:
        for rank in range(world):
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")] + [str(a) for a in worker_args],
                                          env=dict(env, RANK=str(rank))))
        end = time.time() + deadline_s
        live = list(range(world))
        while live:
            for r in list(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.remove(r)
                assert c == 0, f"rank {r} of {world} exited with status {c}"
            assert time.time() < end, f"ranks {live} of {world} still running after {deadline_s:.0f} s"
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
