"""Self-launching one process per GPU on one node (``python bench.py --gpus N`` / ``python main.py ... --gpus N``).

The reference fans work out inside one process (``ParallelBlender``'s thread pool, blending_module.py:1665-1705); the
MI355X path is one process per GPU over RCCL, so an entry point asked for N GPUs without a launcher around it starts
its own ranks: CHILD processes (env:// rendezvous on 127.0.0.1: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT) created BEFORE the parent has touched torch or HIP.  Never exec: a process that has initialised the GPU
must not be replaced -- and the parent never initialises it; the children are always fresh processes."""
from __future__ import annotations

import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from typing import List, Sequence

STAGE_ENV = "SR_STAGE_FILE"


def stage(name: str) -> None:
    """Progress marker of one rank: the launcher prints each rank's last completed stage when its deadline passes, so a
    rank stuck in a collective is named instead of the run dying silently at the caller's limit."""
    path = os.environ.get(STAGE_ENV) or os.environ.get("SR_BENCH_STAGE_FILE")
    if path:
        try:
            with open(path, "a") as f:
                f.write(f"{time.time():.3f} {name}\n")
        except OSError:
            pass


def dist_env():
    """(rank, world, local_rank) from the launcher's environment; (0, 1, 0) without one."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0"))


def launch_ranks(n: int, script: str, argv: Sequence[str], deadline_s: float, who: str = "launcher",
                 extra_env: dict | None = None) -> int:
    """Starts ``python script argv`` n times (ranks 0 .. n-1).  The children write straight to our stdout / stderr.
    Returns the first non-zero child status (the other ranks are then stopped), else 0; after ``deadline_s`` seconds
    the ranks still running are terminated, every rank's last completed stage is printed to stderr and the status is 124."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    stage_dir = tempfile.mkdtemp(prefix="sr_stage_")
    procs: List[subprocess.Popen] = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    STAGE_ENV: os.path.join(stage_dir, f"rank{r}"),
                    "SR_BENCH_STAGE_FILE": os.path.join(stage_dir, f"rank{r}")})
        env.update(extra_env or {})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))

    def last_stages():
        out = []
        for r in range(n):
            try:
                lines = open(os.path.join(stage_dir, f"rank{r}")).read().split("\n")
                lines = [ln for ln in lines if ln]
                out.append(lines[-1].split(" ", 1)[1] if lines else "(nothing: not past the interpreter start)")
            except OSError:
                out.append("(no stage file)")
        return out

    rc = 0
    t_end = time.monotonic() + deadline_s
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print(f"{who}: rank {procs.index(p)} exited with status {code}; stopping the others "
                      f"(last stages: {last_stages()})", file=sys.stderr)
                for q in alive:                              # a rank failed: the others would wait on it forever
                    q.terminate()
        if alive and time.monotonic() > t_end:
            stages = last_stages()
            print(f"{who}: deadline of {deadline_s:.0f} s passed with {len(alive)} of {n} ranks still running", file=sys.stderr)
            for r, st in enumerate(stages):
                state = "running" if procs[r] in alive else f"exited {procs[r].returncode}"
                print(f"{who}:   rank {r} [{state}] last completed stage: {st}", file=sys.stderr)
            for q in alive:
                q.terminate()
            for q in alive:
                try:
                    q.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    q.kill()
            rc = rc or 124
            break
    shutil.rmtree(stage_dir, ignore_errors=True)
    return rc
