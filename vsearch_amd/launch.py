"""One process per GPU, started from a plain shell.

``python bench.py --gpus N`` (and the search CLIs) must work without an outer ``torch.distributed.run``:
the parent process -- which has NOT touched the GPU (no HIP call, no ``torch.cuda.is_available()``) -- starts
N fresh children, one per LOCAL_RANK, with the rendezvous variables ``torch.distributed`` reads from the
environment, waits for them and returns the worst exit code.  Children inherit stdout/stderr, so rank 0's
JSON line is the parent's output.  Nothing is re-exec'ed: a process that initialised the GPU is never replaced.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def under_launcher() -> bool:
    """True when this process already is a rank (torch.distributed.run or spawn_ranks set the variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
    })
    # dmabuf IPC between the ranks' GPUs (RCCL's peer buffers): the build / GPU images of this project export it because their host
    # driver has no legacy IPC (`hipIpcGetMemHandle: invalid argument` otherwise).  A DEFAULT only: a value the user has set stands.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def spawn_ranks(argv: Sequence[str], world: int, timeout_s: Optional[float] = None, extra_env: Optional[Dict[str, str]] = None) -> int:
    """Run ``argv`` (a full command line, e.g. [sys.executable, 'bench.py', '--gpus', '8']) as `world` ranks.

    Returns 0 when every rank exits 0; otherwise the first non-zero exit code (the other ranks are terminated so
    that a failed rank does not leave its peers waiting in a collective)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    import signal
    port = free_port()
    procs: List[subprocess.Popen] = []
    rc = 0
    # SIGTERM / SIGINT to the parent reach the ranks (exact PIDs we started): a rank left behind holds its GPU and may sit in a collective
    def forward(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    old = {}
    try:
        for sig in (signal.SIGTERM, signal.SIGINT):
            try:
                old[sig] = signal.signal(sig, forward)
            except ValueError:                          # not the main thread: no handlers, the finally below still cleans up
                pass
        for r in range(world):
            env = rank_env(r, world, port)
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(list(argv), env=env))
        deadline = None if timeout_s is None else time.monotonic() + timeout_s
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    for o in live:                      # exact PIDs we started, nothing else
                        procs[o].terminate()
            if deadline is not None and time.monotonic() > deadline and live:
                for o in live:
                    procs[o].kill()
                rc = rc or 124
                deadline = None
            if live:
                time.sleep(0.05)
        return rc
    finally:
        # an exception or KeyboardInterrupt in the loop above: terminate, then kill, what is still running
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t_end = time.monotonic() + 5.0
        for p in alive:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
        for sig, h in old.items():
            signal.signal(sig, h)


def relaunch_as_ranks(world: int, script: str, args: Sequence[str]) -> None:
    """Call first thing in a CLI's main(): when N > 1 ranks are asked for and we are not a rank yet, become the
    parent of N ranks and exit with their code."""
    if world <= 1 or under_launcher():
        return
    sys.stdout.flush()
    raise SystemExit(spawn_ranks([sys.executable, script, *args], world))
