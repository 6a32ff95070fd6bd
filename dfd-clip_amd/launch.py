"""One process per GPU: the single-node launcher of this path.

The reference starts its ranks with `accelerate launch` (reference
`scripts/cross-manipulation-train.sh:6`, `main.py:283-287`; DDP wrap `src/trainer.py:73-75`).
Here a parent process that has made NO GPU call starts N fresh children, one per GPU, with
the `torch.distributed` rendezvous variables set (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT), waits for them and returns their worst exit code.  Children are
new interpreters (never a fork or an exec of a process that has touched the device), the
rendezvous is always 127.0.0.1, and a rank that dies takes the others down with it (by PID)
instead of leaving them blocked in a collective.

Importing this module imports neither torch nor the HIP library.
"""
import os
import socket
import subprocess
import sys
import time


def under_launcher(environ=None):
    """True when this process already is a rank (started by this launcher or by torch.distributed.run)."""
    environ = os.environ if environ is None else environ
    return "WORLD_SIZE" in environ and "RANK" in environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's only working path on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))
    return env


def spawn_ranks(world, argv, port=None, poll_s=0.2, grace_s=10.0, executable=None):
    """Start `world` children running `argv` (a command line WITHOUT the interpreter), rank r with LOCAL_RANK r.
    Rank 0 inherits stdout (it prints the result line); every rank inherits stderr.  Returns 0 when all ranks
    exited 0, else the first non-zero exit code seen (after terminating the remaining ranks)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port() if port is None else port
    exe = executable or sys.executable
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([exe] + list(argv), env=rank_env(r, world, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = set(range(world))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code  # a signal: the shell's convention
        if rc != 0:
            break
        time.sleep(poll_s)
    if live:  # a rank failed: the others would wait in a collective forever
        for r in live:
            procs[r].terminate()
        t_end = time.time() + grace_s
        for r in live:
            try:
                procs[r].wait(max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc
