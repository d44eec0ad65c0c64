"""One-process-per-GPU launcher used by bench.py (and testable on CPU with gloo).

The reference starts its ranks with ``mpirun`` and binds ``rank % device_count``
(/root/reference/scenes/spheres.cu:83-98).  Here the job either runs under
``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) or starts its own ranks: the parent process creates N children with that same
environment BEFORE it makes any GPU call, waits for them, and returns rank 0's exit code.
Never falls back to fewer ranks than asked for.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def under_launcher():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def visible_gpus():
    """Device count without initialising the GPU runtime in this process."""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(n, script, argv, need_gpus=True, timeout=None, poll_s=0.05):
    """Start ``n`` ranks of ``script argv`` on this node and wait for them.

    Returns the worst exit code (0 only when every rank exited 0).  All children are watched together: as soon as one
    of them exits non-zero (or is killed) the others -- which would sit in a collective waiting for it until the
    driver's limit -- are terminated and its code is returned; ``timeout`` (seconds, also ``--rank-timeout`` of
    bench.py / RTMI_RANK_TIMEOUT) bounds the whole job and yields 124.  Raises SystemExit with a clear message when
    ``need_gpus`` and fewer than ``n`` GPUs are visible."""
    import time
    if need_gpus and os.environ.get("RTMI_BENCH_TEST_ONE_GPU") != "1":  # (bench.py's one-GPU test hook shares cuda:0)
        have = visible_gpus()
        if have < n:
            raise SystemExit("--gpus %d requested but only %d GPU(s) are visible: refusing to run fewer ranks"
                             % (n, have))
    if timeout is None and os.environ.get("RTMI_RANK_TIMEOUT"):
        timeout = float(os.environ["RTMI_RANK_TIMEOUT"])
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    worst = 0
    t0 = time.monotonic()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0:
                    worst = worst or rc
            if worst:
                break  # a rank failed: the rest cannot finish their collectives
            if timeout is not None and time.monotonic() - t0 > timeout:
                worst = 124
                break
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:  # exactly the children started here, by PID
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return worst
