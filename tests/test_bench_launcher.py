"""bench.py's own launcher (rtmi/launch.py): ``--gpus N`` must start N ranks itself, run the
N-rank exchange, and never fall back to fewer ranks.  CPU only: the exchange self-test runs the
gather + untile of one step on gloo tensors (no rendering: there is no CPU render path)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_spawns_two_ranks_and_runs_the_exchange():
    r = _run(["--gpus", "2", "--selftest-exchange"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints, nobody else
    out = json.loads(lines[0])
    assert out["selftest"] == "ok" and out["n_gpus"] == 2 and out["n_ranks_seen"] == 2


def test_gpus_3_exchange():
    r = _run(["--gpus", "3", "--selftest-exchange"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_ranks_seen"] == 3


def test_more_gpus_than_visible_fails_loudly():
    """On a box with fewer GPUs than asked for (here: none or one) the bench refuses to run."""
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "refusing to run fewer ranks" in (r.stdout + r.stderr)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    """Under an external launcher WORLD_SIZE must equal --gpus (no silent one-rank line)."""
    r = _run(["--gpus", "4", "--selftest-exchange"],
             env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in (r.stdout + r.stderr)


def test_a_rank_that_dies_takes_the_job_down_at_once():
    """One rank exits before the rendezvous: the others would wait in init_process_group for the store's timeout
    (minutes); the launcher sees the death, terminates them and returns the dead rank's code."""
    import time
    t0 = time.time()
    r = _run(["--gpus", "3", "--selftest-exchange", "--selftest-die-rank", "1"], timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout + r.stderr)
    assert time.time() - t0 < 60
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rank_timeout_bounds_the_job():
    """--rank-timeout: ranks that hang (here: rank 0 alone waits for a peer that was told a different port) are
    terminated and the launcher returns 124."""
    sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
    from rtmi import launch
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write("import time\ntime.sleep(600)\n")
    try:
        rc = launch.spawn_ranks(2, f.name, [], need_gpus=False, timeout=1.0)
    finally:
        os.unlink(f.name)
    assert rc == 124
