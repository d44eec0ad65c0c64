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
