"""bench.py's multi-rank plumbing without a GPU: `--gpus 2` with no launcher in the
environment must start two ranks by itself, rendezvous on 127.0.0.1, run the same
double-buffered StepGather the GPU path uses (gloo here), and print exactly one JSON
line that says n_gpus = 2.  A rank that fails must fail the command."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-gloo", "--channels", "5",
                           "--samples", "8192", "--steps", "4", "--warmup", "1"] + extra,
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_gpus_2_launches_two_ranks_and_gathers():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1 and j["scaling"] == "weak"
    ag = j["config"]["all_gather"]
    assert ag["gathered_equals_pattern"] is True
    assert ag["bytes_per_rank"] == 2 * 5 * 8 * 2 * 2


def test_single_rank_dry_run_needs_no_launcher():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode().strip())["n_gpus"] == 1


def test_launcher_environment_is_respected():
    """Under torch.distributed.run the ranks exist already: WORLD_SIZE in the environment
    means bench.py must not start ranks of its own (a lone rank 0 of a world of 1 here)."""
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": "29533"})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert json.loads(r.stdout.decode().strip())["n_gpus"] == 1
