"""CPU test of bench.py's own rank launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment
must start N ranks (fresh processes), rendezvous them, run the length all-gather and the window hand-over of the
sharded inflate, and print one line that says n_gpus == N.  --dry keeps the GPU out of it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


def test_gpus_flag_starts_that_many_ranks():
    r = _run("--gpus", "2", "--dry", "--backend", "gloo", "--chunks-per-gpu", "37")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_group"] == 2 and d["plumbing_ok"] is True and d["dry"] is True


def test_three_ranks_hand_the_window_down_the_line():
    r = _run("--gpus", "3", "--dry", "--backend", "gloo", "--chunks-per-gpu", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["ranks_in_group"] == 3 and d["plumbing_ok"] is True


def test_world_size_that_contradicts_gpus_is_an_error():
    r = _run("--gpus", "2", "--dry", env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_single_rank_dry_line():
    r = _run("--dry")
    assert r.returncode == 0
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["plumbing_ok"] is True
