"""GPU test: bench.py prints one JSON line with the fields its callers read, on a small shard."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chunks-per-gpu", "128", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-mib", "16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GB/s" and d["dtype"] == "u8"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    assert d["inflate"]["verified_equal_to_input"] is True
    cb = d["cpu_baseline"]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "cpu_baseline")):
        assert cb["kind"] == "reference" and cb["cores"] >= 1 and cb["value"] > 0


def test_gpus_2_runs_two_ranks_with_the_window_hand_over():
    """`bench.py --gpus 2` on a one-GPU box: both ranks on GPU 0, collectives over gloo (RCCL refuses two ranks on one
    device).  The line must say n_gpus 2, count both shards, and the sharded inflate (windows handed from rank 0 to
    rank 1) must reproduce the input on every rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ZAMD_BENCH_ONE_GPU"] = "1"
    env["ZAMD_BENCH_WS_GIB"] = "8"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--chunks-per-gpu", "64",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["out_bytes_all"] > d["out_bytes_rank0"] > 0
    assert d["inflate"]["verified_equal_to_input"] is True and d["inflate"]["window_handover"]
