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
