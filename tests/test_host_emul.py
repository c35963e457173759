"""Runs the engine's per-lane building blocks (zlib_amd/csrc/zamd_core.h) on the
host, chained as the kernels chain them, against the oracle.  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipeline_decomposition_matches_oracle():
    subprocess.check_call(["make", "-s", "tests/host_emul"], cwd=ROOT)
    r = subprocess.run([os.path.join(ROOT, "tests", "host_emul"), "40", "31337"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
