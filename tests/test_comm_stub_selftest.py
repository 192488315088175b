"""The RCCL stand-in used by tests/test_gpu_comm_stub.py must itself be right: forked processes all-reduce through it on
the CPU (no GPU, no RCCL involved)."""
import subprocess

import pytest

from _stub import build_selftest


@pytest.mark.parametrize("ranks", [1, 2, 4])
def test_stub_all_reduce_sums_in_every_rank(ranks):
    exe = build_selftest()
    r = subprocess.run([exe, str(ranks), "60"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
