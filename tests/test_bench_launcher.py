"""bench.py's self-launcher (`python bench.py --gpus N` with no torchrun around it) on a box without GPUs: the parent
must not touch a GPU, must refuse when the GPUs are not there, and must turn a failing worker into a non-zero exit with
no JSON line on stdout (the driver would otherwise record a silent success)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="covers the no-GPU behaviour of the launcher")
def test_launcher_refuses_when_the_gpus_are_not_there():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny"],
                       capture_output=True, text=True, timeout=300, cwd=REPO, env=_env())
    assert r.returncode == 2 and "only 0 GPU(s) visible" in r.stderr and not r.stdout.strip()


@pytest.mark.skipif(not _no_gpu(), reason="covers the no-GPU behaviour of the launcher")
def test_launcher_propagates_worker_failure():
    # rehearsal mode lets the parent start its workers on any box; without a GPU every worker exits non-zero
    # ("bench.py needs an MI355X"), which the parent must report as a failure of the whole run
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny", "--launch-timeout", "240"],
                       capture_output=True, text=True, timeout=300, cwd=REPO, env=_env(ALPINE_BENCH_REHEARSAL_ONE_GPU="1"))
    assert r.returncode != 0 and not r.stdout.strip()
    assert "exited with" in r.stderr and "needs an MI355X" in r.stderr


def test_single_gpu_invocation_is_not_a_launcher():
    """--gpus 1 (the default) never spawns workers: on a GPU-less box it fails in-process with the no-fallback message."""
    if not _no_gpu():
        pytest.skip("needs a box without GPUs")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "tiny"], capture_output=True, text=True,
                       timeout=300, cwd=REPO, env=_env())
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
