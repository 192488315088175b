"""bench.py's output contract (the driver parses this line): one JSON object on stdout with the agreed keys, a roofline
object and a cpu_baseline object, here on the tiny workload so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "tiny", "--steps", "4", "--warmup", "1",
                        "--cpu-sample-cells", "500"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    # every storage mode ran on the same workload next to the headline
    # ... plus the headline's and the float32 MFMA's sweeps on full-significand data (throughput is value-dependent)
    assert set(d["other_modes"]) == {"f32", "split", "bf16", "x3_fullsig", "f32_fullsig"}
    assert all("error" not in v for v in d["other_modes"].values()), d["other_modes"]
    assert d["other_modes"]["x3_fullsig"]["x_scale"] != 1.0 and d["config"]["x_scale"] == 1.0
    assert cb["sample_cells"] == 500 and cb["peak_rss_GiB"] > 0


def test_bench_gpus_2_launches_its_own_workers():
    """`python bench.py --gpus 2` with no launcher around it (the driver's invocation): the parent starts two fresh worker
    processes, relays rank 0's single JSON line and exits 0.  On a one-GPU box the rehearsal switch puts both ranks on
    cuda:0 with gloo carrying the reduce block (RCCL refuses two ranks on one device); without it the launcher must refuse."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "4", "--warmup", "1"]
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
        env["ALPINE_BENCH_REHEARSAL_ONE_GPU"] = "1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["parallelism"] == "cells/2" and d["config"]["cells_per_gpu"] in (2496, 2504)
    ar = d["allreduce"]
    assert ar is not None and ar["avg_ms_on_rank0"] > 0 and ar["bytes"] > 0 and ar["carrier"]
    assert len(ar["per_rank"]) == 2 and all(r["ms_per_step"] > 0 for r in ar["per_rank"])
    if ar["carrier"].startswith("native"):
        assert ar["rccl_ranks"] == 2
    else:
        assert ar["rccl_ranks"] is None and all(r["carrier"] == "torch" for r in ar["per_rank"])


def test_bench_under_the_drivers_launcher_command():
    """The driver's N > 1 invocation verbatim -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with two ranks: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the
    launcher, rank 0 prints the one JSON line, the other rank prints nothing on stdout.  On a one-GPU box the rehearsal switch puts both
    ranks on cuda:0 with gloo carrying the reduce block; with two GPUs it runs as the driver runs it (RCCL, one device per rank)."""
    import socket
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.device_count() < 2:
        env["ALPINE_BENCH_REHEARSAL_ONE_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "4", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["cells_per_gpu_by_rank"] == [d["config"]["cells_per_gpu"], 5000 - d["config"]["cells_per_gpu"]]
    ar = d["allreduce"]
    for key in ("first_call_ms", "first_full_block_ms", "standalone_ms", "avg_ms_on_rank0", "bytes", "carrier"):
        assert key in ar and ar[key], key
    assert d["rccl_version"]
