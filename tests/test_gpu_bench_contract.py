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
    assert set(d["other_modes"]) == {"f32", "split", "bf16"} and all("error" not in v for v in d["other_modes"].values())
