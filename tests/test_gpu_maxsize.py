"""Maximum size: BASELINE configs[3] (20 000 genes x 1 000 000 cells, K = 100 + [5]) whole on ONE MI355X --
2e10 elements of X (> 2^32: 64-bit indexing everywhere), 152 GiB resident in float32, 77 GiB as one exact bf16
plane.  The checks (exact XH^T checksum, W^TX checksum, trace == direct loss, split == float32 loss rows) live in
tools/huge_check.py so that the same file is the command-line tool; ~30 s on the device."""
import importlib.util
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_cfg4_whole_on_one_gpu():
    free, total = torch.cuda.mem_get_info(0)
    if free < 170 * 2 ** 30:
        pytest.skip(f"needs 170 GiB of free HBM, {free / 2**30:.0f} GiB free")
    spec = importlib.util.spec_from_file_location("huge_check", Path(__file__).resolve().parent.parent / "tools" / "huge_check.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(["--modes", "split,x3,f32", "--iters", "4"])
    assert rep["elements"] > 2 ** 32
    assert rep["modes"]["split"]["xht_exact"] and rep["modes"]["f32"]["xht_exact"]
    assert rep["split_vs_f32_loss_rows_max_rel"] < 5e-5 and rep["x3_vs_f32_loss_rows_max_rel"] < 5e-5
    assert rep["modes"]["x3"]["xht_exact"]
    assert rep["modes"]["split"]["device_GiB"] < 0.55 * rep["modes"]["f32"]["device_GiB"]     # unused second plane was freed
