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


@pytest.mark.parametrize("K,scale", [(100, 0.37)])         # (K > 128 takes the same guard; its 34 GB factors + 51 GB of panel planes at this length are left out)
def test_very_long_cell_axis_keeps_64_bit_lane_addresses(K, scale):
    """More than 2^25 cells in one shard (a few genes x tens of millions of cells: 17 GB per copy of X): the round-4 kernels that address X as a
    scalar row base + a 32-bit lane offset (stream_gemm_x3v_kernel, stream_gemm_x3w2_kernel) would wrap around there -- alpine_create /
    alpine_finalize_X must keep such a shard on the kernels with 64-bit lane addresses (alpine_info.sweep_waves_per_simd == 1, one launch per
    component block for K > 128), and an iteration must run (XH^T checked against a float64 sample: row sums of H weighted by one gene's row)."""
    import numpy as np
    from alpine_amd import _native as nat
    free, total = torch.cuda.mem_get_info(0)
    if free < 60 * 2 ** 30:
        pytest.skip(f"needs 60 GiB of free HBM, {free / 2**30:.0f} GiB free")
    G, N = 96, (1 << 25) + 1000
    rng = np.random.default_rng(3)
    chunk = (rng.poisson(1.0, size=(1 << 20, G)) * scale).astype(np.float32)          # the same million cells over and over
    eng = nat.NativeShard(n_genes=G, n_cells=N, n_components=K, cov_components=[], cov_levels=[], lam=[], x_dtype="x3")
    try:
        for c0 in range(0, N, chunk.shape[0]):
            n = min(chunk.shape[0], N - c0)
            eng.upload_X_host(chunk[:n], cell0=c0)
        eng.finalize_X()
        info = eng.info()
        assert info.sweep_waves_per_simd == 1, info.sweep_waves_per_simd
        W0 = rng.uniform(0.1, 1.0, size=(G, K)).astype(np.float32)
        h_row = rng.uniform(0.1, 1.0, size=(K, 1 << 20)).astype(np.float32)
        H0 = np.tile(h_row, (1, N // (1 << 20) + 1))[:, :N]
        eng.set_factors(W0, H0, [])
        eng.iter_begin()                                   # XH^T sweep + reduce block
        KP = info.k_padded
        blk = eng.read_buffer(nat.BUF_REDUCE_BLOCK, 0, info.genes_padded * KP)
        nh = max(1, KP // 128) if K > 128 else 1
        XHt = (np.concatenate(list(blk.reshape(nh, info.genes_padded, 128)), axis=1) if K > 128 else blk.reshape(info.genes_padded, KP))[:G, :K]
        reps, rem = divmod(N, 1 << 20)
        full = chunk.astype(np.float64).T @ h_row.astype(np.float64).T                        # one million-cell period
        want = reps * full + chunk[:rem].astype(np.float64).T @ h_row[:, :rem].astype(np.float64).T
        assert np.abs(XHt - want).max() <= 2e-5 * np.abs(want).max()
        eng.iter_end(True)
        eng.run(1, with_loss=True)
        assert np.isfinite(eng.losses()).all()
    finally:
        eng.close()
