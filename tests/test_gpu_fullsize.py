"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot run
there in seconds).  Shape = configs[1]: 20 000 genes x 50 000 cells, K = 50 + [5]; data generated on
the device.  Properties:
  * exact checksums: X holds small integer counts, so with H == 1 the XH^T sweep must return the exact
    row sums of X in every column, and with W == 1 the W^TX sweep (observed through the H update)
    must reproduce the exact column sums -- every element of both 4 GB streams is accounted for, bit-exact;
  * the trace-form loss row equals the direct-form float64 evaluation of the same factors;
  * pure Frobenius objective without regularisers is non-increasing under MU;
  * factors stay non-negative and finite; two runs are bitwise identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G, N, KU, KC = 20000, 50000, 50, [5]


@pytest.fixture(scope="module")
def big():
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    dev = torch.device("cuda", 0)
    rowsum = torch.zeros(G, dtype=torch.float64, device=dev)
    colsum = torch.zeros(N, dtype=torch.float64, device=dev)

    def make(loss_type="kl-divergence", lam=1e3):
        eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=KU, cov_components=KC, cov_levels=[2], lam=[lam],
                                  loss_type=loss_type)
        rowsum.zero_()
        for off, chunk in synth_counts_device_chunks(N, G, rank=KU, seed=0, device=dev, chunk_cells=8192):
            torch.cuda.synchronize()
            eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
            eng.synchronize()
            rowsum.add_(chunk.sum(dim=0, dtype=torch.float64))
            colsum[off:off + chunk.shape[0]] = chunk.sum(dim=1, dtype=torch.float64)
            del chunk
        eng.finalize_X()
        rng = np.random.default_rng(1)
        lab = rng.integers(0, 2, size=N)
        Y = np.zeros((2, N), dtype=np.float32)
        Y[lab, np.arange(N)] = 1.0
        eng.upload_Y(0, Y)
        return eng
    yield make, rowsum, colsum
    torch.cuda.empty_cache()


def test_exact_checksums_of_both_sweeps(big):
    from alpine_amd import _native
    make, rowsum, colsum = big
    eng = make()
    K = KU + sum(KC)
    info = eng.info()
    KP = info.k_padded
    ones_W = np.ones((G, K), dtype=np.float32)
    ones_H = np.ones((K, N), dtype=np.float32)
    B0 = [np.full((2, KC[0]), 0.5, dtype=np.float32)]
    eng.set_factors(ones_W, ones_H, B0)
    eng.iter_begin()
    XHt = eng.read_buffer(_native.BUF_REDUCE_BLOCK, 0, info.genes_padded * KP).reshape(info.genes_padded, KP)
    rs = rowsum.cpu().numpy()
    assert rs.max() < 2 ** 24                      # exactly representable -> the sweep must be bit-exact
    assert np.array_equal(XHt[:G, :K].astype(np.float64), np.repeat(rs[:, None], K, axis=1))
    assert not XHt[G:].any() and not XHt[:, K:].any()
    # W^TX observed through the H update with lam = 0-like neutral guidance is awkward; read it via the
    # unguided columns: H_new = H * (2 WtX) / max((2 WtW) H, eps) with W == 1 after a W "update" we undo:
    eng.close()
    eng = make(loss_type="frobenius", lam=0.0)
    eng.set_factors(ones_W, ones_H, B0)
    # one full iteration: W <- W * 2 rowsum / (2 N K)  (all columns equal), then H <- H * (2 W^T X) / ((2 W^T W) H)
    eng.run(1, with_loss=False)
    W1, H1, _ = eng.get_factors()
    w = (2.0 * rs) / (2.0 * N * K)                                 # new W column (float64 model of the update)
    assert np.allclose(W1[:, 0], w, rtol=3e-6) and np.allclose(W1[:, K - 1], w, rtol=3e-6)
    # every cell: H_new = (2 sum_g w_g X_gn) / (2 K sum_g w_g^2)  -- a weighted checksum of column n of X
    wtx = H1[K - 1].astype(np.float64) * (K * np.sum(W1[:, 0].astype(np.float64) ** 2))
    # compare against float64 recomputation from exact per-chunk data on the device
    from alpine_amd.datasets import synth_counts_device_chunks
    dev = torch.device("cuda", 0)
    wt = torch.tensor(W1[:, 0], dtype=torch.float64, device=dev)
    want = torch.empty(N, dtype=torch.float64, device=dev)
    for off, chunk in synth_counts_device_chunks(N, G, rank=KU, seed=0, device=dev, chunk_cells=8192):
        want[off:off + chunk.shape[0]] = chunk.double() @ wt
    want = want.cpu().numpy()
    assert np.allclose(wtx, want, rtol=5e-6)
    eng.close()


def test_trace_loss_equals_direct_loss_and_reproducible(big):
    make, _, _ = big
    from alpine_amd.model import draw_initial_factors
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, KC + [KU], [2])
    outs = []
    for rep in range(2):
        eng = make()
        eng.set_factors(W0, H0, B0)
        eng.run(6, with_loss=True)
        losses = eng.losses()
        direct = eng.eval_recon_direct()
        W, H, Bs = eng.get_factors()
        outs.append((W, H, losses))
        assert np.isfinite(losses).all() and np.isfinite(W).all() and np.isfinite(H).all()
        assert (W >= 0).all() and (H >= 0).all()
        assert abs(losses[-1, 1] - direct) <= 2e-5 * direct, (losses[-1, 1], direct)
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


def test_frobenius_objective_non_increasing(big):
    make, _, _ = big
    from alpine_amd.model import draw_initial_factors
    W0, H0, B0 = draw_initial_factors(7, 1e-6, G, N, KC + [KU], [2])
    eng = make(loss_type="frobenius", lam=10.0)
    eng.set_factors(W0, H0, B0)
    eng.run(8, with_loss=True)
    total = eng.losses()[:, 0]
    assert (np.diff(total) <= 1e-6 * total[:-1]).all(), total
    eng.close()


def test_metric_shape_cfg3_checksum_and_loss():
    """BASELINE's metric shape itself (20 000 x 200 000, K = 50 + [5, 5]): exact XH^T checksum on integer counts,
    trace-form loss == direct float64 loss, for both the float32 and the bf16 storage path (counts are bf16-exact)."""
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    from _golden import rel_fro
    dev = torch.device("cuda", 0)
    Gc, Nc, ku, kc = 20000, 200000, 50, [5, 5]
    K = ku + sum(kc)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, Gc, Nc, kc + [ku], [2, 2])
    rng = np.random.default_rng(1)
    Ys = []
    for i in range(2):
        lab = rng.integers(0, 2, size=Nc)
        Y = np.zeros((2, Nc), dtype=np.float32)
        Y[lab, np.arange(Nc)] = 1.0
        Ys.append(Y)
    results = {}
    for dt in ("f32", "x3", "bf16"):
        eng = _native.NativeShard(n_genes=Gc, n_cells=Nc, n_components=ku, cov_components=kc, cov_levels=[2, 2], lam=[1e3, 1e3],
                                  alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5, x_dtype=dt)
        rowsum = torch.zeros(Gc, dtype=torch.float64, device=dev)
        xmax = 0.0
        for off, chunk in synth_counts_device_chunks(Nc, Gc, rank=ku, seed=0, device=dev, chunk_cells=8192):
            torch.cuda.synchronize()
            eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
            eng.synchronize()
            rowsum.add_(chunk.sum(dim=0, dtype=torch.float64))
            xmax = max(xmax, float(chunk.max()))
            del chunk
        assert xmax <= 256                      # small integer counts: exactly representable in bf16 as well
        eng.finalize_X()
        for i in range(2):
            eng.upload_Y(i, Ys[i])
        info = eng.info()
        KP = info.k_padded
        eng.set_factors(np.ones((Gc, K), np.float32), np.ones((K, Nc), np.float32), [np.full((2, 5), 0.5, np.float32)] * 2)
        eng.iter_begin()
        XHt = eng.read_buffer(_native.BUF_REDUCE_BLOCK, 0, info.genes_padded * KP).reshape(info.genes_padded, KP)
        rs = rowsum.cpu().numpy()
        assert rs.max() < 2 ** 24
        assert np.array_equal(XHt[:Gc, :K].astype(np.float64), np.repeat(rs[:, None], K, axis=1)), dt
        eng.set_factors(W0, H0, B0)
        eng.run(4, with_loss=True)
        losses = eng.losses()
        W, H, _ = eng.get_factors()
        assert np.isfinite(losses).all() and (W >= 0).all() and (H >= 0).all()
        if dt in ("f32", "x3"):
            direct = eng.eval_recon_direct()
            assert abs(losses[-1, 1] - direct) <= 2e-5 * direct
        results[dt] = (losses, W, H)
        eng.close()
        torch.cuda.empty_cache()
    lx = results["x3"][0]
    assert np.max(np.abs(lx - results["f32"][0]) / np.abs(results["f32"][0])) < 5e-5     # x3: float32-grade loss rows
    assert rel_fro(results["x3"][2], results["f32"][2]) < 1e-4 and rel_fro(results["x3"][1], results["f32"][1]) < 1e-4
    lf, lb = results["f32"][0], results["bf16"][0]
    assert np.max(np.abs(lb[:, 1] - lf[:, 1]) / lf[:, 1]) < 1e-3            # bf16 operands: loss rows within 1e-3
    assert rel_fro(results["bf16"][2], results["f32"][2]) < 2e-2
