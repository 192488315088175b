"""Randomised shapes through the C ABI against the CPU oracle (fused form = the kernels' numerical spec): ragged G / N
(not multiples of any tile), K from 1 to ~100, 0-2 covariates with 1-4 levels, both loss types, regularisers on/off,
several stream-K span sizes; float32 MFMA, the x3 plane-product sweeps and (on integer data) the exact-split path.  Two MU steps + loss rows each."""
import os

import numpy as np
import pytest
import torch

from _golden import rel_fro
from oracle import alpine_oracle as orc

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(3, 300))
    N = int(rng.integers(3, 420))
    n_cov = int(rng.integers(0, 3))
    ks = [int(rng.integers(1, 7)) for _ in range(n_cov)]
    levels = [int(rng.integers(1, 5)) for _ in range(n_cov)]
    Ku = int(rng.choice([1, 2, 5, 17, 30, 33, 60, 64, 65, 90]))
    Ku = min(Ku, 128 - sum(ks))
    loss = ["kl-divergence", "frobenius"][int(rng.integers(0, 2))]
    reg = bool(rng.integers(0, 2))
    integer = bool(rng.integers(0, 2))
    X = rng.gamma(0.5, 2.0, size=(N, G)).astype(np.float32)
    if integer:
        X = np.floor(X * (3 if rng.integers(0, 2) else 400)).astype(np.float32)
    Ys = []
    for C in levels:
        lab = rng.integers(-1 if C > 1 else 0, C, size=N)          # -1 = missing label -> all-zero row
        Y = np.zeros((N, C), dtype=np.float32)
        ok = lab >= 0
        Y[np.flatnonzero(ok), lab[ok]] = 1.0
        Ys.append(Y)
    p = orc.OracleParams(n_components=Ku, n_covariate_components=ks, lam=[float(rng.choice([1.0, 50.0, 1e3])) for _ in ks],
                         orth_W=0.1 if reg else 0.0, alpha_W=0.7 if reg else 0.0, l1_ratio_W=0.4 if reg else 0.0,
                         loss_type=loss, random_state=int(seed))
    return p, X, Ys, integer, (int(rng.integers(0, 4)), int(rng.integers(0, 4)))


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_shape_two_steps_vs_oracle(seed):
    from alpine_amd import _native as nat
    p, X, Ys, integer, splits = _case(seed)
    s = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    orc.fit_fused(p, s, 2, with_loss=True)
    modes = ["f32", "x3"] + (["split"] if integer and float(X.max()) < 65536 else [])
    for mode in modes:
        eng = nat.NativeShard(n_genes=X.shape[1], n_cells=X.shape[0], n_components=p.n_components,
                              cov_components=p.n_covariate_components, cov_levels=[y.shape[1] for y in Ys], lam=p.lam,
                              orth_W=p.orth_W, alpha_W=p.alpha_W, l1_ratio_W=p.l1_ratio_W, eps=p.eps, loss_type=p.loss_type,
                              split_a=splits[0], split_b=splits[1], x_dtype=mode)
        eng.upload_X_host(X)
        eng.finalize_X()
        for i, y in enumerate(Ys):
            eng.upload_Y(i, np.ascontiguousarray(y.T))
        eng.set_factors(W0, H0, B0)
        eng.run(2, with_loss=True)
        W, H, Bs = eng.get_factors()
        losses = eng.losses()
        eng.close()
        tag = f"seed {seed} mode {mode} G={X.shape[1]} N={X.shape[0]} K={p.total_components} cov={p.n_covariate_components} {p.loss_type}"
        assert np.isfinite(W).all() and np.isfinite(H).all(), tag
        assert rel_fro(W, s.W.numpy()) < 2e-5, tag
        assert rel_fro(H, s.H.numpy()) < 2e-5, tag
        for b, bo in zip(Bs, s.Bs):
            assert rel_fro(b, bo.numpy()) < 5e-5, tag
        want = np.array(s.losses)
        assert losses.shape == want.shape, tag
        np.testing.assert_allclose(losses[:, :2], want[:, :2], rtol=1e-4, err_msg=tag)
        np.testing.assert_allclose(losses[:, 2:], want[:, 2:], rtol=2e-3, atol=1e-6 * X.shape[0], err_msg=tag)


@pytest.mark.parametrize("seed", list(range(100, 110)))
def test_random_shape_als_and_minibatch_vs_oracle(seed):
    """Block-coordinate branch (use_als) and mini-batch steps on random shapes against the oracle's op-for-op restatement
    (als_step_faithful / mu_step_faithful on explicit index batches)."""
    _als_and_minibatch_vs_oracle(seed)


@pytest.mark.parametrize("seed,Ku", [(120, 129), (121, 140), (122, 200), (123, 250), (124, 131), (125, 170),
                                     (126, 257), (127, 300), (128, 385), (129, 512), (130, 640)])          # round 4: 3, 4 and 5 column blocks
def test_wide_als_and_minibatch_vs_oracle(seed, Ku):
    """The same on the blocked path (128 < K <= 1024, ceil(K / 128) column blocks): gathered views of the blocked H, the group loop with
    block-local orthogonality over all blocks, epoch loss rows."""
    _als_and_minibatch_vs_oracle(seed, Ku=Ku)


def _als_and_minibatch_vs_oracle(seed, Ku=None):
    from alpine_amd import _native as nat
    p, X, Ys, _, _ = _case(seed)
    if Ku is not None:
        p.n_components = Ku - sum(p.n_covariate_components) if seed % 3 else Ku      # K = Ku exactly, or Ku unguided + the guided ones
        p.n_components = min(p.n_components, 1024 - sum(p.n_covariate_components))
    if not p.n_covariate_components:            # the reference's ALS / sampler code needs at least one covariate
        p.n_covariate_components, p.lam = [2], [10.0]
        Ys = [np.eye(2, dtype=np.float32)[np.random.default_rng(seed).integers(0, 2, size=X.shape[0])]]
    use_als = seed % 2 == 0
    p.use_als = use_als
    N = X.shape[0]
    rng = np.random.default_rng(seed + 7)
    bs = int(rng.integers(max(2, N // 4), N + 1))
    batches = []
    for _ in range(2):                           # two epochs of explicit index batches, with replacement in the second
        epoch = rng.permutation(N) if not batches else rng.integers(0, N, size=N)
        batches.append([epoch[b0:b0 + bs] for b0 in range(0, N, bs)])
    s = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    step = orc.als_step_faithful if use_als else orc.mu_step_faithful
    with torch.no_grad():
        for epoch in batches:
            for idx in epoch:
                step(p, s, torch.tensor(idx, dtype=torch.long))
            s.losses.append(orc.loss_row(p, s))
    for mode in ("f32", "x3"):
        eng = nat.NativeShard(n_genes=X.shape[1], n_cells=N, n_components=p.n_components, cov_components=p.n_covariate_components,
                              cov_levels=[y.shape[1] for y in Ys], lam=p.lam, orth_W=p.orth_W, alpha_W=p.alpha_W,
                              l1_ratio_W=p.l1_ratio_W, eps=p.eps, loss_type=p.loss_type, use_als=use_als, batch_capacity=bs, x_dtype=mode)
        eng.upload_X_host(X)
        eng.finalize_X()
        for i, y in enumerate(Ys):
            eng.upload_Y(i, np.ascontiguousarray(y.T))
        eng.set_factors(W0, H0, B0)
        for epoch in batches:
            for idx in epoch:
                eng.batch_step(idx)
            eng.epoch_loss()
        W, H, Bs = eng.get_factors()
        losses = eng.losses()
        eng.close()
        tag = f"seed {seed} mode {mode} als={use_als} G={X.shape[1]} N={N} bs={bs} K={p.total_components} cov={p.n_covariate_components} {p.loss_type}"
        assert rel_fro(W, s.W.numpy()) < 5e-5, tag
        assert rel_fro(H, s.H.numpy()) < 5e-5, tag
        for b, bo in zip(Bs, s.Bs):
            assert rel_fro(b, bo.numpy()) < 1e-4, tag
        want = np.array(s.losses)
        np.testing.assert_allclose(losses[:, :2], want[:, :2], rtol=1e-4, err_msg=tag)


@pytest.mark.parametrize("loss_type", ["kl-divergence", "frobenius"])
def test_many_label_levels_vs_oracle(loss_type):
    """More label levels than the H update keeps in LDS (sum of levels = 43 > 32): the fused tail then takes the stand-alone
    statistics arithmetic with Y from global memory, and the guided terms read Y from global memory too.  Five MU steps, so
    the tail's output feeds four phase 1s and the closing loss row."""
    from alpine_amd import _native as nat
    rng = np.random.default_rng(11)
    G, N = 150, 700
    levels, ks = [40, 3], [4, 2]
    X = rng.gamma(0.5, 2.0, size=(N, G)).astype(np.float32)
    Ys = []
    for C in levels:
        lab = rng.integers(0, C, size=N)
        Y = np.zeros((N, C), dtype=np.float32)
        Y[np.arange(N), lab] = 1.0
        Ys.append(Y)
    p = orc.OracleParams(n_components=9, n_covariate_components=ks, lam=[50.0, 1e3], orth_W=0.1, alpha_W=0.5, l1_ratio_W=0.3,
                         loss_type=loss_type, random_state=5)
    s = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    orc.fit_fused(p, s, 5, with_loss=True)
    eng = nat.NativeShard(n_genes=G, n_cells=N, n_components=p.n_components, cov_components=ks, cov_levels=levels, lam=p.lam,
                          orth_W=p.orth_W, alpha_W=p.alpha_W, l1_ratio_W=p.l1_ratio_W, eps=p.eps, loss_type=loss_type, x_dtype="x3")
    eng.upload_X_host(X)
    eng.finalize_X()
    for i, y in enumerate(Ys):
        eng.upload_Y(i, np.ascontiguousarray(y.T))
    eng.set_factors(W0, H0, B0)
    eng.run(5, with_loss=True)
    W, H, Bs = eng.get_factors()
    losses = eng.losses()
    eng.close()
    assert rel_fro(W, s.W.numpy()) < 5e-5 and rel_fro(H, s.H.numpy()) < 5e-5
    for b, bo in zip(Bs, s.Bs):
        assert rel_fro(b, bo.numpy()) < 1e-4
    want = np.array(s.losses)
    np.testing.assert_allclose(losses[:, :2], want[:, :2], rtol=1e-4)
    np.testing.assert_allclose(losses[:, 2:], want[:, 2:], rtol=2e-3, atol=1e-6 * N)


def _fit_vs_oracle(seed, G, N, Ku, ks, levels, loss, iters, env=None, monkeypatch=None, reg=False, counts=False, expect_waves=None):
    """`iters` MU iterations of a random problem through the C ABI against the oracle's fused iteration (shared by the LDS
    budget and wide-guidance cases below).  counts: small integer X (every element one bf16 plane: the one-plane sweep forms);
    expect_waves: what alpine_info.sweep_waves_per_simd must say for the x3 engine."""
    from alpine_amd import _native as nat
    rng = np.random.default_rng(seed)
    X = rng.gamma(0.4, 2.5, size=(N, G)).astype(np.float32)
    if counts:
        X = np.floor(X * 3).astype(np.float32)
    Ys = []
    for C in levels:
        Y = np.zeros((N, C), dtype=np.float32)
        Y[np.arange(N), rng.integers(0, C, size=N)] = 1.0
        Ys.append(Y)
    p = orc.OracleParams(n_components=Ku, n_covariate_components=list(ks), lam=[float(rng.choice([1.0, 30.0, 1e3])) for _ in ks],
                         orth_W=0.1 if reg else 0.0, alpha_W=0.6 if reg else 0.0, l1_ratio_W=0.3 if reg else 0.0,
                         loss_type=loss, random_state=seed)
    s = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    orc.fit_fused(p, s, iters, with_loss=True)
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    out = {}
    for mode in ("x3", "f32"):
        eng = nat.NativeShard(n_genes=G, n_cells=N, n_components=Ku, cov_components=list(ks), cov_levels=list(levels), lam=p.lam,
                              orth_W=p.orth_W, alpha_W=p.alpha_W, l1_ratio_W=p.l1_ratio_W, eps=p.eps, loss_type=loss, x_dtype=mode)
        try:
            eng.upload_X_host(X)
            eng.finalize_X()
            if mode == "x3" and expect_waves is not None:
                assert eng.info().sweep_waves_per_simd == expect_waves, eng.info().sweep_waves_per_simd
            for i, y in enumerate(Ys):
                eng.upload_Y(i, np.ascontiguousarray(y.T))
            eng.set_factors(W0, H0, B0)
            eng.run(iters, with_loss=True)
            W, H, Bs = eng.get_factors()
            losses = eng.losses()
        finally:
            eng.close()
        tol = 2e-5 * max(1, iters)
        assert rel_fro(W, s.W.numpy()) < tol and rel_fro(H, s.H.numpy()) < tol, (mode, rel_fro(W, s.W.numpy()), rel_fro(H, s.H.numpy()))
        for b, bo in zip(Bs, s.Bs):
            assert rel_fro(b, bo.numpy()) < 5 * tol, mode
        want = np.array(s.losses)
        np.testing.assert_allclose(losses[:, :2], want[:, :2], rtol=2e-4, err_msg=mode)
        np.testing.assert_allclose(losses[:, 2:], want[:, 2:], rtol=2e-3, atol=1e-6 * N, err_msg=mode)
        out[mode] = (W, H, losses)
    return out


@pytest.mark.parametrize("ks,levels,Ku", [
    ([10, 3], [30, 4], 90),        # K = 103 (KT = 4), 34 rows of Y (> 32: Y stays in global memory), k = 10 x 16 classes of scratch:
                                   # 134 KB of update + 2 x 12 KB of statistics scratch > 160 KB -> the tail is dropped
    ([60], [2], 50),               # K = 110, one covariate of k = 60: Y copy + tail scratch push the total to 166 KB -> Y copy dropped first
    ([64], [40], 20),              # K = 84 (KT = 3), k = 64 with 40 levels: the largest statistics scratch the limits allow
])
def test_h_update_lds_budget_fallbacks(ks, levels, Ku):
    """the fused H update's optional LDS (Y copy, tail scratch) on top of ~134 KB at K > 96 could exceed the CU's
    160 KB for configurations _check_supported accepts; the launch failed and fit() raised.  launch_h_update now drops the Y
    copy and / or the tail when the total does not fit (same results through phase1_open_kernel)."""
    _fit_vs_oracle(101 + len(levels), G=150, N=700, Ku=Ku, ks=ks, levels=levels, loss="kl-divergence", iters=3)
    _fit_vs_oracle(201 + len(levels), G=150, N=700, Ku=Ku, ks=ks, levels=levels, loss="frobenius", iters=3, reg=True)


def test_forced_small_lds_limit_equals_the_default_path(monkeypatch):
    """The same fall-back forced on a small model (ALPINE_HIP_LDS_LIMIT, read once in alpine_create): without the fused tail
    (H H^T partials and statistics from phase1_open_kernel: another summation order) the factors agree to rounding."""
    a = _fit_vs_oracle(77, G=130, N=600, Ku=20, ks=[3, 2], levels=[3, 2], loss="kl-divergence", iters=4)
    b = _fit_vs_oracle(77, G=130, N=600, Ku=20, ks=[3, 2], levels=[3, 2], loss="kl-divergence", iters=4,
                       env={"ALPINE_HIP_LDS_LIMIT": "26000"}, monkeypatch=monkeypatch)
    for mode in ("x3", "f32"):
        assert rel_fro(a[mode][0], b[mode][0]) < 2e-6 and rel_fro(a[mode][1], b[mode][1]) < 2e-6
        assert not (np.array_equal(a[mode][0], b[mode][0]) and np.array_equal(a[mode][1], b[mode][1]))      # the knob did change the launch structure


@pytest.mark.parametrize("seed", [3465])
def test_fuzz_regressions(seed):
    """Cases that tests/fuzz_gpu.py found, replayed whole (every sweep mode against the oracle, float64 arbiter, mini-batch leg).
    3465 (round 3): G = 2047, N = 5362, K = 102, one forced share per tile -- 11 shares of exactly one tile; the even/odd span bias
    then needs a 12th workgroup AND a third piece per span, which the pieces buffer (sized for the unbiased division plus a margin)
    did not hold: an out-of-bounds write on the device.  The buffers are now sized for every division the ctx may use and
    launch_sweep refuses a division that does not fit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_gpu as fz
    fz.run_case(seed)


@pytest.mark.parametrize("Ku,ks,levels,loss,reg", [
    (129, [], [], "kl-divergence", False),               # one component in the second half
    (150, [5, 4], [3, 2], "kl-divergence", True),
    (120, [60, 60], [2, 4], "frobenius", True),           # K = 240, the guided components fill 120 of the first half's 128 columns
    (256, [], [], "frobenius", False),                    # the largest model: both halves full
    (200, [0, 7], [2, 3], "kl-divergence", False),
    (257, [], [], "kl-divergence", True),                 # round 4, more than 256: one component in the third block
    (384, [], [], "frobenius", False),                    # three full blocks
    (300, [60, 60], [2, 4], "kl-divergence", True),       # K = 420: four blocks, 36 components in the last, guided ones fill 120 of the first
    (1024, [], [], "frobenius", False),                   # the largest model: eight full blocks
    (700, [0, 7], [2, 3], "kl-divergence", False),        # six blocks, 67 components in the last
])
def test_wide_models_vs_oracle(Ku, ks, levels, loss, reg):
    """128 < K <= 1024: the blocked path (kernels_wide.hpp; every update as den = A.M on the MFMA + an elementwise apply, the
    sweeps once per half) against the oracle's fused iteration, ragged G and N, x3 and f32 sweeps, loss rows included."""
    _fit_vs_oracle(300 + Ku, G=203, N=517, Ku=Ku, ks=ks, levels=levels, loss=loss, iters=3, reg=reg)


@pytest.mark.parametrize("G,N,K,counts,waves", [
    (1031, 4097, 150, True, 2),        # 128 < K <= 160 on one-plane data: the 8-wave one-pass sweep, ragged in both axes (3 / 9 tiles of 512, partly filled)
    (513, 20000, 129, True, 2),        # one component in the second block; 40 column tiles of cells
    (2049, 700, 160, True, 2),         # the largest model of that form; fewer cells than two tiles
    (900, 3000, 161, True, 1),         # one component more: 11 tiles, the 4-wave form
    (1031, 4097, 150, False, 1),       # full significands: the general 4-wave form
    (777, 5000, 100, False, 2),        # 64 < K <= 128 on full significands: the two-wave sweep (x3v), ragged
    (777, 5000, 100, True, 1),         # ... on one-plane data: x3w's one-plane form
    (3000, 1200, 70, False, 2),        # three 32-component tiles, the last 16-component tile all padding (M16A = 5)
    (640, 9000, 96, False, 2),         # three full tiles (M16A = 6)
    (640, 9000, 128, False, 2),        # four full tiles (M16A = 8: no fragment prefetch)
])
def test_two_wave_sweeps_on_ragged_shapes_vs_oracle(G, N, K, counts, waves):
    """Round 4's two-waves-per-SIMD sweeps (stream_gemm_x3v_kernel for 64 < K <= 128 on data with more than one plane; the 8-wave form of the
    one-pass sweep for 128 < K <= 160 on one-plane data) and their one-wave neighbours on shapes that are not multiples of any tile, several
    workgroup tiles wide, against the oracle's fused iteration; alpine_info says which form ran."""
    _fit_vs_oracle(1200 + K + G % 7, G=G, N=N, Ku=K - 4, ks=[4], levels=[3], loss="kl-divergence", iters=3, reg=(K % 2 == 0), counts=counts, expect_waves=waves)


@pytest.mark.parametrize("G,N,Ku,ks,levels,loss", [
    (70001, 1030, 9, [3], [2], "kl-divergence"),          # more genes than a 16-bit index, ragged in both axes
    (140003, 700, 33, [], [], "frobenius"),               # 137 gene tiles of 1024, fewer cells than one tile
    (131072, 257, 60, [2, 2], [2, 3], "kl-divergence"),   # 2^17 genes exactly, one cell past two 128-cell blocks
    (40, 150001, 5, [1], [2], "frobenius"),               # the other extreme: one partial gene tile, 147 cell tiles, spans capped at 16 384 rows
])
def test_extreme_aspect_ratios_vs_oracle(G, N, Ku, ks, levels, loss):
    """Matrices far from the benchmark's aspect ratio (the fuzz campaign keeps G <= 3200): very tall (many gene tiles, few cells) and very
    flat; 2 iterations against the oracle's fused iteration in both sweep modes."""
    _fit_vs_oracle(900 + G % 97, G=G, N=N, Ku=Ku, ks=ks, levels=levels, loss=loss, iters=2)


@pytest.mark.parametrize("Ku", [20, 150])
def test_minibatch_view_larger_than_the_shard(Ku):
    """A mini-batch drawn with replacement may hold MORE cells than the ctx (a sharded rank whose block receives most of a global batch;
    batch_capacity > n_cells): every buffer a view touches must be sized for the view, not for the shard.  Found by
    tests/fuzz_sharded_gpu.py (seed 30429, round 3): the blocked path's den buffer was sized for max(Gp, Np) rows -- a device fault at
    K = 239 with a 8 120-cell batch capacity on a 4 096-cell shard."""
    from alpine_amd import _native as nat
    rng = np.random.default_rng(77 + Ku)
    G, N, cap = 90, 300, 1000
    X = rng.gamma(0.5, 2.0, size=(N, G)).astype(np.float32)
    Y = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=N)]
    p = orc.OracleParams(n_components=Ku, n_covariate_components=[4], lam=[30.0], orth_W=0.05, loss_type="kl-divergence", random_state=5)
    s = orc.init_factors(p, np.ascontiguousarray(X.T), [Y])
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    batches = [rng.integers(0, N, size=n) for n in (cap, 650, 129, cap)]
    with torch.no_grad():
        for idx in batches:
            orc.mu_step_faithful(p, s, torch.tensor(idx, dtype=torch.long))
        s.losses.append(orc.loss_row(p, s))
    for mode in ("x3", "f32"):
        eng = nat.NativeShard(n_genes=G, n_cells=N, n_components=Ku, cov_components=[4], cov_levels=[3], lam=p.lam, orth_W=p.orth_W,
                              eps=p.eps, loss_type=p.loss_type, batch_capacity=cap, x_dtype=mode)
        eng.upload_X_host(X)
        eng.finalize_X()
        eng.upload_Y(0, np.ascontiguousarray(Y.T))
        eng.set_factors(W0, H0, B0)
        for idx in batches:
            eng.batch_step(idx)
        eng.epoch_loss()
        W, H, Bs = eng.get_factors()
        losses = eng.losses()
        eng.close()
        assert rel_fro(W, s.W.numpy()) < 5e-5 and rel_fro(H, s.H.numpy()) < 5e-5, mode
        assert rel_fro(Bs[0], s.Bs[0].numpy()) < 1e-4, mode
        np.testing.assert_allclose(losses[:, :2], np.array(s.losses)[:, :2], rtol=1e-4, err_msg=mode)
