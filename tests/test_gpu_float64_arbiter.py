"""The factors themselves against a FLOAT64 run of the same iteration (the arbiter between two float32 computations),
under the driver's eyes:

  * BASELINE configs[1] whole (20 000 genes x 50 000 cells, K = 50 + [5]), 3 iterations, x3 and f32 sweeps: W / H within
    1e-5 of the float64 host run (measured 3e-7 .. 1e-6; the float32 CPU oracle itself is at 8e-7);
  * the shape that tests/fuzz_gpu.py found in round 2 (seed 1583: 1 100 genes x 77 631 cells, K = 31, gamma X, split_a =
    split_b = 1 -> ONE workgroup share per tile): before the span cap (SweepGeom::sub, SG_MAX_CHAIN = 16 384 rows) one float32
    accumulator ran over all 77 631 cells and W was off by 9.4e-5 after 6 iterations in f32 mode; with the cap the forced
    division obeys the same bound as the library's own: within 1e-5 after 6 iterations in both modes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_cfg2_factors_vs_float64_host_run():
    import fullsize_vs_float64 as fv
    out = fv.compare("cfg2", iters=3, modes=("x3", "f32"), with_float32_oracle=False, log=lambda m: print(m, flush=True))
    for name, d in out["vs_float64"].items():
        assert d["W"] < 1e-5 and d["H"] < 1e-5, (name, d)
        assert all(b < 1e-5 for b in d["B"]), (name, d)
        assert d["loss_total_rel"] < 2e-5 and d["loss_recon_rel"] < 2e-5, (name, d)


@pytest.mark.parametrize("mode", ["f32", "x3"])
def test_one_share_per_tile_obeys_the_accumulation_cap(mode):
    import fuzz_gpu as fz
    from _golden import rel_fro
    from alpine_amd import _native as nat
    from oracle import alpine_oracle as orc
    p, X, Ys, kind, iters, _splits, _rng = fz.make_case(1583)
    N, G = X.shape
    assert (G, N, p.total_components, kind, iters) == (1100, 77631, 31, "gamma", 6)
    s = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
    W0, H0, B0 = s.W.numpy().copy(), s.H.numpy().copy(), [b.numpy().copy() for b in s.Bs]
    s64 = fz.fit_fused_f64(p, X, Ys, W0, H0, B0, iters)
    eng = nat.NativeShard(n_genes=G, n_cells=N, n_components=p.n_components, cov_components=p.n_covariate_components,
                          cov_levels=[y.shape[1] for y in Ys], lam=p.lam, orth_W=p.orth_W, alpha_W=p.alpha_W,
                          l1_ratio_W=p.l1_ratio_W, eps=p.eps, loss_type=p.loss_type, split_a=1, split_b=1, x_dtype=mode)
    try:
        eng.upload_X_host(X)
        eng.finalize_X()
        for i, y in enumerate(Ys):
            eng.upload_Y(i, np.ascontiguousarray(y.T))
        eng.set_factors(W0, H0, B0)
        info = eng.info()
        # one share per tile was asked for; the share is cut into spans of <= 16 384 rows, each with its own piece
        assert info.span_rows_a <= 16384 and info.span_rows_b <= 16384
        assert info.spans_per_workgroup_a == -(-77696 // 16384)                 # Np = 77 696 rows in one share -> 5 spans
        eng.run(iters, with_loss=False)
        W, H, _ = eng.get_factors()
    finally:
        eng.close()
    eW, eH = rel_fro(W, s64.W.numpy()), rel_fro(H, s64.H.numpy())
    print(f"{mode}: vs float64 after {iters} iterations W {eW:.2e} H {eH:.2e}", flush=True)
    assert eW < 1e-5 and eH < 1e-5, (eW, eH)
