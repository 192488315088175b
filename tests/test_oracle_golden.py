"""Pin the CPU oracle (oracle/alpine_oracle.py) against golden vectors produced by the REAL
reference (ALPINE v0.2.0, torch-CPU) -- the reference itself ships no tests or fixtures
(SURVEY.md section 4), so these committed outputs are the pin."""
import numpy as np
import pytest
import torch

from _golden import ALL_CASES, ALS_CASES, BATCH_CASES, SMALL_CASES, assert_loss_rows_close, load_case, rel_fro
from oracle import alpine_oracle as orc


def _params(c):
    return orc.OracleParams(**{k: v for k, v in c.params.items()})


def _state(c, p):
    return orc.init_factors(p, np.ascontiguousarray(c.X.T), [y.T for y in c.Ys])


@pytest.mark.parametrize("name", ALL_CASES)
def test_init_matches_reference_bitwise(name):
    c = load_case(name)
    assert c.x_ok, "regenerated X differs from the matrix the fixture was made from"
    p = _params(c)
    s = _state(c, p)
    assert np.array_equal(s.W.numpy(), c.W0)
    assert np.array_equal(s.H.numpy(), c.H0)
    for b, b0 in zip(s.Bs, c.B0):
        assert np.array_equal(b.numpy(), b0)


@pytest.mark.parametrize("name", ALL_CASES)
def test_faithful_reproduces_reference(name):
    """Same op order, same RNG stream (randperm per iteration) -> agreement at rounding level
    (MKL may pick different blockings run to run, so not asserted bitwise)."""
    c = load_case(name)
    p = _params(c)
    s = _state(c, p)
    orc.fit_faithful(p, s, c.T, use_perm=True)
    assert rel_fro(s.W.numpy(), c.WT_unscaled) < 2e-5
    assert rel_fro(s.H.numpy(), c.HT_unscaled) < 2e-5
    for b, bt in zip(s.Bs, c.BT_unscaled):
        assert rel_fro(b.numpy(), bt) < 5e-5
    got = np.array(s.losses)
    np.testing.assert_allclose(got, c.loss_history, rtol=2e-5)
    orc.scale_factors(p, s)
    assert rel_fro(s.W.numpy(), c.WT) < 2e-5
    assert rel_fro(s.H.numpy(), c.HT) < 2e-5
    for b, bt in zip(s.Bs, c.BT):
        assert rel_fro(b.numpy(), bt) < 5e-5


@pytest.mark.parametrize("name", SMALL_CASES)
def test_single_step_faithful(name):
    c = load_case(name)
    p = _params(c)
    s = _state(c, p)
    orc.fit_faithful(p, s, 1, use_perm=True, with_loss=False)
    assert rel_fro(s.W.numpy(), c.W1) < 1e-6
    assert rel_fro(s.H.numpy(), c.H1) < 1e-6
    for b, b1 in zip(s.Bs, c.B1):
        assert rel_fro(b.numpy(), b1) < 1e-6


@pytest.mark.parametrize("name", SMALL_CASES)
def test_fused_spec_within_stated_tolerance(name):
    """The re-associated form the HIP kernels implement: 1e-5 per step, 1e-4 after T<=50
    iterations (SURVEY.md section 7 'fp32 tolerance')."""
    c = load_case(name)
    p = _params(c)
    s = _state(c, p)
    orc.fit_fused(p, s, 1, with_loss=False)
    assert rel_fro(s.W.numpy(), c.W1) < 1e-5
    assert rel_fro(s.H.numpy(), c.H1) < 1e-5
    s = _state(c, p)
    orc.fit_fused(p, s, c.T)
    assert rel_fro(s.W.numpy(), c.WT_unscaled) < 1e-4
    assert rel_fro(s.H.numpy(), c.HT_unscaled) < 1e-4
    for b, bt in zip(s.Bs, c.BT_unscaled):
        assert rel_fro(b.numpy(), bt) < 2e-4
    # trace-form loss vs the reference's direct-form loss rows (small shapes: torch.norm is accurate)
    assert_loss_rows_close(np.array(s.losses), c.loss_history, n_cells=c.X.shape[0])


def test_permutation_is_a_numerical_noop():
    c = load_case("kl_2cov_nan")
    p = _params(c)
    a = orc.fit_faithful(p, _state(c, p), c.T, use_perm=True, with_loss=False)
    b = orc.fit_faithful(p, _state(c, p), c.T, use_perm=False, with_loss=False)
    assert rel_fro(a.W.numpy(), b.W.numpy()) < 2e-5
    assert rel_fro(a.H.numpy(), b.H.numpy()) < 2e-5


def test_common_evaluator_matches_direct_loss():
    c = load_case("kl_1cov")
    got = orc.recon_loss_f64(np.ascontiguousarray(c.X.T), c.WT_unscaled, c.HT_unscaled)
    assert abs(got - c.loss_history[-1, 1]) / c.loss_history[-1, 1] < 1e-5


@pytest.mark.parametrize("name", [n for n in SMALL_CASES if load_case(n).H_transform is not None])
def test_transform_oracle_reproduces_reference(name):
    """fit (same RNG stream: init draws + one randperm per iteration), scale, then the unseeded torch.rand init
    of transform on the first 2/3 of the cells -- must land on the reference's transform output."""
    c = load_case(name)
    p = _params(c)
    s = _state(c, p)
    orc.fit_faithful(p, s, c.T, use_perm=True, with_loss=False)
    orc.scale_factors(p, s)
    n_t = (2 * c.X.shape[0]) // 3
    H0 = torch.rand((p.total_components, n_t), dtype=torch.float32)
    Ht = orc.transform_faithful(p.eps, s.W, torch.tensor(np.ascontiguousarray(c.X[:n_t].T)), H0, c.transform_iters)
    assert Ht.shape == c.H_transform.shape
    assert rel_fro(Ht.numpy(), c.H_transform) < 5e-5


@pytest.mark.parametrize("name", BATCH_CASES)
def test_minibatch_oracle_reproduces_reference(name):
    """Mini-batch / weighted sampling: same index streams (global torch generator), same per-batch steps."""
    c = load_case(name)
    p = _params(c)
    s = _state(c, p)
    assert np.array_equal(s.W.numpy(), c.W0) and np.array_equal(s.H.numpy(), c.H0)
    orc.fit_faithful_batches(p, s, c.T, c.fit_kwargs.get("batch_size"), c.fit_kwargs.get("sampling_method", "random"))
    assert rel_fro(s.W.numpy(), c.WT_unscaled) < 2e-5
    assert rel_fro(s.H.numpy(), c.HT_unscaled) < 2e-5
    for b, bt in zip(s.Bs, c.BT_unscaled):
        assert rel_fro(b.numpy(), bt) < 5e-5
    np.testing.assert_allclose(np.array(s.losses), c.loss_history, rtol=5e-5)


@pytest.mark.parametrize("name", ALS_CASES)
def test_als_oracle_reproduces_reference(name):
    c = load_case(name)
    p = _params(c)
    assert p.use_als
    s = _state(c, p)
    orc.fit_faithful_batches(p, s, c.T, c.fit_kwargs.get("batch_size"), c.fit_kwargs.get("sampling_method", "random"))
    assert rel_fro(s.W.numpy(), c.WT_unscaled) < 2e-5
    assert rel_fro(s.H.numpy(), c.HT_unscaled) < 2e-5
    for b, bt in zip(s.Bs, c.BT_unscaled):
        assert rel_fro(b.numpy(), bt) < 5e-5
    np.testing.assert_allclose(np.array(s.losses), c.loss_history, rtol=5e-5)
