"""The committed, fixed-seed slice of tests/fuzz_model_gpu.py: random calls of the drop-in API -- `ALPINE(**random arguments).fit` with random
label columns (missing values, one-level covariates, covariates without guided components), full-batch / mini-batch / weighted epochs, the
block-coordinate branch, scaling on or off, every admissible x_dtype, then an unseeded `transform`, `compute_loss` and (where the case
keeps X resident) a second transform on the resident copy, gene scores and `release` -- against the oracle's op-for-op restatement of the
reference started from the same seed (tests/test_oracle_golden.py pins that restatement on the reference's own outputs)."""
import pytest

import fuzz_model_gpu as fm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(40000, 40032)) + [40775, 40885, 40962] + [60000, 60007, 60036, 60041, 60044, 60060])
def test_random_api_call_vs_oracle(seed):
    """60000 ...: more than 256 components in total (round 4: the blocked path with 3 - 8 column blocks).  40775: a degenerate fit (the reference's scaling divides by zero: NaN pattern must match); 40885 / 40962: the cases during which the
    round-3 campaign's process died of heap corruption (device-to-host copies landing in freed memory; fixed in alpine_hip.hip)."""
    fm.run_case(seed)


def test_random_api_call_with_larger_shapes():
    fm.BIG = True
    try:
        fm.run_case(70003)
    finally:
        fm.BIG = False
