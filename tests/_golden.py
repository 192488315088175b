"""Loader for the committed golden vectors (tests/golden/*.npz, written by oracle/gen_golden.py
from the real reference).  Fixtures are data only: seeded inputs + the reference's outputs."""
from __future__ import annotations

import hashlib
import json
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_CASES = ["kl_1cov", "fro_1cov", "kl_reg", "kl_2cov_nan", "fro_2cov_reg", "ragged",
               "one_iter", "k74", "k105", "counts_2cov", "many_levels", "fro_3cov", "mid_counts", "nondefault",
               "k0_split", "k0_fro", "guided_wide", "guided_wide_fro",       # round 3: k_i = 0, more than 64 guided components
               "wide_k150", "wide_k200_fro",                                 # ... and more than 128 components in total
               "wide_k300", "wide_k520_fro",                                 # round 4: more than 256 (3 and 5 column blocks of 128)
               "zeros_kl", "zeros_fro_reg", "one_level_sparse_labels", "tiny"]  # degenerate inputs: all-zero genes / cells, one-level covariate, 9 x 5 matrix
BATCH_CASES = ["mb_random", "mb_weighted", "full_weighted", "weighted_skew",        # mini-batch / weighted sampling (stochastic)
               "mb_wide_k150", "mb_weighted_wide_k140", "mb_wide_k300"]            # ... with more than 128 / 256 components
ALS_CASES = ["als_kl", "als_fro_2cov", "als_k74_mb", "als_weighted_mb", "als_guided_wide",                  # use_als=True (block-coordinate branch)
             "als_wide_k150", "als_wide_k160_fro", "als_mb_wide_k140", "als_wide_k270", "tiny_als"]                                       # ... with more than 128 components
ALL_CASES = SMALL_CASES + ["cfg1"]


def _regen_inputs(meta):
    """Same seeded recipe as oracle/gen_golden.py:make_case_inputs (gamma data only)."""
    rng = np.random.default_rng(meta["seed"])
    n, g = meta["n_cells"], meta["n_genes"]
    assert meta["data"] == "gamma"
    X = rng.gamma(0.3, 3.0, size=(n, g)).astype(np.float32)
    obs = {}
    for key, levels, nan_frac in meta["covariates"]:
        lab = rng.choice(levels, size=n).astype(object)
        if nan_frac > 0:
            lab[rng.random(n) < nan_frac] = np.nan
        obs[key] = lab
    return X, pd.DataFrame(obs)


def load_case(name: str) -> SimpleNamespace:
    z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    keys = meta["covariate_keys"]
    if "X" in z.files:
        X = z["X"]
        obs = pd.DataFrame({k: np.array([np.nan if v == "__nan__" else v for v in z[f"obs_{k}"].tolist()],
                                        dtype=object) for k in keys})
    else:
        X, obs = _regen_inputs(meta)
    sha = hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest()
    n_cov = len(keys)
    c = SimpleNamespace(name=name, meta=meta, params=dict(meta["params"]), keys=keys, X=X, obs=obs,
                        x_ok=(sha == meta["x_sha256"]), T=meta["T"],
                        loss_history=z["loss_history"], loss_columns=meta["loss_columns"])
    for tag in ("0", "1", "T_unscaled", "T"):
        setattr(c, f"W{tag}", z[f"W{tag}"])
        setattr(c, f"H{tag}", z[f"H{tag}"])
        setattr(c, f"B{tag}", [z[f"B{tag}_{i}"] for i in range(n_cov)])
    c.Ys = [z[f"Y_{i}"] for i in range(n_cov)]          # C_i x N
    c.transform_iters = meta.get("transform_iters", 0)
    c.fit_kwargs = meta.get("fit_kwargs", {})
    c.H_transform = z["H_transform"] if "H_transform" in z.files else None
    return c


POSTHOC_CASES = ["kl_1cov", "kl_2cov_nan", "fro_2cov_reg", "counts_2cov", "many_levels", "fro_3cov", "k0_split", "wide_k150"]


def load_posthoc(name: str) -> SimpleNamespace:
    """compute_loss / get_covariate_gene_scores outputs of the reference on the fitted model (oracle/gen_golden.py --posthoc)."""
    z = np.load(os.path.join(GOLDEN_DIR, f"posthoc_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    return SimpleNamespace(compute_loss_fit=float(z["compute_loss_fit"]),
                           compute_loss_transform=float(z["compute_loss_transform"]) if "compute_loss_transform" in z.files else None,
                           gene_scores={k: z[f"gene_scores_{k}"] for k in meta["gene_score_columns"]},
                           gene_score_columns=meta["gene_score_columns"], varm_keys=meta["gene_score_varm_keys"])


def rel_fro(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def assert_loss_rows_close(got, want, n_cells, rtol=5e-5):
    """Columns [total, recon] relative; prediction-loss columns are sums over cells of terms
    that cancel to ~0 at convergence, so they get an absolute floor of 1e-7 per cell."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    np.testing.assert_allclose(got[:, :2], want[:, :2], rtol=rtol)
    np.testing.assert_allclose(got[:, 2:], want[:, 2:], rtol=20 * rtol, atol=1e-7 * n_cells)
