#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (ylaboratory/ALPINE v0.2.0) on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only where ``/root/reference`` exists (the build
container); the GPU box never sees the reference, it gets the ``.npz`` files this script
writes into ``tests/golden/``.  Nothing of the reference's source is copied: the script
imports it, feeds it seeded inputs and records inputs + outputs.

Four third-party modules the reference imports at module level are absent from this image
(``anndata``, ``scanpy``, ``kneed``, ``hyperopt``); none of them is touched by
``fit(max_iter=<int>)`` / ``store_embeddings`` (only ``isinstance`` checks on
``ad.AnnData``, main.py:307,392), so empty stand-in modules are registered first
(SURVEY.md 8c recipe).

Usage:  python oracle/gen_golden.py [--out tests/golden] [--only NAME ...]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import types

import numpy as np
import pandas as pd

REFERENCE = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _install_stubs():
    sys.dont_write_bytecode = True          # the reference tree is read-only
    stubs = {
        "scanpy": {},
        "kneed": {"KneeLocator": object},
        "hyperopt": dict(fmin=None, tpe=None, hp=None, Trials=None, STATUS_OK=None, STATUS_FAIL=None),
    }
    for name, attrs in stubs.items():
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class AnnData:                           # minimal stand-in: only what fit/store_embeddings touch
        def __init__(self, X, obs=None, var_names=None):
            self.X = X
            self.obs = obs
            self.var_names = pd.Index(var_names if var_names is not None
                                      else [f"gene{i}" for i in range(X.shape[1])])
            self.shape = X.shape
            self.obsm, self.varm, self.layers = {}, {}, {}

    ad = types.ModuleType("anndata")
    ad.AnnData = AnnData
    sys.modules["anndata"] = ad
    sys.path.insert(0, REFERENCE)
    return AnnData


# --------------------------------------------------------------------------- cases
def _labels(rng, n, levels, nan_frac=0.0):
    lab = rng.choice(levels, size=n).astype(object)
    if nan_frac > 0:
        lab[rng.random(n) < nan_frac] = np.nan
    return lab


def make_case_inputs(case: dict):
    """Seeded inputs of a case: X (cells x genes, float32) and the obs columns."""
    sys.path.insert(0, REPO)
    from alpine_amd.datasets import synth_counts_host
    rng = np.random.default_rng(case["seed"])
    n, g = case["n_cells"], case["n_genes"]
    if case.get("data", "gamma") == "gamma":
        X = rng.gamma(0.3, 3.0, size=(n, g)).astype(np.float32)
    else:
        X = synth_counts_host(n, g, rank=case["params"]["n_components"], seed=case["seed"])
    obs = {}
    for key, levels, nan_frac in case["covariates"]:
        obs[key] = _labels(rng, n, levels, nan_frac)
    # degenerate inputs: genes never expressed / empty cells (exact zeros through every product and both updates)
    for gidx in case.get("zero_genes", []):
        X[:, gidx] = 0.0
    for cidx in case.get("zero_cells", []):
        X[cidx, :] = 0.0
    if case.get("skew_first"):
        # the first covariate becomes: its first level on the first `skew_first` cells, its second level everywhere else (a
        # rare, heavily weighted joint label that lives in the FIRST shard of a cell-sharded run)
        key, levels, _ = case["covariates"][0]
        lab = np.array([levels[1]] * n, dtype=object)
        lab[:case["skew_first"]] = levels[0]
        obs[key] = lab
    return X, pd.DataFrame(obs)


CASES = [
    dict(name="kl_1cov", transform_iters=15, n_cells=96, n_genes=64, seed=0, T=50,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3])),
    dict(name="fro_1cov", n_cells=96, n_genes=64, seed=1, T=50,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[10.0], loss_type="frobenius")),
    dict(name="kl_reg", n_cells=96, n_genes=64, seed=2, T=50,
         covariates=[("cond", ["ctl", "stim"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3],
                     orth_W=0.1, alpha_W=0.5, l1_ratio_W=0.3)),
    dict(name="kl_2cov_nan", transform_iters=15, n_cells=120, n_genes=80, seed=3, T=30,
         covariates=[("c1", ["x", "y", "z"], 0.1), ("c2", ["p", "q"], 0.05)],
         params=dict(n_components=5, n_covariate_components=[2, 3], lam=[1e3, 5e2])),
    dict(name="fro_2cov_reg", n_cells=120, n_genes=80, seed=4, T=30,
         covariates=[("c1", ["x", "y", "z"], 0.1), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[2, 3], lam=[5.0, 2.0],
                     loss_type="frobenius", orth_W=0.05, alpha_W=0.2, l1_ratio_W=0.5)),
    dict(name="ragged", transform_iters=15, n_cells=257, n_genes=130, seed=5, T=20,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=7, n_covariate_components=[3], lam=[1e3])),
    dict(name="one_iter", n_cells=96, n_genes=64, seed=6, T=1,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3])),
    dict(name="k74", transform_iters=15, n_cells=300, n_genes=150, seed=7, T=10,
         covariates=[("cond", ["a", "b", "c"], 0.0)],
         params=dict(n_components=70, n_covariate_components=[4], lam=[1e3])),
    dict(name="k105", n_cells=200, n_genes=160, seed=8, T=5,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=100, n_covariate_components=[5], lam=[1e3])),
    dict(name="counts_2cov", n_cells=400, n_genes=300, seed=9, T=40, data="counts",
         covariates=[("cond", ["a", "b"], 0.0), ("batch", ["b0", "b1"], 0.0)],
         params=dict(n_components=12, n_covariate_components=[3, 3], lam=[1e3, 1e3],
                     alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5)),
    # mini-batch and weighted sampling (main.py:509-521, sampling.py): stochastic W updates, duplicates under replacement
    dict(name="mb_random", n_cells=96, n_genes=64, seed=10, T=12, fit_kwargs=dict(batch_size=32),
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3])),
    dict(name="mb_weighted", n_cells=150, n_genes=80, seed=11, T=10, fit_kwargs=dict(batch_size=40, sampling_method="weighted"),
         covariates=[("c1", ["x", "y", "z"], 0.1), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[2, 3], lam=[1e3, 5e2], alpha_W=0.3, orth_W=0.05, l1_ratio_W=0.5)),
    dict(name="full_weighted", n_cells=96, n_genes=64, seed=12, T=10, fit_kwargs=dict(sampling_method="weighted"),
         covariates=[("cond", ["a", "b", "c"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[10.0], loss_type="frobenius")),
    # weighted sampling with a rare label confined to the first cells: in a 2-rank run rank 0 receives ~74 % of every
    # epoch's draws, i.e. MORE cells than its shard holds (draws with replacement) -- the mini-batch view's partial-block
    # buffers must be sized for the batch, not for the shard
    dict(name="weighted_skew", n_cells=640, n_genes=64, seed=16, T=5, fit_kwargs=dict(sampling_method="weighted"), skew_first=32,
         covariates=[("cond", ["rare", "common"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3])),
    # block-coordinate branch (use_als=True, main.py:523-588)
    dict(name="als_kl", n_cells=96, n_genes=64, seed=13, T=20,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=4, n_covariate_components=[2], lam=[1e3], use_als=True, orth_W=0.1, alpha_W=0.5, l1_ratio_W=0.3)),
    dict(name="als_fro_2cov", n_cells=120, n_genes=80, seed=14, T=15,
         covariates=[("c1", ["x", "y", "z"], 0.1), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[2, 3], lam=[5.0, 2.0], loss_type="frobenius", use_als=True)),
    dict(name="als_k74_mb", n_cells=300, n_genes=150, seed=15, T=6, fit_kwargs=dict(batch_size=128),
         covariates=[("cond", ["a", "b", "c"], 0.0)],
         params=dict(n_components=70, n_covariate_components=[4], lam=[1e3], use_als=True, orth_W=0.05)),
    # round 2: paths that only the oracle had covered so far, pinned on the reference itself
    # more label levels in total (36 + 7 = 43) than the H update keeps in LDS (32): guided terms and statistics read Y from memory
    dict(name="many_levels", transform_iters=10, n_cells=260, n_genes=90, seed=21, T=12,
         covariates=[("celltype", [f"t{i:02d}" for i in range(36)], 0.05), ("batch", [f"b{i}" for i in range(7)], 0.0)],
         params=dict(n_components=6, n_covariate_components=[4, 2], lam=[1e3, 2e2])),
    # three covariates, Frobenius loss, missing labels, all regularisers
    dict(name="fro_3cov", n_cells=200, n_genes=96, seed=22, T=20,
         covariates=[("c1", ["x", "y", "z"], 0.1), ("c2", ["p", "q"], 0.05), ("c3", ["u", "v", "w", "s"], 0.0)],
         params=dict(n_components=6, n_covariate_components=[2, 3, 2], lam=[5.0, 2.0, 8.0],
                     loss_type="frobenius", orth_W=0.05, alpha_W=0.2, l1_ratio_W=0.5)),
    # several workgroup tiles in both sweeps, multi-piece tiles, 12 blocks of 128 cells in the fused tails: counts, K = 47
    dict(name="mid_counts", transform_iters=8, n_cells=1500, n_genes=700, seed=23, T=12, data="counts",
         covariates=[("cond", ["a", "b", "c"], 0.02), ("batch", ["b0", "b1"], 0.0)],
         params=dict(n_components=40, n_covariate_components=[4, 3], lam=[1e3, 5e2], alpha_W=1.0, orth_W=0.1, l1_ratio_W=0.5)),
    # block-coordinate branch on weighted mini-batches (duplicates under replacement inside the group loop)
    dict(name="als_weighted_mb", n_cells=160, n_genes=72, seed=24, T=6, fit_kwargs=dict(batch_size=48, sampling_method="weighted"),
         covariates=[("c1", ["x", "y", "z"], 0.0), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[2, 3], lam=[1e3, 5e2], use_als=True, orth_W=0.05, alpha_W=0.3, l1_ratio_W=0.5)),
    # non-default eps and seed, pure-L1 regulariser (l1_ratio_W = 1)
    dict(name="nondefault", transform_iters=6, n_cells=130, n_genes=70, seed=25, T=25,
         covariates=[("cond", ["a", "b", "c"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[3], lam=[50.0], eps=1e-4, random_state=7,
                     orth_W=0.2, alpha_W=0.1, l1_ratio_W=1.0)),
    # round 3: model sizes the reference accepts and the build used to refuse (main.py:331-336 rejects only n < 0)
    # a covariate WITHOUT guided components (k_i = 0): it contributes only its prediction-loss column
    dict(name="k0_split", transform_iters=8, n_cells=140, n_genes=70, seed=31, T=20,
         covariates=[("c1", ["x", "y", "z"], 0.08), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[0, 3], lam=[1e3, 5e2])),
    dict(name="k0_fro", n_cells=120, n_genes=64, seed=32, T=15,
         covariates=[("c1", ["x", "y"], 0.0), ("c2", ["p", "q", "r"], 0.05)],
         params=dict(n_components=4, n_covariate_components=[2, 0], lam=[5.0, 2.0],
                     loss_type="frobenius", orth_W=0.05, alpha_W=0.2, l1_ratio_W=0.5)),
    # more than 64 guided components in total (40 + 30 of K = 90): guided columns in every 32-column tile of H
    dict(name="guided_wide", transform_iters=6, n_cells=200, n_genes=100, seed=33, T=8,
         covariates=[("c1", ["x", "y", "z"], 0.0), ("c2", ["p", "q"], 0.02)],
         params=dict(n_components=20, n_covariate_components=[40, 30], lam=[1e3, 2e2], orth_W=0.05, alpha_W=0.3, l1_ratio_W=0.5)),
    dict(name="guided_wide_fro", n_cells=160, n_genes=96, seed=34, T=8,
         covariates=[("c1", ["x", "y", "z", "w"], 0.0), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=24, n_covariate_components=[36, 36], lam=[4.0, 2.0], loss_type="frobenius")),
    dict(name="als_guided_wide", n_cells=150, n_genes=90, seed=35, T=5,
         covariates=[("c1", ["x", "y", "z"], 0.0), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=10, n_covariate_components=[40, 30], lam=[1e3, 5e2], use_als=True, orth_W=0.05)),
    # more than 128 components in total (the build's blocked two-half path, kernels_wide.hpp)
    dict(name="wide_k150", transform_iters=5, n_cells=120, n_genes=100, seed=36, T=6,
         covariates=[("c1", ["x", "y", "z"], 0.05), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=140, n_covariate_components=[6, 4], lam=[1e3, 2e2], orth_W=0.05, alpha_W=0.3, l1_ratio_W=0.5)),
    dict(name="wide_k200_fro", n_cells=130, n_genes=90, seed=37, T=5,
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=190, n_covariate_components=[10], lam=[4.0], loss_type="frobenius")),
    # ... the block-coordinate branch and mini-batches on that path
    dict(name="als_wide_k150", n_cells=120, n_genes=100, seed=38, T=4,
         covariates=[("c1", ["x", "y", "z"], 0.05), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=138, n_covariate_components=[7, 5], lam=[1e3, 2e2], use_als=True, orth_W=0.05, alpha_W=0.2, l1_ratio_W=0.5)),
    dict(name="als_wide_k160_fro", n_cells=110, n_genes=96, seed=39, T=4,
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=150, n_covariate_components=[10], lam=[3.0], use_als=True, loss_type="frobenius")),
    dict(name="mb_wide_k150", n_cells=150, n_genes=90, seed=40, T=6, fit_kwargs=dict(batch_size=48),
         covariates=[("c1", ["x", "y", "z"], 0.0), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=141, n_covariate_components=[5, 4], lam=[1e3, 5e2], orth_W=0.05)),
    dict(name="mb_weighted_wide_k140", n_cells=160, n_genes=80, seed=41, T=5, fit_kwargs=dict(batch_size=50, sampling_method="weighted"),
         covariates=[("c1", ["x", "y", "z"], 0.1)],
         params=dict(n_components=134, n_covariate_components=[6], lam=[1e2], alpha_W=0.3, l1_ratio_W=0.5, loss_type="frobenius")),
    dict(name="als_mb_wide_k140", n_cells=140, n_genes=80, seed=42, T=4, fit_kwargs=dict(batch_size=64),
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=132, n_covariate_components=[8], lam=[1e2], use_als=True, orth_W=0.02)),
    # more than 256 components (round 4: the blocked path generalised to ceil(K / 128) <= 8 column blocks): an odd number of blocks with a
    # partly filled last one, five blocks with 8 components in the last, the block-coordinate branch, mini-batches
    dict(name="wide_k300", transform_iters=4, n_cells=130, n_genes=100, seed=50, T=5,
         covariates=[("c1", ["x", "y", "z"], 0.05), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=290, n_covariate_components=[6, 4], lam=[1e3, 2e2], orth_W=0.05, alpha_W=0.3, l1_ratio_W=0.5)),
    dict(name="wide_k520_fro", n_cells=110, n_genes=90, seed=51, T=4,
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=510, n_covariate_components=[10], lam=[4.0], loss_type="frobenius")),
    dict(name="als_wide_k270", n_cells=120, n_genes=96, seed=52, T=3,
         covariates=[("c1", ["x", "y", "z"], 0.05), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=258, n_covariate_components=[7, 5], lam=[1e3, 2e2], use_als=True, orth_W=0.05, alpha_W=0.2, l1_ratio_W=0.5)),
    dict(name="mb_wide_k300", n_cells=150, n_genes=90, seed=53, T=4, fit_kwargs=dict(batch_size=48),
         covariates=[("c1", ["x", "y", "z"], 0.0), ("c2", ["p", "q"], 0.0)],
         params=dict(n_components=291, n_covariate_components=[5, 4], lam=[1e3, 5e2], orth_W=0.05)),
    # degenerate inputs the reference accepts: genes that are zero in every cell, cells that are zero in every gene, a covariate
    # with ONE level, labels that are mostly missing, a matrix smaller than any tile
    dict(name="zeros_kl", transform_iters=5, n_cells=100, n_genes=70, seed=43, T=12, zero_genes=[0, 3, 40, 69], zero_cells=[0, 5, 77, 99],
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=5, n_covariate_components=[2], lam=[1e3])),
    dict(name="zeros_fro_reg", n_cells=100, n_genes=70, seed=44, T=12, zero_genes=[1, 2, 33], zero_cells=[10, 11, 98],
         covariates=[("c1", ["x", "y", "z"], 0.1)],
         params=dict(n_components=6, n_covariate_components=[3], lam=[5.0], loss_type="frobenius", orth_W=0.05, alpha_W=0.3, l1_ratio_W=0.5)),
    dict(name="one_level_sparse_labels", n_cells=90, n_genes=60, seed=45, T=10,
         covariates=[("only", ["a"], 0.0), ("rare", ["p", "q", "r"], 0.9)],
         params=dict(n_components=4, n_covariate_components=[2, 2], lam=[1e2, 1e3])),
    dict(name="tiny", transform_iters=4, n_cells=9, n_genes=5, seed=46, T=8,
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=2, n_covariate_components=[1], lam=[10.0])),
    dict(name="tiny_als", n_cells=7, n_genes=11, seed=47, T=6,
         covariates=[("c1", ["x", "y"], 0.0)],
         params=dict(n_components=3, n_covariate_components=[2], lam=[10.0], use_als=True, orth_W=0.1)),
    # BASELINE.json configs[0]: the reference's own CPU-runnable case.  X is regenerated from
    # the seed by the tests (40 MB is not a fixture); only outputs + an input checksum are stored.
    dict(name="cfg1", n_cells=5000, n_genes=2000, seed=0, T=50, store_X=False,
         covariates=[("cond", ["a", "b"], 0.0)],
         params=dict(n_components=20, n_covariate_components=[2], lam=[1e3])),
]


def _cat(mats_w, mats_h):
    return (np.concatenate([np.asarray(w, dtype=np.float32) for w in mats_w], axis=1),
            np.concatenate([np.asarray(h, dtype=np.float32) for h in mats_h], axis=0))


def run_case(case: dict, AnnData, out_dir: str):
    import alpine  # the reference
    X, obs = make_case_inputs(case)
    keys = [c[0] for c in case["covariates"]]
    params = dict(case["params"])
    out = {}

    def run(T, scale):
        a = AnnData(X.copy(), obs.copy())
        m = alpine.ALPINE(device="cpu", scale_needed=scale, **params)
        m.fit(a, covariate_keys=keys, max_iter=T, **case.get("fit_kwargs", {}))
        return m, a

    # init exactly as fit() produces it (main.py:135 -> :436-472)
    m0 = alpine.ALPINE(device="cpu", **params)
    m0.fe = alpine.main.FeatureEncoders(keys)
    Y0 = m0.fe.fit_transform(obs)
    mats = m0._initialize_matrices(np.ascontiguousarray(X.T), Y0)
    out["W0"], out["H0"] = _cat([w.numpy() for w in mats.Ws], [h.numpy() for h in mats.Hs])
    for i, b in enumerate(mats.Bs):
        out[f"B0_{i}"] = b.numpy().astype(np.float32)

    # one step, unscaled
    m1, _ = run(1, False)
    out["W1"], out["H1"] = _cat(m1.matrices["Ws"], m1.matrices["Hs"])
    for i, b in enumerate(m1.matrices["Bs"]):
        out[f"B1_{i}"] = b
    # T steps, unscaled and scaled
    mu, _ = run(case["T"], False)
    out["WT_unscaled"], out["HT_unscaled"] = _cat(mu.matrices["Ws"], mu.matrices["Hs"])
    for i, b in enumerate(mu.matrices["Bs"]):
        out[f"BT_unscaled_{i}"] = b
    ms, a_s = run(case["T"], True)
    out["WT"], out["HT"] = _cat(ms.matrices["Ws"], ms.matrices["Hs"])
    for i, b in enumerate(ms.matrices["Bs"]):
        out[f"BT_{i}"] = b
    for i, y in enumerate(ms.matrices["Ys"]):
        out[f"Y_{i}"] = y                                   # C_i x N as the reference holds it
    # transform (main.py:149-167, :678-724) right after fit, on a different set of cells (the first 2/3), so the
    # unseeded torch.rand init continues the fit's RNG stream exactly as a user would see it
    if case.get("transform_iters"):
        n_t = (2 * case["n_cells"]) // 3
        a_t = AnnData(X[:n_t].copy(), obs.iloc[:n_t].copy())
        ms.transform(a_t, n_iter=case["transform_iters"])
        Ht = [np.asarray(a_t.obsm[k]).T for k in keys] + [np.asarray(a_t.obsm["ALPINE_embedding"]).T]
        out["H_transform"] = np.concatenate(Ht, axis=0).astype(np.float32)
    out["loss_history"] = ms.loss_history.to_numpy(dtype=np.float64)
    assert np.array_equal(mu.loss_history.to_numpy(), ms.loss_history.to_numpy())

    meta = dict(
        name=case["name"], n_cells=case["n_cells"], n_genes=case["n_genes"], seed=case["seed"],
        T=case["T"], data=case.get("data", "gamma"), covariate_keys=keys, transform_iters=case.get("transform_iters", 0),
        fit_kwargs=case.get("fit_kwargs", {}),
        covariates=[[k, lv, nf] for k, lv, nf in case["covariates"]],
        params=params, loss_columns=list(ms.loss_history.columns),
        encoded_labels=ms.fe.encoded_labels,
        obsm_shapes={k: list(np.asarray(v).shape) for k, v in a_s.obsm.items()},
        varm_shapes={k: list(np.asarray(v).shape) for k, v in a_s.varm.items()},
        x_sha256=hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest(),
        generator="oracle/gen_golden.py running /root/reference (ALPINE v0.2.0) on torch-CPU",
    )
    import torch
    meta["torch"] = torch.__version__
    meta["torch_threads"] = torch.get_num_threads()
    if case.get("store_X", True):
        out["X"] = X
        for k in keys:
            out[f"obs_{k}"] = np.array(["__nan__" if (isinstance(v, float) and np.isnan(v)) else v
                                        for v in obs[k].tolist()], dtype="U16")
    out["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(out_dir, f"{case['name']}.npz")
    np.savez_compressed(path, **out)
    print(f"{case['name']:>14}: {os.path.getsize(path) / 1024:8.1f} KiB  "
          f"loss[-1]={out['loss_history'][-1].tolist()}")


POSTHOC_CASES = ["kl_1cov", "kl_2cov_nan", "fro_2cov_reg", "counts_2cov", "many_levels", "fro_3cov", "k0_split", "wide_k150"]


def run_posthoc(case: dict, AnnData, out_dir: str):
    """Post-fit helpers of the reference on the fitted model: compute_loss(adata) on the training
    cells (main.py:187-236) and, where the case has a transform, on the transformed cells; get_covariate_gene_scores()
    (main.py:246-273).  Written to a separate fixture so that the fit fixtures stay byte-identical."""
    import alpine  # the reference
    X, obs = make_case_inputs(case)
    keys = [c[0] for c in case["covariates"]]
    a = AnnData(X.copy(), obs.copy())
    m = alpine.ALPINE(device="cpu", **case["params"])
    m.fit(a, covariate_keys=keys, max_iter=case["T"], **case.get("fit_kwargs", {}))
    out = {"compute_loss_fit": np.float64(m.compute_loss(a))}
    scores = m.get_covariate_gene_scores()
    for k in keys:
        out[f"gene_scores_{k}"] = scores[k].to_numpy(dtype=np.float64)
    cols = {k: [str(c) for c in scores[k].columns] for k in keys}
    a2 = AnnData(X.copy(), obs.copy())
    assert m.get_covariate_gene_scores(a2) is None
    varm_keys = sorted(a2.varm)
    if case.get("transform_iters"):
        n_t = (2 * case["n_cells"]) // 3
        a_t = AnnData(X[:n_t].copy(), obs.iloc[:n_t].copy())
        m.transform(a_t, n_iter=case["transform_iters"])
        out["compute_loss_transform"] = np.float64(m.compute_loss(a_t))
    out["meta_json"] = np.array(json.dumps(dict(name=case["name"], gene_score_columns=cols, gene_score_varm_keys=varm_keys,
                                                generator="oracle/gen_golden.py --posthoc running /root/reference (ALPINE v0.2.0) on torch-CPU")))
    path = os.path.join(out_dir, f"posthoc_{case['name']}.npz")
    np.savez_compressed(path, **out)
    print(f"posthoc {case['name']:>14}: compute_loss={float(out['compute_loss_fit']):.9g} "
          f"{'transform=' + format(float(out['compute_loss_transform']), '.9g') if 'compute_loss_transform' in out else ''}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--posthoc", action="store_true", help="write only the posthoc_<case>.npz fixtures")
    args = ap.parse_args()
    if not os.path.isdir(REFERENCE):
        sys.exit("the reference is not present here; golden vectors can only be regenerated in the build container")
    AnnData = _install_stubs()
    os.makedirs(args.out, exist_ok=True)
    for case in CASES:
        if args.only and case["name"] not in args.only:
            continue
        if args.posthoc:
            if case["name"] in POSTHOC_CASES:
                run_posthoc(case, AnnData, args.out)
        else:
            run_case(case, AnnData, args.out)


if __name__ == "__main__":
    main()
