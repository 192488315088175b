"""Synthetic non-negative gene-expression-like matrices for tests and bench.

``X = Poisson(A @ S / c)`` with ``A ~ Gamma(0.3, 1)`` (cells x rank) and
``S ~ Gamma(0.3, 1)`` (rank x genes), ``c = 0.09 * rank`` so that the mean is ~1 and most
entries are zero, like count data.  Two generators with the same distribution: numpy on the
host (small test cases; bit-stable for a seed) and torch on the device (bench sizes: a
20 000 x 200 000 matrix is generated on the GPU in cell chunks, never on the host).
"""
from __future__ import annotations

import numpy as np


def synth_counts_host(n_cells: int, n_genes: int, rank: int, seed: int = 0) -> np.ndarray:
    """cells x genes float32, C-contiguous (AnnData's layout)."""
    rng = np.random.default_rng(seed)
    A = rng.gamma(0.3, 1.0, size=(n_cells, rank))
    S = rng.gamma(0.3, 1.0, size=(rank, n_genes))
    lam = (A @ S) / (0.09 * rank)
    return rng.poisson(lam).astype(np.float32)


def synth_labels_host(n_cells: int, levels, seed: int = 1) -> np.ndarray:
    """object-dtype label column, i.i.d. uniform over ``levels`` (main.py:415 wants kind 'O')."""
    rng = np.random.default_rng(seed)
    return rng.choice(np.asarray(levels, dtype=object), size=n_cells).astype(object)


def synth_counts_device_chunks(n_cells: int, n_genes: int, rank: int, seed: int, device,
                               chunk_cells: int = 8192, cell_offset: int = 0):
    """Yield ``(local_cell0, X_chunk)`` for the cells ``[cell_offset, cell_offset + n_cells)`` of ONE global synthetic
    matrix, with ``X_chunk`` a (rows x genes) float32 torch tensor on ``device``.  ``S`` is drawn once from ``seed``;
    cells are generated in fixed GLOBAL blocks of ``chunk_cells`` cells, each from a generator keyed by (seed, global
    block index), so every sharding of the cell axis sees the identical matrix (a shard that starts or ends inside
    a block generates the whole block and keeps its rows)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    S = torch._standard_gamma(torch.full((rank, n_genes), 0.3, device=device), generator=g) / (0.09 * rank)
    c_lo, c_hi = cell_offset, cell_offset + n_cells
    for blk in range(c_lo // chunk_cells, (c_hi + chunk_cells - 1) // chunk_cells):
        b0 = blk * chunk_cells
        gc = torch.Generator(device=device)
        gc.manual_seed(seed * 1_000_003 + blk + 1)
        A = torch._standard_gamma(torch.full((chunk_cells, rank), 0.3, device=device), generator=gc)
        Xb = torch.poisson(A @ S, generator=gc)
        lo, hi = max(c_lo, b0), min(c_hi, b0 + chunk_cells)
        yield lo - c_lo, Xb[lo - b0:hi - b0]
