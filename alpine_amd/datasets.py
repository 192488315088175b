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
                               chunk_cells: int = 16384, cell_offset: int = 0):
    """Yield ``(cell0, X_chunk)`` with ``X_chunk`` a (chunk x genes) float32 torch tensor on
    ``device``.  ``S`` is drawn once from ``seed``; ``A`` rows are drawn per chunk from a
    generator keyed by (seed, global cell index of the chunk) so that any sharding of the
    cell axis sees the same matrix."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    conc_s = torch.full((rank, n_genes), 0.3, device=device)
    S = torch._standard_gamma(conc_s, generator=g) / (0.09 * rank)
    for c0 in range(0, n_cells, chunk_cells):
        c1 = min(n_cells, c0 + chunk_cells)
        gc = torch.Generator(device=device)
        gc.manual_seed(seed * 1_000_003 + (cell_offset + c0) + 1)
        A = torch._standard_gamma(torch.full((c1 - c0, rank), 0.3, device=device), generator=gc)
        yield c0, torch.poisson(A @ S, generator=gc)
