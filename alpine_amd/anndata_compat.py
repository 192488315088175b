"""AnnData access.  ``anndata`` is an install-time dependency of the reference
(pyproject.toml:17) but is not present in every image this package runs in, so the real class
is used when importable and a minimal stand-in (same attribute surface the fit path touches:
``X, obs, var_names, shape, obsm, varm, layers``) otherwise.  ``is_anndata`` is what the
``isinstance(adata, ad.AnnData)`` checks of alpine/main.py:307,392 become."""
from __future__ import annotations

import numpy as np
import pandas as pd

try:  # pragma: no cover - depends on the environment
    from anndata import AnnData as _RealAnnData
except Exception:  # ImportError or a broken optional dependency
    _RealAnnData = None


class MiniAnnData:
    """Stand-in with the attributes ALPINE.fit / store_embeddings read and write."""

    def __init__(self, X, obs=None, var_names=None):
        self.X = X
        self.obs = obs if obs is not None else pd.DataFrame(index=np.arange(X.shape[0]))
        self.var_names = pd.Index(var_names if var_names is not None else [f"gene{i}" for i in range(X.shape[1])])
        self.obsm, self.varm, self.layers = {}, {}, {}

    @property
    def shape(self):
        return self.X.shape


AnnData = _RealAnnData if _RealAnnData is not None else MiniAnnData


def is_anndata(obj) -> bool:
    if isinstance(obj, MiniAnnData):
        return True
    return _RealAnnData is not None and isinstance(obj, _RealAnnData)
