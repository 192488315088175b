"""alpine_amd -- MI355X-native implementation of ALPINE's multiplicative-update NMF fit loop.

Drop-in for ``ALPINE(**params).fit(adata, covariate_keys=...)`` / ``store_embeddings`` of
ylaboratory/ALPINE; the loop runs in libalpine_hip.so (hand-written gfx950 kernels)."""
from .anndata_compat import AnnData, MiniAnnData
from .model import ALPINE

__all__ = ["ALPINE", "AnnData", "MiniAnnData"]
__version__ = "0.1.0"
