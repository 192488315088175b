"""alpine_amd -- MI355X-native implementation of ALPINE's multiplicative-update NMF fit loop.

Drop-in for ``ALPINE(**params).fit(adata, covariate_keys=...)`` / ``store_embeddings`` of
ylaboratory/ALPINE; the loop runs in libalpine_hip.so (hand-written gfx950 kernels)."""
# (Importing this package changes nothing in the process environment.  Multi-process GPU work needs dmabuf IPC --
# HSA_ENABLE_IPC_MODE_LEGACY=0, read when the HIP runtime starts; the sharded entry points -- fit(shard_cells=...), bench.py's
# workers -- set that default themselves via alpine_amd.sharded.ensure_dmabuf_ipc.)
from .anndata_compat import AnnData, MiniAnnData
from .model import ALPINE

__all__ = ["ALPINE", "AnnData", "MiniAnnData"]
__version__ = "0.1.0"
