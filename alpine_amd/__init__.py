"""alpine_amd -- MI355X-native implementation of ALPINE's multiplicative-update NMF fit loop.

Drop-in for ``ALPINE(**params).fit(adata, covariate_keys=...)`` / ``store_embeddings`` of
ylaboratory/ALPINE; the loop runs in libalpine_hip.so (hand-written gfx950 kernels)."""
import os as _os

# Multi-process GPU work on this platform needs dmabuf IPC (RCCL's and torch's cross-process handles fail with
# "hipIpcGetMemHandle: invalid argument" in the legacy mode).  The HIP runtime reads the variable when it starts, i.e. at the
# first GPU call, so a default set at import time is in effect unless the caller has already touched the GPU or chosen a value.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .anndata_compat import AnnData, MiniAnnData        # noqa: E402
from .model import ALPINE                                # noqa: E402

__all__ = ["ALPINE", "AnnData", "MiniAnnData"]
__version__ = "0.1.0"
