"""Quick start: the reference's README workflow (README.md:105-126 of ylaboratory/ALPINE) on the MI355X path.

    python examples/quickstart.py            # needs one MI355X; synthetic counts, 2 covariates

With anndata installed, pass a real AnnData instead of MiniAnnData; nothing else changes.  Multi-GPU: launch with
torchrun, call torch.distributed.init_process_group("nccl") and construct ALPINE(..., shard_cells=True) (every rank
holds the full adata) or shard_cells="local" (every rank holds only its own cells)."""
import sys
from pathlib import Path

import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from alpine_amd import ALPINE, MiniAnnData                                # noqa: E402
from alpine_amd.datasets import synth_counts_host, synth_labels_host     # noqa: E402


def main():
    n_cells, n_genes = 5000, 2000
    X = synth_counts_host(n_cells, n_genes, rank=20, seed=0)                       # cells x genes, float32, >= 0
    obs = pd.DataFrame({"condition": synth_labels_host(n_cells, ["ctl", "stim"], seed=1),
                        "batch": synth_labels_host(n_cells, ["b0", "b1", "b2"], seed=2)})
    adata = MiniAnnData(X, obs)

    model = ALPINE(n_components=20, n_covariate_components=[3, 3], lam=[1e3, 1e3], alpha_W=0.0, orth_W=0.0, l1_ratio_W=0.0,
                   device="cuda")                                                  # x_dtype="x3" by default; "auto" for counts
    model.fit(adata, covariate_keys=["condition", "batch"], max_iter=200)
    print(model.loss_history.tail(3))
    print({k: v.shape for k, v in adata.obsm.items()})                             # ALPINE_embedding, condition, batch, ..._dummy_matrix
    print("objective recomputed from adata:", model.compute_loss(adata))
    scores = model.get_covariate_gene_scores()
    print(scores["condition"].head(3))

    # embed new cells with the trained gene signatures (main.py:149-167)
    new = MiniAnnData(synth_counts_host(1000, n_genes, rank=20, seed=7),
                      pd.DataFrame({"condition": synth_labels_host(1000, ["ctl", "stim"], seed=8),
                                    "batch": synth_labels_host(1000, ["b0", "b1", "b2"], seed=9)}))
    model.transform(new)
    print(new.obsm["ALPINE_embedding"].shape)


if __name__ == "__main__":
    main()
