import time, numpy as np, pandas as pd, sys
sys.path.insert(0, ".")
import torch
from alpine_amd import ALPINE, MiniAnnData
from alpine_amd.datasets import synth_counts_device_chunks, synth_labels_host
N, G = 50000, 20000
dev = torch.device("cuda", 0)
t = time.perf_counter()
X = np.empty((N, G), dtype=np.float32)
for off, ch in synth_counts_device_chunks(N, G, rank=50, seed=0, device=dev):
    X[off:off + ch.shape[0]] = ch.cpu().numpy()
print("host X built in %.1f s" % (time.perf_counter() - t))
obs = pd.DataFrame({"cond": synth_labels_host(N, ["a", "b"], 1)})
for dtype in ("x3", "f32", "auto"):
    for rep in range(2):
        a = MiniAnnData(X, obs.copy())
        t = time.perf_counter()
        m = ALPINE(n_components=50, n_covariate_components=[5], lam=[1e3], device="cuda", x_dtype=dtype).fit(a, covariate_keys=["cond"], max_iter=50)
        print(dtype, "fit(50 it) wall %.2f s" % (time.perf_counter() - t), "loss", m.loss_history.iloc[-1, 0], getattr(m, "x_dtype_used", None))
import cProfile, pstats
a = MiniAnnData(X, obs.copy())
pr = cProfile.Profile(); pr.enable()
ALPINE(n_components=50, n_covariate_components=[5], lam=[1e3], device="cuda").fit(a, covariate_keys=["cond"], max_iter=50)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
