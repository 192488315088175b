"""Wall time of whole ALPINE calls INCLUDING host ingest (PCIe upload of X, validation, read-back) on one MI355X.

    python tools/e2e_fit_timing.py [--cells 50000] [--genes 20000]

Reports, for the default x3 sweeps:
  * fit(max_iter=50): wall, number of X uploads
  * fit(max_iter=None) (200-iteration warm-up + final run): wall, uploads -- ONE upload, the warm-up and the final run share
    the resident copy of X (round 1 uploaded twice)
  * compute_loss(adata) and transform(adata) right after fit on the same adata: served from the resident engine (no upload)
    vs a model created with keep_resident=False (one upload each)
"""
import argparse
import json
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, ".")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--profile", action="store_true", help="cProfile of one fit(max_iter=50): top host-side entries by cumulative time")
    args = ap.parse_args()
    import torch
    from alpine_amd import ALPINE, MiniAnnData, _native
    from alpine_amd.datasets import synth_counts_device_chunks, synth_labels_host
    N, G = args.cells, args.genes
    dev = torch.device("cuda", 0)
    X = np.empty((N, G), dtype=np.float32)
    for off, ch in synth_counts_device_chunks(N, G, rank=50, seed=0, device=dev):
        X[off:off + ch.shape[0]] = ch.cpu().numpy()
    obs = pd.DataFrame({"cond": synth_labels_host(N, ["a", "b"], 1)})
    uploads = {"n": 0, "bytes": 0, "s": 0.0}
    real = _native.NativeShard.upload_X_host

    def counting(self, Xc, *a, **kw):
        t = time.perf_counter()
        r = real(self, Xc, *a, **kw)
        uploads["n"] += 1
        uploads["bytes"] += Xc.nbytes
        uploads["s"] += time.perf_counter() - t
        return r
    _native.NativeShard.upload_X_host = counting

    def timed(label, fn):
        torch.cuda.synchronize()
        u0 = dict(uploads)
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        rec = {"call": label, "wall_s": round(dt, 3), "upload_calls": uploads["n"] - u0["n"],
               "upload_GB": round((uploads["bytes"] - u0["bytes"]) / 1e9, 2), "upload_s": round(uploads["s"] - u0["s"], 3)}
        print(json.dumps(rec), flush=True)
        return out

    kw = dict(n_components=50, n_covariate_components=[5], lam=[1e3], device="cuda")
    a = MiniAnnData(X, obs.copy())
    timed("warm-up fit(max_iter=5) [first call: library load, allocator]", lambda: ALPINE(**kw).fit(a, covariate_keys=["cond"], max_iter=5).release())
    timed("fit(max_iter=50)", lambda: ALPINE(**kw).fit(a, covariate_keys=["cond"], max_iter=50).release())
    if args.profile:
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        ALPINE(**kw).fit(a, covariate_keys=["cond"], max_iter=50).release()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(18)
        print(buf.getvalue())
    m = timed("fit(max_iter=None) = 200-iteration warm-up + final run, resident X", lambda: ALPINE(**kw).fit(a, covariate_keys=["cond"], max_iter=None))
    print(json.dumps({"max_iter_chosen": m.max_iter}))
    timed("compute_loss(adata), resident engine", lambda: m.compute_loss(a))
    timed("transform(adata, n_iter=50), resident engine", lambda: m.transform(a, n_iter=50))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        m.transform(a, n_iter=50)
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(16)
        print(buf.getvalue())
    m.release()
    m2 = timed("fit(max_iter=50, keep_resident=False)", lambda: ALPINE(keep_resident=False, **kw).fit(a, covariate_keys=["cond"], max_iter=50))
    timed("compute_loss(adata), fresh engine", lambda: m2.compute_loss(a))
    timed("transform(adata, n_iter=50), fresh engine", lambda: m2.transform(a, n_iter=50))


if __name__ == "__main__":
    main()
