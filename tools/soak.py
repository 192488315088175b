"""Soak: thousands of iterations of the default (x3) path at bench sizes; losses must stay finite and the total loss
must not increase by more than rounding from one iteration to the next (pure Frobenius run) -- a hang, a NaN or a slow
drift shows up here, not in 50-iteration benchmarks.  python tools/soak.py [--workload cfg2] [--iters 3000]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--dtype", default="x3")
    args = ap.parse_args()
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = bench.WORKLOADS[args.workload]
    G, N, ku, kcov = wl["genes"], wl["cells"], wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    for loss_type, lam in (("frobenius", 10.0), ("kl-divergence", 1e3)):
        eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=[2] * len(kcov), lam=[lam] * len(kcov),
                                  loss_type=loss_type, x_dtype=args.dtype)
        for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
            torch.cuda.synchronize()
            eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
            eng.synchronize()
            del chunk
        eng.finalize_X()
        for i in range(len(kcov)):
            eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
        W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], [2] * len(kcov))
        eng.set_factors(W0, H0, B0)
        t0 = time.time()
        done = 0
        while done < args.iters:
            k = min(500, args.iters - done)
            eng.run(k, with_loss=True)
            eng.synchronize()
            done += k
            print(f"{loss_type}: {done} iterations, {time.time() - t0:.1f} s", flush=True)
        L = eng.losses()
        W, H, Bs = eng.get_factors()
        eng.close()
        assert np.isfinite(L).all() and np.isfinite(W).all() and np.isfinite(H).all() and (W >= 0).all() and (H >= 0).all()
        up = np.diff(L[:, 0]) / L[:-1, 0]
        print(f"{loss_type}: total loss {L[0, 0]:.6e} -> {L[-1, 0]:.6e}; largest relative increase between iterations {up.max():.2e}; "
              f"{args.iters / (time.time() - t0):.1f} it/s incl. syncs")
        if loss_type == "frobenius":
            assert up.max() <= 1e-6, up.max()
    print("soak ok")


if __name__ == "__main__":
    main()
