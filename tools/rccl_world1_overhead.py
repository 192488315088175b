"""What the library's own all-reduce costs per iteration when RCCL has nothing to move: a communicator of ONE rank attached to the engine
(ncclAllReduce of the 5 MB reduce block enqueued by alpine_run on the ctx stream, every iteration) against the same engine without it,
at the 25 000-cell shard of cfg3, rounds interleaved.  A lower bound on the exposure `a` of DESIGN.md 5: the launch + the in-place
kernel of a one-rank collective, no wire.

    python tools/rccl_world1_overhead.py [--cells 25000] [--rounds 6] [--steps 200]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=25000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS["cfg3"])
    G, N, ku, kcov = wl["genes"], a.cells, wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    lev = [2] * len(kcov)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], lev)
    engines = []
    for with_comm in (False, True):
        eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=lev, lam=[1e3] * len(kcov),
                                  orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
        for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
            torch.cuda.synchronize()
            eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
            eng.synchronize()
        eng.finalize_X()
        for i in range(len(kcov)):
            eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
        if with_comm:
            eng.comm_init(_native.comm_unique_id(), 1, 0)
            assert eng.comm_count() == (1, 0)
        engines.append(("RCCL communicator of one rank" if with_comm else "no communicator", eng, []))
    for rnd in range(a.rounds):
        for name, eng, ms in engines:
            eng.set_factors(W0, H0, B0)
            eng.run(5, with_loss=True)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run(a.steps, with_loss=True)
            eng.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0) / a.steps)
    base = None
    for name, eng, ms in engines:
        med = float(np.median(ms))
        base = med if base is None else base
        print(f"{name:32s}: {med:.4f} ms per iteration (min {min(ms):.4f}; {1e3 * (med - base):+.1f} us), reduce block {eng.info().reduce_block_floats * 4 / 1e6:.2f} MB")
        eng.close()


if __name__ == "__main__":
    main()
