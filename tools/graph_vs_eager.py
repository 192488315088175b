"""Would a hipGraph shorten the shard-size iteration?  The same 2 x n steady-state MU iterations (no loss rows) submitted
eagerly (alpine_run: six launches per iteration enqueued by the library's C loop) and replayed from a hipGraph of two
iterations (alpine_debug_run_graph), interleaved rounds on ONE engine; also how far the host runs ahead of the device in the
eager loop (time for alpine_run to RETURN vs time until the device is done).

    python tools/graph_vs_eager.py [--cells 25000] [--pairs 100] [--rounds 6]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=25000)
    ap.add_argument("--pairs", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS["cfg3"])
    G, N, ku, kcov = wl["genes"], a.cells, wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], [2, 2])
    eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=[2, 2], lam=[1e3, 1e3],
                              orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(2):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    eng.set_factors(W0, H0, B0)
    eng.run(10, with_loss=False)
    eng.synchronize()
    n_it = 2 * a.pairs
    eager, graph, enq = [], [], []
    for _ in range(a.rounds):
        eng.synchronize()
        t0 = time.perf_counter()
        eng.run(n_it + 2, with_loss=False)               # (+2: alpine_debug_run_graph runs two eager iterations before its capture)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        eager.append(1e3 * (t2 - t0) / (n_it + 2))
        enq.append((t1 - t0) / (t2 - t0))
        t0 = time.perf_counter()
        eng.debug_run_graph(a.pairs)                     # synchronises itself; includes capture + instantiate (once per call)
        graph.append(1e3 * (time.perf_counter() - t0) / (n_it + 2))
    W1, H1, _ = eng.get_factors()
    print(f"cells {N}: {n_it} iterations per leg, {a.rounds} interleaved rounds on one engine")
    print(f"  eager  (alpine_run):             {np.median(eager):.4f} ms per iteration (min {np.min(eager):.4f}); the host has enqueued "
          f"everything after {100 * np.median(enq):.0f} % of the leg's wall time")
    print(f"  hipGraph of two iterations:      {np.median(graph):.4f} ms per iteration (min {np.min(graph):.4f}), capture + instantiate included")
    print(f"  factors finite: {bool(np.isfinite(W1).all() and np.isfinite(H1).all())}")
    eng.close()


if __name__ == "__main__":
    main()
