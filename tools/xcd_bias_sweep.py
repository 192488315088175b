"""Does giving the sweep workgroups of the slower XCDs fewer rows pay?  ONE engine (one physical placement of X: two
engines differ by +-2 % in sweep time from placement alone), the division of both sweeps switched between rounds with
alpine_debug_set_xcd_bias: even workgroups get spans of L (1 + b), odd ones L (1 - b).  On MI355X workgroup w runs on XCC
(w - 1) mod 8 and the odd XCCs stream this access pattern 6-7 % slower (tools/stamps.py sweep), i.e. EVEN workgroups are the
slow ones: b < 0 gives them less.

    python tools/xcd_bias_sweep.py [--cells 200000] [--x-scale 1.0] [--bias 0,-10,-20,-30,-40,20] [--rounds 4] [--steps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=200000)
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--bias", default="0,-10,-20,-30,-40,20")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
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
        if a.x_scale != 1.0:
            chunk = (chunk * a.x_scale).contiguous()
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(2):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    biases = []
    for b in a.bias.split(","):
        try:
            eng.debug_set_xcd_bias(int(b))
            biases.append(int(b))
        except _native.AlpineNativeError as e:      # e.g. the longer span would exceed the accumulation cap -> twice the pieces
            print(f"  bias {b}: skipped ({e})")
    res = {b: {"iter": [], "xht": [], "wtx": []} for b in biases}
    for rnd in range(a.rounds):
        for b in biases:
            eng.debug_set_xcd_bias(b)
            eng.set_factors(W0, H0, B0)
            eng.run(3, with_loss=True)
            eng.set_profiling(True)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run(a.steps, with_loss=True)
            eng.synchronize()
            res[b]["iter"].append(1e3 * (time.perf_counter() - t0) / a.steps)
            ma, na = eng.kernel_time(_native.KERNEL_SWEEP_XHT)
            mb, nb = eng.kernel_time(_native.KERNEL_SWEEP_WTX)
            res[b]["xht"].append(ma / max(1, na))
            res[b]["wtx"].append(mb / max(1, nb))
            eng.set_profiling(False)
    info = eng.info()
    print(f"cells {N}, x_scale {a.x_scale}, x3_wide {info.x3_wide}; one engine, {a.rounds} interleaved rounds of {a.steps} iterations per bias")
    base = np.median(res[biases[0]]["iter"])
    for b in biases:
        r = res[b]
        print(f"  bias {b:+4d} per mille: iteration {np.median(r['iter']):.4f} ms ({100 * (np.median(r['iter']) / base - 1):+.2f} %), "
              f"XH^T sweep {np.median(r['xht']):.4f} ms, W^TX sweep {np.median(r['wtx']):.4f} ms; rounds {[round(x, 4) for x in r['iter']]}")
    eng.close()


if __name__ == "__main__":
    main()
