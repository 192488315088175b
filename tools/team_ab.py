"""Teams of sweep workgroups (SweepGeom::gw) x the even/odd span bias, in the LIBRARY's own iteration: ONE engine (one physical
placement of X: two engines differ by +-2 % in sweep time from placement alone), the division of both sweeps switched between
interleaved rounds with alpine_debug_set_team_width / alpine_debug_set_xcd_bias.  Variant "w:b" = team width w (0 = the library's
choice, 1 = none) and bias b per mille ("p" = what the placement probe chose).

    python tools/team_ab.py [--workload cfg4 --cells 125000] [--x-scale 1.0] [--variants 1:p,0:p,0:0,8:0,4:0] [--rounds 4] [--steps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--dtype", default="x3", choices=["x3", "bf16", "split"])
    ap.add_argument("--variants", default="1:p,0:p,0:0,8:0,4:0,2:0")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS[a.workload])
    G, N, ku, kcov = wl["genes"], a.cells or wl["cells"], wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    lev = [2] * len(kcov)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], lev)
    eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=lev, lam=[1e3] * len(kcov),
                              orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype=a.dtype)
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        if a.x_scale != 1.0:
            chunk = (chunk * a.x_scale).contiguous()
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(len(kcov)):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    probe_bias = eng.info().xcd_bias_per_mille
    biases = []
    for v in a.variants.split(","):
        w, b = v.split(":")
        bb = probe_bias if b == "p" else int(b)
        try:
            eng.debug_set_team_width(int(w))
            eng.debug_set_xcd_bias(bb)
            biases.append((int(w), bb, v))
        except _native.AlpineNativeError as e:
            print(f"  variant {v}: skipped ({e})")
    res = {b: {"iter": [], "xht": [], "wtx": [], "tw": None} for b in biases}
    for rnd in range(a.rounds):
        for b in biases:
            eng.debug_set_team_width(b[0])
            eng.debug_set_xcd_bias(b[1])
            res[b]["tw"] = (eng.info().team_width_a, eng.info().team_width_b)
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
    print(f"{a.workload}: cells {N}, x_scale {a.x_scale}, x3_wide {info.x3_wide}, multi-plane fraction {info.x_multi_plane_fraction}; probe bias {probe_bias}; "
          f"one engine, {a.rounds} interleaved rounds of {a.steps} iterations per variant")
    base = np.median(res[biases[0]]["iter"])
    for b in biases:
        r = res[b]
        print(f"  {b[2]:>6s} (teams {r['tw']}, bias {b[1]:+4d}): iteration {np.median(r['iter']):.4f} ms ({100 * (np.median(r['iter']) / base - 1):+.2f} %), "
              f"XH^T sweep {np.median(r['xht']):.4f} ms, W^TX sweep {np.median(r['wtx']):.4f} ms; rounds {[round(x, 4) for x in r['iter']]}")
    eng.close()


if __name__ == "__main__":
    main()
