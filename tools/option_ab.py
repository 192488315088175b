"""A/B of options that must be set BEFORE alpine_finalize_X (x3_two_wave, x3_variant, x3_narrow, wide_one_pass) in the library's own
iteration: one engine per variant (the options pick kernels and layouts at finalize), rounds interleaved across the engines.  Two
engines differ by +-2 % in sweep time from the physical placement of X alone, so every variant is created TWICE (engines A, B, A', B')
and the spread between a variant's two engines is printed beside the difference between variants.
Variant = "opt=val+opt=val" (empty = the library's defaults), optionally "@w" = team width w for both sweeps.

    python tools/option_ab.py --workload cfg4 --cells 125000 [--x-scale 1.37] --variants "x3_two_wave=0,," [--rounds 4] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--dtype", default="x3", choices=["x3", "f32", "bf16", "split"])
    ap.add_argument("--variants", default="x3_two_wave=0,")
    ap.add_argument("--copies", type=int, default=2)
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
    engines = []
    for copy in range(a.copies):
        for v in a.variants.split(","):
            opts, _, tw = v.partition("@")
            eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=lev, lam=[1e3] * len(kcov),
                                      orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype=a.dtype)
            for kv in filter(None, opts.split("+")):
                k, val = kv.split("=")
                eng.debug_set_option(k, int(val))
            for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
                if a.x_scale != 1.0:
                    chunk = (chunk * a.x_scale).contiguous()
                torch.cuda.synchronize()
                eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
                eng.synchronize()
            eng.finalize_X()
            for i in range(len(kcov)):
                eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
            if tw:
                eng.debug_set_team_width(int(tw))
            engines.append((f"{v or '(defaults)'} #{copy}", eng, {"iter": [], "xht": [], "wtx": []}))
    for rnd in range(a.rounds):
        for name, eng, r in engines:
            eng.set_factors(W0, H0, B0)
            eng.run(3, with_loss=True)
            eng.set_profiling(True)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run(a.steps, with_loss=True)
            eng.synchronize()
            r["iter"].append(1e3 * (time.perf_counter() - t0) / a.steps)
            ma, na = eng.kernel_time(_native.KERNEL_SWEEP_XHT)
            mb, nb = eng.kernel_time(_native.KERNEL_SWEEP_WTX)
            r["xht"].append(ma / max(1, na))
            r["wtx"].append(mb / max(1, nb))
            eng.set_profiling(False)
    print(f"{a.workload}: cells {N}, x_scale {a.x_scale}; {len(engines)} engines, {a.rounds} interleaved rounds of {a.steps} iterations each")
    for name, eng, r in engines:
        info = eng.info()
        med = lambda k: float(np.median(r[k]))
        print(f"  {name:36s} waves/SIMD {info.sweep_waves_per_simd} x3_wide {info.x3_wide} teams ({info.team_width_a}, {info.team_width_b}) bias {info.xcd_bias_per_mille:+d}: "
              f"iteration {med('iter'):.4f} ms (min {min(r['iter']):.4f})  XH^T {med('xht'):.4f}  W^TX {med('wtx'):.4f}  multi-plane {info.x_multi_plane_fraction:.3f}")
        eng.close()


if __name__ == "__main__":
    main()
