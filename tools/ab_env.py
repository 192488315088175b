"""(Round 4: the production library takes these knobs as explicit alpine_debug_set_option calls; the environment variables below are read
by the DIAGNOSTICS build only -- run with ALPINE_HIP_LIBRARY=alpine_amd/libalpine_hip_diag.so.)

A/B of library knobs in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): one engine per variant
(each created under its own environment setting -- the library reads its knobs once, in alpine_create), same synthetic
input, R rounds of `steps` iterations each, alternating between the variants.

    python tools/ab_env.py --var ALPINE_HIP_X3_VARIANT=0 --var ALPINE_HIP_X3_VARIANT=1 [--x-scale 0.3712345] [--cells 25000]

Prints per variant: median / min ms per iteration, median ms per sweep launch (hipEvents inside the library), final loss
row (variants that only change the schedule must agree bitwise)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--var", action="append", required=True, help="NAME=VALUE[,NAME=VALUE...] -- one engine per --var")
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--dtype", default="x3")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS[args.workload])
    if args.cells:
        wl["cells"] = args.cells
    G, N, ku, kcov = wl["genes"], wl["cells"], wl["ku"], wl["kcov"]
    levels = [2] * len(kcov)
    dev = torch.device("cuda", 0)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], levels)
    Ys = [bench.labels_onehot(N, seed=1 + i) for i in range(len(kcov))]
    engines = []
    for spec in args.var:
        pairs = [p.split("=", 1) for p in spec.split(",") if p and "=" in p]
        old = {k: os.environ.get(k) for k, _ in pairs}
        for k, v in pairs:
            os.environ[k] = v
        eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=levels, lam=[1e3] * len(kcov),
                                  orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype=args.dtype)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
            if args.x_scale != 1.0:
                chunk = (chunk * args.x_scale).contiguous()
            torch.cuda.synchronize()
            eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
            eng.synchronize()
            del chunk
        eng.finalize_X()
        torch.cuda.empty_cache()
        for i in range(len(kcov)):
            eng.upload_Y(i, Ys[i])
        eng.set_factors(W0, H0, B0)
        eng.run(5, with_loss=True)
        eng.synchronize()
        engines.append((spec, eng, [], []))
    for _ in range(args.rounds):
        for spec, eng, its, sweeps in engines:
            eng.set_profiling(True)
            eng.synchronize()
            t0 = time.perf_counter()
            eng.run(args.steps, with_loss=True)
            eng.synchronize()
            its.append(1e3 * (time.perf_counter() - t0) / args.steps)
            a, na = eng.kernel_time(_native.KERNEL_SWEEP_XHT)
            b, nb = eng.kernel_time(_native.KERNEL_SWEEP_WTX)
            sweeps.append((a + b) / max(1, na + nb))
            eng.set_profiling(False)
    out = []
    for spec, eng, its, sweeps in engines:
        # all engines ran the same number of iterations from the same start: schedule-only variants agree bitwise
        out.append({"var": spec, "ms_per_iter_median": float(np.median(its)), "ms_per_iter_min": float(np.min(its)),
                    "ms_per_sweep_median": float(np.median(sweeps)), "rounds_ms_per_iter": [round(x, 4) for x in its],
                    "last_loss_row": eng.losses()[-1].tolist()})
        eng.close()
    print(json.dumps({"workload": args.workload, "cells": N, "x_scale": args.x_scale, "dtype": args.dtype, "steps": args.steps, "variants": out}))


if __name__ == "__main__":
    main()
