"""Parity AT THE HEADLINE SIZE, by hand on a GPU box: BASELINE config 3 (20 000 genes x 200 000 cells, K = 50 + [5, 5],
regularisers on) for a few iterations on the GPU in every float32-grade sweep mode against the oracle's fused iteration
run in FLOAT64 on the host from the same initial factors (the float32 CPU oracle beside it, for scale).

By hand at cfg3 (32 GB of float64 X on the host, ~1e12 flops per float64 iteration); the cfg2-size cut (20 000 x 50 000,
3 iterations: 8 GB, seconds) is collected by pytest: tests/test_gpu_float64_arbiter.py calls compare().  Usage:

    python tests/fullsize_vs_float64.py [--workload cfg3] [--iters 3] [--cells N] [--x-scale 1.0] [--out file.json]

Prints and writes: relative Frobenius error of W, H and every B_i after `iters` iterations, and the loss rows against the
float64 run's own rows."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import bench                                       # noqa: E402
from _golden import rel_fro                        # noqa: E402
from oracle import alpine_oracle as orc            # noqa: E402


def compare(workload="cfg3", iters=3, cells=0, x_scale=1.0, modes=None, with_float32_oracle=True, log=print):
    """Run `iters` iterations of `workload` on the GPU in every mode of `modes` and in float64 on the host from the same
    initial factors; returns {"workload", "iters", "cpu_seconds", "float64_loss_rows", "vs_float64": {name: errors}}."""
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    wl = dict(bench.WORKLOADS[workload])
    G, N, ku, kcov = wl["genes"], cells or wl["cells"], wl["ku"], wl["kcov"]
    levels, lam = [2] * len(kcov), [1e3] * len(kcov)
    dev = torch.device("cuda", 0)
    p = orc.OracleParams(n_components=ku, n_covariate_components=kcov, lam=lam, orth_W=wl["orth_W"], alpha_W=wl["alpha_W"],
                         l1_ratio_W=wl["l1_ratio_W"], loss_type="kl-divergence", random_state=42)
    Ys = [bench.labels_onehot(N, seed=1 + i) for i in range(len(kcov))]            # C x N

    # the matrix, once: generated on the device (bench.py's generator), kept on the host for the oracle
    t0 = time.perf_counter()
    X = np.empty((N, G), dtype=np.float32)
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        if x_scale != 1.0:
            chunk = chunk * x_scale
        X[off:off + chunk.shape[0]] = chunk.cpu().numpy()
    log(f"X {X.shape} generated in {time.perf_counter() - t0:.1f} s; mean {float(X[:4096].mean()):.3f}")

    s32 = orc.init_factors(p, np.ascontiguousarray(X.T), [y.T for y in Ys])
    W0, H0, B0 = s32.W.numpy().copy(), s32.H.numpy().copy(), [b.numpy().copy() for b in s32.Bs]
    if not with_float32_oracle:
        s32 = None

    if modes is None:
        modes = ("x3", "f32") + (("split",) if x_scale == 1.0 else ())
    res = {"workload": f"{workload}: {G} genes x {N} cells, K={ku}+{kcov}, x_scale={x_scale}", "iters": iters, "modes": {}}
    for mode in modes:
        eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=levels, lam=lam,
                                  orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype=mode)
        for c0 in range(0, N, 16384):
            eng.upload_X_host(X[c0:c0 + 16384], cell0=c0)
        eng.finalize_X()
        for i, y in enumerate(Ys):
            eng.upload_Y(i, y)
        eng.set_factors(W0, H0, B0)
        eng.run(iters, with_loss=True)
        W, H, Bs = eng.get_factors()
        res["modes"][mode] = dict(W=W, H=H, Bs=Bs, losses=eng.losses(), x3_wide=int(eng.info().x3_wide))
        eng.close()
        log(f"GPU {mode}: done")

    t32 = None
    if s32 is not None:
        t0 = time.perf_counter()
        orc.fit_fused(p, s32, iters, with_loss=True)
        t32 = time.perf_counter() - t0
        log(f"float32 CPU oracle: {iters} iterations in {t32:.1f} s")

    # float64 arbiter (the oracle's fused iteration with every tensor in float64)
    t0 = time.perf_counter()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        s64 = orc.OracleState(torch.tensor(X.T, dtype=torch.float64), [torch.tensor(y, dtype=torch.float64) for y in Ys],
                              torch.tensor(W0, dtype=torch.float64), torch.tensor(H0, dtype=torch.float64),
                              [torch.tensor(b, dtype=torch.float64) for b in B0])
        del X
        orc.fit_fused(p, s64, iters, with_loss=True)
    finally:
        torch.set_default_dtype(old)
    t64 = time.perf_counter() - t0
    log(f"float64 CPU run: {iters} iterations in {t64:.1f} s")
    W64, H64, B64 = s64.W.numpy(), s64.H.numpy(), [b.numpy() for b in s64.Bs]
    L64 = np.array(s64.losses)

    out = {"workload": res["workload"], "iters": iters, "cpu_seconds": {"float32_oracle": t32, "float64": t64},
           "float64_loss_rows": L64.tolist(), "vs_float64": {}}

    def row(name, W, H, Bs, L):
        d = dict(W=rel_fro(W, W64), H=rel_fro(H, H64), B=[rel_fro(b, b64) for b, b64 in zip(Bs, B64)],
                 loss_total_rel=float(np.max(np.abs(L[:, 0] - L64[:, 0]) / np.abs(L64[:, 0]))),
                 loss_recon_rel=float(np.max(np.abs(L[:, 1] - L64[:, 1]) / np.abs(L64[:, 1]))))
        out["vs_float64"][name] = d
        log(f"{name:>22}: W {d['W']:.2e}  H {d['H']:.2e}  B {['%.1e' % b for b in d['B']]}  loss rows: total {d['loss_total_rel']:.1e} recon {d['loss_recon_rel']:.1e}")

    if s32 is not None:
        row("float32 CPU oracle", s32.W.numpy(), s32.H.numpy(), [b.numpy() for b in s32.Bs], np.array(s32.losses))
    for mode, r in res["modes"].items():
        row(f"GPU {mode}" + (" (x3w)" if r["x3_wide"] else ""), r["W"], r["H"], r["Bs"], r["losses"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--cells", type=int, default=0, help="0 = the workload's own (cfg4: 1 000 000 -- pass its per-GPU share, 125000)")
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = compare(a.workload, a.iters, a.cells, a.x_scale, log=lambda m: print(m, flush=True))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
