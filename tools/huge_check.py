"""Maximum-size check: BASELINE configs[3] (20 000 genes x 1 000 000 cells, K = 100 + [5]) WHOLE on one MI355X.

The matrix has 2e10 elements (> 2^32): every index computation that touches X must be 64-bit.  float32 keeps two
copies (160 GB of the 288 GB of HBM3E), the exact split one bf16 plane per copy (80 GB).  Size-independent
properties, as in tests/test_gpu_fullsize.py:
  * XH^T with H == 1 equals the row sums of X (exact while they stay below 2^24, else to 1e-6),
  * the W^TX side through one Frobenius iteration from W == 1, H == 1 (a weighted column checksum per cell),
  * trace-form loss row == direct float64 evaluation of ||X - WH||^2 (float32 engine),
  * losses finite, factors non-negative; split vs float32 loss rows agree to 5e-5.
Run on the GPU box:  python tools/huge_check.py [--cells 1000000] [--modes split,f32]
About 30 s of device time and 152 GiB of HBM; tests/test_gpu_maxsize.py runs it."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=20_000)
    ap.add_argument("--ku", type=int, default=100)
    ap.add_argument("--modes", default="split,x3,f32")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--out", default="")
    args = ap.parse_args(argv)
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors

    G, N, ku, kc = args.genes, args.cells, args.ku, [5]
    K = ku + sum(kc)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    lab = rng.integers(0, 2, size=N)
    Y = np.zeros((2, N), dtype=np.float32)
    Y[lab, np.arange(N)] = 1.0
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kc + [ku], [2])
    report = {"genes": G, "cells": N, "K": K, "elements": G * N, "modes": {}}
    loss_rows = {}
    for mode in args.modes.split(","):
        t0 = time.time()
        res = {}
        for loss_type, lam in (("kl-divergence", 1e3), ("frobenius", 0.0)):
            eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kc, cov_levels=[2], lam=[lam],
                                      loss_type=loss_type, x_dtype=mode)
            rowsum = torch.zeros(G, dtype=torch.float64, device=dev)
            for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
                torch.cuda.synchronize()
                eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
                eng.synchronize()
                if loss_type == "kl-divergence":
                    rowsum.add_(chunk.sum(dim=0, dtype=torch.float64))
                del chunk
            eng.finalize_X()
            eng.upload_Y(0, Y)
            info = eng.info()
            KP = info.k_padded
            ones_W = np.ones((G, K), np.float32)
            ones_H = np.ones((K, N), np.float32)
            Bh = [np.full((2, kc[0]), 0.5, np.float32)]
            if loss_type == "kl-divergence":
                res["device_GiB"] = info.device_bytes / 2 ** 30
                res["ingest_s"] = time.time() - t0
                eng.set_factors(ones_W, ones_H, Bh)
                eng.iter_begin()
                XHt = eng.read_buffer(_native.BUF_REDUCE_BLOCK, 0, info.genes_padded * KP).reshape(info.genes_padded, KP)
                rs = rowsum.cpu().numpy()
                want = np.repeat(rs[:, None], K, axis=1)
                got = XHt[:G, :K].astype(np.float64)
                res["rowsum_max"] = float(rs.max())
                res["xht_exact"] = bool(np.array_equal(got, want))
                res["xht_max_rel"] = float(np.max(np.abs(got - want) / np.maximum(want, 1.0)))
                assert res["xht_exact"] or res["xht_max_rel"] < 1e-6, res
                assert not XHt[G:].any() and not XHt[:, K:].any()
                eng.set_factors(W0, H0, B0)
                eng.synchronize()
                t1 = time.time()
                eng.run(args.iters, with_loss=True)
                eng.synchronize()
                res["ms_per_iter"] = 1e3 * (time.time() - t1) / (args.iters + 1)      # + the loss-only pass
                losses = eng.losses()
                W, H, _ = eng.get_factors()
                assert np.isfinite(losses).all() and (W >= 0).all() and (H >= 0).all() and np.isfinite(H).all()
                assert (np.diff(losses[:, 1]) < 0).all(), losses[:, 1]
                res["recon_loss"] = [float(v) for v in losses[:, 1]]
                if mode in ("f32", "x3"):
                    direct = eng.eval_recon_direct()
                    res["direct_vs_trace_rel"] = abs(losses[-1, 1] - direct) / direct
                    assert res["direct_vs_trace_rel"] < 2e-5, res
                loss_rows[mode] = losses
            else:
                # W == 1, H == 1, one Frobenius iteration without guidance: W <- rowsum / (N K) in every column, then
                # H[k][n] <- (sum_g w_g X_gn) / (K sum_g w_g^2): a weighted checksum of column n of X through W^TX
                eng.set_factors(ones_W, ones_H, Bh)
                eng.run(1, with_loss=False)
                W1, H1, _ = eng.get_factors()
                wt = torch.tensor(W1[:, 0], dtype=torch.float64, device=dev)
                wtx = H1[K - 1].astype(np.float64) * (K * float((wt * wt).sum()))
                want = torch.empty(N, dtype=torch.float64, device=dev)
                for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
                    want[off:off + chunk.shape[0]] = chunk.double() @ wt
                    del chunk
                want = want.cpu().numpy()
                res["wtx_max_rel"] = float(np.max(np.abs(wtx - want) / np.maximum(np.abs(want), 1e-30)))
                assert res["wtx_max_rel"] < 1e-5, res
            eng.close()
            torch.cuda.empty_cache()
        res["wall_s"] = time.time() - t0
        report["modes"][mode] = res
        print(mode, json.dumps(res), flush=True)
    for other in ("split", "x3"):
        if "f32" in loss_rows and other in loss_rows:
            d = np.max(np.abs(loss_rows[other] - loss_rows["f32"]) / np.abs(loss_rows["f32"]))
            report[f"{other}_vs_f32_loss_rows_max_rel"] = float(d)
            assert d < 5e-5, d
    print(json.dumps(report))
    if args.out:
        Path(args.out).write_text(json.dumps(report, indent=1))
    return report


if __name__ == "__main__":
    main()
