"""In-kernel timelines from the stamped throwaway build (python alpine_amd/build.py --stamps -> libalpine_hip_stamps.so,
-DALPINE_STAMPS; the product library carries no stamps).  s_memrealtime ticks are 10 ns.

    python tools/stamps.py sweep [--cells 25000] [--x-scale 1.0] [--env NAME=VALUE ...]
        per workgroup of the LAST sweep launch (the W^TX sweep): start skew, time to the first stage, streaming time,
        flush time (summed over the workgroup's pieces), end skew; grouped by XCD
    python tools/stamps.py hupdate [--cells 25000]
        median duration of every phase of h_update_mfma_kernel over the blocks of the last launch"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "alpine_amd", "libalpine_hip_stamps.so")
HU_PHASES = ["fills (Y, 2W^TW, guided tables) + barrier", "H issue + pieces + H to C/D", "2W^TW.H on the MFMA", "guided terms + update",
             "tile store", "barrier + H H^T partial", "covariate statistics"]


def engine(cells, x_scale, env, workload="cfg3"):
    os.environ["ALPINE_HIP_LIBRARY"] = LIB
    for kv in env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    sys.path.insert(0, REPO)
    import torch
    import bench
    from alpine_amd import _native
    from alpine_amd.datasets import synth_counts_device_chunks
    from alpine_amd.model import draw_initial_factors
    wl = dict(bench.WORKLOADS[workload])
    G, N, ku, kcov = wl["genes"], cells, wl["ku"], wl["kcov"]
    dev = torch.device("cuda", 0)
    lev = [2] * len(kcov)
    W0, H0, B0 = draw_initial_factors(42, 1e-6, G, N, kcov + [ku], lev)
    eng = _native.NativeShard(n_genes=G, n_cells=N, n_components=ku, cov_components=kcov, cov_levels=lev, lam=[1e3] * len(kcov),
                              orth_W=wl["orth_W"], alpha_W=wl["alpha_W"], l1_ratio_W=wl["l1_ratio_W"], x_dtype="x3")
    for off, chunk in synth_counts_device_chunks(N, G, rank=ku, seed=0, device=dev, chunk_cells=8192):
        if x_scale != 1.0:
            chunk = (chunk * x_scale).contiguous()
        torch.cuda.synchronize()
        eng.upload_X_device(chunk.data_ptr(), chunk.stride(0), chunk.shape[0], _native.X_CELLS_BY_GENES, off)
        eng.synchronize()
    eng.finalize_X()
    for i in range(len(kcov)):
        eng.upload_Y(i, bench.labels_onehot(N, seed=1 + i))
    eng.set_factors(W0, H0, B0)
    return eng, _native


def pc(v):
    return " / ".join(f"{np.percentile(v, q):.1f}" for q in (0, 10, 50, 90, 100))


def sweep(a):
    eng, nat = engine(a.cells, a.x_scale, a.env, a.workload)
    eng.run(20, with_loss=True)
    eng.synchronize()
    info = eng.info()
    lib = nat.load()
    nwg = info.grid_b
    buf = (C.c_ulonglong * (8 * 2048))()
    lib.alpine_debug_read_sweep_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.alpine_debug_read_sweep_stamps(buf, 8 * 2048) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8)[:nwg].astype(np.int64)
    t0 = raw[:, 0].min()
    start = (raw[:, 0] - t0) / 100.0
    first = (raw[:, 1] - raw[:, 0]) / 100.0
    life = (raw[:, 3] - raw[:, 0]) / 100.0
    flush = raw[:, 4] / 100.0
    segs = raw[:, 5]
    last_flush = (raw[:, 3] - raw[:, 2]) / 100.0
    end = (raw[:, 3] - t0) / 100.0
    xcc = raw[:, 6]
    print(f"cells {a.cells}, x_scale {a.x_scale}, env {a.env}: last W^TX sweep, {nwg} workgroups, spans of {info.span_rows_b} rows x {info.spans_per_workgroup_b}; x3_wide={info.x3_wide}")
    print(f"  (min / p10 / p50 / p90 / max, us)")
    print(f"  start offset            {pc(start)}")
    print(f"  start -> first stage    {pc(first)}     (panel stage 0 + X ring loads issued + barrier)")
    print(f"  lifetime                {pc(life)}")
    print(f"  flush, all pieces       {pc(flush)}     pieces per workgroup: {np.bincount(segs)[1:].tolist()} with 1, 2, ... pieces")
    print(f"  flush, last piece       {pc(last_flush)}")
    print(f"  streaming = lifetime - first - flush   {pc(life - first - flush)}")
    print(f"  end offset              {pc(end)}     -> kernel span by stamps {end.max():.1f} us; mean lifetime {life.mean():.1f} us")
    one = segs == 1
    if one.any() and (~one).any():
        print(f"  workgroups with one piece end at {pc(end[one])}; with two or more at {pc(end[~one])}")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  XCD {x}: {int(m.sum())} workgroups, lifetime median {np.median(life[m]):.1f}, end median {np.median(end[m]):.1f} (max {end[m].max():.1f})")
    print("  XCC id of workgroups 0..23:", xcc[:24].tolist())
    print(f"  alpine_info: xcd_bias_per_mille {info.xcd_bias_per_mille}, placement probe saw workgroup 0 on XCC {info.xcc_of_workgroup0}")
    hist = (C.c_uint * (4 + 4096))()
    lib.alpine_debug_read_sweep_hist.argtypes = [C.c_void_p, C.c_int]
    assert lib.alpine_debug_read_sweep_hist(hist, 4 + 4096) == 0
    nl = int(hist[0])
    seq = [(int(hist[4 + i]) & 15, int(hist[4 + i]) >> 16) for i in range(max(0, nl - 24), nl)]
    print(f"  sweep launches so far {nl}; (XCC id of workgroup 0, grid) of the last {len(seq)} sweep launches: {seq}")
    wg = np.arange(nwg)
    for r in range(8):
        m = wg % 8 == r
        print(f"  blockIdx % 8 == {r}: XCC ids {sorted(set(xcc[m].tolist()))}, lifetime median {np.median(life[m]):.1f}, rows/us {np.median(info.span_rows_b * info.spans_per_workgroup_b / life[m]):.2f}")
    eng.close()


def hupdate(a):
    eng, nat = engine(a.cells, 1.0, a.env, a.workload)
    eng.run(10, with_loss=True)
    eng.synchronize()
    lib = nat.load()
    nb = (a.cells + 127) // 128
    buf = (C.c_ulonglong * (16 * 8192))()
    lib.alpine_debug_read_hu_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.alpine_debug_read_hu_stamps(buf, 16 * 8192) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16)[:min(nb, 8192), :8].astype(np.int64)
    d = np.diff(st, axis=1) / 100.0
    print(f"cells {a.cells}: {nb} blocks, kernel span (first block start -> last block end) {(st[:, 7].max() - st[:, 0].min()) / 100.0:.1f} us")
    for i, n in enumerate(HU_PHASES):
        print(f"  {n:44s} median {np.median(d[:, i]):6.2f} us   p90 {np.percentile(d[:, i], 90):6.2f}")
    print(f"  block total: median {np.median((st[:, 7] - st[:, 0]) / 100.0):.2f} us; start skew p90 - p10 "
          f"{(np.percentile(st[:, 0], 90) - np.percentile(st[:, 0], 10)) / 100.0:.2f} us")
    eng.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["sweep", "hupdate"])
    ap.add_argument("--cells", type=int, default=25000)
    ap.add_argument("--x-scale", type=float, default=1.0)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--workload", default="cfg3", help="model of this bench.py workload (cfg4: K = 105)")
    a = ap.parse_args()
    if not os.path.exists(LIB):
        sys.exit("build the stamped library first: python alpine_amd/build.py --stamps")
    {"sweep": sweep, "hupdate": hupdate}[a.what](a)
