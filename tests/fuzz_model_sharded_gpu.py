"""Time-boxed randomised campaign of the SHARDED drop-in API on a one-GPU box (by hand; not collected by pytest):

    python tests/fuzz_model_sharded_gpu.py --seconds 600 [--ranks 2|3] [--seed0 S] [--stub]

R rank processes share cuda:0.  Per random case (tests/fuzz_model_gpu.py:make_case, the same on every rank) every rank runs
`ALPINE(shard_cells=True | "local", ...).fit(...)` -- all ranks hold the whole adata, or each passes only its own uneven block of
cells -- with the all-reduce carried by torch.distributed over gloo, or (--stub) by the library's own communicator over the
shared-memory stand-in for RCCL (tests/stub_rccl); then `transform` of its own cells; then the same problem UNSHARDED in the same
process.  W, B and the loss history must agree between the two (replicated), H and the transformed H on the rank's own columns.
What this covers beyond tests/fuzz_sharded_gpu.py (C ABI level): the host logic of the sharded model -- category merging across ranks,
the shared index streams of mini-batch / weighted epochs, the generator position of `transform`, carrier selection."""
import argparse
import os
import socket
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, seconds, seed0, stub, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fuzz_model_gpu as fm
    from _golden import rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    t0 = time.perf_counter()
    n, bad = 0, None
    while True:
        go = [time.perf_counter() - t0 < seconds and bad is None]
        dist.broadcast_object_list(go, src=0)
        if not go[0]:
            break
        seed = seed0 + n
        n += 1
        params, X, obs, bs, sampling, T, x_dtype, n_t, kind = fm.make_case(seed)
        params.pop("keep_resident")
        N = X.shape[0]
        rng = np.random.default_rng(seed + 99)
        local = bool(rng.random() < 0.5)
        if N < 16 * world:
            continue
        if params["use_als"] and (bs is not None or sampling == "weighted"):
            bs, sampling = None, "random"                         # block-coordinate mini-batches are single-device
        if x_dtype == "split" and bs is not None:
            x_dtype = "auto"
        keys = list(obs.columns)
        cuts = [0] + sorted(int(v) for v in rng.choice(np.arange(8, N - 8), size=world - 1, replace=False)) + [N]
        if min(b - a for a, b in zip(cuts, cuts[1:])) < 4:
            continue
        a, b = (cuts[rank], cuts[rank + 1]) if local else (0, N)
        tag = (f"seed {seed} ranks={world} {'local' if local else 'replicated'} {'stub' if stub else 'gloo'} G={X.shape[1]} N={N} "
               f"K={params['n_components']}+{params['n_covariate_components']} {params['loss_type'][:2]} als={int(params['use_als'])} bs={bs} {sampling} T={T} x={x_dtype}")
        try:
            kw = dict(device="cuda:0", x_dtype=x_dtype, shard_comm="native" if stub else "torch", **params)
            ad_s = MiniAnnData(X[a:b].copy(), obs.iloc[a:b].reset_index(drop=True))
            m_s = ALPINE(shard_cells="local" if local else True, **kw).fit(ad_s, covariate_keys=keys, batch_size=bs, max_iter=T, sampling_method=sampling)
            at_s = MiniAnnData(X[a:b].copy(), obs.iloc[a:b].reset_index(drop=True))
            if local:
                m_s.transform(at_s, n_iter=2)
            kw1 = dict(kw, shard_comm="auto")
            ad_1 = MiniAnnData(X.copy(), obs.copy())
            m_1 = ALPINE(shard_cells=False, **kw1).fit(ad_1, covariate_keys=keys, batch_size=bs, max_iter=T, sampling_method=sampling)
            at_1 = MiniAnnData(X.copy(), obs.copy())
            if local:
                m_1.transform(at_1, n_iter=2)
            cat = lambda m: (np.concatenate(m.matrices["Ws"], axis=1), np.concatenate(m.matrices["Hs"], axis=0))      # noqa: E731
            (Ws, Hs), (W1, H1) = cat(m_s), cat(m_1)
            H1own = H1[:, a:b] if local else H1
            if not (np.isfinite(W1).all() and np.isfinite(H1).all()):
                assert np.array_equal(np.isnan(Ws), np.isnan(W1)) and np.array_equal(np.isnan(Hs), np.isnan(H1own)), "NaN pattern"
            else:
                tol = 5e-5 * max(2, T)
                eW, eH = rel_fro(Ws, W1), rel_fro(Hs, H1own)
                assert eW < tol and eH < tol, f"W {eW:.2e} H {eH:.2e}"
                for x, y in zip(m_s.matrices["Bs"], m_1.matrices["Bs"]):
                    if np.asarray(y).size:
                        assert rel_fro(np.asarray(x), np.asarray(y)) < 5 * tol, "B"
                Ls, L1 = m_s.loss_history.to_numpy(), m_1.loss_history.to_numpy()
                assert Ls.shape == L1.shape, f"loss rows {Ls.shape} vs {L1.shape}"
                np.testing.assert_allclose(Ls[:, :2], L1[:, :2], rtol=1e-4)
                if local:
                    emb = lambda ad: np.concatenate([np.asarray(ad.obsm[k]).T for k in keys] + [np.asarray(ad.obsm["ALPINE_embedding"]).T], axis=0)   # noqa: E731
                    eT = rel_fro(emb(at_s), emb(at_1)[:, a:b])
                    assert eT < 10 * tol, f"transform {eT:.2e}"
                if rank == 0:
                    print(f"{tag} carrier={m_s.shard_comm_used}: W {eW:.1e} H {eH:.1e}", flush=True)
        except Exception as e:          # noqa: BLE001 -- reported by rank, the campaign stops at the next agreement point
            bad = f"{tag}: {type(e).__name__}: {e}"
            print(f"MISMATCH on rank {rank}: {bad}", flush=True)
        flags = [None] * world
        dist.all_gather_object(flags, bad)
        if any(flags):
            bad = next(f for f in flags if f)
    if rank == 0:
        print(("FAILED: " + bad) if bad else f"{n} cases passed in {time.perf_counter() - t0:.0f} s (seeds {seed0}..{seed0 + n - 1}, {world} ranks)", flush=True)
    open(os.path.join(out_dir, f"rank{rank}.status"), "w").write("bad" if bad else "ok")
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--seed0", type=int, default=60000)
    ap.add_argument("--stub", action="store_true", help="the library's own communicator over the shared-memory stand-in for RCCL")
    a = ap.parse_args()
    import tempfile
    import torch.multiprocessing as mp
    if a.stub:
        from _stub import build_rccl_stub
        lib = build_rccl_stub()
        os.environ["LD_PRELOAD"] = lib + (":" + os.environ["LD_PRELOAD"] if os.environ.get("LD_PRELOAD") else "")
    out = tempfile.mkdtemp()
    mp.spawn(worker, args=(a.ranks, _free_port(), a.seconds, a.seed0, a.stub, out), nprocs=a.ranks, join=True)
    sys.exit(0 if all(open(os.path.join(out, f"rank{r}.status")).read() == "ok" for r in range(a.ranks)) else 1)


if __name__ == "__main__":
    main()
