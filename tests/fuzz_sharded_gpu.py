"""Time-boxed randomised campaign of the NATIVE multi-rank loop on a one-GPU box (by hand; not collected by pytest):

    python tests/fuzz_sharded_gpu.py --seconds 600 [--ranks 2|3] [--seed0 S]

R rank processes share cuda:0; tests/stub_rccl/rccl_stub.cpp (a shared-memory stand-in for the five RCCL entry points,
see tests/test_gpu_comm_stub.py) is preloaded into them, everything above it is the product path: per random case every
rank takes its block of cells (sharded.shard_bounds), attaches the library's communicator and runs `alpine_run` (MU or
block-coordinate branch) or `alpine_batch_step` / `alpine_epoch_loss` epochs on shared index streams.  Every rank also
runs the same problem unsharded and compares: W, B, loss rows (replicated) and its own columns of H."""
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


def worker(rank, world, port, seconds, seed0, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fuzz_gpu as fz
    from _golden import rel_fro
    from alpine_amd import _native as nat
    from alpine_amd.sharded import attach_native_comm, shard_bounds

    def build(p, X, Ys, W0, H0, B0, c0, c1, mode, use_als, batch_capacity):
        eng = nat.NativeShard(n_genes=X.shape[1], n_cells=c1 - c0, n_components=p.n_components, cov_components=p.n_covariate_components,
                              cov_levels=[y.shape[1] for y in Ys], lam=p.lam, orth_W=p.orth_W, alpha_W=p.alpha_W, l1_ratio_W=p.l1_ratio_W,
                              eps=p.eps, loss_type=p.loss_type, use_als=use_als, batch_capacity=batch_capacity, x_dtype=mode)
        eng.upload_X_host(X[c0:c1])
        eng.finalize_X()
        for i, y in enumerate(Ys):
            eng.upload_Y(i, np.ascontiguousarray(y.T[:, c0:c1]))
        eng.set_factors(W0, H0, B0, h_col0=c0)
        return eng

    t0 = time.perf_counter()
    n = 0
    bad = None
    while True:
        go = [time.perf_counter() - t0 < seconds and bad is None]
        dist.broadcast_object_list(go, src=0)                       # every rank takes the same number of cases
        if not go[0]:
            break
        seed = seed0 + n
        p, X, Ys, kind, iters, _, rng = fz.make_case(seed)          # same seed -> same case and same draws on every rank
        N, G = X.shape
        if N > 20000:                                               # keep a case at a second or two
            X, Ys, N = X[:20000], [y[:20000] for y in Ys], 20000
        if N < 8 * world:
            n += 1
            continue
        from oracle import alpine_oracle as orc
        s0 = orc.init_factors(p, np.ascontiguousarray(X.T), Ys)
        W0, H0, B0 = s0.W.numpy().copy(), s0.H.numpy().copy(), [b.numpy().copy() for b in s0.Bs]
        mode = str(rng.choice(["x3", "f32"]))
        branch = str(rng.choice(["mu", "mu", "als", "minibatch"])) if p.n_covariate_components else "mu"
        use_als = branch == "als"
        bs = int(rng.integers(max(2, N // 5), N + 1)) if branch == "minibatch" else 0
        epochs = []
        if bs:
            for e in range(2):
                order = rng.permutation(N) if e == 0 else rng.integers(0, N, size=N)
                epochs.append([order[b0:b0 + bs] for b0 in range(0, N, bs)])
        c0, c1 = shard_bounds(N, world, rank)
        tag = f"seed {seed} {mode} {branch} G={G} N={N} K={p.total_components} cov={p.n_covariate_components} {p.loss_type} ranks={world} block=[{c0},{c1})"
        try:
            results = []
            for sharded in (True, False):
                a, b = (c0, c1) if sharded else (0, N)
                eng = build(p, X, Ys, W0, H0, B0, a, b, mode, use_als, bs)
                if sharded:
                    attach_native_comm(eng, dist)
                if bs:
                    for ep in epochs:
                        for idx in ep:
                            loc = idx[(idx >= a) & (idx < b)] - a
                            eng.batch_step(loc)                    # sharded: possibly empty; the library exchanges inside
                        eng.epoch_loss()
                else:
                    eng.run(iters, with_loss=True)
                W, H, Bs = eng.get_factors()
                results.append((W, H, Bs, eng.losses()))
                eng.close()
            (Ws, Hs, Bss, Ls), (W1, H1, Bs1, L1) = results
            eW, eH = rel_fro(Ws, W1), rel_fro(Hs, H1[:, c0:c1])
            assert eW < 3e-5 and eH < 3e-5, f"W {eW:.2e} H {eH:.2e}"
            for x, y in zip(Bss, Bs1):
                assert rel_fro(x, y) < 1e-4, f"B {rel_fro(x, y):.2e}"
            assert Ls.shape == L1.shape, f"loss rows {Ls.shape} vs {L1.shape}"
            np.testing.assert_allclose(Ls[:, :2], L1[:, :2], rtol=2e-5)
            if rank == 0:
                print(f"{tag}: W {eW:.1e} H {eH:.1e}", flush=True)
        except Exception as e:          # noqa: BLE001 -- reported by rank, the campaign stops at the next agreement point
            bad = f"{tag}: {type(e).__name__}: {e}"
            print(f"MISMATCH on rank {rank}: {bad}", flush=True)
        flags = [None] * world
        dist.all_gather_object(flags, bad)
        if any(flags):
            bad = next(f for f in flags if f)
        n += 1
    if rank == 0:
        print(("FAILED: " + bad) if bad else f"{n} cases passed in {time.perf_counter() - t0:.0f} s (seeds {seed0}..{seed0 + n - 1}, {world} ranks)", flush=True)
    open(os.path.join(out_dir, f"rank{rank}.status"), "w").write("bad" if bad else "ok")
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--seed0", type=int, default=20000)
    a = ap.parse_args()
    import tempfile
    import torch.multiprocessing as mp
    from _stub import build_rccl_stub
    lib = build_rccl_stub()
    os.environ["LD_PRELOAD"] = lib + (":" + os.environ["LD_PRELOAD"] if os.environ.get("LD_PRELOAD") else "")
    out = tempfile.mkdtemp()
    mp.spawn(worker, args=(a.ranks, _free_port(), a.seconds, a.seed0, out), nprocs=a.ranks, join=True)
    sys.exit(0 if all(open(os.path.join(out, f"rank{r}.status")).read() == "ok" for r in range(a.ranks)) else 1)


if __name__ == "__main__":
    main()
