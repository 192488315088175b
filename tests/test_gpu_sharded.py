"""Sharded HIP path on ONE GPU box: two ranks (processes) both drive cuda:0 with the real engine
(external reduce block in a torch tensor, engine on torch's stream), all-reduce over gloo (RCCL needs
one device per rank; the protocol above the collective is what this covers).  Result must equal the
single-shard HIP run to rounding, and the reference golden vectors to the stated tolerance."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case_name, out_dir, local=False, shard_comm="auto", device_per_rank=False):
    # (fit_kwargs of the golden case -- batch_size / sampling_method -- are passed through: sharded mini-batches)
    # device_per_rank: rank r drives cuda:r and torch.distributed runs on the nccl (= RCCL) backend -- the production layout
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    dev_i = rank if device_per_rank else 0
    torch.cuda.set_device(dev_i)
    if device_per_rank:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_i))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from _golden import load_case
    from alpine_amd import ALPINE, MiniAnnData
    c = load_case(case_name)
    if local:
        # rank-local input: an UNEVEN, unaligned split in rank order (rank 0 gets the first third of the cells)
        n = c.X.shape[0]
        cut = [0, n // 3 + 1, n]
        adata = MiniAnnData(c.X[cut[rank]:cut[rank + 1]].copy(), c.obs.iloc[cut[rank]:cut[rank + 1]].reset_index(drop=True))
        m = ALPINE(device=f"cuda:{dev_i}", shard_cells="local", shard_comm=shard_comm, **c.params).fit(adata, covariate_keys=c.keys, max_iter=c.T, **c.fit_kwargs)
        assert adata.obsm["ALPINE_embedding"].shape[0] == cut[rank + 1] - cut[rank]
        if c.transform_iters:                      # transform of the rank's own cells right after the fit
            a_t = MiniAnnData(c.X[cut[rank]:cut[rank + 1]].copy(), c.obs.iloc[cut[rank]:cut[rank + 1]].reset_index(drop=True))
            m.transform(a_t, n_iter=c.transform_iters)
            np.save(os.path.join(out_dir, f"transform_rank{rank}.npy"),
                    np.concatenate([np.asarray(a_t.obsm[k]).T for k in c.keys] + [np.asarray(a_t.obsm["ALPINE_embedding"]).T], axis=0))
    else:
        adata = MiniAnnData(c.X.copy(), c.obs.copy())
        m = ALPINE(device=f"cuda:{dev_i}", shard_cells=True, shard_comm=shard_comm, **c.params).fit(adata, covariate_keys=c.keys, max_iter=c.T, **c.fit_kwargs)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=np.concatenate(m.matrices["Ws"], axis=1),
             H=np.concatenate(m.matrices["Hs"], axis=0), losses=m.loss_history.to_numpy(), comm=np.array(m.shard_comm_used),
             comm_note=np.array(str(m.shard_comm_note)),
             **{f"B{i}": b for i, b in enumerate(m.matrices["Bs"])})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case_name", ["kl_2cov_nan", "ragged", "wide_k150", "wide_k300"])
def test_two_ranks_one_gpu(case_name, tmp_path):
    import torch.multiprocessing as mp
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path)), nprocs=world, join=True)
    c = load_case(case_name)
    single = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
    W1 = np.concatenate(single.matrices["Ws"], axis=1)
    H1 = np.concatenate(single.matrices["Hs"], axis=0)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["H"], r[1]["H"])
    assert rel_fro(r[0]["W"], W1) < 2e-5 and rel_fro(r[0]["H"], H1) < 2e-5      # summation order differs between shardings
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(r[0]["H"], c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(r[0][f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])


@pytest.mark.parametrize("case_name", ["kl_2cov_nan", "counts_2cov"])
def test_two_ranks_local_input(case_name, tmp_path):
    """shard_cells='local': each rank passes only its own cells (uneven split); W, B and the loss history equal the
    single-process run, the ranks' H columns concatenate to its H, labels missing on one rank still get their column."""
    import torch.multiprocessing as mp
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path), True), nprocs=world, join=True)
    c = load_case(case_name)
    single = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
    W1 = np.concatenate(single.matrices["Ws"], axis=1)
    H1 = np.concatenate(single.matrices["Hs"], axis=0)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["losses"], r[1]["losses"])
    H = np.concatenate([r[0]["H"], r[1]["H"]], axis=1)
    assert H.shape == H1.shape
    assert rel_fro(r[0]["W"], W1) < 2e-5 and rel_fro(H, H1) < 2e-5
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])
    if c.transform_iters:
        # transform of ALL cells in one process right after a fit == the ranks' local transforms side by side
        a_t = MiniAnnData(c.X.copy(), c.obs.copy())
        single.transform(a_t, n_iter=c.transform_iters)
        Ht1 = np.concatenate([np.asarray(a_t.obsm[k]).T for k in c.keys] + [np.asarray(a_t.obsm["ALPINE_embedding"]).T], axis=0)
        Ht = np.concatenate([np.load(tmp_path / f"transform_rank{i}.npy") for i in range(world)], axis=1)
        assert rel_fro(Ht, Ht1) < 2e-5


@pytest.mark.parametrize("case_name,local", [("mb_random", False), ("mb_weighted", False), ("mb_weighted", True), ("full_weighted", True),
                                             ("weighted_skew", False), ("mb_wide_k150", False), ("mb_weighted_wide_k140", True)])
def test_two_ranks_minibatch(case_name, local, tmp_path):
    """Mini-batch / weighted sampling sharded over two ranks: both draw the same global index stream, each takes the
    batch's cells that fall into its block (sometimes none), the reduce block is all-reduced between batch_begin and
    batch_end.  Must reproduce the REFERENCE's stochastic run (same tolerances as the single-device mini-batch test).
    weighted_skew: a rare, heavily weighted label lives in rank 0's block, so rank 0 receives ~470 of every epoch's 640 draws
    -- more cells than its 320-cell shard holds and more 128-cell statistic blocks than the shard has."""
    import torch.multiprocessing as mp
    from _golden import assert_loss_rows_close, load_case, rel_fro
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path), local), nprocs=world, join=True)
    c = load_case(case_name)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["losses"], r[1]["losses"])
    H = np.concatenate([r[0]["H"], r[1]["H"]], axis=1) if local else r[0]["H"]
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(r[0][f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])


@pytest.mark.parametrize("case_name,local", [("als_kl", False), ("als_fro_2cov", True), ("als_wide_k150", False), ("als_wide_k270", False)])
def test_two_ranks_use_als(case_name, local, tmp_path):
    """Block-coordinate branch sharded over two ranks: one extra all-reduce of the K x K H H^T slot per component group."""
    import torch.multiprocessing as mp
    from _golden import assert_loss_rows_close, load_case, rel_fro
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path), local), nprocs=world, join=True)
    c = load_case(case_name)
    assert c.params.get("use_als")
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["losses"], r[1]["losses"])
    H = np.concatenate([r[0]["H"], r[1]["H"]], axis=1) if local else r[0]["H"]
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(r[0][f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])
