"""The RCCL paths on ONE GPU: a communicator / process group of world size 1 on the real engine and its real stream.

A one-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device), so what runs here is everything around
the wire: ncclGetUniqueId / ncclCommInitRank through the C ABI, ncclAllReduce enqueued by the library's own loop on the
ctx stream (alpine_run, alpine_iter, alpine_batch_step, alpine_epoch_loss, the per-group exchange of the block-coordinate
branch), and torch.distributed's nccl backend on a tensor that aliases the external reduce block (stream ordering between
the engine's kernels and the collective).  A world-size-1 sum must leave every result BITWISE equal to the plain run.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _engine(c, x_dtype="x3", stream=None, block=None, use_als=False, batch_capacity=0):
    from alpine_amd import _native
    p = c.params
    levels = [y.shape[0] for y in c.Ys]
    eng = _native.NativeShard(n_genes=c.X.shape[1], n_cells=c.X.shape[0], n_components=p["n_components"],
                              cov_components=p["n_covariate_components"], cov_levels=levels, lam=p["lam"],
                              orth_W=p.get("orth_W", 0.0), alpha_W=p.get("alpha_W", 0.0), l1_ratio_W=p.get("l1_ratio_W", 0.0),
                              loss_type=p.get("loss_type", "kl-divergence"), x_dtype=x_dtype, stream=stream, reduce_block=block,
                              use_als=use_als, batch_capacity=batch_capacity)
    eng.upload_X_host(c.X)
    eng.finalize_X()
    for i, y in enumerate(c.Ys):
        eng.upload_Y(i, y)
    eng.set_factors(c.W0, c.H0, c.B0)
    return eng


def _result(eng):
    W, H, Bs = eng.get_factors()
    return W, H, Bs, eng.losses()


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[3], b[3])


@pytest.mark.parametrize("case_name,x_dtype", [("counts_2cov", "x3"), ("kl_2cov_nan", "f32")])
def test_torch_nccl_allreduce_on_the_engine_stream(nccl_group, case_name, x_dtype):
    """ShardedLoop + TorchDistComm over an nccl group of one rank, external reduce block in a torch tensor, engine on a
    non-default torch stream (what bench.py --comm torch and ALPINE(shard_comm='torch') do at N > 1)."""
    import torch
    from _golden import load_case
    from alpine_amd import _native
    from alpine_amd.sharded import ShardedLoop, TorchDistComm
    c = load_case(case_name)
    plain = _engine(c, x_dtype)
    plain.run(c.T, with_loss=True)
    want = _result(plain)
    plain.close()

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    assert stream.cuda_stream != 0
    p = c.params
    nfl = _native.reduce_block_floats(c.X.shape[1], c.X.shape[0], p["n_components"], p["n_covariate_components"], [y.shape[0] for y in c.Ys])
    with torch.cuda.stream(stream):
        block = torch.zeros(nfl, dtype=torch.float32, device=dev)
    stream.synchronize()
    eng = _engine(c, x_dtype, stream=stream.cuda_stream, block=block.data_ptr())
    assert eng.reduce_block()[0] == block.data_ptr()
    with torch.cuda.stream(stream):
        ShardedLoop(eng, TorchDistComm(block)).run(c.T, with_loss=True)
    got = _result(eng)
    eng.close()
    _same(got, want)
    assert nccl_group.get_backend() == "nccl"


@pytest.mark.parametrize("case_name", ["counts_2cov", "als_fro_2cov"])
def test_native_rccl_communicator_in_alpine_run(nccl_group, case_name):
    """alpine_comm_init_rank with nranks = 1, then the plain C loop (alpine_run) enqueues ncclAllReduce itself: MU branch
    and block-coordinate branch (one more exchange per component group)."""
    from _golden import load_case
    from alpine_amd import _native
    from alpine_amd.sharded import NativeComm, ShardedLoop, attach_native_comm
    c = load_case(case_name)
    als = bool(c.params.get("use_als"))
    plain = _engine(c, use_als=als)
    plain.run(c.T, with_loss=True)
    want = _result(plain)
    plain.close()

    eng = _engine(c, use_als=als)
    attach_native_comm(eng, nccl_group)              # torch only carries the 128-byte id
    assert eng.comm_ranks == 1
    eng.set_profiling(True)
    ShardedLoop(eng, NativeComm(eng), als_groups=(len(c.Ys) + 1 if als else 0)).run(c.T, with_loss=True)
    got = _result(eng)
    ms, n = eng.kernel_time(_native.KERNEL_ALLREDUCE)
    # T iterations + the closing loss row; the block-coordinate branch adds one K x K exchange per covariate group
    assert n == (c.T + 1) + (c.T * len(c.Ys) if als else 0) and ms >= 0
    eng.comm_destroy()
    eng.close()
    _same(got, want)


def test_native_rccl_split_entry_points_and_minibatches(nccl_group):
    """The caller-driven form (begin / alpine_comm_all_reduce / end) and the mini-batch composites with a communicator."""
    import torch
    from _golden import load_case
    from alpine_amd import _native
    c = load_case("mb_random")
    bs = c.fit_kwargs["batch_size"]
    n = c.X.shape[0]

    def epochs(eng, split):
        torch.manual_seed(1234)
        for _ in range(3):
            perm = torch.randperm(n).numpy()
            for b0 in range(0, n, bs):
                idx = perm[b0:b0 + bs]
                if split:
                    eng.batch_begin(idx)
                    eng.comm_all_reduce()
                    eng.batch_end()
                else:
                    eng.batch_step(idx)
            if split:
                eng.epoch_loss_begin()
                eng.comm_all_reduce()
                eng.epoch_loss_end()
            else:
                eng.epoch_loss()
        return _result(eng)

    plain = _engine(c, batch_capacity=bs)
    want = epochs(plain, False)
    plain.close()
    for split in (False, True):
        eng = _engine(c, batch_capacity=bs)
        eng.comm_init(_native.comm_unique_id(), 1, 0)
        got = epochs(eng, split)
        eng.close()                                   # alpine_destroy also destroys the communicator
        _same(got, want)

    # full-batch split form
    c2 = load_case("kl_1cov")
    plain = _engine(c2)
    plain.run(5, with_loss=True)
    want = _result(plain)
    plain.close()
    eng = _engine(c2)
    eng.comm_init(_native.comm_unique_id(), 1, 0)
    for _ in range(5):
        eng.iter_begin()
        eng.comm_all_reduce()
        eng.iter_end(True)
    eng.iter(False)
    _same(_result(eng), want)
    with pytest.raises(_native.AlpineNativeError):
        eng.comm_init(_native.comm_unique_id(), 1, 0)      # a ctx holds one communicator
    with pytest.raises(_native.AlpineNativeError):
        eng.comm_all_reduce(0, 10**12)
    eng.close()


def test_comm_calls_without_a_communicator_fail_loudly():
    from _golden import load_case
    from alpine_amd import _native
    c = load_case("kl_1cov")
    eng = _engine(c)
    with pytest.raises(_native.AlpineNativeError) as ei:
        eng.comm_all_reduce()
    assert ei.value.code == -4
    with pytest.raises(_native.AlpineNativeError):
        eng.batch_step(np.zeros(0, dtype=np.int64))
    eng.comm_destroy()                                   # nothing attached: a no-op
    eng.close()


def test_drop_in_fit_with_native_comm_world_size_one(nccl_group):
    """ALPINE(shard_cells=True) under an nccl group of one rank picks the native communicator and equals the plain fit."""
    from _golden import load_case
    from alpine_amd import ALPINE, MiniAnnData
    c = load_case("kl_2cov_nan")
    a = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
    m = ALPINE(device="cuda:0", shard_cells=True, **c.params)
    # world size 1 short-circuits to the single-device path by design; force the sharded code path through a 1-rank group
    m._dist_world = lambda: (nccl_group, 0, 1) if m.shard_cells else (None, 0, 1)
    m.fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
    assert m.shard_comm_used == "native"
    for x, y in zip(a.matrices["Ws"] + a.matrices["Hs"] + a.matrices["Bs"], m.matrices["Ws"] + m.matrices["Hs"] + m.matrices["Bs"]):
        assert np.array_equal(x, y)
    assert np.array_equal(a.loss_history.to_numpy(), m.loss_history.to_numpy())


@pytest.mark.parametrize("case_name", ["kl_2cov_nan", "mb_random", "als_kl"])
def test_drop_in_fit_with_devices_world_size_one_on_real_rccl(case_name):
    """ALPINE(devices=[0]): the single-process multi-GPU form with ONE device -- ncclCommInitAll on real RCCL, the engine driven from a
    worker thread, ncclAllReduce enqueued by alpine_run / alpine_batch_step / alpine_epoch_loss.  Against the reference's goldens and
    BITWISE against the plain single-device fit."""
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    c = load_case(case_name)
    fk = dict(c.meta.get("fit_kwargs") or {})
    res = []
    for devices in (None, [0]):
        ad = MiniAnnData(c.X.copy(), c.obs.copy())
        m = ALPINE(device="cuda:0", devices=devices, **c.params).fit(ad, covariate_keys=c.keys, max_iter=c.T, **fk)
        res.append((np.concatenate(m.matrices["Ws"], axis=1), np.concatenate(m.matrices["Hs"], axis=0), m.loss_history.to_numpy()))
    assert m.shard_comm_used.startswith("native (one process") and m.fit_info["devices"] == [0]
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert rel_fro(res[1][0], c.WT) < 1e-4 and rel_fro(res[1][1], c.HT) < 1e-4
    assert_loss_rows_close(res[1][2], c.loss_history, n_cells=c.X.shape[0])


def test_comm_count_reports_what_the_communicator_says(nccl_group):
    """alpine_comm_count = ncclCommCount / ncclCommUserRank of the attached communicator (what bench.py prints for N > 1 runs)."""
    from _golden import load_case
    from alpine_amd import _native
    from alpine_amd.sharded import attach_native_comm
    c = load_case("kl_1cov")
    eng = _engine(c)
    with pytest.raises(_native.AlpineNativeError, match="no communicator"):
        eng.comm_count()
    attach_native_comm(eng, nccl_group)
    assert eng.comm_count() == (1, 0)
    eng.comm_destroy()
    eng.close()
