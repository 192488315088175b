"""The library's NATIVE multi-rank loop with more than one rank, on a one-GPU box.

RCCL refuses two ranks on one device, so on this pool the communicator inside libalpine_hip.so could otherwise only run at
world size 1 (tests/test_gpu_nccl.py), where every all-reduce is the identity.  Here the rank processes are started with
tests/stub_rccl/rccl_stub.cpp preloaded: a shared-memory stand-in for the five RCCL entry points the library calls (its own
CPU check: tests/test_comm_stub_selftest.py).  Everything above those five calls is the product path, unchanged: the same
libalpine_hip.so, `alpine_comm_init_rank`, and the C loops of `alpine_run` / `alpine_iter` / `alpine_batch_step` /
`alpine_epoch_loss` that decide which slot of the reduce block is exchanged when.  Two ranks share cuda:0; results must
equal the single-device run to rounding and the reference's golden vectors to the stated tolerance.

What this does NOT show: anything about RCCL or xGMI (bandwidth, stream ordering inside RCCL) -- see DESIGN.md 5."""
import os
import subprocess

import numpy as np
import pytest

from _golden import assert_loss_rows_close, load_case, rel_fro
from _stub import build_rccl_stub

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class preload_stub:
    """LD_PRELOAD for the processes started inside the block (spawned children inherit the environment at exec)."""

    def __enter__(self):
        self.old = os.environ.get("LD_PRELOAD")
        lib = build_rccl_stub()
        os.environ["LD_PRELOAD"] = lib if not self.old else lib + ":" + self.old
        return lib

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("LD_PRELOAD", None)
        else:
            os.environ["LD_PRELOAD"] = self.old
        return False


def _spawn(case_name, tmp_path, local):
    import torch.multiprocessing as mp
    from test_gpu_sharded import _free_port, _worker
    world = 2
    with preload_stub():
        mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path), local, "native"), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert all(str(x["comm"]) == "native" for x in r)          # the library's own communicator carried every exchange
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["losses"], r[1]["losses"])
    return r


def _check_against_golden(c, r, local):
    H = np.concatenate([r[0]["H"], r[1]["H"]], axis=1) if local else r[0]["H"]
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(r[0][f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])
    return H


@pytest.mark.parametrize("case_name,local", [("kl_2cov_nan", False), ("counts_2cov", True), ("fro_2cov_reg", False), ("wide_k200_fro", False), ("wide_k520_fro", False)])
def test_native_loop_two_ranks_full_batch(case_name, local, tmp_path):
    """alpine_run with a two-rank communicator: one exchange of the whole reduce block per iteration (+ one for the last
    loss row).  Equal to the single-device run up to the summation order of the two shards."""
    from alpine_amd import ALPINE, MiniAnnData
    r = _spawn(case_name, tmp_path, local)
    c = load_case(case_name)
    H = _check_against_golden(c, r, local)
    single = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
    assert rel_fro(r[0]["W"], np.concatenate(single.matrices["Ws"], axis=1)) < 2e-5
    assert rel_fro(H, np.concatenate(single.matrices["Hs"], axis=0)) < 2e-5


@pytest.mark.parametrize("case_name,local", [("als_kl", False), ("als_fro_2cov", True), ("als_wide_k160_fro", False)])
def test_native_loop_two_ranks_block_coordinate(case_name, local, tmp_path):
    """use_als: alpine_iter exchanges the K x K H H^T slot after every component group, in C."""
    r = _spawn(case_name, tmp_path, local)
    c = load_case(case_name)
    assert c.params.get("use_als")
    _check_against_golden(c, r, local)


@pytest.mark.parametrize("case_name,local", [("mb_random", False), ("mb_weighted", True), ("weighted_skew", False), ("mb_wide_k150", True), ("mb_wide_k300", False)])
def test_native_loop_two_ranks_minibatch(case_name, local, tmp_path):
    """Mini-batches: alpine_batch_step (gather, phase 1, exchange, phase 2, scatter) and alpine_epoch_loss with the
    communicator attached, including batches of which a rank holds no cell (weighted_skew: most draws fall into rank 0's
    block)."""
    r = _spawn(case_name, tmp_path, local)
    _check_against_golden(load_case(case_name), r, local)


@pytest.mark.parametrize("name", ["kl_2cov_nan", "als_kl"])
def test_c_host_two_ranks(name, tmp_path):
    """examples/fit_c --ranks 2: a plain-C host forks two rank processes, passes the communicator id through a file and
    splices the ranks' H blocks; no Python and no torch in those processes."""
    from test_c_abi_example import build_example, read_result, write_problem
    exe = build_example()
    c = load_case(name)
    flags = 16 | (4 if c.params.get("use_als") else 0)
    prob, res1, res2 = tmp_path / "p.bin", tmp_path / "r1.bin", tmp_path / "r2.bin"
    write_problem(prob, c, flags)
    r = subprocess.run([exe, str(prob), str(res1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with preload_stub():
        r = subprocess.run([exe, "--ranks", "2", str(prob), str(res2)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, FIT_C_DEVICE_COUNT="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    l1, W1, H1, _ = read_result(res1, c)
    l2, W2, H2, Bs = read_result(res2, c)
    assert rel_fro(W2, W1) < 2e-5 and rel_fro(H2, H1) < 2e-5
    assert rel_fro(W2, c.WT) < 1e-4 and rel_fro(H2, c.HT) < 1e-4
    for b, bt in zip(Bs, c.BT):
        assert rel_fro(b, bt) < 2e-4
    assert_loss_rows_close(l2, c.loss_history, n_cells=c.X.shape[0])


def test_bench_two_ranks_native_carrier():
    """`python bench.py --gpus 2 --comm native` as the driver's N > 1 runs use it (own launcher, the library's communicator,
    the all-reduce timed by the library's events), rehearsed on one GPU with the stand-in carrying the bytes."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ALPINE_BENCH_REHEARSAL_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--comm", "native", "--workload", "tiny", "--steps", "4", "--warmup", "1"]
    with preload_stub() as lib:
        env["LD_PRELOAD"] = lib
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    ar = d["allreduce"]
    assert ar["carrier"].startswith("native") and ar["fallback_note"] is None
    assert ar["avg_ms_on_rank0"] > 0 and ar["bytes"] > 0
    # the line verifies itself: what the communicator reports, and every rank's own numbers
    assert ar["rccl_ranks"] == 2
    pr = ar["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["carrier"] == "native" and r["rccl_ranks"] == 2 and r["rccl_rank"] == r["rank"] for r in pr)
    assert all(r["ms_per_step"] > 0 and r["avg_ms_xht"] > 0 and r["avg_ms_wtx"] > 0 and r["cells"] > 0 for r in pr)
    assert abs(max(r["ms_per_step"] for r in pr) - d["ms_per_step"]) <= 1e-6 * d["ms_per_step"]


def _als_minibatch_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from alpine_amd import _native
    from alpine_amd.sharded import attach_native_comm
    rng = np.random.default_rng(3)
    eng = _native.NativeShard(n_genes=40, n_cells=64, n_components=3, cov_components=[2], cov_levels=[2], lam=[1.0], use_als=True,
                              batch_capacity=32, x_dtype="x3")
    eng.upload_X_host(rng.random((64, 40), dtype=np.float32))
    eng.finalize_X()
    eng.upload_Y(0, np.eye(2, dtype=np.float32)[rng.integers(0, 2, size=64)].T.copy())
    eng.set_factors(rng.random((40, 5), dtype=np.float32), rng.random((5, 64), dtype=np.float32), [rng.random((2, 2), dtype=np.float32)])
    attach_native_comm(eng, dist)
    try:
        eng.batch_step(np.arange(16))
        msg = "no error"
    except _native.AlpineNativeError as e:
        msg = f"{e.code}: {e}"
    eng.close()
    open(os.path.join(out_dir, f"als_mb_rank{rank}.txt"), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_als_minibatch_is_refused_on_a_two_rank_communicator(tmp_path):
    """The group loop of a mini-batch step has no exchange point, so with more than one rank it would silently use local
    sums: alpine_batch_begin must say ALPINE_ERR_UNSUPPORTED (-5) instead -- on every rank, before any collective."""
    import torch.multiprocessing as mp
    from test_gpu_sharded import _free_port
    with preload_stub():
        mp.spawn(_als_minibatch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(tmp_path / f"als_mb_rank{r}.txt").read()
        assert msg.startswith("-5:") and "single-shard" in msg, msg


@pytest.mark.parametrize("name,D", [("kl_2cov_nan", 2), ("als_kl", 2), ("counts_2cov", 3)])
def test_c_host_devices_in_one_process(name, D, tmp_path):
    """examples/fit_c --devices D: ONE process, D ctxs, alpine_comm_init_all, D host threads each running the one-GPU call sequence -- the
    single-process form of SURVEY.md 8b from plain C (the stand-in carries the bytes: D ctxs share cuda:0 here)."""
    from test_c_abi_example import build_example, read_result, write_problem
    exe = build_example()
    c = load_case(name)
    flags = 16 | (4 if c.params.get("use_als") else 0)
    prob, res1, res2 = tmp_path / "p.bin", tmp_path / "r1.bin", tmp_path / "r2.bin"
    write_problem(prob, c, flags)
    r = subprocess.run([exe, str(prob), str(res1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with preload_stub():
        r = subprocess.run([exe, "--devices", str(D), str(prob), str(res2)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, FIT_C_DEVICE_COUNT="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    l1, W1, H1, _ = read_result(res1, c)
    l2, W2, H2, Bs = read_result(res2, c)
    assert rel_fro(W2, W1) < 2e-5 and rel_fro(H2, H1) < 2e-5
    assert rel_fro(W2, c.WT) < 1e-4 and rel_fro(H2, c.HT) < 1e-4
    for b, bt in zip(Bs, c.BT):
        assert rel_fro(b, bt) < 2e-4
    assert_loss_rows_close(l2, c.loss_history, n_cells=c.X.shape[0])


_DEVICES_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import numpy as np
from _golden import load_case
from alpine_amd import ALPINE, MiniAnnData
c = load_case({case!r})
fk = dict(c.meta.get("fit_kwargs") or {{}})
ad = MiniAnnData(c.X.copy(), c.obs.copy())
m = ALPINE(device="cuda:0", devices={devices!r}, **c.params).fit(ad, covariate_keys=c.keys, max_iter=c.T, **fk)
np.savez({out!r}, W=np.concatenate(m.matrices["Ws"], axis=1), H=np.concatenate(m.matrices["Hs"], axis=0), losses=m.loss_history.to_numpy(),
         emb=np.asarray(ad.obsm["ALPINE_embedding"]), comm=str(m.shard_comm_used), devices=np.array(m.fit_info["devices"]),
         **{{f"B{{i}}": b for i, b in enumerate(m.matrices["Bs"])}})
"""


@pytest.mark.parametrize("case_name,devices", [("kl_2cov_nan", [0, 0]), ("counts_2cov", [0, 0, 0]), ("als_fro_2cov", [0, 0]), ("mb_weighted", [0, 0]),
                                               ("wide_k150", [0, 0]), ("k105", [0, 0]), ("wide_k300", [0, 0, 0])])
def test_drop_in_fit_with_devices_in_one_process(case_name, devices, tmp_path):
    """ALPINE(devices=[...]).fit(adata): one process, one engine and one host thread per listed GPU, the library's communicator over them
    (alpine_comm_init_all), no launcher and no torch.distributed -- the reference's one blocking call (main.py:82-147).  Rehearsed on one
    GPU: the engines share cuda:0 (ALPINE_AMD_TEST_SHARED_DEVICE=1) and the stand-in carries the bytes.  Full batch, block-coordinate
    branch, weighted mini-batches (the epoch's index stream drawn once, every engine takes its cells), K > 64, K > 128 and K > 256."""
    import sys
    c = load_case(case_name)
    out = str(tmp_path / "r.npz")
    script = _DEVICES_SCRIPT.format(repo=REPO, case=case_name, devices=devices, out=out)
    with preload_stub():
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, ALPINE_AMD_TEST_SHARED_DEVICE="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert str(z["comm"]).startswith("native (one process") and list(z["devices"]) == devices
    assert rel_fro(z["W"], c.WT) < 1e-4 and rel_fro(z["H"], c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(z[f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(z["losses"], c.loss_history, n_cells=c.X.shape[0])
    assert z["emb"].shape == (c.X.shape[0], c.params["n_components"])
