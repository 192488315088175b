"""The first REAL multi-rank run, whenever a box with two GPUs runs the suite: two fresh rank processes, rank r on cuda:r,
torch.distributed on the nccl (= RCCL) backend for the control plane, the library's own RCCL communicator for the
per-iteration all-reduce (`shard_comm="native"`: ncclCommInitRank + ncclAllReduce inside libalpine_hip.so, over xGMI).
Skipped on a one-GPU box -- there the same protocol runs over a shared-memory stand-in (tests/test_gpu_comm_stub.py) or
over gloo (tests/test_gpu_sharded.py), which says nothing about RCCL itself.

The ranks are started with torch.multiprocessing.spawn (fresh interpreters: nothing GPU-related is inherited)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _two_gpus() -> bool:
    import torch
    return torch.cuda.device_count() >= 2


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("case_name,local,comm", [("kl_2cov_nan", False, "native"), ("counts_2cov", True, "native"),
                                                  ("als_kl", False, "native"), ("mb_weighted", False, "native"),
                                                  ("kl_2cov_nan", False, "torch")])
def test_two_ranks_on_two_gpus(case_name, local, comm, tmp_path):
    import torch.multiprocessing as mp
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    from test_gpu_sharded import _free_port, _worker
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, str(tmp_path), local, comm, True), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert all(str(x["comm"]) == comm for x in r), [str(x["comm_note"]) for x in r]
    assert np.array_equal(r[0]["W"], r[1]["W"]) and np.array_equal(r[0]["losses"], r[1]["losses"])     # replicated state stays in step
    c = load_case(case_name)
    H = np.concatenate([r[0]["H"], r[1]["H"]], axis=1) if local else r[0]["H"]
    assert rel_fro(r[0]["W"], c.WT) < 1e-4 and rel_fro(H, c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(r[0][f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(r[0]["losses"], c.loss_history, n_cells=c.X.shape[0])
    if not c.fit_kwargs and not c.params.get("use_als"):
        single = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
        assert rel_fro(r[0]["W"], np.concatenate(single.matrices["Ws"], axis=1)) < 2e-5
        assert rel_fro(H, np.concatenate(single.matrices["Hs"], axis=0)) < 2e-5


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("case_name", ["kl_2cov_nan", "counts_2cov", "als_kl", "mb_weighted", "k105"])
def test_devices_in_one_process_on_two_gpus(case_name, tmp_path):
    """ALPINE(devices=[0, 1]).fit(adata) in a fresh process (no launcher, no torch.distributed): ncclCommInitAll over the two GPUs, one host
    thread per engine, the all-reduce over xGMI enqueued by the library's C loop.  Against the goldens and the single-device fit."""
    import os
    import subprocess
    import sys
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from alpine_amd import ALPINE, MiniAnnData
    from test_gpu_comm_stub import _DEVICES_SCRIPT, REPO
    c = load_case(case_name)
    out = str(tmp_path / "r.npz")
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    r = subprocess.run([sys.executable, "-c", _DEVICES_SCRIPT.format(repo=REPO, case=case_name, devices=[0, 1], out=out)], capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert str(z["comm"]).startswith("native (one process") and list(z["devices"]) == [0, 1]
    assert rel_fro(z["W"], c.WT) < 1e-4 and rel_fro(z["H"], c.HT) < 1e-4
    for i, bt in enumerate(c.BT):
        assert rel_fro(z[f"B{i}"], bt) < 2e-4
    assert_loss_rows_close(z["losses"], c.loss_history, n_cells=c.X.shape[0])
    if not c.fit_kwargs and not c.params.get("use_als"):
        single = ALPINE(device="cuda:0", **c.params).fit(MiniAnnData(c.X.copy(), c.obs.copy()), covariate_keys=c.keys, max_iter=c.T)
        assert rel_fro(z["W"], np.concatenate(single.matrices["Ws"], axis=1)) < 2e-5
        assert rel_fro(z["H"], np.concatenate(single.matrices["Hs"], axis=0)) < 2e-5
