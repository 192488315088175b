"""N>1 path on CPU: the sharded loop (alpine_amd/sharded.py) driven with a CPU engine over gloo,
world_size 2.  The engine here wraps the ORACLE's fused step (tests may use the oracle); the product
engine is _native.NativeShard, which exposes the same iter_begin / reduce block / iter_end protocol.
What is covered: shard bounds, the packed reduce-block protocol (everything that crosses shards is a
sum over cells), the extra half-step for the last loss row, and equality with the single-shard run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


class OracleShardEngine:
    """CPU stand-in with NativeShard's protocol: iter_begin() fills a flat float32 block with LOCAL sums,
    the caller all-reduces it, iter_end() consumes the GLOBAL sums."""

    def __init__(self, p, X_gn, Ys, W, H, Bs):
        from oracle import alpine_oracle as orc
        self.orc, self.p = orc, p
        self.s = orc.OracleState(torch.tensor(X_gn), [torch.tensor(y) for y in Ys], torch.tensor(W), torch.tensor(H),
                                 [torch.tensor(b) for b in Bs])
        self.xnorm2_local = float(torch.sum(self.s.X.double() ** 2))
        G, K = W.shape
        self.sizes = [G * K, K * K] + [b.size for b in Bs] + [b.shape[1] for b in Bs] + [1] * len(Bs) + [1]
        self.block = torch.zeros(sum(self.sizes), dtype=torch.float64)     # fp64 block: keeps the test about logic
        self.pending = False
        self.losses = []

    def iter_begin(self):
        XHt, HHt, bnum, bden, pred = self.orc.fused_reduce_terms(self.p, self.s)
        parts = [XHt.flatten(), HHt.flatten()] + [b.flatten() for b in bnum] + list(bden) + \
                [x.reshape(1) for x in pred] + [torch.tensor([self.xnorm2_local])]
        self.block.copy_(torch.cat([t.double().flatten() for t in parts]))

    def _unpack(self):
        G, K = self.s.W.shape
        out, o = [], 0
        for n in self.sizes:
            out.append(self.block[o:o + n])
            o += n
        nb = len(self.s.Bs)
        XHt = out[0].reshape(G, K).float()
        HHt = out[1].reshape(K, K).float()
        bnum = [out[2 + i].reshape(self.s.Bs[i].shape).float() for i in range(nb)]
        bden = [out[2 + nb + i].float() for i in range(nb)]
        pred = [out[2 + 2 * nb + i][0] for i in range(nb)]
        return (XHt, HHt, bnum, bden, pred), float(out[-1][0])

    def iter_end(self, update=True):
        terms, xnorm2 = self._unpack()
        if self.pending:
            self.losses.append(self.orc.trace_loss_row(self.p, self.s, xnorm2, terms))
        self.pending = False
        if update:
            self.orc.mu_step_fused(self.p, self.s, terms)
            self.pending = True


class GlooComm:
    def __init__(self, block):
        self.block = block

    def all_reduce(self):
        dist.all_reduce(self.block, op=dist.ReduceOp.SUM)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case_name, T, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _golden import load_case
    from oracle import alpine_oracle as orc
    from alpine_amd.sharded import ShardedLoop, shard_bounds
    c = load_case(case_name)
    p = orc.OracleParams(**c.params)
    c0, c1 = shard_bounds(c.X.shape[0], world, rank)
    eng = OracleShardEngine(p, np.ascontiguousarray(c.X[c0:c1].T), [y[:, c0:c1].copy() for y in c.Ys],
                            c.W0.copy(), c.H0[:, c0:c1].copy(), [b.copy() for b in c.B0])
    ShardedLoop(eng, GlooComm(eng.block)).run(T, with_loss=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=eng.s.W.numpy(), H=eng.s.H.numpy(), c0=c0, c1=c1,
             losses=np.array(eng.losses), **{f"B{i}": b.numpy() for i, b in enumerate(eng.s.Bs)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case_name", ["kl_2cov_nan", "fro_2cov_reg"])
def test_two_shards_equal_one_shard(case_name, tmp_path):
    from _golden import assert_loss_rows_close, load_case, rel_fro
    from oracle import alpine_oracle as orc
    T, world = 8, 2
    mp.spawn(_worker, args=(world, _free_port(), case_name, T, str(tmp_path)), nprocs=world, join=True)
    c = load_case(case_name)
    p = orc.OracleParams(**c.params)
    s = orc.init_factors(p, np.ascontiguousarray(c.X.T), [y.T for y in c.Ys])
    orc.fit_fused(p, s, T)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    H = np.concatenate([x["H"] for x in r], axis=1)
    assert int(r[0]["c0"]) == 0 and int(r[0]["c1"]) == int(r[1]["c0"]) and int(r[1]["c1"]) == c.X.shape[0]
    for x in r:                                   # replicated state is identical on every rank
        assert np.array_equal(x["W"], r[0]["W"]) and np.array_equal(x["losses"], r[0]["losses"])
        assert rel_fro(x["W"], s.W.numpy()) < 5e-6
        for i, b in enumerate(s.Bs):
            assert rel_fro(x[f"B{i}"], b.numpy()) < 5e-6
    assert rel_fro(H, s.H.numpy()) < 5e-6
    assert r[0]["losses"].shape == (T, 2 + len(c.Ys))
    assert_loss_rows_close(r[0]["losses"], np.array(s.losses), n_cells=c.X.shape[0], rtol=1e-5)


def test_shard_bounds_partition():
    from alpine_amd.sharded import shard_bounds
    for n in (1, 7, 64, 200000, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 16
            assert all(s % 8 == 0 or s == n for s, _ in b)


def test_loop_call_sequence():
    from alpine_amd.sharded import ShardedLoop
    log = []

    class E:
        def iter_begin(self): log.append("b")
        def iter_end(self, update=True): log.append("e1" if update else "e0")

    class Cm:
        def all_reduce(self): log.append("r")

    ShardedLoop(E(), Cm()).run(2, with_loss=True)
    assert log == ["b", "r", "e1", "b", "r", "e1", "b", "r", "e0"]
    log.clear()
    ShardedLoop(E(), Cm()).run(1, with_loss=False)
    assert log == ["b", "r", "e1"]


def test_loop_call_sequence_use_als():
    """use_als: after the big all-reduce, one K x K all-reduce per component group except the first."""
    from alpine_amd.sharded import ShardedLoop
    log = []

    class E:
        def iter_begin(self): log.append("b")
        def iter_end(self, update=True): log.append("e1" if update else "e0")
        def als_begin(self): log.append("a")
        def als_group_begin(self, g): log.append(f"g{g}")
        def als_group_end(self, g): log.append(f"u{g}")
        def reduce_block_hht(self): return (100, 16)

    class Cm:
        def all_reduce(self): log.append("r")
        def all_reduce_slice(self, off, n): log.append(f"s{off}:{n}")

    ShardedLoop(E(), Cm(), als_groups=3).run(1, with_loss=True)
    assert log == ["b", "r", "a", "g0", "u0", "g1", "s100:16", "u1", "g2", "s100:16", "u2", "b", "r", "e0"]
