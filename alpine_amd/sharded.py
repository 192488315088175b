"""Cell-axis sharding of the MU loop over the GPUs of one node: one process per GPU, one
all-reduce (sum) of the packed reduce block per iteration over RCCL/xGMI.

Everything an iteration needs from OTHER shards is a sum over cells of quantities of the old H
(XH^T, HH^T, the B-update sums, the prediction-loss sums, ||X||^2); the engine writes its local
sums into one contiguous float32 block in ``iter_begin``; after the all-reduce every rank holds
the global sums and ``iter_end`` updates the replicated W, B and the local columns of H with no
further communication.  The reference has no counterpart (single device, main.py:70).

``ShardedLoop`` only orchestrates; the engine is ``_native.NativeShard`` in production.  Tests
drive the same loop with a CPU engine over gloo to cover the N>1 logic without GPUs.
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous cell block of rank r: [N*r/P, N*(r+1)/P) with interior boundaries rounded to multiples of 8 cells
    (the bf16 path packs 8 cells per 16-byte granule; 8 cells of imbalance are nothing)."""
    def cut(r: int) -> int:
        if r <= 0:
            return 0
        if r >= world:
            return n_total
        return min(n_total, ((n_total * r) // world + 4) // 8 * 8)
    return cut(rank), cut(rank + 1)


class TorchDistComm:
    """Sum-all-reduce of a torch tensor that aliases the engine's reduce block.  With the nccl
    backend (= RCCL on ROCm) the collective is enqueued relative to the current torch stream,
    which is the stream the engine was created on, so no host synchronisation is needed."""

    def __init__(self, block, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.block = block
        self.group = group

    def all_reduce(self) -> None:
        self._dist.all_reduce(self.block, op=self._dist.ReduceOp.SUM, group=self.group)

    def all_reduce_slice(self, offset: int, n: int) -> None:
        """Sum-all-reduce of block[offset : offset + n] only (the H H^T slot between the groups of the use_als branch)."""
        self._dist.all_reduce(self.block[offset:offset + n], op=self._dist.ReduceOp.SUM, group=self.group)


class ShardedLoop:
    def __init__(self, engine, comm, als_groups: int = 0):
        """``als_groups`` > 0: block-coordinate branch (use_als) with that many component groups (covariates + 1); the
        group loop needs H H^T of all cells after every group, i.e. one small extra all-reduce per group."""
        self.engine = engine
        self.comm = comm
        self.als_groups = als_groups
        self._hht = engine.reduce_block_hht() if als_groups > 0 else None

    def step(self, update: bool = True) -> None:
        self.engine.iter_begin()
        self.comm.all_reduce()
        if not update or self.als_groups == 0:
            self.engine.iter_end(update)
            return
        self.engine.als_begin()
        for grp in range(self.als_groups):
            self.engine.als_group_begin(grp)
            if grp > 0:
                self.comm.all_reduce_slice(*self._hht)
            self.engine.als_group_end(grp)

    def run(self, n_iters: int, with_loss: bool = True) -> None:
        for _ in range(n_iters):
            self.step(True)
        if with_loss and n_iters > 0:
            self.step(False)        # loss row of the last iteration needs the sums of the final H
