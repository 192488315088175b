"""Cell-axis sharding of the MU loop over the GPUs of one node: one process per GPU, one
all-reduce (sum) of the packed reduce block per iteration over RCCL/xGMI.

Everything an iteration needs from OTHER shards is a sum over cells of quantities of the old H
(XH^T, HH^T, the B-update sums, the prediction-loss sums, ||X||^2); the engine writes its local
sums into one contiguous float32 block in ``iter_begin``; after the all-reduce every rank holds
the global sums and ``iter_end`` updates the replicated W, B and the local columns of H with no
further communication.  The reference has no counterpart (single device, main.py:70).

Two carriers for that one collective:

* **native** (default when every rank owns its own GPU): the library holds an RCCL communicator
  (``alpine_comm_init_rank``) and its composite entry points (``alpine_run``, ``alpine_iter``,
  ``alpine_batch_step``, ``alpine_epoch_loss``) enqueue ``ncclAllReduce`` on the ctx stream themselves --
  the per-iteration loop is plain C, torch only carries the 128-byte unique id at start-up
  (``attach_native_comm``).
* **torch** (``TorchDistComm``): ``torch.distributed.all_reduce`` on a tensor that aliases the reduce
  block, between the split entry points (``iter_begin`` / ``iter_end``): any backend, e.g. gloo for the
  rehearsal where several ranks share one GPU, which RCCL refuses.

``ShardedLoop`` only orchestrates; the engine is ``_native.NativeShard`` in production.  Tests
drive the same loop with a CPU engine over gloo to cover the N>1 logic without GPUs.
"""
from __future__ import annotations

from typing import Optional, Tuple


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


def check_shardable(n_total: int, world: int) -> None:
    """Raise the SAME error on every rank when some rank would get no cells (interior cuts are rounded to multiples of
    8, so small N over many ranks leaves empty blocks): a rank-local failure before a collective would hang the peers."""
    empty = [r for r in range(world) if shard_bounds(n_total, world, r)[0] >= shard_bounds(n_total, world, r)[1]]
    if empty:
        raise ValueError(f"cannot shard {n_total} cells over {world} ranks: rank(s) {empty} would hold no cells "
                         f"(need at least 8 cells per rank)")


def all_ranks_ok(dist, ok: bool, what: str, err: Optional[BaseException] = None) -> None:
    """Agree on a rank-local outcome before the next collective: every rank raises if any rank failed."""
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, (bool(ok), "" if err is None else f"{type(err).__name__}: {err}"))
    bad = [(r, m) for r, (f, m) in enumerate(flags) if not f]
    if bad:
        if err is not None:
            raise err
        raise RuntimeError(f"{what} failed on rank(s) " + ", ".join(f"{r} ({m})" for r, m in bad))


def native_comm_possible(dist, device_index: int) -> bool:
    """RCCL needs one distinct GPU per rank (it refuses two ranks on one device); single node assumed."""
    devs = [None] * dist.get_world_size()
    dist.all_gather_object(devs, int(device_index))
    return len(set(devs)) == len(devs)


def ensure_dmabuf_ipc() -> None:
    """Multi-process GPU work on this platform needs dmabuf IPC (RCCL's and torch's cross-process handles fail with
    "hipIpcGetMemHandle: invalid argument" in the legacy mode).  The HIP runtime reads the variable when it starts, so the
    sharded entry points set the default before their first GPU call; a value chosen by the caller is left alone."""
    import os
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        try:
            import torch
            late = torch.cuda.is_initialized()
        except Exception:       # noqa: BLE001
            late = False
        if late:
            os.environ["_ALPINE_AMD_IPC_DEFAULT_SET_LATE"] = "1"     # the runtime is already up: it never saw the value (dmabuf_ipc_problem)


def dmabuf_ipc_problem(gpu_runtime_started: bool):
    """None, or what to tell the user: the HIP runtime reads HSA_ENABLE_IPC_MODE_LEGACY when it starts, so once this process has made a GPU
    call without it the default set by ensure_dmabuf_ipc comes too late -- and RCCL / torch then fail with "hipIpcGetMemHandle: invalid
    argument" in the first collective.  (A runtime that has not started yet will see the default: no problem.)"""
    import os
    v = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if v == "0":
        # set by the user, or by ensure_dmabuf_ipc -- which only helps if the runtime had not started when it was set; the caller passes
        # whether it had, and ensure_dmabuf_ipc records whether the value is its own late default
        if gpu_runtime_started and os.environ.get("_ALPINE_AMD_IPC_DEFAULT_SET_LATE") == "1":
            return ("this process initialised the GPU runtime before HSA_ENABLE_IPC_MODE_LEGACY=0 was set: multi-process GPU work on this platform "
                    "needs dmabuf IPC.  export HSA_ENABLE_IPC_MODE_LEGACY=0 BEFORE the first GPU call (before torch.cuda.set_device / "
                    "init_process_group), e.g. in the launcher's environment")
        return None
    return (f"HSA_ENABLE_IPC_MODE_LEGACY={v!r}: multi-process GPU work on this platform needs dmabuf IPC (RCCL and torch fail with "
            f"'hipIpcGetMemHandle: invalid argument' otherwise).  export HSA_ENABLE_IPC_MODE_LEGACY=0 before the first GPU call")


def attach_native_comm(engine, dist, group=None) -> None:
    """Give ``engine`` an RCCL communicator over the ranks of the torch.distributed group: rank 0 draws the unique id
    through the library, torch broadcasts its 128 bytes (control plane only), every rank joins.  Collective: a failure on
    any rank (rank 0 drawing the id, any rank joining) raises on EVERY rank, so nobody is left blocked in a later collective."""
    from . import _native
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [None]
    if rank == 0:
        try:
            box = [("id", _native.comm_unique_id())]
        except Exception as e:      # noqa: BLE001 -- librccl missing, ncclGetUniqueId error: tell the peers instead of leaving them in the broadcast
            box = [("error", f"{type(e).__name__}: {e}")]
    dist.broadcast_object_list(box, src=0, group=group)
    kind, payload = box[0]
    if kind != "id":
        raise RuntimeError(f"rank 0 could not create the communicator id: {payload}")
    err = None
    try:
        engine.comm_init(payload, world, rank)
    except Exception as e:          # noqa: BLE001 -- agree with the peers before raising
        err = e
    all_ranks_ok(dist, err is None, "alpine_comm_init_rank", err)


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


class NativeComm:
    """The library's own communicator behind the same two calls as ``TorchDistComm`` (for callers that keep the split
    begin / end entry points, e.g. the mini-batch epochs); ``ShardedLoop`` skips it and calls ``engine.run`` directly."""

    native = True

    def __init__(self, engine):
        self.engine = engine

    def all_reduce(self) -> None:
        self.engine.comm_all_reduce(0, None)

    def all_reduce_slice(self, offset: int, n: int) -> None:
        self.engine.comm_all_reduce(offset, n)


class ShardedLoop:
    def __init__(self, engine, comm, als_groups: int = 0):
        """``als_groups`` > 0: block-coordinate branch (use_als) with that many component groups (covariates + 1); the
        group loop needs H H^T of all cells after every group, i.e. one small extra all-reduce per group."""
        self.engine = engine
        self.comm = comm
        self.als_groups = als_groups
        self.native = bool(getattr(comm, "native", False))
        self._hht = engine.reduce_block_hht() if als_groups > 0 and not self.native else None

    def step(self, update: bool = True) -> None:
        if self.native:
            self.engine.iter(update)          # begin, ncclAllReduce on the ctx stream, end (+ group exchanges): all in C
            return
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
        if self.native:
            self.engine.run(n_iters, with_loss=with_loss)
            return
        for _ in range(n_iters):
            self.step(True)
        if with_loss and n_iters > 0:
            self.step(False)        # loss row of the last iteration needs the sums of the final H
