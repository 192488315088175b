"""``ALPINE`` -- host-side mirror of the reference's model class for the path this package
accelerates: ``ALPINE(**params).fit(adata, covariate_keys=...)`` / ``store_embeddings`` with the
full-batch multiplicative-update loop running on one MI355X (or cell-sharded over several, one
process per GPU) through libalpine_hip.so.

Mirrors alpine/main.py:46-147, :303-434 (constructor, validators with the same messages and
quirks, fit orchestration, result dict, AnnData write-back).  The numerical loop itself
(main.py:486-676) is NOT here -- it is the HIP library.  There is no CPU execution path in this
class: ``device="cpu"`` raises.
"""
from __future__ import annotations

import math
from copy import copy, deepcopy
from typing import Dict, List, Optional, Union

import numpy as np
import pandas as pd
import torch

from . import _native
from .anndata_compat import is_anndata
from .encoder import FeatureEncoders
from .sharded import (NativeComm, ShardedLoop, TorchDistComm, all_ranks_ok, attach_native_comm, check_shardable, dmabuf_ipc_problem,
                      ensure_dmabuf_ipc, native_comm_possible, shard_bounds)

Float32Array = np.ndarray


def all_nonnegative(X: np.ndarray) -> bool:
    """``np.all(X >= 0)`` (main.py:399, :682 -- False for any negative element or NaN) without the boolean temporary and on
    several threads: row blocks of ~64 MB, one ``min`` each (numpy releases the GIL in reductions).  At 20 000 x 200 000 the
    plain form takes several seconds on the host -- more than a 200-iteration fit takes on the device."""
    if X.size == 0:
        return True
    if X.ndim == 2 and 0 < X.strides[0] < X.strides[1]:
        X = X.T                                   # column-major input: cut along the long stride
    if X.ndim != 2 or X.shape[0] < 2 or X.nbytes < (64 << 20):
        return bool(X.min() >= 0)
    import os
    from concurrent.futures import ThreadPoolExecutor
    rows = max(1, (64 << 20) // max(1, X.strides[0] if X.strides[0] > 0 else X.shape[1] * X.itemsize))
    blocks = [(r, min(X.shape[0], r + rows)) for r in range(0, X.shape[0], rows)]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1, len(blocks))) as pool:
        return all(pool.map(lambda ab: bool(X[ab[0]:ab[1]].min() >= 0), blocks))


def _skip_cpu_generator(draws: int) -> None:
    """Advance torch's global CPU generator by ``draws`` 32-bit outputs of its mt19937 without producing anything: the state
    (CPUGeneratorImplState: seed, left, seeded, next, 624 state words in 64-bit slots, normal-distribution cache) is handed
    to numpy's MT19937, which steps it at ~1 ns per output, and written back.  Raises on any layout it does not recognise."""
    b = torch.get_rng_state().numpy().copy()
    if b.size != 5056:
        raise RuntimeError("unexpected CPU generator state size")
    left = int(np.frombuffer(b[8:12].tobytes(), dtype=np.int32)[0])
    nxt = int(np.frombuffer(b[16:24].tobytes(), dtype=np.uint64)[0])
    if not ((left == 1 and nxt == 0) or (1 <= left <= 624 and left + nxt == 625)):
        raise RuntimeError("unexpected mt19937 position fields")
    words = np.frombuffer(b[24:24 + 624 * 8].tobytes(), dtype=np.uint64)
    if int(words.max()) >> 32:
        raise RuntimeError("unexpected mt19937 state words")
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": words.astype(np.uint32), "pos": 624 if nxt == 0 else nxt}}
    while draws > 0:
        k = min(draws, 1 << 22)
        bg.random_raw(k)
        draws -= k
    st = bg.state["state"]
    pos = int(st["pos"])
    b[24:24 + 624 * 8] = np.frombuffer(st["key"].astype(np.uint64).tobytes(), dtype=np.uint8)
    b[16:24] = np.frombuffer(np.uint64(pos).tobytes(), dtype=np.uint8)
    b[8:12] = np.frombuffer(np.int32(625 - pos).tobytes(), dtype=np.uint8)
    torch.set_rng_state(torch.from_numpy(b))


_RANDPERM_SKIP_OK: Optional[bool] = None


def replay_randperms(n: int, times: int) -> None:
    """Leave the global torch generator where ``times`` calls of ``torch.randperm(n)`` would (sampling.py:14: one per
    iteration of the reference's loop).  ATen's CPU randperm draws one 32-bit output per swap, n - 1 per call, so the
    generator can be stepped directly (17x faster at n = 200 000: 300 permutations cost 1.5 s otherwise).  The shortcut is
    checked against the real call once per process on a small n and is never used if it disagrees or cannot read the state."""
    global _RANDPERM_SKIP_OK
    if n < 2 or times <= 0:
        for _ in range(max(0, times)):
            torch.randperm(n)
        return
    if _RANDPERM_SKIP_OK is None:
        keep = torch.get_rng_state()
        try:
            torch.manual_seed(987654321)
            torch.rand(3)
            s0 = torch.get_rng_state()
            for _ in range(3):
                torch.randperm(701)
            want = torch.get_rng_state()
            torch.set_rng_state(s0)
            _skip_cpu_generator(3 * 700)
            _RANDPERM_SKIP_OK = bool(torch.equal(want, torch.get_rng_state()))
        except Exception:       # noqa: BLE001 -- any surprise in the state layout: keep the plain replay
            _RANDPERM_SKIP_OK = False
        finally:
            torch.set_rng_state(keep)
    if _RANDPERM_SKIP_OK:
        keep = torch.get_rng_state()
        try:
            _skip_cpu_generator((n - 1) * times)
            return
        except Exception:       # noqa: BLE001
            torch.set_rng_state(keep)
    for _ in range(times):
        torch.randperm(n)


def draw_initial_factors(random_state: int, eps: float, n_features: int, n_samples: int,
                         n_all_components: List[int], cov_levels: List[int]):
    """The reference's initial draws (main.py:440, :454-470) on the torch CPU generator: reseed,
    then every W_j (G x k_j), every H_j (k_j x N), every B_i (C_i x k_i), each ``U[0,1)`` clamped
    from below at eps.  Returns W (G x K), H (K x N), [B_i] as float32 numpy arrays."""
    torch.manual_seed(random_state)
    Ws = [torch.rand((n_features, k), dtype=torch.float32).clamp(min=eps) for k in n_all_components]
    Hs = [torch.rand((k, n_samples), dtype=torch.float32).clamp(min=eps) for k in n_all_components]
    Bs = [torch.rand((c, k), dtype=torch.float32).clamp(min=eps) for c, k in zip(cov_levels, n_all_components)]
    return torch.cat(Ws, dim=1).numpy(), torch.cat(Hs, dim=0).numpy(), [b.numpy() for b in Bs]


def _maybe_two_bf16_planes(X: np.ndarray, max_rows: int = 64) -> bool:
    """Cheap host-side screen for x_dtype="auto": could X be stored exactly as two bf16 planes (16 significant bits: integer counts
    < 65 536 and the like)?  Looks at up to `max_rows` rows spread over the matrix: False as soon as one float32 value has any of its low
    8 significand bits set -- then the exact-split storage would be refused by the library anyway (alpine_finalize_X) and trying it
    first would cost a whole extra upload.  True only means "worth trying": the library still checks every element."""
    if X.size == 0:
        return True
    n = X.shape[0]
    rows = np.unique(np.linspace(0, n - 1, num=min(n, max_rows)).astype(np.int64))
    sample = np.ascontiguousarray(X[rows], dtype=np.float32)
    return not bool((sample.view(np.uint32) & np.uint32(0xFF)).any())


def _parse_device(device: str) -> int:
    d = torch.device(device)
    if d.type == "cpu":
        raise ValueError(
            "alpine_amd runs the fit loop on an MI355X through libalpine_hip.so and has no CPU path; "
            "use device='cuda' (or 'cuda:<i>' / 'hip').")
    if d.type not in ("cuda", "hip"):
        raise ValueError(f"unsupported device {device!r}")
    return d.index if d.index is not None else -1


class ALPINE:
    def __init__(
        self,
        n_components: int,
        n_covariate_components: List[int],
        lam: List[float],
        orth_W: float = 0.0,
        alpha_W: float = 0.0,
        l1_ratio_W: float = 0.0,
        use_als: bool = False,
        scale_needed: bool = True,
        loss_type: str = "kl-divergence",
        device: str = "cuda",
        eps: float = 1e-6,
        random_state: int = 42,
        shard_cells: Union[bool, str] = False,
        x_dtype: str = "auto",
        shard_comm: str = "auto",
        keep_resident: bool = False,
        devices: Optional[List[int]] = None,
    ):
        self.n_components = n_components
        self.n_covariate_components = n_covariate_components
        self.lam = lam
        self.orth_W = orth_W
        self.alpha_W = alpha_W
        self.l1_ratio_W = l1_ratio_W
        self.use_als = use_als
        self.scale_needed = scale_needed
        ds = str(device)
        if ds == "hip" or ds.startswith("hip:"):
            ds = "cuda" + ds[3:]
        self.device = torch.device(ds)
        self.loss_type = loss_type
        self.eps = eps
        self.random_state = random_state
        # extension (not in the reference): shard the cell axis over the ranks of the default torch.distributed process
        # group, one process per GPU.  False: single device.  True: every rank passes the SAME full adata and keeps its
        # contiguous block of cells; all ranks return the full factors.  "local": every rank passes ONLY its own cells
        # (rank order = cell order; a 20k x 1M matrix never has to exist in one process) and gets its own columns of H.
        if shard_cells not in (False, True, "local"):
            raise ValueError("shard_cells must be False, True or 'local'")
        self.shard_cells = shard_cells
        # carrier of the per-iteration all-reduce when shard_cells is on: "native" = the library's own RCCL communicator
        # (ncclAllReduce enqueued by the C loop itself), "torch" = torch.distributed.all_reduce on the process group's
        # backend, "auto" = native when every rank owns a distinct GPU, else torch (e.g. gloo with ranks sharing a GPU)
        if shard_comm not in ("auto", "native", "torch"):
            raise ValueError("shard_comm must be 'auto', 'native' or 'torch'")
        self.shard_comm = shard_comm
        # extension (off by default, as in the reference, which frees everything at the end of fit): keep_resident=True leaves
        # the engine -- with both float32 copies of X in HBM, ~30 GiB at 20k x 200k -- alive after a single-device fit, so
        # that a following compute_loss(adata) / transform(adata) ON THE SAME adata.X does not upload X again (one upload
        # costs as much as 50-100 iterations at that size).  The resident copy is used only while adata.X is the same
        # object with the same buffer, shape, strides, dtype AND position-sensitive digest of every byte; any mismatch releases it.
        # release() frees it explicitly.
        # extension (the reference's fit is ONE blocking call in ONE process, main.py:82-147): devices=[0, 1, ..., P-1] shards
        # the cell axis over those GPUs of this process -- no launcher, no torch.distributed.  One engine and one host thread per device,
        # the library's own RCCL communicator over them (ncclCommInitAll), the per-iteration all-reduce enqueued by each engine's C loop.
        # None (default): the single device named by `device`.  transform / compute_loss run on devices[0].
        if devices is not None:
            if not isinstance(devices, (list, tuple)) or len(devices) == 0 or any(not isinstance(d, int) or isinstance(d, bool) or d < 0 for d in devices):
                raise ValueError("devices must be a non-empty list of non-negative GPU ordinals, e.g. [0, 1, 2, 3]")
            import os
            if len(set(devices)) != len(devices) and os.environ.get("ALPINE_AMD_TEST_SHARED_DEVICE") != "1":
                # (RCCL refuses two ranks on one GPU; the variable is the rehearsal switch of the tests, which preload a stand-in communicator)
                raise ValueError("devices must not name a GPU twice")
            if shard_cells:
                raise ValueError("devices=[...] (one process, several GPUs) and shard_cells (one process per GPU) are alternatives: use one of them")
            devices = [int(d) for d in devices]
            self.device = torch.device(f"cuda:{devices[0]}")
        self.devices = devices
        if not isinstance(keep_resident, bool):
            raise TypeError("keep_resident must be a boolean.")
        self.keep_resident = keep_resident
        self._resident = None
        # extension: storage / matrix-pipe mode of the two sweeps (the reference has float32 only).
        #   "auto" (default) "split" when X allows it (integer counts: bit-identical to "x3" at half the memory and 1.8 x the
        #                   iterations per second), else "x3".  A sample of X decides up front whether "split" is worth trying
        #                   (_maybe_two_bf16_planes), so data with full significands costs no second upload.
        #   "x3"            X float32 in HBM; every product is formed from the exact bf16 planes of both factors on the
        #                   bf16 matrix pipe (six plane products, float32 accumulate): float32-grade results for ANY X at
        #                   HBM-bound instead of float32-MFMA-bound speed.
        #   "f32"           the float32 MFMA (v_mfma_f32_32x32x2_f32) on the same float32 storage.
        #   "split"         X stored as 1-2 bf16 planes that sum EXACTLY to the input (integer counts): float32-grade
        #                   results at half the memory and traffic; raises if X has more than 16 significant bits.
        #   "bf16"          X and the operand copies of W/H rounded to bf16 (fp32 accumulation, fp32 masters): NOT
        #                   float32-grade, tolerance in DESIGN.md.
        if x_dtype not in ("f32", "x3", "bf16", "split", "auto"):
            raise ValueError("x_dtype must be 'f32', 'x3', 'bf16', 'split' or 'auto'")
        self.x_dtype = x_dtype

        self._validate_init_args()

        self.n_all_components = self.n_covariate_components + [self.n_components]     # main.py:79
        self.total_components = sum(self.n_all_components)                            # main.py:80

    # ------------------------------------------------------------------ fit (main.py:82-147)
    def fit(
        self,
        adata,
        covariate_keys: List[str],
        batch_size: Optional[int] = None,
        max_iter: Optional[int] = None,
        sampling_method: str = "random",
        verbose: bool = False,
    ) -> "ALPINE":
        self._validate_fit_args(adata, covariate_keys, batch_size, max_iter, sampling_method, verbose)
        self.feature_names = adata.var_names.tolist()
        self.n_features = adata.shape[1]
        self.covariate_keys = covariate_keys
        self.sampling_method = sampling_method
        self.verbose = verbose

        n_sample = adata.shape[0]
        self.fe = FeatureEncoders(covariate_keys)
        Y = self.fe.fit_transform(adata.obs, merge_categories=self._category_merger())   # list of N x C_i float32 (main.py:108-109)
        if len(Y) == 0:
            # the reference cannot fit without covariates: _fit's prologue indexes Ys[0] (sampling.py:40, called from
            # main.py:496) and raises this very error before the first iteration; reproduced, not "fixed"
            raise IndexError("list index out of range")
        if len(self.lam) < len(Y):
            # len(lam) is never validated by the reference (main.py:322-381); its first iteration then fails on self.lam[i]
            raise IndexError("list index out of range")
        n_global = self._global_cells(n_sample)                    # == n_sample unless shard_cells="local"
        self.batch_size = batch_size if batch_size is not None else n_global
        self._check_supported(n_global)

        self.release()                                  # a previous fit's resident engine
        sess = self._open_session(adata.X, Y)           # X is uploaded ONCE, also when the warm-up runs first
        try:
            if max_iter is None:
                # main.py:116-129: 200-iteration warm-up, Kneedle elbow on log10(recon loss); the final run starts again from
                # the seeded initial factors (the reference re-runs _initialize_matrices, main.py:135) on the SAME resident X
                warm = self._run_session(sess, 200, scale=False)
                self.max_iter = self._compute_best_iter(warm["loss_history"]["reconstruction loss"].values)
                del warm
            else:
                self.max_iter = max_iter
            res = self._run_session(sess, self.max_iter, scale=self.scale_needed)
        except BaseException:
            self._close_session(sess)
            raise
        if self.keep_resident and not sess["sharded"] and not sess.get("multi") and sess["x_dtype"] in ("f32", "x3") and sess["batch_capacity"] == 0:
            self._resident = dict(eng=sess["eng"], X=adata.X, fingerprint=self._x_fingerprint(adata.X), x_dtype=sess["x_dtype"])
        else:
            self._close_session(sess)
        self.loss_history = res["loss_history"]
        offs = np.cumsum([0] + self.n_all_components)
        X32 = adata.X if adata.X.dtype == np.float32 else adata.X.astype(np.float32)
        self.matrices: Dict[str, Union[Float32Array, List[Float32Array]]] = {
            "X": X32.T,                                                     # G x N view, not a device read-back (main.py:38)
            "Ys": [np.ascontiguousarray(y.T) for y in Y],                   # C_i x N
            "Ws": [np.ascontiguousarray(res["W"][:, offs[j]:offs[j + 1]]) for j in range(len(self.n_all_components))],
            "Hs": [np.ascontiguousarray(res["H"][offs[j]:offs[j + 1]]) for j in range(len(self.n_all_components))],
            "Bs": res["Bs"],
        }
        self.fit_info = res["info"]
        self.store_embeddings(adata)
        return self

    @staticmethod
    def _close_session(sess: dict) -> None:
        for e in sess.get("engs") or [sess["eng"]]:
            e.close()
        if sess.get("pool") is not None:
            sess["pool"].shutdown(wait=False)

    def _dist_world(self):
        import torch.distributed as dist
        if bool(self.shard_cells) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist, dist.get_rank(), dist.get_world_size()
        return None, 0, 1

    def _global_cells(self, n_local: int) -> int:
        dist, _, world = self._dist_world()
        if dist is None or self.shard_cells != "local":
            return n_local
        sizes = [None] * world
        dist.all_gather_object(sizes, int(n_local))
        return int(sum(sizes))

    def _category_merger(self):
        """shard_cells='local': labels that occur only on other ranks must still get a one-hot column here."""
        dist, _, world = self._dist_world()
        if self.shard_cells != "local" or dist is None:
            return None

        def merge(cats: np.ndarray) -> np.ndarray:
            parts = [None] * world
            dist.all_gather_object(parts, cats.tolist())
            return np.unique(np.array([c for p in parts for c in p], dtype=object if cats.dtype == object else None))
        return merge

    # limits of libalpine_hip.so that the reference does not have ("Deviations"): checked here so that the
    # user gets a Python-side message before any device work, instead of a late native status
    MAX_TOTAL_COMPONENTS = 1024
    MAX_FAST_COMPONENTS = 128              # up to here: the fused MFMA path; above: the blocked path, ceil(K / 128) column blocks (float32 storage)
    MAX_COVARIATE_COMPONENTS = 64           # per covariate; their sum may reach the total
    MAX_COVARIATES = 16

    def _check_supported(self, n_sample: int) -> None:
        ks = list(self.n_covariate_components)
        if len(ks) > self.MAX_COVARIATES:
            raise NotImplementedError(f"more than {self.MAX_COVARIATES} covariates are not supported by the MI355X build (got {len(ks)})")
        if any(k > self.MAX_COVARIATE_COMPONENTS for k in ks):
            raise NotImplementedError(f"more than {self.MAX_COVARIATE_COMPONENTS} guided components for ONE covariate are not supported by the "
                                      f"MI355X build (got {max(ks)}; their sum over the covariates may reach the total)")
        if self.total_components > self.MAX_TOTAL_COMPONENTS:
            raise NotImplementedError(f"n_components + sum(n_covariate_components) = {self.total_components} > "
                                      f"{self.MAX_TOTAL_COMPONENTS} is not supported by the MI355X build")
        if self.total_components > self.MAX_FAST_COMPONENTS:
            # 128 < K <= 1024 runs on the blocked path (kernels_wide.hpp)
            if sum(ks) > self.MAX_FAST_COMPONENTS:
                raise NotImplementedError(f"with more than {self.MAX_FAST_COMPONENTS} components in total, sum(n_covariate_components) must be <= "
                                          f"{self.MAX_FAST_COMPONENTS} in the MI355X build (got {sum(ks)})")
            if self.x_dtype not in ("x3", "f32", "auto"):
                raise NotImplementedError(f"more than {self.MAX_FAST_COMPONENTS} components need float32 storage: x_dtype='x3', 'f32' or 'auto'")
        if self.sampling_method not in ("random", "weighted"):
            raise ValueError(f"Unknown sampling method: {self.sampling_method}. Only 'weighted', and 'random' are supported.")
        if self._uses_batches(n_sample):
            if self.use_als and (self.shard_cells or (self.devices is not None and len(self.devices) > 1)):
                raise NotImplementedError("use_als=True with mini-batches is single-device (sharded: full batch only)")
            if self.x_dtype not in ("f32", "x3", "auto"):
                raise NotImplementedError("mini-batch / weighted sampling needs float32 storage: x_dtype='f32', 'x3' or 'auto'")

    def _uses_batches(self, n_sample: int) -> bool:
        """Full batch with the 'random' permutation is the in-place fast path (the permutation only re-orders sums);
        anything else changes the mathematics (stochastic W updates, or duplicates under weighted resampling) and goes
        through the gathered mini-batch view."""
        return self.batch_size < n_sample or self.sampling_method == "weighted"

    # ------------------------------------------------------------------ resident engine
    @staticmethod
    def _x_fingerprint(X: np.ndarray):
        """Identity of an input matrix: buffer address, shape, strides, dtype and a POSITION-SENSITIVE digest of every byte -- one
        64-bit xxh3 (or, without the xxhash package, one zlib.crc32) per 64 MB block, blocks hashed on several threads (both release
        the GIL) -- so that any in-place edit of X is seen: a value change, but also a row swap or a shuffle, which the plain
        per-block sums of round 3 could not see (`X[[0, 1]] = X[[1, 0]]` left them unchanged, and compute_loss / transform
        then ran on the stale copy in HBM).  The reference re-reads adata.X on every call; ~0.15 s at 20k x 200k with xxh3 on 16 threads."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            from xxhash import xxh3_64_intdigest as digest
        except ImportError:                       # noqa: BLE001 -- slower, same guarantee per block (32 bits)
            from zlib import crc32 as digest
        A = X if X.flags.c_contiguous else (X.T if X.T.flags.c_contiguous else np.ascontiguousarray(X))
        flat = A.reshape(-1).view(np.uint8)
        step = 64 << 20
        blocks = [(a, min(flat.shape[0], a + step)) for a in range(0, flat.shape[0], step)]
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1, len(blocks)))) as pool:
            digests = tuple(pool.map(lambda ab: int(digest(flat[ab[0]:ab[1]])), blocks))
        return (X.ctypes.data, X.shape, X.strides, X.dtype.str, digests)

    def _resident_engine_for(self, X: np.ndarray):
        r = self._resident
        if r is None:
            return None
        if r["X"] is not X or r["fingerprint"] != self._x_fingerprint(X):
            self.release()                            # another matrix, or the fitted one was edited: the HBM copy is stale
            return None
        return r["eng"]

    def release(self) -> None:
        """Free the engine (and the copies of X in HBM) that fit() left resident for compute_loss / transform."""
        r, self._resident = getattr(self, "_resident", None), None
        if r is not None:
            r["eng"].close()

    def __del__(self):
        try:
            self.release()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    # ------------------------------------------------------------------ one process, several GPUs (devices=[...])
    def _open_session_devices(self, X_cells_genes: np.ndarray, Y: List[np.ndarray]) -> dict:
        """devices=[d_0 .. d_{P-1}]: engine r on GPU d_r holds the contiguous cell block r (the same cuts as the one-process-per-GPU form),
        uploads run on P host threads, and ONE ncclCommInitAll attaches the library's communicator to the P engines (rank r = engine r)."""
        from concurrent.futures import ThreadPoolExecutor
        N_total, G = X_cells_genes.shape
        devs = list(self.devices)
        P = len(devs)
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: alpine_amd needs an MI355X (there is no CPU fallback)")
        n_vis = torch.cuda.device_count()
        if max(devs) >= n_vis:
            raise ValueError(f"devices={devs}: this process sees {n_vis} GPU(s)")
        check_shardable(N_total, P)
        bounds = [shard_bounds(N_total, P, r) for r in range(P)]
        uses_batches = self._uses_batches(N_total)
        cov_levels = [y.shape[1] for y in Y]
        x_dtype = self.x_dtype
        if x_dtype == "auto":
            no_split = uses_batches or self.total_components > self.MAX_FAST_COMPONENTS or not _maybe_two_bf16_planes(X_cells_genes)
            x_dtype = "x3" if no_split else "split"
        batch_capacity = min(self.batch_size, N_total) if uses_batches else 0

        def make_engine(r: int, dtype: str):
            c0, c1 = bounds[r]
            e = _native.NativeShard(n_genes=G, n_cells=c1 - c0, n_components=self.n_components, cov_components=self.n_covariate_components,
                                    cov_levels=cov_levels, lam=self.lam, orth_W=self.orth_W, alpha_W=self.alpha_W, l1_ratio_W=self.l1_ratio_W,
                                    eps=self.eps, loss_type=self.loss_type, device_id=devs[r], x_dtype=dtype, batch_capacity=batch_capacity,
                                    use_als=self.use_als)
            try:
                chunk = max(8, ((1 << 28) // (4 * G)) // 8 * 8)
                for r0 in range(c0, c1, chunk):
                    r1 = min(c1, r0 + chunk)
                    e.upload_X_host(np.ascontiguousarray(X_cells_genes[r0:r1], dtype=np.float32), _native.X_CELLS_BY_GENES, r0 - c0)
                e.finalize_X()
            except _native.AlpineNativeError as err:
                e.close()
                if self.x_dtype == "auto" and dtype == "split" and err.code == -5:
                    return None                          # this shard's X is not bf16-plane exact: every shard takes x3 (below)
                raise
            except Exception:
                e.close()
                raise
            return e

        pool = ThreadPoolExecutor(max_workers=P)
        engs: List = []
        try:
            def build(dtype):
                futs = [pool.submit(make_engine, r, dtype) for r in range(P)]
                out, first_err = [], None
                for f in futs:
                    try:
                        out.append(f.result())
                    except BaseException as e:      # noqa: BLE001 -- collect every engine before raising, so that none leaks
                        out.append(None)
                        first_err = first_err or e
                if first_err is not None:
                    for e in out:
                        if e is not None:
                            e.close()
                    raise first_err
                return out
            engs = build(x_dtype)
            if any(e is None for e in engs):             # "auto": some shard is not exact in two bf16 planes
                for e in engs:
                    if e is not None:
                        e.close()
                x_dtype = "x3"
                engs = build(x_dtype)
            self.x_dtype_used = x_dtype
            _native.comm_init_all(engs)
            counts = [e.comm_count() for e in engs]
            if any(c != (P, r) for r, c in enumerate(counts)):
                raise RuntimeError(f"the communicator over devices {devs} reports (ranks, rank) = {counts}")
            self.shard_comm_used, self.shard_comm_note = "native (one process, ncclCommInitAll)", None
            for r, e in enumerate(engs):
                c0, c1 = bounds[r]
                for i, y in enumerate(Y):
                    e.upload_Y(i, np.ascontiguousarray(y[c0:c1].T))
        except BaseException:
            for e in engs:
                if e is not None:
                    e.close()
            pool.shutdown(wait=False)
            raise
        return dict(multi=True, engs=engs, eng=engs[0], pool=pool, bounds=bounds, devices=devs, sharded=False, N_total=N_total, G=G,
                    cov_levels=cov_levels, Y=Y, x_dtype=x_dtype, batch_capacity=batch_capacity)

    def _run_session_devices(self, sess: dict, n_iter: int, scale: bool) -> dict:
        """The run of _run_session with one host thread per engine: every composite call (alpine_run, alpine_batch_step, alpine_epoch_loss)
        is made on all P engines at once and the ranks meet inside the library's all-reduce."""
        engs, pool, bounds = sess["engs"], sess["pool"], sess["bounds"]
        N_total, G, cov_levels, Y = sess["N_total"], sess["G"], sess["cov_levels"], sess["Y"]
        P = len(engs)
        W0, H0, B0 = draw_initial_factors(self.random_state, self.eps, G, N_total, self.n_all_components, cov_levels)
        self._rng_post_init = torch.get_rng_state()
        self._rng_replay = (N_total, n_iter)

        def on_all(fn):
            """fn(rank, engine) on every engine concurrently (P threads: all of them must be inside a collective together)"""
            futs = [pool.submit(fn, r, engs[r]) for r in range(P)]
            errs = []
            for f in futs:
                try:
                    f.result()
                except BaseException as e:      # noqa: BLE001
                    errs.append(e)
            if errs:
                raise errs[0]

        def init(r, e):
            e.set_factors(W0, H0, B0, h_col0=bounds[r][0])
            e.reset_losses()
        on_all(init)
        if sess["batch_capacity"] > 0:
            self._rng_replay = None
            weights = None
            if self.sampling_method == "weighted":
                weights = torch.as_tensor(self._balanced_joint_weights(Y), dtype=torch.double)
            bs = self.batch_size
            for _ in range(n_iter):
                # the reference's index stream, drawn ONCE per epoch on the calling thread (main.py:502-506); every engine takes its cells
                epoch = (torch.multinomial(weights, N_total, True) if weights is not None else torch.randperm(N_total)).numpy()

                def one_epoch(r, e, epoch=epoch):
                    c0, c1 = bounds[r]
                    for b0 in range(0, N_total, bs):
                        batch = epoch[b0:min(b0 + bs, N_total)]
                        e.batch_step(batch[(batch >= c0) & (batch < c1)] - c0)
                    e.epoch_loss()
                on_all(one_epoch)
        else:
            on_all(lambda r, e: e.run(n_iter, with_loss=True))
        if scale:
            on_all(lambda r, e: e.scale())
        parts = [None] * P

        def read(r, e):
            parts[r] = e.get_factors()
        on_all(read)
        W, _, Bs = parts[0]
        H = np.empty((self.total_components, N_total), dtype=np.float32)
        for r, (c0, c1) in enumerate(bounds):
            H[:, c0:c1] = parts[r][1]
        losses = engs[0].losses()
        info = engs[0].info()
        info_d = {f: getattr(info, f) for f, _ in info._fields_}
        info_d["devices"] = list(sess["devices"])
        colnames = ["total loss", "reconstruction loss"] + [f"prediction loss({k})" for k in self.covariate_keys]
        return dict(W=W, H=H, Bs=Bs, loss_history=pd.DataFrame(losses, columns=colnames), info=info_d)

    def _open_session(self, X_cells_genes: np.ndarray, Y: List[np.ndarray]) -> dict:
        """Create the engine(s) for this input and upload X and Y once (main.py:445-449); the factors are set per run."""
        if self.devices is not None:
            return self._open_session_devices(X_cells_genes, Y)
        N_total, G = X_cells_genes.shape
        dev_index = _parse_device(str(self.device))
        dist, rank, world = self._dist_world()
        sharded = dist is not None
        local_input = sharded and self.shard_cells == "local"
        if sharded:
            ensure_dmabuf_ipc()                        # before this process's first GPU call, if it has not made one yet
            # ... and if it HAS (torch.cuda.set_device / init_process_group in every normal torchrun script) without the variable, say so
            # on every rank now instead of "hipIpcGetMemHandle: invalid argument" from deep inside the first collective
            problem = dmabuf_ipc_problem(torch.cuda.is_initialized())
            all_ranks_ok(dist, problem is None, "HSA_ENABLE_IPC_MODE_LEGACY check", RuntimeError(problem) if problem else None)
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: alpine_amd needs an MI355X (there is no CPU fallback)")
        if dev_index < 0:
            dev_index = torch.cuda.current_device()
        if local_input:
            # every rank holds only its own cells: the global cell order is rank order
            sizes = [None] * world
            dist.all_gather_object(sizes, (int(N_total), int(G)))
            if any(g != G for _, g in sizes):
                raise ValueError("shard_cells='local': all ranks must pass the same genes")
            row0 = 0                                   # first local row of X / Y that belongs to this shard
            c0 = sum(n for n, _ in sizes[:rank])
            N_total = sum(n for n, _ in sizes)
            c1 = c0 + sizes[rank][0]
        else:
            if sharded:
                check_shardable(N_total, world)        # same error on every rank, before any collective
            c0, c1 = shard_bounds(N_total, world, rank)
            row0 = c0
        n_loc = c1 - c0
        uses_batches = self._uses_batches(N_total)          # N_total is the global cell count from here on
        cov_levels = [y.shape[1] for y in Y]

        x_dtype = self.x_dtype
        if x_dtype == "auto":
            # the exact-split storage needs full batches on K <= 128, does not serve a resident engine (compute_loss evaluates on the
            # float32 copy), and is only tried when a sample of this rank's rows says X may qualify
            no_split = (uses_batches or self.total_components > self.MAX_FAST_COMPONENTS or self.keep_resident
                        or not _maybe_two_bf16_planes(X_cells_genes[row0:row0 + n_loc]))
            if sharded:                                # one choice for all ranks (a rank-local one would desynchronise the collectives below)
                votes = [None] * world
                dist.all_gather_object(votes, bool(no_split))
                no_split = any(votes)
            x_dtype = "x3" if no_split else "split"
        kw = dict(n_genes=G, n_cells=n_loc, n_components=self.n_components,
                  cov_components=self.n_covariate_components, cov_levels=cov_levels, lam=self.lam,
                  orth_W=self.orth_W, alpha_W=self.alpha_W, l1_ratio_W=self.l1_ratio_W, eps=self.eps,
                  loss_type=self.loss_type, device_id=dev_index, x_dtype=x_dtype,
                  batch_capacity=(min(self.batch_size, N_total) if uses_batches else 0),     # a whole batch may fall into one shard
                  use_als=self.use_als)
        block, stream = None, None
        if sharded:
            # The engine and a torch-carried collective must share ONE explicit stream: the default stream's handle is 0,
            # which the C ABI reads as "create a private stream", and that would leave the all-reduce unordered with the kernels.
            with torch.cuda.device(dev_index):
                stream = torch.cuda.Stream()
                nfl = _native.reduce_block_floats(G, n_loc, self.n_components, self.n_covariate_components, cov_levels)
                with torch.cuda.stream(stream):
                    block = torch.zeros(nfl, dtype=torch.float32, device=f"cuda:{dev_index}")
                stream.synchronize()
                kw.update(stream=stream.cuda_stream, reduce_block=block.data_ptr())
                assert stream.cuda_stream != 0

        def make_engine(dtype):
            e = _native.NativeShard(**{**kw, "x_dtype": dtype})
            try:
                chunk = max(8, ((1 << 28) // (4 * G)) // 8 * 8)    # multiple of 8 cells (bf16 paths pack 8 rows per granule)
                for r0 in range(row0, row0 + n_loc, chunk):
                    r1 = min(row0 + n_loc, r0 + chunk)
                    e.upload_X_host(np.ascontiguousarray(X_cells_genes[r0:r1], dtype=np.float32), _native.X_CELLS_BY_GENES, r0 - row0)
                e.finalize_X()
            except Exception:
                e.close()
                raise
            return e

        eng, create_err = None, None
        try:
            try:
                eng = make_engine(x_dtype)
            except _native.AlpineNativeError as err:
                # "auto" only: X has more than 16 significant bits somewhere -> the pre-split storage does not apply; keep X
                # in float32 and split it inside the sweeps instead.  Sharded: the choice must be the same on every rank
                # (a rank-local exception would leave the peers blocked in the first all-reduce), so it is agreed on below.
                if not (self.x_dtype == "auto" and x_dtype == "split" and err.code == -5):
                    raise
                if not sharded:
                    x_dtype = "x3"
                    eng = make_engine(x_dtype)
        except Exception as e:          # noqa: BLE001
            if not sharded:
                raise
            create_err = e
        if sharded:
            all_ranks_ok(dist, create_err is None, "engine creation", create_err)
            if self.x_dtype == "auto" and x_dtype == "split":
                fits = [None] * world
                dist.all_gather_object(fits, eng is not None)
                if not all(fits):                       # some shard's X is not bf16-plane exact: all ranks take x3
                    if eng is not None:
                        eng.close()
                    x_dtype = "x3"
                    eng = make_engine(x_dtype)
        self.x_dtype_used = x_dtype
        try:
            comm = None
            if sharded:
                mode = self.shard_comm
                self.shard_comm_note = None
                if mode == "auto":
                    mode = "native" if native_comm_possible(dist, dev_index) else "torch"
                    if mode == "native":
                        # the library's communicator is the default carrier, but "auto" must not turn a communicator
                        # set-up failure into a failed fit: attach_native_comm raises on EVERY rank together, so all
                        # ranks take the torch.distributed carrier together
                        try:
                            attach_native_comm(eng, dist)
                        except Exception as e:          # noqa: BLE001
                            mode = "torch"
                            try:
                                eng.comm_destroy()      # a rank whose own join succeeded must not keep all-reducing natively
                            except Exception:           # noqa: BLE001 -- nothing was attached
                                pass
                            self.shard_comm_note = f"native communicator failed ({type(e).__name__}: {e}); fell back to torch.distributed"
                elif mode == "native":
                    attach_native_comm(eng, dist)
                comm = NativeComm(eng) if mode == "native" else TorchDistComm(block)
                self.shard_comm_used = mode
            for i, y in enumerate(Y):
                eng.upload_Y(i, np.ascontiguousarray(y[row0:row0 + n_loc].T))
        except BaseException:
            eng.close()
            raise
        return dict(eng=eng, comm=comm, block=block, stream=stream, dist=dist, rank=rank, world=world, sharded=sharded,
                    local_input=local_input, dev_index=dev_index, N_total=N_total, G=G, c0=c0, c1=c1, n_loc=n_loc,
                    cov_levels=cov_levels, Y=Y, x_dtype=x_dtype, batch_capacity=kw["batch_capacity"])

    def _run_session(self, sess: dict, n_iter: int, scale: bool) -> dict:
        """Initialise exactly like main.py:436-472, run the MU loop on the resident input, read the factors back."""
        if sess.get("multi"):
            return self._run_session_devices(sess, n_iter, scale)
        eng, comm, dist = sess["eng"], sess["comm"], sess["dist"]
        sharded, local_input, dev_index, stream = sess["sharded"], sess["local_input"], sess["dev_index"], sess["stream"]
        N_total, G, c0, c1, cov_levels, Y = sess["N_total"], sess["G"], sess["c0"], sess["c1"], sess["cov_levels"], sess["Y"]
        W0, H0, B0 = draw_initial_factors(self.random_state, self.eps, G, N_total, self.n_all_components, cov_levels)
        # The reference's loop also draws torch.randperm(N) once per iteration from the global generator
        # (sampling.py:14).  Full batch makes the permutation a numerical no-op, so it is not applied, but a later
        # unseeded transform() (main.py:687) continues that stream: remember how to advance it lazily.
        self._rng_post_init = torch.get_rng_state()
        self._rng_replay = (N_total, n_iter)
        eng.set_factors(W0, H0, B0, h_col0=c0)
        eng.reset_losses()
        if sess["batch_capacity"] > 0:
            if sharded:
                with torch.cuda.device(dev_index), torch.cuda.stream(stream):
                    self._run_epochs(eng, Y, N_total, n_iter, comm=comm, c0=c0, c1=c1,
                                     gather_labels=dist if local_input else None)
            else:
                self._run_epochs(eng, Y, N_total, n_iter)
        elif sharded:
            with torch.cuda.device(dev_index), torch.cuda.stream(stream):
                ShardedLoop(eng, comm, als_groups=(len(cov_levels) + 1 if self.use_als else 0)).run(n_iter, with_loss=True)
        elif self.verbose:
            # main.py:490-494, :669-671: tqdm bar with the objective loss.  The loop runs asynchronously on the device,
            # so the bar advances in chunks (one host sync per chunk instead of one per iteration).
            from tqdm import tqdm
            step = max(1, n_iter // 20)
            with tqdm(total=n_iter, desc="Iteration", ncols=100) as pbar:
                done = 0
                while done < n_iter:
                    k = min(step, n_iter - done)
                    eng.run(k, with_loss=True)
                    done += k
                    pbar.set_postfix({"objective loss": float(eng.losses()[-1, 0])})
                    pbar.update(k)
        else:
            eng.run(n_iter, with_loss=True)
        if scale:
            eng.scale()
        W, H_loc, Bs = eng.get_factors()
        losses = eng.losses()
        info = eng.info()
        info_d = {f: getattr(info, f) for f, _ in info._fields_}
        if sharded and not local_input:
            H = np.empty((self.total_components, N_total), dtype=np.float32)
            parts = [None] * sess["world"]
            dist.all_gather_object(parts, (c0, c1, H_loc))
            for a, b, h in parts:
                H[:, a:b] = h
        else:
            H = H_loc                                   # single device, or rank-local input: this rank's cells only
        colnames = ["total loss", "reconstruction loss"] + [f"prediction loss({k})" for k in self.covariate_keys]
        return dict(W=W, H=H, Bs=Bs, loss_history=pd.DataFrame(losses, columns=colnames), info=info_d)

    # ------------------------------------------------------ transform (main.py:149-185, :678-724)
    def transform(self, adata, n_iter: Optional[int] = None) -> None:
        if not hasattr(self, "matrices"):
            raise RuntimeError("Model is not trained yet. Please fit the model first.")
        if not is_anndata(adata):
            raise TypeError("adata must be an AnnData object.")
        if not isinstance(n_iter, (int, type(None))) or (n_iter is not None and n_iter <= 0):
            raise ValueError("n_iter must be a positive integer or None.")
        n_iter = n_iter if n_iter is not None else self.max_iter
        self._transform(adata, n_iter)

    def fit_transform(self, adata, covariate_keys: List[str], batch_size: Optional[int] = None,
                      max_iter: Optional[int] = None, sampling_method: str = "random", verbose: bool = False) -> None:
        self.fit(adata, covariate_keys, batch_size=batch_size, max_iter=max_iter, sampling_method=sampling_method,
                 verbose=verbose).transform(adata)

    def _advance_rng_like_reference_fit(self) -> None:
        """Bring the global torch generator to where the reference's fit() would have left it (init draws, then
        one randperm(N) per iteration), provided nobody else has drawn from it since our init draws."""
        replay = getattr(self, "_rng_replay", None)
        if replay is None:
            return
        self._rng_replay = None
        if torch.equal(torch.get_rng_state(), self._rng_post_init):
            n_total, n_iter = replay
            replay_randperms(n_total, n_iter)

    def _transform(self, adata, n_iter: int) -> None:
        X = adata.X
        if not all_nonnegative(X):
            raise ValueError("All elements in adata.X must be non-negative.")
        n_sample, G = X.shape
        dev_index = _parse_device(str(self.device))
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: alpine_amd needs an MI355X (there is no CPU fallback)")
        if dev_index < 0:
            dev_index = torch.cuda.current_device()
        self._advance_rng_like_reference_fit()
        # main.py:687-689: U[0,1) from the global generator, NOT reseeded, NOT clamped.  Rank-local inputs: every rank
        # draws the init of ALL cells (rank order = cell order) and keeps its columns, so the result equals the
        # single-process transform of the concatenated cells; cells are independent, no collective is needed.
        dist, rank, world = self._dist_world()
        if dist is not None and self.shard_cells == "local":
            sizes = [None] * world
            dist.all_gather_object(sizes, int(n_sample))
            c0 = sum(sizes[:rank])
            H0 = np.ascontiguousarray(torch.rand((self.total_components, sum(sizes)), dtype=torch.float32).numpy()[:, c0:c0 + n_sample])
        else:
            H0 = torch.rand((self.total_components, n_sample), dtype=torch.float32).numpy()
        W = np.ascontiguousarray(np.concatenate(self.matrices["Ws"], axis=1), dtype=np.float32)
        def make_engine(dtype):
            e = _native.NativeShard(n_genes=G, n_cells=n_sample, n_components=self.total_components, cov_components=[],
                                    cov_levels=[], lam=[], eps=self.eps, device_id=dev_index, transform_only=True, x_dtype=dtype)
            try:
                chunk = max(8, ((1 << 28) // (4 * G)) // 8 * 8)
                for r0 in range(0, n_sample, chunk):
                    e.upload_X_host(np.ascontiguousarray(X[r0:r0 + chunk], dtype=np.float32), _native.X_CELLS_BY_GENES, r0)
                e.finalize_X()
            except Exception:
                e.close()
                raise
            return e

        resident = self._resident_engine_for(X) if (dist is None or self.shard_cells != "local") else None
        if resident is not None:
            # adata.X is the very matrix fit() left in HBM: one W^TX sweep on the resident copy, no upload
            resident.set_factors(W, H0, self.matrices["Bs"])
            resident.transform(n_iter)
            _, H, _ = resident.get_factors()
        else:
            x_dtype = (("x3" if (self.total_components > self.MAX_FAST_COMPONENTS or not _maybe_two_bf16_planes(X)) else "split")
                       if self.x_dtype == "auto" else self.x_dtype)
            try:
                eng = make_engine(x_dtype)
            except _native.AlpineNativeError as err:
                if not (self.x_dtype == "auto" and err.code == -5):
                    raise
                eng = make_engine("x3")
            try:
                eng.set_factors(W, H0, [])
                eng.transform(n_iter)
                _, H, _ = eng.get_factors()
            finally:
                eng.close()
        offs = np.cumsum([0] + self.n_all_components)
        Hs = [H[offs[j]:offs[j + 1]] for j in range(len(self.n_all_components))]
        for i, covariate in enumerate(self.covariate_keys):
            adata.obsm[covariate] = np.ascontiguousarray(Hs[i].T)
            adata.varm[covariate] = deepcopy(self.matrices["Ws"][i])
        adata.obsm["ALPINE_embedding"] = np.ascontiguousarray(Hs[-1].T)
        adata.varm["ALPINE_weights"] = deepcopy(self.matrices["Ws"][-1])

    # ----------------------------------------- mini-batch epochs (main.py:500-521, sampling.py:6-71)
    @staticmethod
    def _balanced_joint_weights(Y: List[np.ndarray]) -> np.ndarray:
        """sampling.py:36-55 + sklearn compute_sample_weight("balanced"): a cell's joint label is the tuple of its
        argmax level per covariate (an all-zero NaN row has argmax 0); weight = n / (n_classes * count(label))."""
        codes = np.stack([np.argmax(y, axis=1) for y in Y], axis=1)
        _, inv, cnt = np.unique(codes, axis=0, return_inverse=True, return_counts=True)
        inv = np.asarray(inv).reshape(-1)
        return (codes.shape[0] / (len(cnt) * cnt.astype(np.float64)))[inv]

    def _run_epochs(self, eng, Y: List[np.ndarray], n_total: int, n_iter: int, comm=None, c0: int = 0, c1: Optional[int] = None,
                    gather_labels=None) -> None:
        """The epoch/batch structure of main.py:500-521 with the reference's own index streams (drawn from the global
        torch generator right after the init draws): 'random' = torch.randperm(N) (sampling.py:14); 'weighted' =
        WeightedRandomSampler(weights, N, replacement=True) == torch.multinomial(weights as float64, N, True)
        (sampling.py:18-33).  Each batch is one alpine_batch_step; one loss row over all cells per epoch.
        Sharded (``comm``): every rank draws the SAME global index stream (same seed), takes the indices that fall into
        its block [c0, c1) -- possibly none -- and the reduce block is all-reduced between alpine_batch_begin and
        alpine_batch_end (and between the two halves of the epoch loss); with the library's own communicator attached
        alpine_batch_step / alpine_epoch_loss do all three steps in C."""
        self._rng_replay = None                                     # the stream is consumed for real here
        weights = None
        if self.sampling_method == "weighted":
            Y_all = Y
            if gather_labels is not None:                           # rank-local inputs: the weights need every cell's labels
                parts = [None] * gather_labels.get_world_size()
                gather_labels.all_gather_object(parts, [np.asarray(y) for y in Y])
                Y_all = [np.concatenate([p[i] for p in parts], axis=0) for i in range(len(Y))]
            weights = torch.as_tensor(self._balanced_joint_weights(Y_all), dtype=torch.double)
        bs = self.batch_size
        c1 = n_total if c1 is None else c1
        # the library's own communicator: alpine_batch_step / alpine_epoch_loss enqueue the all-reduce themselves
        composite = bool(getattr(comm, "native", False))
        for _ in range(n_iter):
            if weights is not None:
                epoch = torch.multinomial(weights, n_total, True).numpy()
            else:
                epoch = torch.randperm(n_total).numpy()
            for b0 in range(0, n_total, bs):
                batch = epoch[b0:min(b0 + bs, n_total)]
                if comm is None:
                    eng.batch_step(batch)
                elif composite:
                    eng.batch_step(batch[(batch >= c0) & (batch < c1)] - c0)
                else:
                    eng.batch_begin(batch[(batch >= c0) & (batch < c1)] - c0)
                    comm.all_reduce()
                    eng.batch_end()
            if comm is None or composite:
                eng.epoch_loss()
            else:
                eng.epoch_loss_begin()
                comm.all_reduce()
                eng.epoch_loss_end()

    # ------------------------------------------------- store_embeddings (main.py:303-320)
    def store_embeddings(self, adata) -> None:
        if not hasattr(self, "matrices"):
            raise RuntimeError("Model is not trained yet. Please fit the model first.")
        elif not is_anndata(adata):
            raise TypeError("adata must be an AnnData object.")
        adata.obsm["ALPINE_embedding"] = copy(self.matrices["Hs"][-1].T)
        adata.varm["ALPINE_weights"] = copy(self.matrices["Ws"][-1])
        dummy_matrices = self.fe.transform(adata.obs)
        for i, covariate in enumerate(self.covariate_keys):
            adata.obsm[covariate] = copy(self.matrices["Hs"][i].T)
            adata.obsm[f"{covariate}_dummy_matrix"] = dummy_matrices[i]
            adata.varm[covariate] = copy(self.matrices["Ws"][i])

    # ------------------------------------------------------ post-fit helpers (main.py:187-273)
    def compute_loss(self, adata):
        """main.py:187-236: total loss of the factors stored in ``adata`` (after fit or transform).  The G x N term
        ||X - W H||_F^2 is evaluated on the device (float64 accumulation, no G x N temporary); the prediction terms are
        C_i x N and follow the reference's numpy expressions."""
        if not hasattr(self, "matrices"):
            raise RuntimeError("Model is not trained yet. Please fit the model first.")
        if not is_anndata(adata):
            raise TypeError("adata must be an AnnData object.")
        if "ALPINE_embedding" not in adata.obsm:
            raise ValueError("ALPINE_embedding not found in adata.obsm. Please transform the data first.")
        Hs = [np.asarray(adata.obsm[k], dtype=np.float32).T for k in self.covariate_keys] + \
             [np.asarray(adata.obsm["ALPINE_embedding"], dtype=np.float32).T]
        Ws = [np.asarray(adata.varm[k], dtype=np.float32) for k in self.covariate_keys] + \
             [np.asarray(adata.varm["ALPINE_weights"], dtype=np.float32)]
        W = np.ascontiguousarray(np.concatenate(Ws, axis=1))
        H = np.ascontiguousarray(np.concatenate(Hs, axis=0))
        recon_loss = self._recon_loss_device(adata.X if isinstance(adata.X, np.ndarray) else np.asarray(adata.X), W, H)

        def kl_divergence(y, y_hat):                                                   # main.py:200-204
            y_hat = np.clip(y_hat, a_min=self.eps, a_max=None)
            return np.sum(y * np.log(np.clip(y / y_hat, a_min=self.eps, a_max=None)) - y + y_hat)

        Ys = self.fe.transform(adata.obs)
        Bs = self.matrices["Bs"]
        if self.loss_type == "kl-divergence":
            pred_loss = [kl_divergence(Ys[i].T, Bs[i] @ Hs[i]) for i in range(len(Ys))]
        else:
            pred_loss = [np.linalg.norm(Ys[i].T - Bs[i] @ Hs[i], ord="fro") ** 2 for i in range(len(Ys))]
        return recon_loss + sum(self.lam[i] * pl for i, pl in enumerate(pred_loss))

    def _recon_loss_device(self, X_cells_genes: np.ndarray, W: np.ndarray, H: np.ndarray) -> float:
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: alpine_amd needs an MI355X (there is no CPU fallback)")
        dev_index = _parse_device(str(self.device))
        if dev_index < 0:
            dev_index = torch.cuda.current_device()
        n_sample, G = X_cells_genes.shape
        resident = self._resident_engine_for(X_cells_genes)
        if resident is not None and W.shape[1] == self.total_components:
            resident.set_factors(W, H, self.matrices["Bs"])          # adata.X is still in HBM from fit(): no upload
            return float(resident.eval_recon_direct())
        eng = _native.NativeShard(n_genes=G, n_cells=n_sample, n_components=W.shape[1], cov_components=[], cov_levels=[], lam=[],
                                  eps=self.eps, device_id=dev_index, transform_only=True, x_dtype="f32")
        try:
            chunk = max(8, ((1 << 28) // (4 * G)) // 8 * 8)
            for r0 in range(0, n_sample, chunk):
                eng.upload_X_host(np.ascontiguousarray(X_cells_genes[r0:r0 + chunk], dtype=np.float32), _native.X_CELLS_BY_GENES, r0)
            eng.finalize_X()
            eng.set_factors(W, H, [])
            return float(eng.eval_recon_direct())
        finally:
            eng.close()

    def get_covariate_gene_scores(self, adata=None):
        """main.py:246-273: per covariate, W_i (H_i Y_i^T / rowsum(Y_i)) as a genes x levels DataFrame (G x k_i x C_i work)."""
        if not hasattr(self, "matrices"):
            raise RuntimeError("Model is not trained yet. Please fit the model first.")
        cov_gene_scores = {}
        for i, covariate in enumerate(self.covariate_keys):
            W = self.matrices["Ws"][i]
            H = self.matrices["Hs"][i]
            Y = self.matrices["Ys"][i]
            HY = H @ Y.T / Y.sum(axis=1)
            cov_gene_scores[covariate] = pd.DataFrame(W @ HY, index=self.feature_names, columns=self.fe.encoded_labels[covariate])
        if adata is None:
            return cov_gene_scores
        for condition, df in cov_gene_scores.items():
            adata.varm[condition + "_gene_scores"] = df
        return None

    def get_decomposed_matrices(self):
        """main.py:238-244."""
        if not hasattr(self, "matrices"):
            raise RuntimeError("Model is not trained yet. Please fit the model first.")
        return self.matrices

    # -------------------------------------------------- warm-up elbow (main.py:755-770)
    def _compute_best_iter(self, train_loss) -> int:
        """main.py:755-770: Kneedle elbow of log10(reconstruction loss) over the warm-up iterations.  Uses the real
        ``kneed.KneeLocator`` (the reference's dependency) when importable, otherwise ``alpine_amd.kneedle`` (a
        restatement of the published algorithm; parity with kneed unpinned, see that module)."""
        import warnings
        x = np.arange(0, len(train_loss))
        y = np.log10(train_loss)
        try:
            from kneed import KneeLocator
            elbow = KneeLocator(x, y, curve="convex", direction="decreasing", interp_method="polynomial",
                                polynomial_degree=2).elbow
        except ImportError:
            from .kneedle import find_elbow
            elbow = find_elbow(x, y, S=1.0, polynomial_degree=2)
        if elbow is not None:
            return int(elbow)
        warnings.warn("Kneedle elbow not found, using default max_iter=200")
        return 200

    # ------------------------------------------------ validators (main.py:322-381, :383-434)
    def _validate_init_args(self) -> None:
        if self.n_components <= 0:
            raise ValueError("n_components must be greater than 0.")
        if not isinstance(self.n_covariate_components, list):
            raise TypeError("n_covariate_components must be a list.")
        for n in self.n_covariate_components:
            if not isinstance(n, int) or n < 0:
                raise ValueError("Each element in n_covariate_components must be a non-negative integer.")
        if not isinstance(self.lam, list):
            raise TypeError("lam must be in a list.")
        for lam in self.lam:
            if not isinstance(lam, float) or lam < 0:
                raise ValueError("Each element in lam must be a non-negative float.")
        if not isinstance(self.alpha_W, float) or self.alpha_W < 0:
            raise ValueError("alpha_W must be a non-negative float.")
        if not isinstance(self.orth_W, float) or self.orth_W < 0:
            raise ValueError("orth_W must be a non-negative float.")
        if not isinstance(self.l1_ratio_W, float) or self.l1_ratio_W < 0 or self.l1_ratio_W > 1:
            raise ValueError("l1_ratio_W must be a float between 0 and 1.")
        if not isinstance(self.scale_needed, bool):
            raise TypeError("scale_needed must be a boolean.")
        if not isinstance(self.loss_type, str):
            raise TypeError("loss_type must be a string.")
        valid_loss_types = ["kl-divergence", "frobenius"]
        if self.loss_type not in valid_loss_types:
            raise ValueError(f"loss_type must be one of {valid_loss_types}.")
        if not isinstance(self.eps, float) or self.eps < 0:
            raise ValueError("eps must be a non-negative float.")
        if not isinstance(self.random_state, int) or self.random_state < 0:
            raise ValueError("random_state must be a non-negative integer.")

    def _validate_fit_args(self, adata, covariate_keys, batch_size, max_iter, sampling_method, verbose) -> None:
        if not is_anndata(adata):
            raise TypeError("adata must be an AnnData object.")
        if not isinstance(adata.X, np.ndarray):
            raise TypeError("adata.X must be a numpy array.")
        elif adata.X.ndim != 2:
            raise ValueError("adata.X must be a 2D numpy array.")
        elif not all_nonnegative(adata.X):
            raise ValueError("All elements in adata.X must be non-negative.")
        if not isinstance(covariate_keys, list):
            raise TypeError("covariate_keys must be a list.")
        elif not len(covariate_keys) == len(self.n_covariate_components):
            raise ValueError("Length of covariate_keys must match length of n_covariate_components.")
        else:
            for key in covariate_keys:
                if not isinstance(key, str):
                    raise TypeError("Each element in covariate_keys must be a string.")
                if key not in adata.obs.columns:
                    raise ValueError(f"Covariate key '{key}' not found in adata.obs.")
                if not adata.obs[key].dtype.kind == "O":
                    raise TypeError(f"Covariate '{key}' in adata.obs must be a categorical or object type variable.")
        # the next two reproduce the reference's (ineffective for ints) conditions, main.py:420-428
        if batch_size is not None and not isinstance(batch_size, int) and batch_size > 0:
            raise TypeError("batch_size must be a positive integer.")
        if max_iter is not None and not isinstance(max_iter, int) and max_iter > 0:
            raise TypeError("max_iter must be a positive integer.")
        if not isinstance(sampling_method, str):
            raise TypeError("sampling_method must be a string.")
        if not isinstance(verbose, bool):
            raise TypeError("verbose must be a boolean.")
