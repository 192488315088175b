"""ctypes binding of libalpine_hip.so (include/alpine_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libalpine_hip.so")
# tools/ may point at the diagnostics build (libalpine_hip_diag.so: timing-only ablations compiled in) -- the product never does
if os.environ.get("ALPINE_HIP_LIBRARY"):
    LIB_PATH = os.path.abspath(os.environ["ALPINE_HIP_LIBRARY"])

LOSS_KL, LOSS_FROBENIUS = 0, 1
X_CELLS_BY_GENES, X_GENES_BY_CELLS = 0, 1
KERNEL_SWEEP_XHT, KERNEL_SWEEP_WTX, KERNEL_ALLREDUCE = 0, 1, 2
COMM_ID_BYTES = 128
ERR_RCCL = -6
FLAG_TRANSFORM_ONLY, FLAG_X_BF16, FLAG_USE_ALS, FLAG_X_SPLIT, FLAG_X3_PRODUCTS = 1, 2, 4, 8, 16
BUF_REDUCE_BLOCK, BUF_WTW, BUF_W, BUF_H, BUF_X_GN, BUF_X_NG = 0, 1, 2, 3, 4, 5

EXPORTS = [
    "alpine_reduce_block_floats", "alpine_create", "alpine_destroy", "alpine_last_error", "alpine_get_info",
    "alpine_upload_X_host", "alpine_upload_X_device", "alpine_finalize_X", "alpine_upload_Y",
    "alpine_set_factors", "alpine_get_factors", "alpine_iter_begin", "alpine_iter_end", "alpine_reduce_block",
    "alpine_als_begin", "alpine_als_group_begin", "alpine_als_group_end", "alpine_reduce_block_hht", "alpine_batch_step", "alpine_batch_begin", "alpine_batch_end", "alpine_epoch_loss", "alpine_epoch_loss_begin", "alpine_epoch_loss_end", "alpine_run", "alpine_transform", "alpine_get_losses", "alpine_reset_losses", "alpine_scale", "alpine_synchronize",
    "alpine_eval_recon_direct", "alpine_set_profiling", "alpine_debug_set_xcd_bias", "alpine_debug_run_graph", "alpine_get_kernel_time", "alpine_read_buffer",
    "alpine_comm_get_unique_id", "alpine_comm_version", "alpine_comm_init_rank", "alpine_comm_destroy", "alpine_comm_all_reduce", "alpine_iter",
    "alpine_comm_count", "alpine_comm_init_all", "alpine_debug_set_team_width", "alpine_debug_set_option",
]
ABI_VERSION = 6          # ALPINE_HIP_ABI_VERSION of include/alpine_hip.h this binding was written against


class AlpineConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device_id", C.c_int32),
        ("n_genes", C.c_int64), ("n_cells", C.c_int64),
        ("n_components", C.c_int32), ("n_covariates", C.c_int32),
        ("cov_components", C.POINTER(C.c_int32)), ("cov_levels", C.POINTER(C.c_int32)),
        ("lam", C.POINTER(C.c_double)),
        ("orth_W", C.c_double), ("alpha_W", C.c_double), ("l1_ratio_W", C.c_double), ("eps", C.c_double),
        ("loss_type", C.c_int32), ("split_a", C.c_int32), ("split_b", C.c_int32), ("flags", C.c_int32),
        ("stream", C.c_void_p), ("reduce_block", C.c_void_p), ("batch_capacity", C.c_int64),
    ]


class AlpineInfo(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("k_total", C.c_int32), ("k_padded", C.c_int32),
        ("split_a", C.c_int32), ("split_b", C.c_int32), ("grid_a", C.c_int32), ("grid_b", C.c_int32),
        ("genes_padded", C.c_int64), ("cells_padded", C.c_int64),
        ("reduce_block_floats", C.c_int64), ("device_bytes", C.c_int64), ("x_sqnorm", C.c_double),
        ("x_multi_plane_fraction", C.c_double), ("x3_wide", C.c_int32), ("sweep_waves_per_simd", C.c_int32),
        ("span_rows_a", C.c_int32), ("span_rows_b", C.c_int32),
        ("spans_per_workgroup_a", C.c_int32), ("spans_per_workgroup_b", C.c_int32),
        ("xcd_bias_per_mille", C.c_int32), ("xcc_of_workgroup0", C.c_int32),
        ("team_width_a", C.c_int32), ("team_width_b", C.c_int32),
    ]


class AlpineNativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libalpine_hip: {msg} (status {code})")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load the shared library or raise -- there is no CPU/eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  alpine_amd has no non-HIP execution path.")
    lib = C.CDLL(LIB_PATH)
    p, i32, i64, f32p = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_float)
    lib.alpine_reduce_block_floats.restype = i64
    lib.alpine_reduce_block_floats.argtypes = [C.POINTER(AlpineConfig)]
    lib.alpine_create.argtypes = [C.POINTER(AlpineConfig), C.POINTER(p)]
    lib.alpine_destroy.argtypes = [p]
    lib.alpine_last_error.restype = C.c_char_p
    lib.alpine_last_error.argtypes = [p]
    lib.alpine_get_info.argtypes = [p, C.POINTER(AlpineInfo)]
    lib.alpine_upload_X_host.argtypes = [p, p, i32, i64, i64, i64]
    lib.alpine_upload_X_device.argtypes = [p, p, i32, i64, i64, i64]
    lib.alpine_finalize_X.argtypes = [p]
    lib.alpine_upload_Y.argtypes = [p, i32, p, i64]
    lib.alpine_set_factors.argtypes = [p, p, p, i64, C.POINTER(p)]
    lib.alpine_get_factors.argtypes = [p, p, p, i64, C.POINTER(p)]
    lib.alpine_iter_begin.argtypes = [p]
    lib.alpine_iter_end.argtypes = [p, i32]
    lib.alpine_reduce_block.argtypes = [p, C.POINTER(p), C.POINTER(i64)]
    lib.alpine_run.argtypes = [p, i32, i32]
    lib.alpine_transform.argtypes = [p, i32]
    lib.alpine_batch_step.argtypes = [p, p, i64]
    lib.alpine_epoch_loss.argtypes = [p]
    lib.alpine_batch_begin.argtypes = [p, p, i64]
    lib.alpine_als_begin.argtypes = [p]
    lib.alpine_als_group_begin.argtypes = [p, C.c_int]
    lib.alpine_als_group_end.argtypes = [p, C.c_int]
    lib.alpine_reduce_block_hht.argtypes = [p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.alpine_batch_end.argtypes = [p]
    lib.alpine_epoch_loss_begin.argtypes = [p]
    lib.alpine_epoch_loss_end.argtypes = [p]
    lib.alpine_get_losses.argtypes = [p, C.POINTER(C.c_double), i64, C.POINTER(i64)]
    lib.alpine_reset_losses.argtypes = [p]
    lib.alpine_scale.argtypes = [p]
    lib.alpine_synchronize.argtypes = [p]
    lib.alpine_eval_recon_direct.argtypes = [p, C.POINTER(C.c_double)]
    lib.alpine_set_profiling.argtypes = [p, i32]
    lib.alpine_debug_set_xcd_bias.argtypes = [p, i32]
    lib.alpine_debug_run_graph.argtypes = [p, i32]
    lib.alpine_get_kernel_time.argtypes = [p, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    lib.alpine_read_buffer.argtypes = [p, i32, i64, i64, p]
    lib.alpine_comm_get_unique_id.argtypes = [p]
    lib.alpine_comm_init_rank.argtypes = [p, p, i32, i32]
    lib.alpine_comm_version.argtypes = [C.POINTER(C.c_int)]
    lib.alpine_comm_destroy.argtypes = [p]
    lib.alpine_comm_all_reduce.argtypes = [p, i64, i64]
    lib.alpine_iter.argtypes = [p, i32]
    lib.alpine_comm_count.argtypes = [p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.alpine_comm_init_all.argtypes = [C.POINTER(p), i32]
    lib.alpine_debug_set_team_width.argtypes = [p, i32]
    lib.alpine_debug_set_option.argtypes = [p, C.c_char_p, i32]
    for name in EXPORTS:
        if name not in ("alpine_reduce_block_floats", "alpine_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _f32c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class NativeShard:
    """One ctx = one GPU = one shard of the cell axis (thin OO wrapper over the C ABI)."""

    def __init__(self, n_genes: int, n_cells: int, n_components: int, cov_components: Sequence[int],
                 cov_levels: Sequence[int], lam: Sequence[float], orth_W: float = 0.0, alpha_W: float = 0.0,
                 l1_ratio_W: float = 0.0, eps: float = 1e-6, loss_type: str = "kl-divergence",
                 device_id: int = 0, stream: Optional[int] = None, reduce_block: Optional[int] = None,
                 split_a: int = 0, split_b: int = 0, transform_only: bool = False, x_dtype: str = "f32",
                 batch_capacity: int = 0, use_als: bool = False):
        if x_dtype not in ("f32", "bf16", "split", "x3"):
            raise ValueError("x_dtype must be 'f32', 'bf16', 'split' or 'x3'")
        self._lib = load()
        self._h = C.c_void_p()
        n_cov = len(cov_components)
        self._k = (C.c_int32 * max(1, n_cov))(*cov_components)
        self._lev = (C.c_int32 * max(1, n_cov))(*cov_levels)
        self._lam = (C.c_double * max(1, n_cov))(*[float(x) for x in lam[:n_cov]])
        cfg = AlpineConfig()
        cfg.struct_size = C.sizeof(AlpineConfig)
        cfg.device_id = device_id
        cfg.n_genes, cfg.n_cells = n_genes, n_cells
        cfg.n_components, cfg.n_covariates = n_components, n_cov
        cfg.cov_components, cfg.cov_levels, cfg.lam = self._k, self._lev, self._lam
        cfg.orth_W, cfg.alpha_W, cfg.l1_ratio_W, cfg.eps = orth_W, alpha_W, l1_ratio_W, eps
        cfg.loss_type = LOSS_KL if loss_type == "kl-divergence" else LOSS_FROBENIUS
        cfg.split_a, cfg.split_b, cfg.flags = split_a, split_b, ((FLAG_TRANSFORM_ONLY if transform_only else 0) | (FLAG_X_BF16 if x_dtype == "bf16" else 0) | (FLAG_X_SPLIT if x_dtype == "split" else 0) | (FLAG_X3_PRODUCTS if x_dtype == "x3" else 0) |
                                                       (FLAG_USE_ALS if use_als else 0))
        cfg.stream = stream
        cfg.reduce_block = reduce_block
        cfg.batch_capacity = batch_capacity
        self._cfg = cfg
        self.n_genes, self.n_cells, self.n_cov = n_genes, n_cells, n_cov
        self.cov_components, self.cov_levels = list(cov_components), list(cov_levels)
        self.k_total = n_components + sum(cov_components)
        self.comm_ranks, self.comm_rank = 1, 0
        rc = self._lib.alpine_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self._lib.alpine_last_error(None).decode()
            self._h = C.c_void_p()
            raise AlpineNativeError(rc, msg)
        try:
            self.info()                      # ABI version of the loaded library against this binding (a stale build, ALPINE_HIP_LIBRARY)
        except AlpineNativeError:
            self.close()
            raise

    # -- plumbing
    def _chk(self, rc: int):
        if rc != 0:
            raise AlpineNativeError(rc, self._lib.alpine_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.alpine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> AlpineInfo:
        out = AlpineInfo()
        out.abi_version = -1
        self._chk(self._lib.alpine_get_info(self._h, C.byref(out)))
        if out.abi_version != ABI_VERSION:
            # a stale build, or another library swapped in through ALPINE_HIP_LIBRARY: the struct layouts may differ
            raise AlpineNativeError(-5, f"{LIB_PATH} reports ABI version {out.abi_version}, this binding expects {ABI_VERSION}: rebuild it "
                                        "(python -c 'import __graft_entry__ as g; g.build()')")
        return out

    # -- ingest
    def upload_X_host(self, X: np.ndarray, layout: int = X_CELLS_BY_GENES, cell0: int = 0):
        if X.dtype != np.float32 or X.ndim != 2 or X.strides[1] != 4 or X.strides[0] % 4:
            X = _f32c(X)
        n = X.shape[0] if layout == X_CELLS_BY_GENES else X.shape[1]
        self._chk(self._lib.alpine_upload_X_host(self._h, X.ctypes.data, layout, X.strides[0] // 4, cell0, n))

    def upload_X_device(self, dev_ptr: int, ld: int, n_cells: int, layout: int = X_CELLS_BY_GENES, cell0: int = 0):
        self._chk(self._lib.alpine_upload_X_device(self._h, dev_ptr, layout, ld, cell0, n_cells))

    def finalize_X(self):
        self._chk(self._lib.alpine_finalize_X(self._h))

    def upload_Y(self, i: int, Y_cn: np.ndarray):
        Y_cn = _f32c(Y_cn)
        assert Y_cn.shape == (self.cov_levels[i], self.n_cells), (Y_cn.shape, self.cov_levels[i], self.n_cells)
        self._chk(self._lib.alpine_upload_Y(self._h, i, Y_cn.ctypes.data, Y_cn.shape[1]))

    def set_factors(self, W: np.ndarray, H: np.ndarray, Bs: Sequence[np.ndarray], h_col0: int = 0):
        """W: G x K; H: K x N_total (this shard takes columns [h_col0, h_col0 + n_cells)); Bs[i]: C_i x k_i."""
        W = _f32c(W)
        if H.dtype != np.float32 or H.strides[1] != 4:
            H = _f32c(H)
        assert W.shape == (self.n_genes, self.k_total) and H.shape[0] == self.k_total
        assert h_col0 + self.n_cells <= H.shape[1]
        Bs = [_f32c(b) for b in Bs]
        arr = (C.c_void_p * max(1, self.n_cov))(*[b.ctypes.data for b in Bs])
        self._chk(self._lib.alpine_set_factors(self._h, W.ctypes.data, H.ctypes.data + 4 * h_col0,
                                               H.strides[0] // 4, arr))

    def get_factors(self):
        W = np.empty((self.n_genes, self.k_total), dtype=np.float32)
        H = np.empty((self.k_total, self.n_cells), dtype=np.float32)
        Bs = [np.empty((c, k), dtype=np.float32) for c, k in zip(self.cov_levels, self.cov_components)]
        arr = (C.c_void_p * max(1, self.n_cov))(*[b.ctypes.data for b in Bs])
        self._chk(self._lib.alpine_get_factors(self._h, W.ctypes.data, H.ctypes.data, self.n_cells, arr))
        return W, H, Bs

    # -- iteration
    def iter_begin(self):
        self._chk(self._lib.alpine_iter_begin(self._h))

    def iter_end(self, update: bool = True):
        self._chk(self._lib.alpine_iter_end(self._h, 1 if update else 0))

    def iter(self, update: bool = True):
        """begin + [all-reduce when a communicator is attached] + end"""
        self._chk(self._lib.alpine_iter(self._h, 1 if update else 0))

    # -- multi-GPU (RCCL inside the library)
    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        """Collective: every rank of the communicator calls this with the id rank 0 got from ``comm_unique_id()``."""
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {COMM_ID_BYTES} bytes")
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._chk(self._lib.alpine_comm_init_rank(self._h, C.cast(buf, C.c_void_p), nranks, rank))
        self.comm_ranks, self.comm_rank = nranks, rank

    def comm_count(self):
        """(ranks, rank) as the attached communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(-1), C.c_int(-1)
        self._chk(self._lib.alpine_comm_count(self._h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def comm_destroy(self):
        self._chk(self._lib.alpine_comm_destroy(self._h))
        self.comm_ranks, self.comm_rank = 1, 0

    def comm_all_reduce(self, offset: int = 0, n: Optional[int] = None):
        if n is None:
            n = self.reduce_block()[1] - offset
        self._chk(self._lib.alpine_comm_all_reduce(self._h, offset, n))

    def reduce_block(self):
        ptr, n = C.c_void_p(), C.c_int64()
        self._chk(self._lib.alpine_reduce_block(self._h, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def run(self, n_iters: int, with_loss: bool = True):
        self._chk(self._lib.alpine_run(self._h, n_iters, 1 if with_loss else 0))

    def batch_step(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self._chk(self._lib.alpine_batch_step(self._h, idx.ctypes.data if idx.size else None, idx.size))

    def epoch_loss(self):
        self._chk(self._lib.alpine_epoch_loss(self._h))

    def als_begin(self):
        self._chk(self._lib.alpine_als_begin(self._h))

    def als_group_begin(self, grp: int):
        self._chk(self._lib.alpine_als_group_begin(self._h, grp))

    def als_group_end(self, grp: int):
        self._chk(self._lib.alpine_als_group_end(self._h, grp))

    def reduce_block_hht(self):
        off, n = C.c_int64(), C.c_int64()
        self._chk(self._lib.alpine_reduce_block_hht(self._h, C.byref(off), C.byref(n)))
        return int(off.value), int(n.value)

    def batch_begin(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        self._chk(self._lib.alpine_batch_begin(self._h, idx.ctypes.data if idx.size else None, idx.size))

    def batch_end(self):
        self._chk(self._lib.alpine_batch_end(self._h))

    def epoch_loss_begin(self):
        self._chk(self._lib.alpine_epoch_loss_begin(self._h))

    def epoch_loss_end(self):
        self._chk(self._lib.alpine_epoch_loss_end(self._h))

    def transform(self, n_iter: int):
        self._chk(self._lib.alpine_transform(self._h, n_iter))

    def losses(self) -> np.ndarray:
        n = C.c_int64()
        self._chk(self._lib.alpine_get_losses(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, self.n_cov + 2), dtype=np.float64)
        if n.value:
            self._chk(self._lib.alpine_get_losses(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return out

    def reset_losses(self):
        self._chk(self._lib.alpine_reset_losses(self._h))

    def scale(self):
        self._chk(self._lib.alpine_scale(self._h))

    def synchronize(self):
        self._chk(self._lib.alpine_synchronize(self._h))

    def eval_recon_direct(self) -> float:
        out = C.c_double()
        self._chk(self._lib.alpine_eval_recon_direct(self._h, C.byref(out)))
        return out.value

    def set_profiling(self, on):
        """False / 0 = off, True / 1 = events around every sweep and all-reduce, n > 1 = every n-th iteration only."""
        self._chk(self._lib.alpine_set_profiling(self._h, int(on)))

    def debug_set_xcd_bias(self, per_mille: int):
        self._chk(self._lib.alpine_debug_set_xcd_bias(self._h, int(per_mille)))

    def debug_set_option(self, name: str, value: int):
        """Result-preserving knobs (tests, tools): 'no_tail', 'fused_w', 'unfused_mid', 'guided_scalar', 'tail_stats_per_covariate',
        'sg_variant'; before finalize_X: 'x3_variant' (-1 | 0 | 2), 'x3_narrow' (0 | 1).  The library takes none of them from the environment."""
        self._chk(self._lib.alpine_debug_set_option(self._h, name.encode(), int(value)))

    def debug_set_team_width(self, width: int):
        self._chk(self._lib.alpine_debug_set_team_width(self._h, int(width)))

    def debug_run_graph(self, n_pairs: int):
        self._chk(self._lib.alpine_debug_run_graph(self._h, int(n_pairs)))

    def kernel_time(self, which: int):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self._lib.alpine_get_kernel_time(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def read_buffer(self, which: int, offset: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        self._chk(self._lib.alpine_read_buffer(self._h, which, offset, n, out.ctypes.data))
        return out


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 calls this and hands the bytes to every rank."""
    lib = load()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.alpine_comm_get_unique_id(C.cast(buf, C.c_void_p))
    if rc != 0:
        raise AlpineNativeError(rc, lib.alpine_last_error(None).decode())
    return buf.raw


def comm_init_all(engines: Sequence["NativeShard"]) -> None:
    """ONE process, one engine per GPU: ncclCommInitAll over the engines' devices, engine i becomes rank i (no unique id)."""
    lib = load()
    arr = (C.c_void_p * len(engines))(*[e._h.value for e in engines])
    rc = lib.alpine_comm_init_all(arr, len(engines))
    if rc != 0:
        msg = (engines[0]._lib.alpine_last_error(engines[0]._h) or lib.alpine_last_error(None) or b"").decode()
        raise AlpineNativeError(rc, msg or lib.alpine_last_error(None).decode())
    for i, e in enumerate(engines):
        e.comm_ranks, e.comm_rank = len(engines), i


def comm_version() -> int:
    """ncclGetVersion of the librccl the library is linked against (an integer such as 22203)."""
    lib = load()
    v = C.c_int()
    rc = lib.alpine_comm_version(C.byref(v))
    if rc != 0:
        raise AlpineNativeError(rc, lib.alpine_last_error(None).decode())
    return int(v.value)


def reduce_block_floats(n_genes: int, n_cells: int, n_components: int, cov_components: Sequence[int],
                        cov_levels: Sequence[int]) -> int:
    lib = load()
    n_cov = len(cov_components)
    k = (C.c_int32 * max(1, n_cov))(*cov_components)
    lev = (C.c_int32 * max(1, n_cov))(*cov_levels)
    lam = (C.c_double * max(1, n_cov))(*([0.0] * max(1, n_cov)))
    cfg = AlpineConfig()
    cfg.struct_size = C.sizeof(AlpineConfig)
    cfg.n_genes, cfg.n_cells, cfg.n_components, cfg.n_covariates = n_genes, n_cells, n_components, n_cov
    cfg.cov_components, cfg.cov_levels, cfg.lam = k, lev, lam
    cfg.eps = 1e-6
    n = lib.alpine_reduce_block_floats(C.byref(cfg))
    if n < 0:
        raise AlpineNativeError(int(n), lib.alpine_last_error(None).decode())
    return int(n)
