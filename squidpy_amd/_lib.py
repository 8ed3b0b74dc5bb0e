"""ctypes binding of ``libsqgr.so`` (C ABI declared in ``include/sqgr.h``).

This is the only way the Python front-end computes anything: there is no CPU fallback.  If the shared
library has not been built (``python -m squidpy_amd._build``) or no HIP device is usable, the functions
here raise :class:`SqgrError` / :class:`OSError` loudly."""

from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Any

import numpy as np

from ._build import LIB_PATH

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u64p = C.POINTER(C.c_uint64)
c_u32p = C.POINTER(C.c_uint32)
c_u8p = C.POINTER(C.c_uint8)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol declared in include/sqgr.h must be listed here
# (tests/test_abi.py cross-checks this table against the header and the built library).
SIGNATURES: dict[str, tuple[Any, list[Any]]] = {
    "sqgr_abi_version": (C.c_int, []),
    "sqgr_last_error": (C.c_char_p, []),
    "sqgr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sqgr_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "sqgr_ctx_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_ctx_trim": (C.c_int, [C.c_void_p, C.c_int64]),
    "sqgr_ctx_sync": (C.c_int, [C.c_void_p]),
    "sqgr_debug_counters": (C.c_int, [c_i64p, C.c_int32]),
    "sqgr_ctx_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), c_i64p]),
    "sqgr_timer_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "sqgr_timer_reset": (C.c_int, [C.c_void_p]),
    "sqgr_timer_get": (C.c_int, [C.c_void_p, C.c_char_p, c_f64p, c_i64p]),
    "sqgr_timer_report": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "sqgr_comm_unique_id": (C.c_int, [c_u8p]),
    "sqgr_comm_create": (C.c_int, [C.c_void_p, c_u8p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "sqgr_comm_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_comm_info": (C.c_int, [C.c_void_p, c_i32p, c_i32p]),
    "sqgr_comm_allreduce_i64": (C.c_int, [C.c_void_p, c_i64p, C.c_int64, C.c_int32]),
    "sqgr_comm_barrier": (C.c_int, [C.c_void_p]),
    "sqgr_nhood_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqgr_nhood_info": (C.c_int, [C.c_void_p, c_i64p]),
    "sqgr_graph_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_i64p, c_i32p, c_f32p, C.POINTER(C.c_void_p)]),
    "sqgr_graph_create_f64": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_i64p, c_i32p, c_f64p, C.POINTER(C.c_void_p)]),
    "sqgr_graph_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_graph_renumbered": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p, C.POINTER(C.c_void_p)]),
    "sqgr_spatial_order": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, c_i32p]),
    "sqgr_nhood_set_spot_map": (C.c_int, [C.c_void_p, c_i32p]),
    "sqgr_nhood_counts": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p, C.c_int32, c_u32p]),
    "sqgr_nhood_counts_batch": (C.c_int, [C.c_void_p, C.c_void_p, c_u8p, C.c_int64, C.c_int32, c_u32p]),
    "sqgr_nhood_create": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p, C.c_int32, c_i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "sqgr_nhood_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_nhood_run": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, c_i64p, c_i64p, c_u64p, c_u32p]),
    "sqgr_nhood_run_pcg64": (C.c_int, [C.c_void_p, c_u64p, C.c_int64, c_i64p, c_i64p, c_u64p, c_u32p]),
    "sqgr_nhood_run_pcg64_stats": (C.c_int, [C.c_void_p, c_u64p, C.c_int64, c_f64p, c_f64p]),
    "sqgr_pcg64_permutations": (C.c_int, [C.c_void_p, C.c_int64, c_u64p, C.c_int64, c_i32p]),
    "sqgr_nhood_shuffled_labels": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, c_u8p]),
    "sqgr_nhood_tune": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "sqgr_interaction_matrix": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p, C.c_int32, C.c_int32, c_f64p]),
    "sqgr_autocorr_create": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_autocorr_create_cm": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_create": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_create_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_alloc_dense": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_upload_columns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "sqgr_matrix_create_csr": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_create_csc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "sqgr_matrix_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_autocorr_create_cols": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_autocorr_create_colidx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i32p, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_autocorr_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_autocorr_scores": (C.c_int, [C.c_void_p, C.c_int32, c_f64p]),
    "sqgr_autocorr_perms": (C.c_int, [C.c_void_p, C.c_int32, c_i32p, C.c_uint64, C.c_int64, C.c_int64, c_f64p]),
    "sqgr_autocorr_perms_pcg64": (C.c_int, [C.c_void_p, C.c_int32, c_u64p, C.c_int64, c_f64p]),
    "sqgr_autocorr_perm_stats": (C.c_int, [C.c_void_p, C.c_int32, c_i32p, c_u64p, C.c_uint64, C.c_int64, C.c_int64, c_f64p, c_i64p, c_f64p, c_f64p, c_f64p, C.c_int32]),
    "sqgr_autocorr_perm_indices": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_int64, C.c_int64, c_i32p]),
    "sqgr_pair_counts": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, c_f64p, C.c_int32, C.c_int32, c_i64p]),
    "sqgr_pair_counts_batch": (C.c_int, [C.c_void_p, c_f64p, c_i64p, C.c_int32, c_f64p, C.c_int32, C.c_int32, c_i64p]),
    "sqgr_knn_dist": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, c_f64p, C.c_int64, C.c_int32, C.c_int32, c_f64p]),
    "sqgr_points_create": (C.c_int, [C.c_void_p, c_f64p, c_i32p, C.c_int64, C.POINTER(C.c_void_p)]),
    "sqgr_points_destroy": (C.c_int, [C.c_void_p]),
    "sqgr_knn_hist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_f64p, C.c_int64, C.c_int32, C.c_int32, c_f64p, C.c_int32, c_i64p]),
    "sqgr_knn_self": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, C.c_int32, c_i32p, c_f64p]),
    "sqgr_radius_self": (C.c_int, [C.c_void_p, c_f64p, C.c_int64, C.c_double, c_i64p, c_i32p, c_f64p, C.c_int64]),
    "sqgr_ligrec_counts": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, c_i64p, c_i32p, c_f64p, c_i32p, c_f64p, c_i32p, C.c_int64, c_i32p, C.c_int32,
         c_f64p, c_u8p, C.c_uint64, c_u64p, C.c_int64, C.c_int64, c_i64p, c_f64p],
    ),
    "sqgr_cooccur_counts": (
        C.c_int,
        [C.c_void_p, c_f32p, c_f32p, c_i32p, C.c_int64, C.c_int32, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_i64p],
    ),
}


ABI_VERSION = 7  # SQGR_ABI_VERSION of include/sqgr.h


class SqgrError(RuntimeError):
    """A libsqgr call returned a negative status."""

    def __init__(self, status: int, message: str):
        super().__init__(f"libsqgr error {status}: {message}")
        self.status = status


_lib: C.CDLL | None = None
_lock = threading.Lock()


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen the in-tree ``libsqgr.so`` and set the prototypes.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or os.environ.get("SQGR_LIBRARY") or LIB_PATH
        if not os.path.exists(p):
            raise OSError(
                f"{p} not found: build the HIP extension first (`python -m squidpy_amd._build`). "
                "squidpy_amd has no CPU fallback."
            )
        lib = C.CDLL(p)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.sqgr_abi_version() != ABI_VERSION:
            raise OSError(f"{p}: ABI version {lib.sqgr_abi_version()} != {ABI_VERSION} (rebuild: python -m squidpy_amd._build)")
        if path is None:
            _lib = lib
        return lib


def _check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise SqgrError(rc, (lib.sqgr_last_error() or b"").decode(errors="replace"))


def _ptr(a: np.ndarray | None, ctype: Any) -> Any:
    if a is None:
        return None
    return a.ctypes.data_as(ctype)


def _as(a: Any, dtype: Any) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.sqgr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context:
    """One device + one HIP stream (``sqgr_ctx``)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.sqgr_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.sqgr_ctx_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        _check(self.lib, self.lib.sqgr_ctx_sync(self.h))

    def device_info(self) -> dict[str, Any]:
        buf = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = C.c_int64(0)
        _check(self.lib, self.lib.sqgr_ctx_device_info(self.h, buf, 256, C.byref(cu), C.byref(mem)))
        return {"name": buf.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    def alloc_counters(self) -> dict[str, int]:
        """What the library has asked of the HIP allocator so far (process-wide): ``sqgr_debug_counters``."""
        out = np.zeros(8, dtype=np.int64)
        _check(self.lib, self.lib.sqgr_debug_counters(out.ctypes.data_as(c_i64p), 8))
        names = ("mallocs", "malloc_bytes", "malloc_ns", "frees", "free_ns", "pool_hits", "pool_parks", "pool_flushes")
        return {k: int(v) for k, v in zip(names, out)}

    # ---- kernel timers (HIP events on this context's stream)
    def trim(self, keep_bytes: int = 0) -> None:
        """Hand parked device buffers (``SQGR_POOL_GB``) back to the driver until at most ``keep_bytes`` stay parked."""
        _check(self.lib, self.lib.sqgr_ctx_trim(self.h, int(keep_bytes)))

    def timer_enable(self, on: bool = True) -> None:
        _check(self.lib, self.lib.sqgr_timer_enable(self.h, int(on)))

    def timer_reset(self) -> None:
        _check(self.lib, self.lib.sqgr_timer_reset(self.h))

    def timer_get(self, prefix: str) -> tuple[float, int]:
        ms = C.c_double(0)
        cnt = C.c_int64(0)
        _check(self.lib, self.lib.sqgr_timer_get(self.h, prefix.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def timer_report(self) -> dict[str, tuple[int, float]]:
        buf = C.create_string_buffer(16384)
        _check(self.lib, self.lib.sqgr_timer_report(self.h, buf, 16384))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                name, cnt, ms = item.rsplit(":", 2)
                out[name] = (int(cnt), float(ms))
        return out


_default_ctx: dict[int, Context] = {}


def trim_device_memory(keep_bytes: int = 0) -> None:
    """Hand the device buffers libsqgr keeps parked between calls (``SQGR_POOL_GB``, default a quarter of the GPU's memory) back to the driver —
    for processes that share the GPU with other HIP users (torch, a second library).  Resident graphs and plans stay."""
    for ctx in _default_ctx.values():
        ctx.trim(keep_bytes)


def default_context(device: int | None = None) -> Context:
    """Process-wide context for ``device`` (default: ``LOCAL_RANK`` or 0) — one process per GPU."""
    if device is None:
        device = int(os.environ.get("SQGR_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """``ncclGetUniqueId`` through the library (rank 0 calls this and hands the bytes to the other ranks)."""
    lib = load_library()
    buf = np.zeros(UNIQUE_ID_BYTES, dtype=np.uint8)
    _check(lib, lib.sqgr_comm_unique_id(_ptr(buf, c_u8p)))
    return buf.tobytes()


class Comm:
    """RCCL communicator owned by libsqgr (``sqgr_comm``): one rank per process and GPU."""

    SUM, MAX = 0, 1

    def __init__(self, ctx: Context, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f"Expected a {UNIQUE_ID_BYTES}-byte unique id, found {len(unique_id)} bytes.")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        uid = np.frombuffer(unique_id, dtype=np.uint8).copy()
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.sqgr_comm_create(ctx.h, _ptr(uid, c_u8p), self.rank, self.world, C.byref(h)))
        self.h = h

    def allreduce_i64(self, buf: np.ndarray, op: int = 0) -> np.ndarray:
        """In-place all-reduce of a C-contiguous int64 array (uint64 through its int64 view for sums)."""
        if buf.dtype != np.int64 or not buf.flags.c_contiguous:
            raise ValueError("allreduce_i64 needs a C-contiguous int64 array")
        _check(self.ctx.lib, self.ctx.lib.sqgr_comm_allreduce_i64(self.h, _ptr(buf, c_i64p), buf.size, int(op)))
        return buf

    def barrier(self) -> None:
        _check(self.ctx.lib, self.ctx.lib.sqgr_comm_barrier(self.h))

    def info(self) -> tuple[int, int]:
        """(rank, world) as RCCL reports them for this communicator (``ncclCommUserRank`` / ``ncclCommCount``)."""
        r, w = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)
        _check(self.ctx.lib, self.ctx.lib.sqgr_comm_info(self.h, _ptr(r, c_i32p), _ptr(w, c_i32p)))
        return int(r[0]), int(w[0])

    def close(self) -> None:
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_comm_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """Device-resident CSR (``sqgr_graph``) built from a scipy sparse matrix."""

    def __init__(self, ctx: Context, adj: Any, with_data: bool = True):
        from scipy import sparse

        adj = sparse.csr_matrix(adj) if not sparse.isspmatrix_csr(adj) else adj
        if adj.shape[0] != adj.shape[1]:
            raise ValueError(f"Expected a square adjacency matrix, found shape `{adj.shape}`.")
        self.ctx = ctx
        self.n = adj.shape[0]
        self.nnz = int(adj.nnz)
        indptr = _as(adj.indptr, np.int64)
        indices = _as(adj.indices, np.int32)
        h = C.c_void_p()
        if with_data and adj.dtype != np.float32:  # float64 (and integer / bool) weights go over as float64: exact
            data = _as(adj.data, np.float64)
            rc = ctx.lib.sqgr_graph_create_f64(ctx.h, self.n, self.nnz, _ptr(indptr, c_i64p), _ptr(indices, c_i32p), _ptr(data, c_f64p), C.byref(h))
        else:
            data = _as(adj.data, np.float32) if with_data else None
            rc = ctx.lib.sqgr_graph_create(ctx.h, self.n, self.nnz, _ptr(indptr, c_i64p), _ptr(indices, c_i32p), _ptr(data, c_f32p), C.byref(h))
        _check(ctx.lib, rc)
        self.h = h
        self._twin = None  # (order, renumbered Graph): see renumbered()

    def renumbered(self, order: np.ndarray) -> "Graph":
        """The twin ``P A P^T`` of this graph (structure only) for ``order[new] = old``, built on the device
        (``sqgr_graph_renumbered``); kept with the graph and reused while ``order`` stays the same."""
        order = _as(order, np.int32)
        if self._twin is not None and np.array_equal(self._twin[0], order):
            return self._twin[1]
        if len(order) != self.n:
            raise ValueError(f"Expected an order of {self.n} observations, found {len(order)}.")
        h = C.c_void_p()
        _check(self.ctx.lib, self.ctx.lib.sqgr_graph_renumbered(self.ctx.h, self.h, _ptr(order, c_i32p), C.byref(h)))
        twin = Graph.__new__(Graph)
        twin.ctx, twin.n, twin.nnz, twin.h, twin._twin = self.ctx, self.n, self.nnz, h, None
        if self._twin is not None:
            self._twin[1].close()
        self._twin = (order.copy(), twin)
        return twin

    def close(self) -> None:
        if getattr(self, "_twin", None) is not None:
            self._twin[1].close()
            self._twin = None
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_graph_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


# ---- resident graphs across calls ---------------------------------------------------------------------------------------------------
# `sq.gr.spatial_neighbors_*` hands its CSR to the statistics through `adata.obsp`; the statistics of one analysis
# (nhood_enrichment, interaction_matrix, spatial_autocorr, ...) all read the same matrix.  Uploading it (host-side validation,
# 12 bytes per edge over PCIe, COO expansion, symmetric half list) once instead of once per call is what this small cache is
# for.  Keyed by CONTENT (xxh3 of indptr / indices / data, ~10 GB/s), so an `adata.obsp` entry edited in place is never served
# stale; a few entries per context, least recently used evicted and freed.
_GRAPH_CACHE_SLOTS = 4
_graph_cache: "dict[tuple, Graph]" = {}


_graph_cache_lock = threading.Lock()


def _fingerprint(adj: Any, with_data: bool) -> tuple:
    try:
        import xxhash

        h = xxhash.xxh3_128()
    except ImportError:  # optional dependency: hashlib is ~5x slower on a 1e6-spot graph, still far below the upload it saves
        import hashlib

        h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(adj.indptr).view(np.uint8))
    h.update(np.ascontiguousarray(adj.indices).view(np.uint8))
    if with_data:
        h.update(np.ascontiguousarray(adj.data).view(np.uint8))
    return (adj.shape, int(adj.nnz), str(adj.indptr.dtype), str(adj.indices.dtype), str(adj.data.dtype) if with_data else "", h.digest())


def cached_graph(ctx: Context, adj: Any, with_data: bool = True) -> "Graph":
    """A device-resident copy of ``adj`` that survives the call: reused when the same matrix (by content) comes again.  The
    caller must NOT close it.  ``SQGR_GRAPH_CACHE=0`` disables the cache (a fresh upload that the cache frees on eviction)."""
    from scipy import sparse

    adj = sparse.csr_matrix(adj) if not sparse.isspmatrix_csr(adj) else adj
    if os.environ.get("SQGR_GRAPH_CACHE", "1") == "0":
        clear_graph_cache()  # (takes the lock itself)
    key = (id(ctx), bool(with_data)) + _fingerprint(adj, with_data)
    with _graph_cache_lock:
        g = _graph_cache.pop(key, None)
        if g is None or getattr(g, "h", None) is None:
            g = Graph(ctx, adj, with_data=with_data)
        _graph_cache[key] = g  # most recently used last
        while len(_graph_cache) > _GRAPH_CACHE_SLOTS:
            _graph_cache.pop(next(iter(_graph_cache))).close()
    return g


def clear_graph_cache() -> None:
    """Free every cached device graph."""
    with _graph_cache_lock:
        while _graph_cache:
            _graph_cache.popitem()[1].close()


import atexit  # noqa: E402

atexit.register(clear_graph_cache)  # before the contexts they live on are torn down


def spatial_order_device(ctx: Context, xy: np.ndarray) -> np.ndarray:
    """``order[new] = old`` along the Z-order curve of the coordinates ``xy`` (n x 2), computed on the device (``sqgr_spatial_order``:
    ~1 ms per million observations + the upload)."""
    xy = np.ascontiguousarray(np.asarray(xy, dtype=np.float64)[:, :2])
    out = np.empty(len(xy), dtype=np.int32)
    _check(ctx.lib, ctx.lib.sqgr_spatial_order(ctx.h, _ptr(xy, c_f64p), len(xy), _ptr(out, c_i32p)))
    return out


def nhood_counts(ctx: Context, g: Graph, labels: np.ndarray, n_cls: int) -> np.ndarray:
    labels = _as(labels, np.int32)
    out = np.zeros((n_cls, n_cls), dtype=np.uint32)
    _check(ctx.lib, ctx.lib.sqgr_nhood_counts(ctx.h, g.h, _ptr(labels, c_i32p), n_cls, _ptr(out, c_u32p)))
    return out


def nhood_counts_batch(ctx: Context, g: Graph, labels: np.ndarray, n_cls: int) -> np.ndarray:
    """labels: (P, N) integer array of shuffled label vectors -> uint32 (P, K, K)."""
    labels = np.asarray(labels)
    if labels.ndim != 2 or labels.shape[1] != g.n:
        raise ValueError(f"Expected labels of shape (n_perms, {g.n}), found {labels.shape}.")
    if labels.size and (labels.min() < 0 or labels.max() >= n_cls):
        raise ValueError("label outside [0, n_cls)")
    lab8 = _as(labels, np.uint8)
    out = np.zeros((labels.shape[0], n_cls, n_cls), dtype=np.uint32)
    _check(
        ctx.lib,
        ctx.lib.sqgr_nhood_counts_batch(ctx.h, g.h, _ptr(lab8, c_u8p), labels.shape[0], n_cls, _ptr(out, c_u32p)),
    )
    return out


def interaction_matrix(ctx: Context, g: Graph, labels: np.ndarray, n_cls: int, weights: bool) -> np.ndarray:
    labels = _as(labels, np.int32)
    out = np.zeros((n_cls, n_cls), dtype=np.float64)
    _check(
        ctx.lib, ctx.lib.sqgr_interaction_matrix(ctx.h, g.h, _ptr(labels, c_i32p), n_cls, int(weights), _ptr(out, c_f64p))
    )
    return out


class NhoodPlan:
    """Resident permutation-test problem (``sqgr_nhood``)."""

    def __init__(self, ctx: Context, g: Graph, labels: np.ndarray, n_cls: int, lib_ids: np.ndarray | None = None, n_libs: int = 0):
        self.ctx, self.g, self.n_cls = ctx, g, int(n_cls)
        labels = _as(labels, np.int32)
        if len(labels) != g.n:
            raise ValueError(f"Expected {g.n} labels, found {len(labels)}.")
        lib = _as(lib_ids, np.int32) if lib_ids is not None else None
        h = C.c_void_p()
        _check(
            ctx.lib,
            ctx.lib.sqgr_nhood_create(
                ctx.h, g.h, _ptr(labels, c_i32p), self.n_cls, _ptr(lib, c_i32p), int(n_libs) if lib is not None else 0, C.byref(h)
            ),
        )
        self.h = h

    def info(self) -> dict[str, Any]:
        """Launch geometry (``sqgr_nhood_info``)."""
        v = np.zeros(12, dtype=np.int64)
        _check(self.ctx.lib, self.ctx.lib.sqgr_nhood_info(self.h, _ptr(v, c_i64p)))
        return {
            "slab_width": int(v[0]), "perms_per_pass": int(v[8]), "row_halves": int(v[9]), "counter_mode": int(v[10]),
            "batches_per_launch": int(v[1]), "blocks_per_batch": int(v[2]), "list_edges": int(v[3]),
            "symmetric": int(v[4]) != 0, "self_loops": int(v[6]), "hist_words": int(v[5]), "generator_group": int(v[7]),
            "partial_bytes_per_chunk": int(v[11]), "partial_bytes_per_launch": int(v[1]) * int(v[2]) * int(v[11]),
        }

    def set_spot_map(self, spot_of: np.ndarray | None) -> None:
        """The plan lives on a renumbered twin of the caller's graph (:meth:`Graph.renumbered`): slab row ``i`` belongs to the
        caller's observation ``spot_of[i]`` — :meth:`run` then returns the moments of the plan on the caller's own graph, bit for bit
        (``sqgr_nhood_set_spot_map``; no libraries, at most 256 clusters; the numpy-stream entry points refuse such a plan)."""
        m = _as(spot_of, np.int32) if spot_of is not None else None
        _check(self.ctx.lib, self.ctx.lib.sqgr_nhood_set_spot_map(self.h, _ptr(m, c_i32p)))

    def set_comm(self, comm: "Comm | None") -> None:
        """Attach an RCCL communicator: :meth:`run` / :meth:`run_pcg64` then return the moments summed over all ranks
        (all-reduced on the device), :meth:`run_pcg64_stats` gathers the ranks' per-permutation counts."""
        self._comm = comm  # keep it alive
        _check(self.ctx.lib, self.ctx.lib.sqgr_nhood_set_comm(self.h, comm.h if comm is not None else None))

    def tune(self, perms_per_pass: int = 0, blocks_per_batch: int = 0, batches_per_launch: int = 0) -> None:
        _check(self.ctx.lib, self.ctx.lib.sqgr_nhood_tune(self.h, perms_per_pass, blocks_per_batch, batches_per_launch))

    def run(
        self, seed: int, perm_begin: int, perm_end: int, shift: np.ndarray | None = None, return_perms: bool = False
    ) -> tuple[np.ndarray, np.ndarray, np.ndarray | None]:
        """-> (sum_d int64 (K,K), sum_d2 uint64 (K,K), perms uint32 (P,K,K) | None) with d = count - shift."""
        k = self.n_cls
        s = _as(shift, np.int64).reshape(-1) if shift is not None else None
        out_sum = np.zeros((k, k), dtype=np.int64)
        out_sq = np.zeros((k, k), dtype=np.uint64)
        perms = np.zeros((max(perm_end - perm_begin, 0), k, k), dtype=np.uint32) if return_perms else None
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_nhood_run(
                self.h,
                C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                int(perm_begin),
                int(perm_end),
                _ptr(s, c_i64p),
                _ptr(out_sum, c_i64p),
                _ptr(out_sq, c_u64p),
                _ptr(perms, c_u32p),
            ),
        )
        return out_sum, out_sq, perms

    def run_pcg64(
        self, states: np.ndarray, shift: np.ndarray | None = None, return_perms: bool = False
    ) -> tuple[np.ndarray, np.ndarray, np.ndarray | None]:
        """Same as :meth:`run` with numpy's own streams reproduced on the device; ``states``: (P, 4) uint64 rows
        ``[state_hi, state_lo, inc_hi, inc_lo]`` of the PCG64 generators (see ``_utils.pcg64_states``)."""
        k = self.n_cls
        states = _as(states, np.uint64)
        if states.ndim != 2 or states.shape[1] != 4:
            raise ValueError(f"Expected states of shape (n_perms, 4), found {states.shape}.")
        s = _as(shift, np.int64).reshape(-1) if shift is not None else None
        out_sum = np.zeros((k, k), dtype=np.int64)
        out_sq = np.zeros((k, k), dtype=np.uint64)
        perms = np.zeros((states.shape[0], k, k), dtype=np.uint32) if return_perms else None
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_nhood_run_pcg64(
                self.h, _ptr(states, c_u64p), states.shape[0], _ptr(s, c_i64p), _ptr(out_sum, c_i64p), _ptr(out_sq, c_u64p),
                _ptr(perms, c_u32p),
            ),
        )
        return out_sum, out_sq, perms

    def run_pcg64_stats(self, states: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """numpy's ``perms.mean(axis=0)`` / ``perms.std(axis=0)`` of the numpy-stream permutation counts, formed on the
        device bit for bit (``sqgr_nhood_run_pcg64_stats``)."""
        k = self.n_cls
        states = _as(states, np.uint64).reshape(-1, 4)
        mean = np.zeros((k, k), dtype=np.float64)
        std = np.zeros((k, k), dtype=np.float64)
        _check(self.ctx.lib, self.ctx.lib.sqgr_nhood_run_pcg64_stats(self.h, _ptr(states, c_u64p), states.shape[0], _ptr(mean, c_f64p), _ptr(std, c_f64p)))
        return mean, std

    def shuffled_labels(self, seed: int, perm: int) -> np.ndarray:
        out = np.zeros(self.g.n, dtype=np.uint8)
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_nhood_shuffled_labels(self.h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), int(perm), _ptr(out, c_u8p)),
        )
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_nhood_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def cooccur_counts(
    ctx: Context,
    x: np.ndarray,
    y: np.ndarray,
    labels: np.ndarray,
    n_cls: int,
    thr2: np.ndarray,
    fma: bool = False,
    shard_index: int = 0,
    shard_count: int = 1,
) -> np.ndarray:
    """`_occur_count` on the GPU -> int64 (K, K, L)."""
    x, y = _as(x, np.float32), _as(y, np.float32)
    labels, thr2 = _as(labels, np.int32), _as(thr2, np.float32)
    if not (len(x) == len(y) == len(labels)):
        raise ValueError("x, y and labels must have the same length")
    out = np.zeros((n_cls, n_cls, len(thr2)), dtype=np.int64)
    _check(
        ctx.lib,
        ctx.lib.sqgr_cooccur_counts(
            ctx.h, _ptr(x, c_f32p), _ptr(y, c_f32p), _ptr(labels, c_i32p), len(x), n_cls, _ptr(thr2, c_f32p), len(thr2),
            int(fma), shard_index, shard_count, _ptr(out, c_i64p),
        ),
    )
    return out


class DeviceMatrix:
    """The (cells x features) expression matrix resident on the device (``sqgr_matrix``), uploaded once per call: a dense
    row-major float64 / float32 array (also a column range of one, through its row pitch) or a scipy CSR / CSC matrix as it
    is — index arrays int32 / int64, values float32 / float64; other value types are converted to float64 on the host.
    Feature blocks are cut out of it (and sparse ones densified, float32 widened) on the device.

    ``stream_columns=w``: a dense matrix of ``STREAM_MIN_BYTES`` and more is uploaded in column blocks of ``w`` columns, first block
    first, by a background thread on the context's copy stream (``sqgr_matrix_upload_columns``) — the constructor returns at once and
    ``wait_columns(c)`` blocks until columns ``[0, c)`` have arrived (``AutocorrPlan.from_columns`` / ``from_column_list`` call it): the
    upload hides behind the work on the first blocks (config 3: 0.29 s of PCIe behind 0.58 s of kernels)."""

    STREAM_MIN_BYTES = 256 << 20

    def __init__(self, ctx: Context, x: Any, stream_columns: int | None = None):
        from scipy import sparse

        self._thread, self._arrived, self._error, self._cond = None, 0, None, None

        self.ctx = ctx
        h = C.c_void_p()
        if sparse.issparse(x):
            if not (sparse.isspmatrix_csr(x) or sparse.isspmatrix_csc(x)):
                x = sparse.csr_matrix(x)
            if x.dtype not in (np.float32, np.float64):
                x = x.astype(np.float64)
            if x.indices.dtype != x.indptr.dtype or x.indices.dtype not in (np.int32, np.int64):
                x = type(x)((x.data, x.indices.astype(np.int64), x.indptr.astype(np.int64)), shape=x.shape)
            fn = ctx.lib.sqgr_matrix_create_csr if sparse.isspmatrix_csr(x) else ctx.lib.sqgr_matrix_create_csc
            for attempt in range(2):
                indptr, indices, data = np.ascontiguousarray(x.indptr), np.ascontiguousarray(x.indices), np.ascontiguousarray(x.data)
                try:
                    _check(
                        ctx.lib,
                        fn(ctx.h, x.shape[0], x.shape[1], int(x.nnz), indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
                           indices.dtype.itemsize, data.ctypes.data_as(C.c_void_p), data.dtype.itemsize, C.byref(h)),
                    )
                    break
                except SqgrError as exc:
                    # the library checks the canonical format where the arrays land (no O(nnz) pass on the host for the usual,
                    # canonical, matrix); a matrix that is not is brought into it here the way `toarray()` would read it
                    if attempt or not ("not sorted" in str(exc) or "duplicate" in str(exc)):
                        raise
                    x = x.copy()
                    x.sum_duplicates()  # sorts the indices as well
            self.kind = "csr" if sparse.isspmatrix_csr(x) else "csc"
        else:
            x = np.asarray(x)
            if x.ndim != 2:
                raise ValueError(f"DeviceMatrix needs a 2-D array, found shape `{x.shape}`.")
            if x.dtype not in (np.float32, np.float64):
                x = np.ascontiguousarray(x, dtype=np.float64)
            if x.strides[1] != x.itemsize or x.strides[0] % x.itemsize or x.strides[0] < x.shape[1] * x.itemsize:
                x = np.ascontiguousarray(x)  # not a row-major array or a column range of one
            if stream_columns and x.shape[0] * x.shape[1] * x.itemsize >= self.STREAM_MIN_BYTES and x.shape[1] > stream_columns:
                import threading

                _check(ctx.lib, ctx.lib.sqgr_matrix_alloc_dense(ctx.h, x.itemsize, x.shape[0], x.shape[1], C.byref(h)))
                self._cond = threading.Condition()
                # pieces of 1/32 of a feature block (~1 ms of PCIe each): the small copies of the block being scored (row sums, scores,
                # the p-value reductions) queue behind whatever piece is on the bus — config 3 with whole feature blocks as pieces
                # 0.79 s, eighths 0.69 s, 1/32 0.67 s (tools/upload_overlap_diag.py; uploaded whole in front: 0.87 s)
                ld, width = x.strides[0] // x.itemsize, max(64, int(stream_columns) // max(1, int(os.environ.get("SQGR_UPLOAD_PIECES", "32"))))

                def upload() -> None:  # (ctypes releases the GIL inside the call; `x` is kept alive by the closure)
                    try:
                        for c0 in range(0, x.shape[1], width):
                            c1 = min(x.shape[1], c0 + width)
                            _check(ctx.lib, ctx.lib.sqgr_matrix_upload_columns(h, C.c_void_p(x.ctypes.data + c0 * x.itemsize), ld, c0, c1 - c0))
                            with self._cond:
                                self._arrived = c1
                                self._cond.notify_all()
                    except BaseException as exc:  # handed to the waiting thread
                        with self._cond:
                            self._error = exc
                            self._cond.notify_all()

                self._thread = threading.Thread(target=upload, name="sqgr-matrix-upload", daemon=True)
                self._thread.start()
            else:
                _check(
                    ctx.lib,
                    ctx.lib.sqgr_matrix_create_dense(ctx.h, x.ctypes.data_as(C.c_void_p), x.itemsize, x.shape[0], x.shape[1], x.strides[0] // x.itemsize, C.byref(h)),
                )
            self.kind = "dense"
        self.shape, self.dtype = x.shape, x.dtype
        self.h = h

    def wait_columns(self, c1: int | None = None) -> None:
        """Block until columns ``[0, c1)`` (default: all) of a streamed matrix have arrived; raises what the upload raised."""
        if self._cond is None:
            return
        want = self.shape[1] if c1 is None else min(int(c1), self.shape[1])
        with self._cond:
            while self._arrived < want and self._error is None:
                self._cond.wait()
            if self._error is not None:
                raise self._error

    def close(self) -> None:
        if getattr(self, "_thread", None) is not None:
            self._thread.join()  # the upload writes into the array that is about to be released
            self._thread = None
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_matrix_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class AutocorrPlan:
    """Resident feature block for Moran's I / Geary's C (``sqgr_autocorr``).  ``vals``: (G, N) float64."""

    MODES = {"moran": 0, "geary": 1}

    @classmethod
    def from_columns(cls, ctx: Context, g: Graph, matrix: "DeviceMatrix", col0: int, n_features: int) -> "AutocorrPlan":
        """Features = columns ``[col0, col0 + n_features)`` of a device-resident (cells x genes) matrix."""
        self = cls.__new__(cls)
        self.ctx, self.g, self.G = ctx, g, int(n_features)
        matrix.wait_columns(int(col0) + int(n_features))
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.sqgr_autocorr_create_cols(ctx.h, g.h, matrix.h, int(col0), int(n_features), C.byref(h)))
        self.h = h
        return self

    @classmethod
    def from_column_list(cls, ctx: Context, g: Graph, matrix: "DeviceMatrix", cols: np.ndarray) -> "AutocorrPlan":
        """Features = columns ``cols`` (any subset, any order) of a device-resident (cells x genes) matrix."""
        self = cls.__new__(cls)
        cols = _as(cols, np.int32)
        self.ctx, self.g, self.G = ctx, g, int(len(cols))
        matrix.wait_columns()
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.sqgr_autocorr_create_colidx(ctx.h, g.h, matrix.h, _ptr(cols, c_i32p), self.G, C.byref(h)))
        self.h = h
        return self

    def __init__(self, ctx: Context, g: Graph, vals: np.ndarray):
        vals = np.asarray(vals, dtype=np.float64)
        if vals.ndim != 2 or vals.shape[1] != g.n:
            raise ValueError(f"Expected vals of shape (n_features, {g.n}), found {vals.shape}.")
        self.ctx, self.g, self.G = ctx, g, vals.shape[0]
        h = C.c_void_p()
        if not vals.flags.c_contiguous and vals.T.strides[1] == vals.itemsize and vals.shape[0] > 1:
            # a slice of a transposed cell-major matrix (`adata.X[:, genes].T`): hand it over cell-major instead of
            # materialising the transpose on the host
            cm = np.ascontiguousarray(vals.T)
            _check(ctx.lib, ctx.lib.sqgr_autocorr_create_cm(ctx.h, g.h, _ptr(cm, c_f64p), self.G, C.byref(h)))
        else:
            vals = _as(vals, np.float64)
            _check(ctx.lib, ctx.lib.sqgr_autocorr_create(ctx.h, g.h, _ptr(vals, c_f64p), self.G, C.byref(h)))
        self.h = h

    def scores(self, mode: str) -> np.ndarray:
        out = np.zeros(self.G, dtype=np.float64)
        _check(self.ctx.lib, self.ctx.lib.sqgr_autocorr_scores(self.h, self.MODES[mode], _ptr(out, c_f64p)))
        return out

    def perms(self, mode: str, perm_idx: np.ndarray | None = None, seed: int = 0, perm_begin: int = 0, perm_end: int = 0) -> np.ndarray:
        if perm_idx is not None:
            perm_idx = _as(perm_idx, np.int32)
            if perm_idx.ndim != 2 or perm_idx.shape[1] != self.g.n:
                raise ValueError(f"Expected perm_idx of shape (n_perms, {self.g.n}), found {perm_idx.shape}.")
            perm_begin, perm_end = 0, perm_idx.shape[0]
        out = np.zeros((perm_end - perm_begin, self.G), dtype=np.float64)
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_autocorr_perms(
                self.h, self.MODES[mode], _ptr(perm_idx, c_i32p), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                int(perm_begin), int(perm_end), _ptr(out, c_f64p),
            ),
        )
        return out

    def perms_pcg64(self, mode: str, pcg_states: np.ndarray) -> np.ndarray:
        """Permutation scores under numpy's own streams, drawn on the device (``sqgr_autocorr_perms_pcg64``);
        ``pcg_states``: (n_perms, 4) uint64 from :func:`squidpy_amd._utils.pcg64_states`."""
        states = _as(pcg_states, np.uint64).reshape(-1, 4)
        out = np.zeros((states.shape[0], self.G), dtype=np.float64)
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_autocorr_perms_pcg64(self.h, self.MODES[mode], _ptr(states, c_u64p), states.shape[0], _ptr(out, c_f64p)),
        )
        return out

    def perm_stats(
        self, mode: str, score: np.ndarray, *, perm_idx: np.ndarray | None = None, pcg_states: np.ndarray | None = None, seed: int = 0,
        perm_begin: int = 0, perm_end: int = 0, only_feature: bool = False,
    ) -> dict[str, np.ndarray]:
        """The permutation test reduced on the device (``sqgr_autocorr_perm_stats``): per feature the number of permutation
        scores ``>= score``, and numpy's ``sum`` / ``std`` / ``var`` of the scores over the permutation axis — the (P, G)
        scores themselves never leave the GPU.  Permutations: injected (``perm_idx``), numpy streams (``pcg_states``) or the
        device generator for ``[perm_begin, perm_end)``.  ``only_feature``: the plan's single feature is the only one of the
        whole call — numpy reduces a (P, 1) array in its contiguous (pairwise) order, not row by row."""
        score = _as(score, np.float64)
        if score.shape != (self.G,):
            raise ValueError(f"Expected `{self.G}` observed scores, found shape `{score.shape}`.")
        states = None
        if perm_idx is not None:
            perm_idx = _as(perm_idx, np.int32)
            if perm_idx.ndim != 2 or perm_idx.shape[1] != self.g.n:
                raise ValueError(f"Expected perm_idx of shape (n_perms, {self.g.n}), found {perm_idx.shape}.")
            perm_begin, perm_end = 0, perm_idx.shape[0]
        elif pcg_states is not None:
            states = _as(pcg_states, np.uint64).reshape(-1, 4)
            perm_begin, perm_end = 0, states.shape[0]
        ge = np.zeros(self.G, dtype=np.int64)
        ssum, sstd, svar = (np.zeros(self.G, dtype=np.float64) for _ in range(3))
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_autocorr_perm_stats(
                self.h, self.MODES[mode], _ptr(perm_idx, c_i32p), _ptr(states, c_u64p), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                int(perm_begin), int(perm_end), _ptr(score, c_f64p), _ptr(ge, c_i64p), _ptr(ssum, c_f64p), _ptr(sstd, c_f64p), _ptr(svar, c_f64p),
                1 if only_feature else 0,
            ),
        )
        return {"n_ge": ge, "sum": ssum, "std": sstd, "var": svar}

    def close(self) -> None:
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_autocorr_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def autocorr_perm_indices(ctx: Context, n: int, seed: int, perm_begin: int, perm_end: int) -> np.ndarray:
    out = np.zeros((perm_end - perm_begin, n), dtype=np.int32)
    _check(
        ctx.lib,
        ctx.lib.sqgr_autocorr_perm_indices(ctx.h, n, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), perm_begin, perm_end, _ptr(out, c_i32p)),
    )
    return out


METRICS = {
    "euclidean": 0, "l2": 0, "minkowski": 0, "p": 0,  # sklearn's default minkowski p=2
    "manhattan": 1, "cityblock": 1, "l1": 1,
    "chebyshev": 2, "infinity": 2,
    "canberra": 3,  # nearest-neighbour statistics only (a BallTree metric; KDTree / Ripley's L do not take it)
}


def sqrt_thresholds(radii: np.ndarray) -> np.ndarray:
    """For every radius r the largest float64 t with ``sqrt(t) <= r`` (IEEE correctly-rounded sqrt), so that the
    device can decide ``sqrt(d2) <= r`` as ``d2 <= t`` without a square root.  Negative radii -> -1 (nothing)."""
    r = np.asarray(radii, dtype=np.float64)
    t = r * r
    t[r < 0] = -1.0
    ok = r >= 0
    for _ in range(8):  # a handful of ulps at most
        up = np.nextafter(t, np.inf)
        grow = ok & np.isfinite(up) & (np.sqrt(np.where(ok, up, 0.0)) <= r)
        shrink = ok & (np.sqrt(np.where(ok, t, 0.0)) > r)
        if not (grow.any() or shrink.any()):
            break
        t = np.where(grow, up, t)
        t = np.where(shrink, np.nextafter(t, -np.inf), t)
    return t


def pair_counts(ctx: Context, xy: np.ndarray, support: np.ndarray, metric: str = "euclidean") -> np.ndarray:
    """#ordered non-self pairs within every radius of ``support`` (ascending) -> int64 (S,)."""
    xy = _as(xy, np.float64)
    support = _as(support, np.float64)
    m = METRICS[metric]
    thr = sqrt_thresholds(support) if m == 0 else support
    out = np.zeros(len(support), dtype=np.int64)
    _check(ctx.lib, ctx.lib.sqgr_pair_counts(ctx.h, _ptr(xy, c_f64p), xy.shape[0], _ptr(thr, c_f64p), len(thr), m, _ptr(out, c_i64p)))
    return out


PAIR_BATCH_MAX_SETS = 65535  # sets per launch of sqgr_pair_counts_batch (the set index is a grid dimension)


def pair_counts_batch(ctx: Context, sets: "list[np.ndarray]", support: np.ndarray, metric: str = "euclidean") -> np.ndarray:
    """:func:`pair_counts` of several point sets -> int64 (n_sets, S).  One launch serves up to 65 535 sets of similar size: the
    launch grid is sized by its largest set, so the sets are grouped by size class (sizes within a factor of 4) — one huge
    cluster among a hundred small ones no longer makes every small set launch the huge set's tile grid — and any number of
    sets (``n_simulations`` has no upper limit in the reference, gr/_ripley.py:171) is cut into launches of at most 65 535."""
    support = _as(support, np.float64)
    m = METRICS[metric]
    thr = sqrt_thresholds(support) if m == 0 else support
    out = np.zeros((len(sets), len(support)), dtype=np.int64)
    if not sets:
        return out
    arrs = [_as(a, np.float64).reshape(-1, 2) for a in sets]
    sizes = np.array([len(a) for a in arrs], dtype=np.int64)
    size_class = np.zeros(len(arrs), dtype=np.int64)
    big = sizes > 256                                        # (one tile or less: all in one class)
    size_class[big] = np.ceil(np.log2(sizes[big] / 256.0) / 2.0).astype(np.int64)
    for cls in np.unique(size_class):
        members = np.nonzero(size_class == cls)[0]
        for c0 in range(0, len(members), PAIR_BATCH_MAX_SETS):
            idx = members[c0 : c0 + PAIR_BATCH_MAX_SETS]
            offsets = np.zeros(len(idx) + 1, dtype=np.int64)
            np.cumsum(sizes[idx], out=offsets[1:])
            xy = np.ascontiguousarray(np.concatenate([arrs[i] for i in idx], axis=0)) if offsets[-1] else np.zeros((1, 2))
            part = np.zeros((len(idx), len(support)), dtype=np.int64)
            _check(ctx.lib, ctx.lib.sqgr_pair_counts_batch(ctx.h, _ptr(xy, c_f64p), _ptr(offsets, c_i64p), len(idx), _ptr(thr, c_f64p), len(thr), m, _ptr(part, c_i64p)))
            out[idx] = part
    return out


def knn_dist(ctx: Context, query: np.ndarray, ref: np.ndarray, k: int, metric: str = "euclidean") -> np.ndarray:
    """k nearest-neighbour distances of every query point -> float64 (nq, k), ascending."""
    query, ref = _as(query, np.float64), _as(ref, np.float64)
    m = METRICS[metric]
    if k > ref.shape[0]:
        raise ValueError(
            f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {ref.shape[0]}, "
            f"n_samples = {query.shape[0]}"
        )
    out = np.zeros((query.shape[0], k), dtype=np.float64)
    _check(ctx.lib, ctx.lib.sqgr_knn_dist(ctx.h, _ptr(query, c_f64p), query.shape[0], _ptr(ref, c_f64p), ref.shape[0], k, m, _ptr(out, c_f64p)))
    return np.sqrt(out) if m == 0 else out


def pcg64_permutations(ctx: Context, n: int, states: np.ndarray) -> np.ndarray:
    """numpy's ``Generator.permutation(n)`` for every generator state in ``states`` (P, 4), computed on the device."""
    states = _as(states, np.uint64)
    out = np.zeros((states.shape[0], n), dtype=np.int32)
    _check(ctx.lib, ctx.lib.sqgr_pcg64_permutations(ctx.h, n, _ptr(states, c_u64p), states.shape[0], _ptr(out, c_i32p)))
    return out


def knn_self(ctx: Context, xy: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """k nearest other samples of every sample -> (dist float64 (n, k) ascending, idx int32 (n, k)); what
    ``NearestNeighbors(n_neighbors=k).fit(xy).kneighbors()`` returns."""
    xy = _as(xy, np.float64)
    n = xy.shape[0]
    if xy.ndim != 2 or xy.shape[1] != 2:
        raise ValueError(f"Expected 2-D coordinates of shape (n, 2), found {xy.shape}.")
    if k + 1 > n:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k + 1}, n_samples_fit = {n}, n_samples = {n}")
    idx = np.zeros((n, k), dtype=np.int32)
    d2 = np.zeros((n, k), dtype=np.float64)
    _check(ctx.lib, ctx.lib.sqgr_knn_self(ctx.h, _ptr(xy, c_f64p), n, k, _ptr(idx, c_i32p), _ptr(d2, c_f64p)))
    return np.sqrt(d2), idx


def radius_self(ctx: Context, xy: np.ndarray, radius: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """all other samples within ``radius`` of every sample -> CSR (indptr int64 (n+1,), idx int32, dist float64)."""
    xy = _as(xy, np.float64)
    n = xy.shape[0]
    if xy.ndim != 2 or xy.shape[1] != 2:
        raise ValueError(f"Expected 2-D coordinates of shape (n, 2), found {xy.shape}.")
    indptr = np.zeros(n + 1, dtype=np.int64)
    _check(ctx.lib, ctx.lib.sqgr_radius_self(ctx.h, _ptr(xy, c_f64p), n, float(radius), _ptr(indptr, c_i64p), None, None, 0))
    nnz = int(indptr[-1])
    idx = np.zeros(max(nnz, 1), dtype=np.int32)
    d2 = np.zeros(max(nnz, 1), dtype=np.float64)
    _check(ctx.lib, ctx.lib.sqgr_radius_self(ctx.h, _ptr(xy, c_f64p), n, float(radius), _ptr(indptr, c_i64p), _ptr(idx, c_i32p), _ptr(d2, c_f64p), max(nnz, 1)))
    return indptr, idx[:nnz], np.sqrt(d2[:nnz])


def ligrec_counts(
    ctx: Context,
    data_csc: Any,
    clustering: np.ndarray,
    n_cls: int,
    inv_counts: np.ndarray,
    interactions: np.ndarray,
    cluster_pairs: np.ndarray,
    obs: np.ndarray,
    valid: np.ndarray,
    *,
    seed: int = 0,
    pcg_states: np.ndarray | None = None,
    perm_begin: int = 0,
    perm_end: int = 0,
    return_first_groups: bool = False,
) -> np.ndarray | tuple[np.ndarray, np.ndarray]:
    """Permutation counts of the ligand-receptor test (``_score_permutations``, gr/_ligrec.py:616-673).

    ``data_csc``: scipy CSC matrix (n_cells, n_genes), float64.  Returns int64 (n_interactions, n_cluster_pairs);
    with ``return_first_groups`` also the (n_cls, n_genes) group means of the first permutation of the range."""
    from scipy import sparse

    m = sparse.csc_matrix(data_csc)
    if not m.has_sorted_indices:
        m = m.sorted_indices()
    n_cells, n_genes = m.shape
    colptr = _as(m.indptr, np.int64)
    rowidx = _as(m.indices, np.int32)
    values = _as(m.data, np.float64)
    clustering = _as(clustering, np.int32)
    if clustering.shape != (n_cells,):
        raise ValueError(f"Expected `{n_cells}` cluster labels, found `{clustering.shape}`.")
    inv_counts = _as(inv_counts, np.float64)
    interactions = _as(interactions, np.int32).reshape(-1, 2)
    cluster_pairs = _as(cluster_pairs, np.int32).reshape(-1, 2)
    n_inter, n_cp = interactions.shape[0], cluster_pairs.shape[0]
    obs = _as(obs, np.float64)
    valid = _as(valid, np.uint8)
    if obs.shape != (n_inter, n_cp) or valid.shape != (n_inter, n_cp) or inv_counts.shape != (n_cls,):
        raise ValueError("`obs`/`valid` must have shape (n_interactions, n_cluster_pairs) and `inv_counts` (n_cls,).")
    states = None
    if pcg_states is not None:
        states = _as(pcg_states, np.uint64).reshape(-1, 4)
        if states.shape[0] != perm_end - perm_begin:
            raise ValueError("`pcg_states` must hold one row per permutation of the range.")
    out = np.zeros((n_inter, n_cp), dtype=np.int64)
    groups = np.zeros((n_cls, n_genes), dtype=np.float64) if return_first_groups else None
    _check(
        ctx.lib,
        ctx.lib.sqgr_ligrec_counts(
            ctx.h, n_cells, n_genes, n_cls, _ptr(colptr, c_i64p), _ptr(rowidx, c_i32p), _ptr(values, c_f64p),
            _ptr(clustering, c_i32p), _ptr(inv_counts, c_f64p), _ptr(interactions, c_i32p), n_inter, _ptr(cluster_pairs, c_i32p),
            n_cp, _ptr(obs, c_f64p), _ptr(valid, c_u8p), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _ptr(states, c_u64p),
            int(perm_begin), int(perm_end), _ptr(out, c_i64p), _ptr(groups, c_f64p),
        ),
    )
    return (out, groups) if return_first_groups else out


class DevicePoints:
    """A 2-D point set (and an optional integer label per point) resident on the device (``sqgr_points``)."""

    def __init__(self, ctx: Context, xy: np.ndarray, labels: np.ndarray | None = None):
        xy = _as(xy, np.float64)
        if xy.ndim != 2 or xy.shape[1] != 2:
            raise ValueError(f"Expected 2-D coordinates of shape (n, 2), found {xy.shape}.")
        lab = _as(labels, np.int32) if labels is not None else None
        if lab is not None and lab.shape != (xy.shape[0],):
            raise ValueError("`labels` must hold one entry per point.")
        self.ctx, self.n = ctx, xy.shape[0]
        h = C.c_void_p()
        _check(ctx.lib, ctx.lib.sqgr_points_create(ctx.h, _ptr(xy, c_f64p), _ptr(lab, c_i32p), self.n, C.byref(h)))
        self.h = h

    def knn_hist(self, refs: np.ndarray, k: int, edges: np.ndarray, metric: str = "euclidean", exclude_label: int = -1) -> np.ndarray:
        """``np.histogram(kneighbors(queries, k) distances, bins=edges)[0]`` for the resident points (those whose label is
        not ``exclude_label``) against ``refs``; int64 (len(edges) - 1,)."""
        refs = _as(refs, np.float64)
        edges = _as(edges, np.float64)
        out = np.zeros(len(edges) - 1, dtype=np.int64)
        _check(
            self.ctx.lib,
            self.ctx.lib.sqgr_knn_hist(self.ctx.h, self.h, int(exclude_label), _ptr(refs, c_f64p), refs.shape[0], int(k), METRICS[metric],
                                       _ptr(edges, c_f64p), len(edges), _ptr(out, c_i64p)),
        )
        return out

    def close(self) -> None:
        if getattr(self, "h", None):
            self.ctx.lib.sqgr_points_destroy(self.h)
            self.h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
