"""TEST INFRASTRUCTURE — not product code.  ctypes binding + build of ``oracle/c/sqgr_cpu.c`` (the C
restatement of the reference's numba kernels used as bench.py's timed ``cpu_baseline`` and as a second
checker).  ``build(native=True)`` compiles with ``-march=native`` on the machine that will time it."""

from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "sqgr_cpu.c")
OUT_DIR = os.path.join(HERE, "_build")


def build(native: bool = False, force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "liboracle_c_native.so" if native else "liboracle_c.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(SRC):
        return out
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        if os.path.exists(out):
            return out
        raise RuntimeError("no C compiler for the oracle C port")
    cmd = [cc, "-O3", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off"] + (["-march=native"] if native else []) + ["-o", out, SRC]
    subprocess.check_call(cmd)
    return out


_libs: dict[bool, C.CDLL] = {}


def lib(native: bool = False) -> C.CDLL:
    if native not in _libs:
        try:
            path = build(native=native)
        except Exception:
            if not native:
                raise
            path = build(native=False)
        _libs[native] = C.CDLL(path)
    return _libs[native]


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def nenrich(indices: np.ndarray, indptr: np.ndarray, clustering: np.ndarray, k: int, parallel: bool = False, native: bool = False) -> np.ndarray:
    """gr/_nhood.py:54-141 -> uint32 (K, K)."""
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    clustering = np.ascontiguousarray(clustering, dtype=np.uint32)
    out = np.zeros((k, k), dtype=np.uint32)
    rc = lib(native).sq_nenrich(_p(indices, C.c_uint32), _p(indptr, C.c_uint32), _p(clustering, C.c_uint32),
                                C.c_int64(len(indptr) - 1), C.c_int(k), _p(out, C.c_uint32), C.c_int(int(parallel)))
    if rc != 0:
        raise MemoryError("sq_nenrich scratch allocation failed")
    return out


def occur_count(x, y, thresholds, labs, k: int, parallel: bool = False, native: bool = False) -> np.ndarray:
    """gr/_ppatterns.py:283-310 -> int64 (K, K, L)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    thr = np.ascontiguousarray(thresholds, dtype=np.float32)
    labs = np.ascontiguousarray(labs, dtype=np.int32)
    out = np.zeros((k, k, len(thr)), dtype=np.int64)
    rc = lib(native).sq_occur_count(_p(x, C.c_float), _p(y, C.c_float), _p(thr, C.c_float), _p(labs, C.c_int32),
                                    C.c_int64(len(x)), C.c_int(k), C.c_int(len(thr)), _p(out, C.c_int64), C.c_int(int(parallel)))
    if rc != 0:
        raise MemoryError("sq_occur_count scratch allocation failed")
    return out


def _autocorr(fn: str, g, vals, parallel: bool, native: bool) -> np.ndarray:
    data = np.ascontiguousarray(g.data, dtype=np.float64)
    indices = np.ascontiguousarray(g.indices, dtype=np.int32)
    indptr = np.ascontiguousarray(g.indptr, dtype=np.int32)
    X = np.ascontiguousarray(vals, dtype=np.float64)
    out = np.zeros(X.shape[0], dtype=np.float64)
    getattr(lib(native), fn)(_p(data, C.c_double), _p(indices, C.c_int32), _p(indptr, C.c_int32), _p(X, C.c_double),
                             C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _p(out, C.c_double), C.c_int(int(parallel)))
    return out


def morans_i(g, vals, parallel: bool = False, native: bool = False) -> np.ndarray:
    return _autocorr("sq_morans_i", g, vals, parallel, native)


def gearys_c(g, vals, parallel: bool = False, native: bool = False) -> np.ndarray:
    return _autocorr("sq_gearys_c", g, vals, parallel, native)


def ligrec_score(data, perm_labels, inv_counts, mean_obs, interactions, cpairs, valid, parallel: bool = False, native: bool = False) -> np.ndarray:
    """gr/_ligrec.py:616-673 for given shuffled label vectors (n_perms, n_cells) -> int64 (n_inter, n_cpairs)."""
    data = np.ascontiguousarray(data, dtype=np.float64)
    perm_labels = np.ascontiguousarray(perm_labels, dtype=np.int32)
    inv_counts = np.ascontiguousarray(inv_counts, dtype=np.float64)
    mean_obs = np.ascontiguousarray(mean_obs, dtype=np.float64)
    interactions = np.ascontiguousarray(interactions, dtype=np.int32)
    cpairs = np.ascontiguousarray(cpairs, dtype=np.int32)
    valid = np.ascontiguousarray(valid, dtype=np.uint8)
    n_cells, n_genes = data.shape
    out = np.zeros((len(interactions), len(cpairs)), dtype=np.int64)
    rc = lib(native).sq_ligrec_score(
        _p(data, C.c_double), C.c_int64(n_cells), C.c_int(n_genes), _p(perm_labels, C.c_int32), C.c_int64(len(perm_labels)),
        C.c_int(len(inv_counts)), _p(inv_counts, C.c_double), _p(mean_obs, C.c_double), _p(interactions, C.c_int32),
        C.c_int64(len(interactions)), _p(cpairs, C.c_int32), C.c_int(len(cpairs)), _p(valid, C.c_uint8), _p(out, C.c_int64),
        C.c_int(int(parallel)),
    )
    if rc != 0:
        raise MemoryError("sq_ligrec_score scratch allocation failed")
    return out
