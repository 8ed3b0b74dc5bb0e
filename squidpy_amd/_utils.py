"""Host-side helpers with the reference's semantics for the hot path.

Cited reference lines are under /root/reference/src/squidpy."""

from __future__ import annotations

import functools
import logging
import os
import warnings
from typing import Any, Callable

import numpy as np
import pandas as pd
from pandas import CategoricalDtype
from pandas.api.types import infer_dtype

logg = logging.getLogger("squidpy_amd")


def extract_adata_if_sdata(adata: Any, *, table_key: str | None = None) -> Any:
    """gr/_utils.py:25-52.  A SpatialData-like object is recognised by its ``tables`` mapping."""
    if hasattr(adata, "tables") and not hasattr(adata, "obs"):
        if table_key is None:
            raise TypeError("missing required keyword-only argument: 'table_key'")
        if table_key not in adata.tables:
            raise ValueError(
                f"Table {table_key!r} not found in SpatialData. Available tables: {list(adata.tables.keys())}"
            )
        return adata.tables[table_key]
    return adata


def _assert_categorical_obs(adata: Any, key: str) -> None:
    """gr/_utils.py:55-60."""
    if key not in adata.obs:
        raise KeyError(f"Cluster key `{key}` not found in `adata.obs`.")
    if not isinstance(adata.obs[key].dtype, CategoricalDtype):
        raise TypeError(f"Expected `adata.obs[{key!r}]` to be `categorical`, found `{infer_dtype(adata.obs[key])}`.")


def _assert_connectivity_key(adata: Any, key: str) -> None:
    """gr/_utils.py:63-69."""
    if key not in adata.obsp:
        key_added = key.replace("_connectivities", "")
        raise KeyError(
            f"Spatial connectivity key `{key}` not found in `adata.obsp`. "
            f"Please run `squidpy.gr.spatial_neighbors(..., key_added={key_added!r})` first."
        )


def _assert_spatial_basis(adata: Any, key: str) -> None:
    """gr/_utils.py:72-74."""
    if key not in adata.obsm:
        raise KeyError(f"Spatial basis `{key}` not found in `adata.obsm`.")


def assert_positive(value: float, *, name: str) -> None:
    """_validators.py:68-70."""
    if value <= 0:
        raise ValueError(f"Expected `{name}` to be positive, found `{value}`.")


def assert_key_in_adata(adata: Any, key: str, attr: str) -> None:
    if key not in getattr(adata, attr):
        raise KeyError(f"Key `{key}` not found in `adata.{attr}`.")


def _save_data(adata: Any, *, attr: str, key: str, data: Any, time: Any | None = None) -> None:
    """gr/_utils.py:77-86."""
    getattr(adata, attr)[key] = data
    logg.info("Adding `adata.%s[%r]`", attr, key)


def spawn_generators(seed: int | None, n: int) -> list[np.random.Generator]:
    """_utils.py:240-241 — the reference's per-permutation numpy streams (``rng="numpy"`` mode)."""
    return [np.random.default_rng(s) for s in np.random.SeedSequence(seed).spawn(n)]


def _cpu_count() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        return os.cpu_count() or 1


def get_n_processes(n_cores: int | None) -> int:
    """_utils.py:336-349: `None` is serial, ``-1`` all cores, 0 / < -1 raise, too many warns + clamps.

    The value is validated for API compatibility; the GPU path does not fan out host processes."""
    if n_cores is None:
        return 1
    max_cores = _cpu_count()
    if n_cores == -1:
        return max_cores
    if n_cores < -1 or n_cores == 0:
        raise ValueError(f"Number of cores must be `-1` or a positive integer, got `{n_cores}`.")
    if n_cores > max_cores:
        logg.warning("Requested `n_jobs=%s`, but only `%s` core(s) are available.", n_cores, max_cores)
        return max_cores
    return n_cores


def deprecated_params(params: dict[str, str]) -> Callable[..., Any]:
    """_utils.py:376-404: warn (FutureWarning) and drop deprecated keyword arguments."""

    def decorator(func: Callable[..., Any]) -> Callable[..., Any]:
        @functools.wraps(func)
        def wrapper(*args: Any, **kwargs: Any) -> Any:
            for k in list(kwargs):
                if k in params:
                    warnings.warn(
                        f"Parameter `{k}` of `{func.__name__}()` is deprecated "
                        f"and has no effect. It will be removed in squidpy v{params[k]}.",
                        FutureWarning,
                        stacklevel=2,
                    )
                    kwargs.pop(k)
            return func(*args, **kwargs)

        return wrapper

    return decorator


def category_codes(series: pd.Series) -> tuple[np.ndarray, int]:
    """Category -> code map in ``cat.categories`` order (gr/_nhood.py:196-197); NaN raises KeyError."""
    codes = series.cat.codes.to_numpy()
    if (codes < 0).any():
        raise KeyError(float("nan"))  # the reference's dict lookup `clust_map[nan]` fails the same way
    return codes.astype(np.int32), len(series.cat.categories)


def resolve_seed(seed: int | None) -> int:
    """64-bit key of the device generator.  ``None`` draws fresh entropy, as SeedSequence(None) does."""
    if seed is None:
        return int.from_bytes(os.urandom(8), "little")
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def _pcg64_states_numpy(seed: Any, n: int, begin: int, end: int) -> np.ndarray:
    """The states as numpy itself produces them: one SeedSequence child and one PCG64 object per stream (13 us each)."""
    mask = (1 << 64) - 1
    seqs = np.random.SeedSequence(seed).spawn(n)[begin:end]
    out = np.empty((len(seqs), 4), dtype=np.uint64)
    for k, s in enumerate(seqs):
        st = np.random.PCG64(s).state["state"]
        out[k] = (st["state"] >> 64, st["state"] & mask, st["inc"] >> 64, st["inc"] & mask)
    return out


_SS_INIT_A, _SS_MULT_A, _SS_INIT_B, _SS_MULT_B = 0x43B0D7E5, 0x931E8875, 0x8B51F9DD, 0x58F38DED
_SS_MIX_L, _SS_MIX_R, _M32 = 0xCA01F9DD, 0x4973F715, 0xFFFFFFFF
_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645


def _u32_words(v: int) -> list[int]:
    """numpy's `_coerce_to_uint32_array` of a non-negative int: little-endian 32-bit words, [0] for 0."""
    out = []
    while True:
        out.append(v & _M32)
        v >>= 32
        if v == 0:
            return out


def pcg64_states(seed: int | None, n: int, begin: int = 0, end: int | None = None) -> np.ndarray:
    """Initial PCG64 states of ``spawn_generators(seed, n)[begin:end]`` (/root/reference/src/squidpy/_utils.py:240-241) as (k, 4)
    uint64 rows ``[state_hi, state_lo, inc_hi, inc_lo]`` — what the device needs to continue numpy's streams bit for bit.

    ``SeedSequence(seed).spawn(n)`` + ``PCG64(child)`` restated in vectorised numpy (round 5): the children of one SeedSequence
    differ only in the LAST word of their entropy (the spawn key), so the entropy pool is mixed once up to that word, the last
    mixing step, ``generate_state(4, uint64)`` and the two LCG steps of PCG64's seeding run over all children at once in 32-bit
    limbs.  100 000 streams: 1.2 s through numpy's objects, ~20 ms here; `==` numpy's own states (tests/test_host_logic_cpu.py),
    and anything this restatement does not cover (non-integer entropy, a pool size other than 4) goes through numpy itself."""
    end = n if end is None else end
    ss = np.random.SeedSequence(seed)
    if not isinstance(ss.entropy, int) or ss.entropy < 0 or ss.pool_size != 4 or ss.spawn_key != () or end - begin < 16 or n >= 2**32:
        return _pcg64_states_numpy(seed if seed is not None else ss.entropy, n, begin, end)
    # ---- the part of `mix_entropy` every child shares: run entropy (padded to the pool size when a spawn key follows)
    words = _u32_words(ss.entropy)
    words += [0] * (4 - len(words))
    hc = _SS_INIT_A

    def hashmix(v: int) -> int:
        nonlocal hc
        v ^= hc
        hc = (hc * _SS_MULT_A) & _M32
        v = (v * hc) & _M32
        return v ^ (v >> 16)

    def mix(x: int, y: int) -> int:
        r = (_SS_MIX_L * x - _SS_MIX_R * y) & _M32
        return r ^ (r >> 16)

    pool = [hashmix(words[i]) for i in range(4)]
    for i_src in range(4):
        for i_dst in range(4):
            if i_src != i_dst:
                pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src]))
    for w in words[4:]:
        for i_dst in range(4):
            pool[i_dst] = mix(pool[i_dst], hashmix(w))
    # ---- the child's own word (spawn key (i,), i < 2^32: one word) over all children at once
    key = np.arange(begin, end, dtype=np.uint64)  # uint64 arithmetic masked to 32 bits
    m32 = np.uint64(_M32)
    P = []
    for i_dst in range(4):
        v = key ^ np.uint64(hc)
        hc = (hc * _SS_MULT_A) & _M32
        v = (v * np.uint64(hc)) & m32
        v ^= v >> np.uint64(16)
        r = (np.uint64(_SS_MIX_L) * np.uint64(pool[i_dst]) - np.uint64(_SS_MIX_R) * v) & m32  # (wraps modulo 2^64, then 2^32)
        P.append(r ^ (r >> np.uint64(16)))
    # ---- generate_state(4, uint64): 8 words cycling through the pool
    hb = _SS_INIT_B
    W = []
    for i in range(8):
        v = P[i & 3] ^ np.uint64(hb)
        hb = (hb * _SS_MULT_B) & _M32
        v = (v * np.uint64(hb)) & m32
        W.append(v ^ (v >> np.uint64(16)))
    val = [W[2 * k] | (W[2 * k + 1] << np.uint64(32)) for k in range(4)]  # little-endian uint64 view of the words
    # ---- pcg64_set_seed: initstate = val[0] << 64 | val[1], initseq = val[2] << 64 | val[3];  inc = initseq << 1 | 1;
    # state = ((0 * M + inc) + initstate) * M + inc  (mod 2^128) — in four 32-bit limbs
    def limbs(hi: np.ndarray, lo: np.ndarray) -> list[np.ndarray]:
        return [lo & m32, lo >> np.uint64(32), hi & m32, hi >> np.uint64(32)]

    def add128(a: list, b: list) -> list:
        out, carry = [], np.zeros_like(a[0])
        for x, y in zip(a, b):
            t = x + y + carry
            out.append(t & m32)
            carry = t >> np.uint64(32)
        return out

    def mul128_const(a: list, c: int) -> list:
        cl = [(c >> (32 * k)) & _M32 for k in range(4)]
        out, carry = [], np.zeros_like(a[0])
        for k in range(4):  # limb k of the product: sum of a[i] * c[k - i], accumulated in two halves so nothing exceeds 2^64
            lo_acc, hi_acc = carry & m32, carry >> np.uint64(32)
            for i in range(k + 1):
                t = a[i] * np.uint64(cl[k - i])
                lo_acc = lo_acc + (t & m32)
                hi_acc = hi_acc + (t >> np.uint64(32))
            out.append(lo_acc & m32)
            carry = hi_acc + (lo_acc >> np.uint64(32))
        return out

    inc_hi = (val[2] << np.uint64(1)) | (val[3] >> np.uint64(63))
    inc_lo = (val[3] << np.uint64(1)) | np.uint64(1)
    inc = limbs(inc_hi, inc_lo)
    state = add128(mul128_const(add128(inc, limbs(val[0], val[1])), _PCG_MULT), inc)
    out = np.empty((end - begin, 4), dtype=np.uint64)
    out[:, 0] = state[2] | (state[3] << np.uint64(32))
    out[:, 1] = state[0] | (state[1] << np.uint64(32))
    out[:, 2] = inc_hi
    out[:, 3] = inc_lo
    return out


class progress:
    """The reference's progress bar (`show_progress_bar`, _utils.py:168-197 drives tqdm from its worker processes): here the
    host loop over permutation batches / feature blocks / clusters advances it.  Silent when disabled, when tqdm is missing,
    when stderr is not a terminal (logs, tests, batch jobs) or on ranks other than 0."""

    def __init__(self, total: int, unit: str, enabled: bool = True):
        self._bar = None
        if enabled and total > 0:
            try:
                import sys

                from tqdm.auto import tqdm

                if sys.stderr.isatty() and os.environ.get("RANK", "0") == "0":
                    self._bar = tqdm(total=int(total), unit=unit)
            except Exception:  # tqdm missing or unusable: the reference degrades the same way (`tqdm = None`)
                self._bar = None

    def update(self, n: int = 1) -> None:
        if self._bar is not None:
            self._bar.update(n)

    def close(self) -> None:
        if self._bar is not None:
            self._bar.close()
            self._bar = None

    def __enter__(self) -> "progress":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()
