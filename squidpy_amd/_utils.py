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


def pcg64_states(seed: int | None, n: int, begin: int = 0, end: int | None = None) -> np.ndarray:
    """Initial PCG64 states of ``spawn_generators(seed, n)[begin:end]`` as (k, 4) uint64 rows
    ``[state_hi, state_lo, inc_hi, inc_lo]`` — what the device needs to continue numpy's streams bit for bit."""
    end = n if end is None else end
    mask = (1 << 64) - 1
    seqs = np.random.SeedSequence(seed).spawn(n)[begin:end]
    out = np.empty((len(seqs), 4), dtype=np.uint64)
    for k, s in enumerate(seqs):
        st = np.random.PCG64(s).state["state"]
        out[k] = (st["state"] >> 64, st["state"] & mask, st["inc"] >> 64, st["inc"] & mask)
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
