"""``co_occurrence`` and ``spatial_autocorr`` with the reference's signatures on the MI355X path.

Reference: /root/reference/src/squidpy/gr/_ppatterns.py — ``co_occurrence`` :361-428 (``_occur_count`` :283-310,
``_co_occurrence_helper`` :313-358, ``_find_min_max`` :431-440); ``spatial_autocorr`` :56-255 (``_score_helper``
:258-280, ``_p_value_calc`` :443-498, ``_analytic_pval`` :501-538, ``_g_moments`` :541-559).
The pair counting / permutation scoring run in ``libsqgr.so``; the host keeps only O(K*K*L) / O(P*G) post-processing."""

from __future__ import annotations

from typing import Any, Sequence

import numpy as np

from .. import _dist
from .._constants import Key
from .._lib import cooccur_counts, default_context
from .._utils import (
    _assert_categorical_obs,
    _assert_spatial_basis,
    _save_data,
    category_codes,
    deprecated_params,
    extract_adata_if_sdata,
)

__all__ = ["co_occurrence"]

fp = np.float32
ip = np.int32


def _find_min_max(spatial: np.ndarray) -> tuple[np.float32, np.float32]:
    """gr/_ppatterns.py:431-440 (same sklearn call as the reference; O(N) host work)."""
    from sklearn.metrics import pairwise_distances

    coord_sum = np.sum(spatial, axis=1)
    min_idx, min_idx2 = np.argpartition(coord_sum, 2)[:2]
    max_idx = np.argmax(coord_sum)
    thres_max = pairwise_distances(spatial[min_idx, :].reshape(1, -1), spatial[max_idx, :].reshape(1, -1))[0, 0] / 2.0
    thres_min = pairwise_distances(spatial[min_idx, :].reshape(1, -1), spatial[min_idx2, :].reshape(1, -1))[0, 0]
    return thres_min.astype(fp), thres_max.astype(fp)


def _occ_from_counts(counts: np.ndarray) -> np.ndarray:
    """gr/_ppatterns.py:343-358: ``occ[i, c, r] = (counts[c, i, r] / row_sums[c, r]) / (row_sums[i, r] / totals[r])``
    where both are non-zero, else 0 — same two float64 divisions per element as the reference loop."""
    counts = counts.astype(np.int64)
    row_sums = counts.sum(axis=0)  # [c, r]
    totals = row_sums.sum(axis=0)  # [r]
    with np.errstate(divide="ignore", invalid="ignore"):
        probs = row_sums / totals  # [i, r]
        cond = counts / row_sums[:, None, :]  # [c, i, r]
        occ_cir = cond / probs[None, :, :]
    ok = (probs[None, :, :] != 0.0) & (row_sums[:, None, :] != 0.0)
    occ_cir = np.where(ok, occ_cir, 0.0)
    return np.ascontiguousarray(np.transpose(occ_cir, (1, 0, 2)))  # -> [i, c, r]


@deprecated_params({"n_splits": "1.10.0", "n_jobs": "1.10.0", "backend": "1.10.0", "show_progress_bar": "1.10.0"})
def co_occurrence(
    adata: Any,
    cluster_key: str,
    spatial_key: str = Key.obsm.spatial,
    interval: int | Sequence[float] | np.ndarray = 50,
    copy: bool = False,
    *,
    table_key: str | None = None,
    fma: bool = False,
    device: int | None = None,
) -> tuple[np.ndarray, np.ndarray] | None:
    """Compute co-occurrence probability of clusters (drop-in for ``squidpy.gr.co_occurrence``).

    Same parameters, validation, deprecated-keyword behaviour (``FutureWarning``) and ``adata.uns`` slot
    (``'{cluster_key}_co_occurrence'`` -> ``{"occ", "interval"}``) as the reference.  The O(N^2 L) pair scan runs on
    the GPU with exact integer counts.

    Extra keyword-only parameters: ``fma`` — evaluate ``d2`` as ``fma(dx, dx, dy*dy)`` instead of separately rounded
    products (only matters for pairs lying exactly on a threshold; default matches numpy semantics); ``device``.
    With a ``torch.distributed`` process group the row tiles are split across ranks and the int64 counts all-reduced.

    Note: the number of clusters is the number of *categories* of ``adata.obs[cluster_key]``; the reference takes
    ``len(np.unique(labels))`` and indexes out of bounds when a category is empty (gr/_ppatterns.py:337-338).
    """
    adata = extract_adata_if_sdata(adata, table_key=table_key)
    _assert_categorical_obs(adata, key=cluster_key)
    _assert_spatial_basis(adata, key=spatial_key)

    spatial = np.asarray(adata.obsm[spatial_key]).astype(fp)
    labs, n_cls = category_codes(adata.obs[cluster_key])

    if isinstance(interval, (int, np.integer)):
        thresh_min, thresh_max = _find_min_max(spatial)
        interval = np.linspace(thresh_min, thresh_max, num=int(interval), dtype=fp)
    else:
        interval = np.array(sorted(interval), dtype=fp, copy=True)
    if len(interval) <= 1:
        raise ValueError(f"Expected interval to be of length `>= 2`, found `{len(interval)}`.")

    thresholds = (interval[1:]) ** 2  # float32, as in gr/_ppatterns.py:341
    ctx = default_context(device)
    rank, world = _dist.world()
    counts = cooccur_counts(
        ctx, spatial[:, 0], spatial[:, 1], labs.astype(ip), n_cls, thresholds, fma=fma, shard_index=rank, shard_count=world
    )
    (counts,) = _dist.allreduce_sum_([counts])
    out = _occ_from_counts(counts)

    if copy:
        return out, interval
    _save_data(adata, attr="uns", key=Key.uns.co_occurrence(cluster_key), data={"occ": out, "interval": interval})
    return None
