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
import pandas as pd
from scipy import sparse

from .._constants import SpatialAutocorr
from .._lib import AutocorrPlan, DeviceMatrix, cached_graph, cooccur_counts, default_context
from .._stats import analytic_columns, multipletests_pvals, permutation_columns
from .._utils import (
    _assert_categorical_obs,
    _assert_connectivity_key,
    _assert_spatial_basis,
    _save_data,
    assert_key_in_adata,
    assert_positive,
    category_codes,
    deprecated_params,
    extract_adata_if_sdata,
    get_n_processes,
    pcg64_states,
    progress,
    resolve_seed,
    spawn_generators,
)

__all__ = ["spatial_autocorr", "co_occurrence"]

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
    shard: str = "tiles",
) -> tuple[np.ndarray, np.ndarray] | None:
    """Compute co-occurrence probability of clusters (drop-in for ``squidpy.gr.co_occurrence``).

    Same parameters, validation, deprecated-keyword behaviour (``FutureWarning``) and ``adata.uns`` slot
    (``'{cluster_key}_co_occurrence'`` -> ``{"occ", "interval"}``) as the reference.  The O(N^2 L) pair scan runs on
    the GPU with exact integer counts.

    Extra keyword-only parameters: ``fma`` — evaluate ``d2`` as ``fma(dx, dx, dy*dy)`` instead of separately rounded
    products (only matters for pairs lying exactly on a threshold; default matches numpy semantics); ``device``.
    With a process group the work is split across ranks and the int64 counts all-reduced: ``shard="tiles"`` (default) gives
    every rank a share of the row tiles of the N x N sweep (each pair evaluated once overall), ``shard="intervals"`` gives
    every rank all pairs and a contiguous batch of the radius intervals (the partition BASELINE's north star names; it
    repeats the distance work on every rank and is kept as the alternative).  The counts are identical either way.

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
    if shard not in ("tiles", "intervals"):
        raise ValueError(f"Invalid option `{shard}` for `shard`. Valid options are: `['tiles', 'intervals']`.")
    ctx = default_context(device)
    rank, world = _dist.world()
    if shard == "intervals" and world > 1:
        # a cumulative count at threshold r only needs the thresholds of the rank's own batch: pairs below its first
        # threshold all fall into its first bin
        lo, hi = _dist.shard_range(len(thresholds), rank, world)
        counts = np.zeros((n_cls, n_cls, len(thresholds)), dtype=np.int64)
        if hi > lo:
            counts[:, :, lo:hi] = cooccur_counts(ctx, spatial[:, 0], spatial[:, 1], labs.astype(ip), n_cls, thresholds[lo:hi], fma=fma)
    else:
        counts = cooccur_counts(
            ctx, spatial[:, 0], spatial[:, 1], labs.astype(ip), n_cls, thresholds, fma=fma, shard_index=rank, shard_count=world
        )
    (counts,) = _dist.allreduce_sum_([counts])
    out = _occ_from_counts(counts)

    if copy:
        return out, interval
    _save_data(adata, attr="uns", key=Key.uns.co_occurrence(cluster_key), data={"occ": out, "interval": interval})
    return None


class _ColumnSelection:
    """``base[:, cols].T`` not yet formed: the reference selects the features with ``adata[:, genes].X`` (gr/_ppatterns.py:156-166),
    an O(nnz) host copy; here ``base`` goes to the device as it is and ``cols`` select there (``sqgr_autocorr_create_colidx``)."""

    def __init__(self, base: Any, cols: np.ndarray):
        self.base, self.cols = base, np.ascontiguousarray(cols, dtype=np.int32)
        self.shape = (len(self.cols), base.shape[0])

    def on_host(self) -> Any:
        return self.base[:, self.cols].T

    def device_bytes(self) -> int:
        b = self.base
        if sparse.issparse(b):  # a CSR matrix gets a by-column twin on the device
            return (b.data.nbytes + b.indices.nbytes + b.indptr.nbytes) * (2 if sparse.isspmatrix_csr(b) else 1)
        return int(b.shape[0]) * int(b.shape[1]) * max(b.itemsize, 4)

    @staticmethod
    def worthwhile(base: Any, cols: np.ndarray) -> bool:
        if sparse.issparse(base):
            return sparse.isspmatrix_csr(base) or sparse.isspmatrix_csc(base)
        # dense rows: the whole matrix is uploaded, which only pays when a good part of it is asked for
        return isinstance(base, np.ndarray) and base.ndim == 2 and base.dtype in (np.float32, np.float64) and 4 * len(cols) >= base.shape[1]


def _column_selection(var_names: Any, genes: Any, matrix: Any) -> _ColumnSelection | None:
    try:
        cols = var_names.get_indexer(np.asarray(genes))
    except Exception:  # non-unique var_names and the like: AnnData's own indexing decides what that means
        return None
    if len(cols) == 0 or (cols < 0).any() or not _ColumnSelection.worthwhile(matrix, cols):
        return None
    return _ColumnSelection(matrix, cols)


def _extract_vals(adata: Any, attr: str, genes: Any, layer: str | None, use_raw: bool) -> tuple[Any, Any]:
    """gr/_ppatterns.py:154-194: ``vals`` as (n_features, N) — or a `_ColumnSelection` standing for it — plus the feature index."""

    def extract_X(genes: Any) -> tuple[Any, Any]:
        if genes is None:
            if "highly_variable" in adata.var:
                # = adata[:, adata.var["highly_variable"]].var_names.values, without forming the subset object
                genes = adata.var_names[np.asarray(adata.var["highly_variable"], dtype=bool)].values
            else:
                genes = adata.var_names.values
        elif isinstance(genes, str):
            genes = [genes]
        if not use_raw:
            if len(genes) == adata.shape[1] and np.array_equal(np.asarray(genes), np.asarray(adata.var_names)):
                subset = adata  # every feature, in order: no subsetting copy of a (possibly very large, sparse) matrix
            else:
                lazy = _column_selection(adata.var_names, genes, adata.X if layer is None else adata.layers[layer])
                if lazy is not None:
                    return lazy, genes
                subset = adata[:, genes]
            return (subset.X if layer is None else subset.layers[layer]).T, genes
        if getattr(adata, "raw", None) is None:
            raise AttributeError("No `.raw` attribute found. Try specifying `use_raw=False`.")
        genes = list(set(genes) & set(adata.raw.var_names))
        lazy = _column_selection(adata.raw.var_names, genes, adata.raw.X)
        if lazy is not None:
            return lazy, genes
        return adata.raw[:, genes].X.T, genes

    def extract_obs(cols: Any) -> tuple[Any, Any]:
        if cols is None:
            df = adata.obs.select_dtypes(include=np.number)
            return df.T.to_numpy(), df.columns
        if isinstance(cols, str):
            cols = [cols]
        return adata.obs[cols].T.to_numpy(), cols

    def extract_obsm(ixs: Any) -> tuple[Any, Any]:
        assert_key_in_adata(adata, layer, attr="obsm")
        if ixs is None:
            ixs = list(np.arange(adata.obsm[layer].shape[1]))
        ixs = list(np.ravel([ixs]))
        return adata.obsm[layer][:, ixs].T, ixs

    if attr == "X":
        return extract_X(genes)
    if attr == "obs":
        return extract_obs(genes)
    if attr == "obsm":
        return extract_obsm(genes)
    raise NotImplementedError(f"Extracting from `adata.{attr}` is not yet implemented.")


def spatial_autocorr(
    adata: Any,
    connectivity_key: str = Key.obsp.spatial_conn(),
    genes: str | int | Sequence[str] | Sequence[int] | None = None,
    mode: str = "moran",
    transformation: bool = True,
    n_perms: int | None = None,
    two_tailed: bool = False,
    corr_method: str | None = "fdr_bh",
    attr: str = "X",
    layer: str | None = None,
    seed: int | None = None,
    use_raw: bool = False,
    copy: bool = False,
    n_jobs: int | None = None,
    backend: str = "loky",
    show_progress_bar: bool = True,
    *,
    table_key: str | None = None,
    rng: str = "numpy",
    device: int | None = None,
    gene_block: int = 2048,
) -> pd.DataFrame | None:
    """Calculate Global Autocorrelation Statistic — Moran's I or Geary's C (drop-in for
    ``squidpy.gr.spatial_autocorr``, gr/_ppatterns.py:56-255).

    Same parameters, value extraction (``attr`` in X/obs/obsm, ``layer``, ``use_raw``, HVG default), statistics
    (``I``/``C``, ``pval_norm``, ``var_norm``, and with ``n_perms``: ``pval_z_sim``, ``pval_sim``, ``var_sim``),
    ``{pval}_{corr_method}`` columns, sort order and ``adata.uns['moranI'|'gearyC']`` slot as the reference.

    Extra keyword-only parameters: ``rng`` — ``"numpy"`` (default since round 5) reproduces the reference's ``rng.permutation(N)``
    streams (gr/_ppatterns.py:269-272) bit for bit on the GPU: Squidpy's permutation columns for that ``seed``; ``"philox"`` draws
    the row permutations with the device's counter-based generator, keyed by ``(seed, permutation index)`` — the throughput mode,
    statistically equivalent, another stream; ``"numpy-host"`` draws numpy's streams on the host; ``device``; ``gene_block`` — features
    resident on the GPU at a time.  With a ``torch.distributed`` process group, feature blocks are split across ranks
    and the score columns gathered.
    """
    adata = extract_adata_if_sdata(adata, table_key=table_key)
    _assert_connectivity_key(adata, connectivity_key)
    if rng not in ("philox", "numpy", "numpy-host"):
        raise ValueError(f"Invalid option `{rng}` for `rng`. Valid options are: `['philox', 'numpy', 'numpy-host']`.")
    vals, index = _extract_vals(adata, attr, genes, layer, use_raw)

    mode = SpatialAutocorr(mode)
    if mode == SpatialAutocorr.MORAN:
        stat, expected, ascending = "I", -1.0 / (adata.shape[0] - 1), False
    else:
        stat, expected, ascending = "C", 1.0, True

    g = sparse.csr_matrix(adata.obsp[connectivity_key]).copy()
    if transformation:  # row-normalize (in the matrix dtype, like sklearn.preprocessing.normalize(copy=False))
        from sklearn.preprocessing import normalize

        normalize(g, norm="l1", axis=1, copy=False)

    get_n_processes(n_jobs)
    if n_perms is not None:
        assert_positive(n_perms, name="n_perms")
    n = g.shape[0]
    n_feat = vals.shape[0]
    perm_idx = None
    states = None
    key = resolve_seed(seed)
    if seed is None:  # fresh entropy: every rank must score its feature blocks against rank 0's permutation set
        from ._nhood import _broadcast_seed

        key = _broadcast_seed(key)
        seed = key
    ctx = default_context(device)
    if n_perms is not None and rng == "numpy-host":
        gens = spawn_generators(seed, n_perms)
        perm_idx = np.stack([gens[p].permutation(n) for p in range(n_perms)]).astype(np.int32)
    elif n_perms is not None and rng == "numpy":  # numpy's `rng.permutation(N)` streams, reproduced on the device per block
        states = pcg64_states(seed if seed is not None else key, n_perms)

    graph = cached_graph(ctx, g, with_data=True)  # stays resident for the next call on the same (normalised) matrix
    rank, world = _dist.world()
    blocks = [(b0, min(n_feat, b0 + gene_block)) for b0 in range(0, n_feat, max(int(gene_block), 1))]
    mine = [bi for bi in range(len(blocks)) if _block_owner(bi, len(blocks), world) == rank]  # a contiguous run of feature blocks
    score = np.full(n_feat, np.nan)
    # the (n_perms, n_feat) permutation scores stay on the device; per feature their exceedance count, sum, std and var come back
    red = None
    if n_perms is not None:
        red = {"n_ge": np.zeros(n_feat, dtype=np.int64), **{k: np.full(n_feat, np.nan) for k in ("sum", "std", "var")}}
    # The expression matrix goes to the device ONCE, as it lies in memory — `adata.X` is usually scipy CSR float32, sometimes a
    # dense row-major array — and the feature blocks are cut out of it (densified, widened to float64) there; every rank
    # uploads the columns of its own blocks only.  Nothing is sliced, densified or transposed on the host per block.
    resident, shift = (None, 0)
    cols = None
    if isinstance(vals, _ColumnSelection):  # a gene subset (the HVG default, an explicit list): selected on the device
        if mine and vals.device_bytes() <= ctx.device_info()["hbm_bytes"] // 4:
            resident, cols = DeviceMatrix(ctx, vals.base), vals.cols
        elif mine:
            vals = vals.on_host()
    if mine and cols is None:
        resident, shift = _resident_features(ctx, vals, blocks[mine[0]][0], blocks[mine[-1]][1], block=max(int(gene_block), 1))
    bar = progress(sum(blocks[bi][1] - blocks[bi][0] for bi in mine), "feature", show_progress_bar and n_perms is not None)
    try:
        for bi in mine:
            b0, b1 = blocks[bi]
            if cols is not None:
                plan = AutocorrPlan.from_column_list(ctx, graph, resident, cols[b0:b1])
            elif resident is not None:
                plan = AutocorrPlan.from_columns(ctx, graph, resident, b0 - shift, b1 - b0)
            else:  # a gene-major host array (contiguous feature rows), or a matrix too large to keep resident
                blk = vals[b0:b1]
                blk = np.asarray(blk.toarray() if sparse.issparse(blk) else blk, dtype=np.float64)
                plan = AutocorrPlan(ctx, graph, blk)
            try:
                score[b0:b1] = plan.scores(mode.s)
                if n_perms is not None:
                    only = n_feat == 1  # a (n_perms, 1) score array: numpy reduces it in its contiguous order
                    if states is not None:
                        part = plan.perm_stats(mode.s, score[b0:b1], pcg_states=states, only_feature=only)
                    elif perm_idx is not None:
                        part = plan.perm_stats(mode.s, score[b0:b1], perm_idx=perm_idx, only_feature=only)
                    else:
                        part = plan.perm_stats(mode.s, score[b0:b1], seed=key, perm_begin=0, perm_end=n_perms, only_feature=only)
                    for k, v in part.items():
                        red[k][b0:b1] = v
                bar.update(b1 - b0)
            finally:
                plan.close()
    finally:
        bar.close()
        if resident is not None:
            resident.close()
    if world > 1:
        score = _merge_blocks(score, blocks, world, axis=0)
        if red is not None:
            red = {k: _merge_blocks(v, blocks, world, axis=0) for k, v in red.items()}
    if np.isnan(score).any():
        import warnings

        warnings.warn("Some features are constant or contain NaN: their statistic is NaN.", UserWarning, stacklevel=2)

    with np.errstate(divide="ignore", invalid="ignore"):
        pval_results = analytic_columns(score, g, mode.s, expected, two_tailed)
        if red is not None:
            pval_results.update(permutation_columns(score, n_perms, red["n_ge"], red["sum"], red["std"], red["var"]))

    df = pd.DataFrame({stat: score, **pval_results}, index=index)
    if corr_method is not None:
        for pv in [c for c in df.columns if "pval" in c]:
            df[f"{pv}_{corr_method}"] = multipletests_pvals(df[pv].values, method=corr_method)
    df.sort_values(by=stat, ascending=ascending, inplace=True)

    if copy:
        return df
    _save_data(adata, attr="uns", key=mode.s + stat, data=df)
    return None


def _block_owner(bi: int, n_blocks: int, world: int) -> int:
    """Feature blocks go to the ranks in contiguous runs (so that a rank's features are one column range of the matrix)."""
    return bi * world // max(n_blocks, 1)


def _resident_features(ctx: Any, vals: Any, c0: int, c1: int, block: int | None = None) -> tuple[DeviceMatrix | None, int]:
    """Columns ``[c0, c1)`` of the (cells x features) matrix behind ``vals`` (features x cells, gr/_ppatterns.py:154-185)
    uploaded as they lie in memory -> (device matrix, index of its first column), or ``(None, 0)`` when the feature rows are
    contiguous on the host anyway (gene-major array: plain block uploads) or the matrix would not fit a quarter of the HBM."""
    budget = ctx.device_info()["hbm_bytes"] // 4
    if sparse.issparse(vals):
        base = vals.T  # CSR <-> CSC view of the same arrays, (cells x features)
        if not (sparse.isspmatrix_csr(base) or sparse.isspmatrix_csc(base)):
            base = sparse.csr_matrix(base)
        shift = 0
        if sparse.isspmatrix_csc(base) and (c0 > 0 or c1 < base.shape[1]):
            base, shift = base[:, c0:c1], c0  # a slice of the column pointer array; CSR keeps all columns (no cheap cut)
        if base.data.nbytes + base.indices.nbytes + base.indptr.nbytes > budget:
            return None, 0
        return DeviceMatrix(ctx, base), shift
    if isinstance(vals, np.ndarray) and vals.ndim == 2:
        base = vals.T
        if base.strides[1] == base.itemsize and base.shape[1] > 0 and base.strides[0] >= base.shape[1] * base.itemsize:
            view = base[:, c0:c1]  # row-major (cells x features): a column range through the row pitch
            if view.shape[0] * view.shape[1] * max(view.itemsize, 4) <= budget:
                # a large dense matrix arrives feature block by feature block while the first blocks are scored (round 6)
                return DeviceMatrix(ctx, view, stream_columns=block), c0
    return None, 0


def _merge_blocks(a: np.ndarray, blocks: list[tuple[int, int]], world: int, axis: int) -> np.ndarray:
    """Every rank filled only its own feature blocks (NaN elsewhere): gather and take each block from its owner."""
    parts = _dist.allgather_object(a)
    out = a.copy()
    for bi, (b0, b1) in enumerate(blocks):
        src = parts[_block_owner(bi, len(blocks), world)]
        if axis == 0:
            out[b0:b1] = src[b0:b1]
        else:
            out[:, b0:b1] = src[:, b0:b1]
    return out
