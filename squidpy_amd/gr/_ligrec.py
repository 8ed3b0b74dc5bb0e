"""``ligrec`` / ``PermutationTest`` with the reference's signatures on the MI355X path.

Reference: /root/reference/src/squidpy/gr/_ligrec.py — ``PermutationTestABC`` :91-474 (``prepare`` :135-228,
``test`` :234-373, complexes :398-460), ``PermutationTest.prepare`` :490-542, ``ligrec`` :548-612, the numba kernel
``_score_permutations`` :616-673 and its driver ``_analysis`` :677-775.

Division of labour: the host (pandas) validates and normalises the interaction table exactly like the reference and
forms the observed statistics; every permutation — label shuffle, per-cluster means, comparison against the observed
statistic — runs in ``libsqgr.so`` (``sqgr_ligrec_counts``).  The expression matrix stays sparse (CSC) instead of the
reference's dense float64 copy: a zero adds nothing to a cluster sum and the device walks the stored entries in cell
order, so the sums are bit-identical to the reference's sequential loop."""

from __future__ import annotations

import warnings
from collections.abc import Iterable, Mapping, Sequence
from itertools import product
from types import MappingProxyType
from typing import Any

import numpy as np
import pandas as pd
from scipy import sparse

from .. import _dist
from .._constants import ComplexPolicy, CorrAxis, Key
from .._lib import default_context, ligrec_counts
from .._stats import multipletests_pvals
from .._utils import (
    _assert_categorical_obs,
    _save_data,
    assert_positive,
    deprecated_params,
    extract_adata_if_sdata,
    logg,
    pcg64_states,
    resolve_seed,
)

__all__ = ["ligrec", "PermutationTest"]

SOURCE = "source"
TARGET = "target"
_MEAN_BLOCK = 512  # genes per dense block when the observed means are formed


def _looks_like_anndata(obj: Any) -> bool:
    return all(hasattr(obj, a) for a in ("obs", "var_names", "obs_names", "X"))


def _interaction_table(interactions: Any) -> pd.DataFrame:
    """The accepted spellings of ``interactions`` -> a fresh DataFrame with `source` / `target` columns
    (gr/_ligrec.py:167-192: same branching, same errors)."""
    if isinstance(interactions, Mapping):
        interactions = pd.DataFrame(interactions)
    if isinstance(interactions, pd.DataFrame):
        for col in (SOURCE, TARGET):
            if col not in interactions.columns:
                raise KeyError(f"Column `{col!r}` is not in `interactions`.")
        return interactions.copy()
    if isinstance(interactions, Iterable):
        items = tuple(interactions)
        if not items:
            raise ValueError("No interactions were specified.")
        if isinstance(items[0], str):
            items = tuple(product(items, repeat=2))  # all ordered pairs of the given genes
        elif len(items) == 2:
            items = tuple(zip(*items))  # (sources, targets)
        if any(len(it) != 2 for it in items):
            raise ValueError("Not all interactions are of length `2`.")
        return pd.DataFrame(list(items), columns=[SOURCE, TARGET])
    raise TypeError(f"Expected either a `pandas.DataFrame`, `dict` or `iterable`, found `{type(interactions).__name__}`")


def _adjust_along(pvals: pd.DataFrame, corr_method: str, axis: CorrAxis) -> pd.DataFrame:
    """FDR correction of the tested (non-NaN -> as 1.0) p-values along ``axis`` (gr/_ligrec.py:52-85); columns stay
    sparse with NaN fill.  ``alpha`` only feeds statsmodels' reject mask, which is unused there."""
    vals = pvals.to_numpy(dtype=np.float64)
    work = vals if axis == CorrAxis.CLUSTERS else vals.T  # correct down the columns of `work`
    out = np.empty_like(work)
    for c in range(work.shape[1]):
        col = work[:, c]
        q = multipletests_pvals(np.nan_to_num(col, copy=True, nan=1.0), method=corr_method)
        q[np.isnan(col)] = np.nan
        out[:, c] = q
    if axis != CorrAxis.CLUSTERS:
        out = out.T
    return pd.DataFrame(
        {c: pd.arrays.SparseArray(out[:, i], fill_value=np.nan) for i, c in enumerate(pvals.columns)}, index=pvals.index
    )


class PermutationTest:
    """Receptor-ligand interaction testing; the expected workflow is ``PermutationTest(adata).prepare(...).test(...)``
    (gr/_ligrec.py:91-133, 477-542).

    Parameters
    ----------
    adata
        Annotated data object (:class:`anndata.AnnData` or the duck-typed stand-in of this package).
    use_raw
        Whether to access ``adata.raw``.
    """

    def __init__(self, adata: Any, use_raw: bool = True):
        if not _looks_like_anndata(adata):
            raise TypeError(f"Expected `adata` to be of type `anndata.AnnData`, found `{type(adata).__name__}`.")
        if not adata.n_obs:
            raise ValueError("No cells are in `adata.obs_names`.")
        if not len(adata.var_names):
            raise ValueError("No genes are in `adata.var_names`.")
        self._adata = adata
        src = adata
        if use_raw:
            if getattr(adata, "raw", None) is None:
                raise AttributeError("No `.raw` attribute found. Try specifying `use_raw=False`.")
            if adata.raw.n_obs != adata.n_obs:
                raise ValueError(f"Expected `{adata.n_obs}` cells in `.raw` object, found `{adata.raw.n_obs}`.")
            src = adata.raw
        x = sparse.csc_matrix(src.X, dtype=np.float64)
        if np.isnan(x.data).any():  # the reference's `.fillna(0.0)` (:127-129)
            x.data = np.nan_to_num(x.data, nan=0.0)
        self._x = x
        self._genes = pd.Index(src.var_names)
        self._interactions: pd.DataFrame | None = None
        self._gene_ids: dict[str, int] | None = None  # gene -> column of the trimmed matrix
        self._trimmed: sparse.csc_matrix | None = None

    # ------------------------------------------------------------------------------------------------ prepare
    def prepare(
        self,
        interactions: Any = None,
        complex_policy: str = ComplexPolicy.MIN.value,
        interactions_params: Mapping[str, Any] = MappingProxyType({}),
        transmitter_params: Mapping[str, Any] = MappingProxyType({"categories": "ligand"}),
        receiver_params: Mapping[str, Any] = MappingProxyType({"categories": "receptor"}),
        **_: Any,
    ) -> "PermutationTest":
        """Normalise ``interactions`` and keep those whose `source` and `target` are both in the data.

        ``interactions`` may be a DataFrame / dict with `source` and `target`, a sequence of gene names (all ordered
        pairs), a sequence of pairs or a pair of sequences.  ``None`` asks `omnipath` for the intercellular network
        exactly as the reference does (needs the `omnipath` package and network access).  Complexes are written
        ``'alpha_beta'``; ``complex_policy='min'`` keeps the member with the lowest mean expression, ``'all'``
        expands to every member combination."""
        policy = ComplexPolicy(complex_policy)
        if interactions is None:
            interactions = self._fetch_omnipath(interactions_params, transmitter_params, receiver_params)
        table = _interaction_table(interactions)
        if table.empty:
            raise ValueError("The interactions are empty")

        # genes and interaction partners are compared in upper case; duplicated genes keep their first column
        genes = pd.Index(self._genes.astype(str).str.upper())
        for col in (SOURCE, TARGET):
            table[col] = table[col].str.upper()
        table = table.dropna(subset=[SOURCE, TARGET], how="any")
        table = table.drop_duplicates(subset=[SOURCE, TARGET], keep="first")
        first = ~genes.duplicated()
        if not first.all():
            logg.warning("Removed `%s` duplicate gene(s)", int((~first).sum()))
        self._x = self._x[:, np.where(first)[0]]
        self._genes = genes[first]
        col_of = {g: i for i, g in enumerate(self._genes)}

        table = self._resolve_complexes(table, policy, col_of)
        present = table[SOURCE].isin(self._genes) & table[TARGET].isin(self._genes)
        table = table[present]
        if table.empty:
            raise ValueError("After filtering by genes, no interactions remain.")
        table = table.drop_duplicates(subset=[SOURCE, TARGET], keep="first")  # complexes may have collapsed onto one pair

        used = sorted(set(table[SOURCE]) | set(table[TARGET]))
        self._trimmed = self._x[:, [col_of[g] for g in used]].tocsc()
        self._gene_ids = {g: i for i, g in enumerate(used)}
        self._interactions = table
        return self

    def _resolve_complexes(self, table: pd.DataFrame, policy: ComplexPolicy, col_of: dict[str, int]) -> pd.DataFrame:
        """gr/_ligrec.py:398-460."""
        if policy == ComplexPolicy.ALL:
            other = [c for c in table.columns if c not in (SOURCE, TARGET)]
            rows, index = [], []
            for ix, rec in zip(table.index, table.to_dict("records")):
                for s, t in product(str(rec[SOURCE]).split("_"), str(rec[TARGET]).split("_")):
                    rows.append([rec[c] for c in other] + [s, t])
                    index.append(ix)
            return pd.DataFrame(rows, columns=[*other, SOURCE, TARGET], index=pd.Index(index, name=table.index.name))
        means_cache: dict[str, float] = {}

        def mean_expr(g: str) -> float:
            if g not in means_cache:
                means_cache[g] = float(self._x[:, col_of[g]].sum() / self._x.shape[0])
            return means_cache[g]

        def pick(member: Any) -> Any:
            if member is None or "_" not in member:
                return member
            found = [c for c in member.split("_") if c in col_of]
            if not found:
                return None
            return min(found, key=mean_expr)  # first minimum, as `argmin`

        table = table.copy()
        table[SOURCE] = table[SOURCE].apply(pick)
        table[TARGET] = table[TARGET].apply(pick)
        return table

    @staticmethod
    def _fetch_omnipath(interactions_params: Mapping[str, Any], transmitter_params: Mapping[str, Any], receiver_params: Mapping[str, Any]) -> pd.DataFrame:
        """gr/_ligrec.py:515-539."""
        from omnipath.interactions import import_intercell_network

        table = import_intercell_network(
            interactions_params=interactions_params, transmitter_params=transmitter_params, receiver_params=receiver_params
        )
        for col in (SOURCE, TARGET):
            if col in table.columns:
                table.pop(col)
        table = table.rename(columns={"genesymbol_intercell_source": SOURCE, "genesymbol_intercell_target": TARGET})
        for col in (SOURCE, TARGET):
            table[col] = table[col].str.replace("^COMPLEX:", "", regex=True)
        return table

    @property
    def interactions(self) -> pd.DataFrame | None:
        """The interactions."""
        return self._interactions

    def __repr__(self) -> str:
        n = len(self._interactions) if self._interactions is not None else None
        return f"<{self.__class__.__name__}[n_interaction={n}]>"

    __str__ = __repr__

    # ------------------------------------------------------------------------------------------------ test
    @deprecated_params({"numba_parallel": "1.10.0", "backend": "1.10.0"})
    def test(
        self,
        cluster_key: str,
        clusters: Any = None,
        n_perms: int = 1000,
        threshold: float = 0.01,
        seed: int | None = None,
        corr_method: str | None = None,
        corr_axis: str = CorrAxis.INTERACTIONS.value,
        alpha: float = 0.05,
        copy: bool = False,
        key_added: str | None = None,
        n_jobs: int | None = None,
        show_progress_bar: bool = True,
        *,
        rng: str = "numpy",
        device: int | None = None,
    ) -> Mapping[str, pd.DataFrame] | None:
        """Perform the permutation test (gr/_ligrec.py:234-373).  Returns / stores ``{'means', 'pvalues', 'metadata'}``.

        ``n_jobs`` and ``show_progress_bar`` are accepted (``n_jobs`` validated) and do not influence the GPU path.
        ``rng='numpy'`` (default since round 5) reproduces the reference's PCG64 streams (gr/_ligrec.py:616-673) bit for bit on
        the device, i.e. Squidpy's p-values for that ``seed``; ``rng='philox'`` shuffles with the device generator keyed by
        ``(seed, permutation)`` (throughput mode, another stream).

        Limits of the GPU path: at most ``65535`` clusters among the requested cluster pairs (16-bit labels; ``NotImplementedError``
        beyond; the reference has no limit; more than 256 run in cluster tiles of 255); a subset that resolves to a single cluster is computed on the host like the reference does."""
        assert_positive(n_perms, name="n_perms")
        _assert_categorical_obs(self._adata, key=cluster_key)
        if rng not in ("philox", "numpy"):
            raise ValueError(f"Invalid option `{rng}` for `rng`. Valid options are: `['philox', 'numpy']`.")
        if n_jobs is not None and n_jobs != -1 and (n_jobs < -1 or n_jobs == 0):
            raise ValueError(f"Number of threads must be `-1` or a positive integer, got `{n_jobs}`.")
        axis = CorrAxis(corr_axis) if corr_method is not None else None
        n_cat = len(self._adata.obs[cluster_key].cat.categories)
        if n_cat <= 1:
            raise ValueError(f"Expected at least `2` clusters, found `{n_cat}`.")
        if self._interactions is None or self._trimmed is None or self._gene_ids is None:
            raise RuntimeError("Run `.prepare()` first.")

        # clusters are compared as strings (the reference casts the column to string categories, :305)
        labels = pd.Series(self._adata.obs[cluster_key].astype("string").astype("category").values)
        if clusters is None:
            clusters = list(map(str, self._adata.obs[cluster_key].cat.categories))
        if all(isinstance(c, str) for c in clusters):
            clusters = list(product(clusters, repeat=2))
        known = set(labels.cat.categories)
        pairs = []
        for pair in clusters:
            if not isinstance(pair, Sequence):
                raise TypeError(f"Expected a `Sequence`, found `{type(pair).__name__}`.")
            if len(pair) != 2:
                raise ValueError(f"Expected a `tuple` of length `2`, found `{len(pair)}`.")
            for c in pair:
                if c not in known:
                    raise ValueError(f"Invalid cluster `{c!r}`.")
            pairs.append((pair[0], pair[1]))
        pairs = sorted(pairs)
        wanted = {c for pair in pairs for c in pair}

        keep = labels.isin(list(wanted)).to_numpy()
        sub = labels[keep].cat.remove_unused_categories()
        codes = sub.cat.codes.to_numpy().astype(np.int32)
        code_of = {c: i for i, c in enumerate(sub.cat.categories)}
        n_cls = len(code_of)
        cpairs = np.array([[code_of[a], code_of[b]] for a, b in pairs], dtype=np.int32)
        table = self._interactions[[SOURCE, TARGET]]
        inter = np.array([[self._gene_ids[s], self._gene_ids[t]] for s, t in table.itertuples(index=False)], dtype=np.int32)
        x = self._trimmed[np.where(keep)[0], :].tocsc() if not keep.all() else self._trimmed
        x.sort_indices()

        logg.info(
            "Running `%s` permutations on `%s` interactions and `%s` cluster combinations on the GPU", n_perms, len(inter), len(pairs)
        )
        means, pvalues = self._analysis(x, codes, n_cls, inter, cpairs, threshold, n_perms, seed, rng, device)

        index = pd.MultiIndex.from_frame(table, names=[SOURCE, TARGET])
        columns = pd.MultiIndex.from_tuples(pairs, names=["cluster_1", "cluster_2"])
        res: dict[str, pd.DataFrame] = {
            "means": pd.DataFrame({c: pd.arrays.SparseArray(means[:, i], fill_value=0) for i, c in enumerate(columns)}, index=index),
            "pvalues": pd.DataFrame(
                {c: pd.arrays.SparseArray(pvalues[:, i], fill_value=np.nan) for i, c in enumerate(columns)}, index=index
            ),
            "metadata": self._interactions[self._interactions.columns.difference([SOURCE, TARGET])],
        }
        res["metadata"].index = res["means"].index.copy()
        if axis is not None:
            logg.info("Performing FDR correction across the `%s` using method `%s` at level `%s`", axis.value, corr_method, alpha)
            res["pvalues"] = _adjust_along(res["pvalues"], corr_method, axis)
        if copy:
            return res
        _save_data(self._adata, attr="uns", key=Key.uns.ligrec(cluster_key, key_added), data=res)
        return None

    @staticmethod
    def _observed(x: sparse.csc_matrix, codes: np.ndarray, n_cls: int, threshold: float) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Observed cluster means, expression-fraction mask and reciprocal cluster sizes (gr/_ligrec.py:712-726).

        The means go through ``pandas.groupby(...).mean()`` on dense float64 columns, like the reference, so that they
        carry pandas' (compensated) summation bit for bit; dense blocks of `_MEAN_BLOCK` genes bound the memory."""
        n_cells, n_genes = x.shape
        sizes = np.bincount(codes, minlength=n_cls)
        mean_obs = np.empty((n_cls, n_genes), dtype=np.float64)
        for g0 in range(0, n_genes, _MEAN_BLOCK):
            block = pd.DataFrame(x[:, g0 : g0 + _MEAN_BLOCK].toarray())
            mean_obs[:, g0 : g0 + _MEAN_BLOCK] = block.groupby(codes, observed=True).mean().to_numpy()
        onehot = sparse.csr_matrix((np.ones(n_cells, dtype=np.int64), (codes, np.arange(n_cells))), shape=(n_cls, n_cells))
        expressed = (onehot @ (x > 0).astype(np.int64)).toarray()  # cells with value > 0 per (cluster, gene)
        mask = (expressed / sizes[:, None]) >= threshold
        inv_counts = 1.0 / np.maximum(sizes.astype(np.float64), 1)
        return mean_obs, mask, inv_counts

    def _analysis(
        self,
        x: sparse.csc_matrix,
        codes: np.ndarray,
        n_cls: int,
        inter: np.ndarray,
        cpairs: np.ndarray,
        threshold: float,
        n_perms: int,
        seed: int | None,
        rng: str,
        device: int | None,
    ) -> tuple[np.ndarray, np.ndarray]:
        """gr/_ligrec.py:677-775 with the permutation loop on the device."""
        from ._nhood import _broadcast_seed

        mean_obs, mask, inv_counts = self._observed(x, codes, n_cls, threshold)
        rec, lig = inter[:, 0], inter[:, 1]
        c1, c2 = cpairs[:, 0], cpairs[:, 1]
        m_rec = mean_obs[c1][:, rec].T  # (n_interactions, n_cluster_pairs)
        m_lig = mean_obs[c2][:, lig].T
        nonzero = (m_rec > 0) & (m_lig > 0)
        valid = nonzero & mask[c1][:, rec].T & mask[c2][:, lig].T
        means = np.where(nonzero, (m_rec + m_lig) / 2.0, 0.0)
        obs = m_rec + m_lig

        if n_cls > 65535:
            raise NotImplementedError(
                f"`{n_cls}` clusters: the label generators of the GPU path write 16-bit labels, at most `65535` clusters per call "
                "(there is no CPU fallback); restrict `clusters`."
            )
        if n_cls == 1:
            # A cluster subset that resolves to ONE cluster (e.g. clusters=[("A", "A")]): the reference has no check here and
            # computes it (gr/_ligrec.py:677-775).  Shuffling a constant label vector changes nothing, so every permutation
            # yields the same group mean — the kernel's sequential sum over the cells times 1/size (gr/_ligrec.py:647-655),
            # which need not equal pandas' `groupby().mean()` of the observed side bit for bit: `shuffled > observed` is
            # then true in all permutations or in none.  Formed here exactly like that, on the host (no shuffles needed).
            sums = np.zeros(x.shape[1], dtype=np.float64)
            for g in range(x.shape[1]):
                col = x.data[x.indptr[g] : x.indptr[g + 1]]
                if len(col):
                    sums[g] = np.cumsum(col.astype(np.float64))[-1]  # strictly sequential float64 accumulation, cell order
            grp = sums * inv_counts[0]
            exceeds = (grp[rec] + grp[lig])[:, None] > obs
            pvalues = np.where(exceeds, 1.0, 0.0)
            pvalues[~valid] = np.nan
            return means, pvalues
        ctx = default_context(device)
        rank, world = _dist.world()
        lo, hi = _dist.shard_range(n_perms, rank, world)
        if rng == "numpy":
            if seed is None:
                seed = _broadcast_seed(resolve_seed(None))
            counts = ligrec_counts(
                ctx, x, codes, n_cls, inv_counts, inter, cpairs, obs, valid, pcg_states=pcg64_states(seed, n_perms, lo, hi),
                perm_begin=lo, perm_end=hi,
            )
        else:
            key = _broadcast_seed(resolve_seed(seed))
            counts = ligrec_counts(ctx, x, codes, n_cls, inv_counts, inter, cpairs, obs, valid, seed=key, perm_begin=lo, perm_end=hi)
        (counts,) = _dist.allreduce_sum_([counts])
        pvalues = counts.astype(np.float64) / n_perms
        pvalues[~valid] = np.nan
        return means, pvalues


@deprecated_params({"numba_parallel": "1.10.0", "backend": "1.10.0"})
def ligrec(
    adata: Any,
    cluster_key: str,
    interactions: Any = None,
    complex_policy: str = ComplexPolicy.MIN.value,
    threshold: float = 0.01,
    corr_method: str | None = None,
    corr_axis: str = CorrAxis.CLUSTERS.value,
    use_raw: bool = True,
    copy: bool = False,
    key_added: str | None = None,
    gene_symbols: str | None = None,
    *,
    n_perms: int = 1000,
    seed: int | None = None,
    clusters: Any = None,
    alpha: float = 0.05,
    n_jobs: int | None = None,
    show_progress_bar: bool = True,
    interactions_params: Mapping[str, Any] = MappingProxyType({}),
    transmitter_params: Mapping[str, Any] = MappingProxyType({"categories": "ligand"}),
    receiver_params: Mapping[str, Any] = MappingProxyType({"categories": "receptor"}),
    table_key: str | None = None,
    rng: str = "numpy",
    device: int | None = None,
) -> Mapping[str, pd.DataFrame] | None:
    """Perform the permutation test as described in CellPhoneDB (drop-in for ``squidpy.gr.ligrec``, gr/_ligrec.py:548-612).

    Same parameters, defaults, validation errors and ``adata.uns['{cluster_key}_ligrec']`` slot as the reference;
    ``gene_symbols`` temporarily takes the gene names from ``adata.var[gene_symbols]`` (``adata.raw.var`` with
    ``use_raw``).  Extra keyword-only ``rng`` / ``device`` as in :func:`nhood_enrichment`."""
    adata = extract_adata_if_sdata(adata, table_key=table_key)
    with _gene_symbols(adata, key=gene_symbols, use_raw=use_raw):
        return (
            PermutationTest(adata, use_raw=use_raw)
            .prepare(
                interactions,
                complex_policy=complex_policy,
                interactions_params=interactions_params,
                transmitter_params=transmitter_params,
                receiver_params=receiver_params,
            )
            .test(
                cluster_key=cluster_key,
                clusters=clusters,
                n_perms=n_perms,
                threshold=threshold,
                seed=seed,
                corr_method=corr_method,
                corr_axis=corr_axis,
                alpha=alpha,
                copy=copy,
                key_added=key_added,
                n_jobs=n_jobs,
                show_progress_bar=show_progress_bar,
                rng=rng,
                device=device,
            )
        )


class _gene_symbols:
    """Context manager: gene names come from ``var[key]`` while it is active (gr/_utils.py:132-183)."""

    def __init__(self, adata: Any, *, key: str | None, use_raw: bool):
        self._key = key
        self._target = None
        if key is None:
            return
        target = adata
        if use_raw:
            if getattr(adata, "raw", None) is None:
                raise AttributeError("No `.raw` attribute found. Try specifying `use_raw=False`.")
            target = adata.raw
        if key not in target.var:
            raise KeyError(f"Unable to find gene symbols in `adata.{'raw.' if use_raw else ''}var[{key!r}]`.")
        self._target = target

    def __enter__(self) -> None:
        if self._target is not None:
            self._saved = self._target.var.index.copy()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self._target.var.index = pd.Index(self._target.var[self._key])

    def __exit__(self, *_exc: Any) -> None:
        if self._target is not None:
            self._target.var.index = self._saved
