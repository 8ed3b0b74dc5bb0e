"""TEST INFRASTRUCTURE — not product code.  CPU oracle for the ``sq.gr`` hot path.

Fast numpy/scipy restatements of the reference algorithms (scverse/squidpy @ /root/reference,
paths below relative to ``src/squidpy``).  Every function cites the reference lines it follows.
The restatements are pinned in ``tests/test_oracle_pinned.py`` against

* the reference's own kernel source executed under a numba stub (``oracle/ref_shim.py``,
  build container only) and the golden vectors generated from it (``tests/golden/``);
* the known-answer vectors the reference's tests hold for this path
  (``tests/graph/test_nhood.py:153-173`` interaction matrices; the closed-form ``var_norm``
  of ``tests/graph/test_ppatterns.py:108-137``).

PARITY UNPINNED for the Moran's I / Geary's C statistic itself: the arithmetic lives in
``scanpy.metrics.morans_i / gearys_c`` (third-party, ``scanpy>=1.9.3`` in
``pyproject.toml:68``, not vendored, no lock file, absent from this image) and no reference
test pins a value of I or C.  ``morans_i`` / ``gearys_c`` below restate scanpy's published
algorithm (scanpy/metrics/_morans_i.py, _gearys_c.py, v1.9.3 – v1.11) and are checked against
the dense textbook formulae; the call sites that anchor them are ``gr/_ppatterns.py:14,200,
205,216,267,272``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product (``squidpy_amd``) never does.
"""

from __future__ import annotations

from typing import Any

import numpy as np
import pandas as pd
from scipy import sparse, stats

from . import devrng

# =========================================================================== RNG streams


def spawn_generators(seed: int | None, n: int) -> list[np.random.Generator]:
    """_utils.py:240-241."""
    return [np.random.default_rng(s) for s in np.random.SeedSequence(seed).spawn(n)]


def shuffle_group(labels: np.ndarray, lib_codes: np.ndarray, n_libs: int, rs: np.random.Generator) -> np.ndarray:
    """gr/_utils.py:185-213 with the categorical replaced by integer codes (category order)."""
    out = np.empty(lib_codes.shape, dtype=labels.dtype)
    for c in range(n_libs):
        idx = np.where(lib_codes == c)[0]
        grp = labels[idx].copy()
        rs.shuffle(grp)
        out[idx] = grp
    return out


# =========================================================================== nhood_enrichment


def nhood_counts(indices: np.ndarray, indptr: np.ndarray, labels: np.ndarray, n_cls: int) -> np.ndarray:
    """count[a, b] = sum_{i: lab_i = a} #{j in N(i): lab_j = b}; uint32 (K, K).

    gr/_nhood.py:79-88 (per-row neighbour-label histogram ``res[N, K]``) followed by the
    generated if/elif accumulation ``g{cl} += res[row]`` (gr/_nhood.py:113-135).  Edge weights are
    never read; stored zeros and self loops count (gr/_nhood.py:205).
    """
    if n_cls <= 1:
        raise ValueError(f"Expected at least `2` clusters, found `{n_cls}`.")  # gr/_nhood.py:107-108
    labels = np.asarray(labels).astype(np.int64)
    indptr = np.asarray(indptr).astype(np.int64)
    deg = np.diff(indptr)
    row_lab = np.repeat(labels, deg)
    col_lab = labels[np.asarray(indices).astype(np.int64)]
    flat = np.bincount(row_lab * n_cls + col_lab, minlength=n_cls * n_cls)
    return flat.reshape(n_cls, n_cls).astype(np.uint32)


def nhood_perm_counts_numpy(
    indices: np.ndarray,
    indptr: np.ndarray,
    labels: np.ndarray,
    n_cls: int,
    seed: int | None,
    n_perms: int,
    lib_codes: np.ndarray | None = None,
    n_libs: int = 0,
    perm_range: tuple[int, int] | None = None,
) -> np.ndarray:
    """gr/_nhood.py:516-547 with the reference's numpy streams: float64 (P, K, K)."""
    gens = spawn_generators(seed, n_perms)
    lo, hi = perm_range if perm_range is not None else (0, n_perms)
    out = np.empty((hi - lo, n_cls, n_cls), dtype=np.float64)
    base = np.asarray(labels).astype(np.uint32)
    for k, ix in enumerate(range(lo, hi)):
        rng = gens[ix]
        if lib_codes is not None:
            shuffled = shuffle_group(base, lib_codes, n_libs, rng)
        else:
            shuffled = base.copy()
            rng.shuffle(shuffled)
        out[k] = nhood_counts(indices, indptr, shuffled, n_cls)
    return out


def nhood_perm_labels_numpy(labels: np.ndarray, seed: int | None, n_perms: int) -> np.ndarray:
    """The shuffled label vectors the reference evaluates (gr/_nhood.py:533-538): (P, N)."""
    gens = spawn_generators(seed, n_perms)
    base = np.asarray(labels).astype(np.uint32)
    out = np.empty((n_perms, len(base)), dtype=np.uint32)
    for ix in range(n_perms):
        s = base.copy()
        gens[ix].shuffle(s)
        out[ix] = s
    return out


def nhood_perm_counts_philox(
    indices: np.ndarray,
    indptr: np.ndarray,
    labels: np.ndarray,
    n_cls: int,
    seed: int,
    perm_begin: int,
    perm_end: int,
    lib_codes: np.ndarray | None = None,
    n_libs: int = 0,
) -> np.ndarray:
    """Same loop as gr/_nhood.py:516-547 but with the device generator (oracle/devrng.py)."""
    out = np.empty((perm_end - perm_begin, n_cls, n_cls), dtype=np.float64)
    for k, p in enumerate(range(perm_begin, perm_end)):
        s = devrng.shuffled_labels(np.asarray(labels), seed, p, lib_codes, n_libs)
        out[k] = nhood_counts(indices, indptr, s, n_cls)
    return out


def nhood_zscore(count: np.ndarray, perms: np.ndarray) -> np.ndarray:
    """gr/_nhood.py:231 — population std, no zero guard."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return (count - perms.mean(axis=0)) / perms.std(axis=0)


def interaction_matrix(
    data: np.ndarray, indices: np.ndarray, indptr: np.ndarray, cats: np.ndarray, n_cats: int, weights: bool
) -> np.ndarray:
    """gr/_nhood.py:396-429 (after NaN masking): output[cat_i, cat_j] += w_ij (or 1)."""
    is_int = np.issubdtype(data.dtype, np.integer) or data.dtype == bool
    out = np.zeros((n_cats, n_cats), dtype=int if is_int else float)
    g_data = data if weights else np.broadcast_to(1, len(data))
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    np.add.at(out, (cats[rows], cats[indices]), g_data)
    return out


# =========================================================================== Moran / Geary


def _as_csr64(g: Any) -> sparse.csr_matrix:
    g = sparse.csr_matrix(g)
    return sparse.csr_matrix((g.data.astype(np.float64), g.indices, g.indptr), shape=g.shape)


def morans_i(g: Any, vals: Any) -> np.ndarray:
    """scanpy.metrics.morans_i restated (third-party; see module docstring).

    ``I_k = N / W * sum_i z_i (sum_j w_ij z_j) / sum_i z_i^2``, ``z = x - mean(x)``,
    ``W = g.data.sum()`` in float64; constant rows -> NaN.  ``vals``: (G, N) dense or sparse.
    """
    g64 = _as_csr64(g)
    X = np.asarray(vals.toarray() if sparse.issparse(vals) else vals, dtype=np.float64)
    if X.ndim == 1:
        X = X[None, :]
    n = X.shape[1]
    W = g64.data.sum()
    z = X - X.mean(axis=1, keepdims=True)
    z2ss = (z * z).sum(axis=1)
    y = (g64 @ z.T).T
    inum = (z * y).sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = n / W * inum / z2ss
    const = (X == X[:, [0]]).all(axis=1)
    out[const] = np.nan
    return out


def gearys_c(g: Any, vals: Any) -> np.ndarray:
    """scanpy.metrics.gearys_c restated.

    ``C_k = (N - 1) * sum_ij w_ij (x_i - x_j)^2 / (2 W sum_i (x_i - mean)^2)``; constant rows -> NaN.
    """
    g64 = _as_csr64(g)
    X = np.asarray(vals.toarray() if sparse.issparse(vals) else vals, dtype=np.float64)
    if X.ndim == 1:
        X = X[None, :]
    n = X.shape[1]
    W = g64.data.sum()
    rows = np.repeat(np.arange(n), np.diff(g64.indptr))
    out = np.empty(X.shape[0])
    for k in range(X.shape[0]):
        x = X[k]
        total = (g64.data * (x[rows] - x[g64.indices]) ** 2).sum()
        denom = 2 * W * ((x - x.mean()) ** 2).sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            out[k] = (n - 1) * total / denom
    const = (X == X[:, [0]]).all(axis=1)
    out[const] = np.nan
    return out


def autocorr_perm_indices(n: int, seed: int | None, n_perms: int) -> np.ndarray:
    """The row permutations the reference draws (gr/_ppatterns.py:269-271): (P, N) int64."""
    gens = spawn_generators(seed, n_perms)
    return np.stack([gens[p].permutation(n) for p in range(n_perms)])


def score_perms(mode: str, g: Any, vals: Any, perm_idx: np.ndarray) -> np.ndarray:
    """gr/_ppatterns.py:258-280: ``func(g[idx, :], vals)`` per permutation, float64 (P, G)."""
    func = morans_i if mode == "moran" else gearys_c
    g = sparse.csr_matrix(g)
    out = np.empty((len(perm_idx), vals.shape[0]))
    for i, idx in enumerate(perm_idx):
        out[i] = func(g[idx, :], vals)
    return out


def g_moments(w: Any) -> tuple[float, float, float]:
    """gr/_ppatterns.py:541-559."""
    s0 = w.sum()
    t = w.transpose() + w
    t2 = t.multiply(t) if sparse.issparse(t) else t * t
    s1 = t2.sum() / 2.0
    s2array = np.array(w.sum(1) + w.sum(0).transpose()) ** 2
    s2 = s2array.sum()
    return s0, s1, s2


def analytic_pval(score: np.ndarray, g: Any, mode: str, expected: float, two_tailed: bool) -> tuple[np.ndarray, float]:
    """gr/_ppatterns.py:501-538."""
    s0, s1, s2 = g_moments(g)
    n = g.shape[0]
    s02 = s0 * s0
    if mode == "geary":
        v = ((2 * s1 + s2) * (n - 1) - 4 * s02) / (2 * (n + 1) * s02)
    elif mode == "moran":
        n2 = n * n
        v = (n2 * s1 - n * s2 + 3 * s02) / ((n - 1) * (n + 1) * s02) - (1.0 / (n - 1)) ** 2
    else:
        raise AssertionError(f"Unexpected mode `{mode}`.")
    se = v ** (1 / 2.0)
    z = (score - expected) / se
    p = np.empty(score.shape)
    p[z > 0] = 1 - stats.norm.cdf(z[z > 0])
    p[z <= 0] = stats.norm.cdf(z[z <= 0])
    if two_tailed:
        p *= 2.0
    return p, v


def p_value_calc(
    score: np.ndarray, sims: np.ndarray | None, g: Any, mode: str, expected: float, two_tailed: bool
) -> dict[str, Any]:
    """gr/_ppatterns.py:443-498."""
    p_norm, var_norm = analytic_pval(score, g, mode, expected, two_tailed)
    res: dict[str, Any] = {"pval_norm": p_norm, "var_norm": var_norm}
    if sims is None:
        return res
    n_perms = sims.shape[0]
    large = (sims >= score).sum(axis=0)
    m = (n_perms - large) < large
    large[m] = n_perms - large[m]
    p_sim = (large + 1) / (n_perms + 1)
    e = sims.sum(axis=0) / n_perms
    se = sims.std(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (score - e) / se
    p_z = np.empty(z.shape)
    p_z[z > 0] = 1 - stats.norm.cdf(z[z > 0])
    p_z[z <= 0] = stats.norm.cdf(z[z <= 0])
    res["pval_z_sim"] = p_z
    res["pval_sim"] = p_sim
    res["var_sim"] = np.var(sims, axis=0)
    return res


def fdr_bh(pvals: np.ndarray) -> np.ndarray:
    """statsmodels.stats.multitest.multipletests(method="fdr_bh")[1] restated
    (third-party, absent here; call site gr/_ppatterns.py:239-245)."""
    pvals = np.asarray(pvals, dtype=float)
    order = np.argsort(pvals)
    ps = pvals[order]
    n = len(ps)
    raw = ps / (np.arange(1, n + 1) / float(n))
    corr = np.minimum.accumulate(raw[::-1])[::-1]
    corr[corr > 1] = 1
    out = np.empty_like(corr)
    out[order] = corr
    return out


def spatial_autocorr(
    g: Any,
    vals: np.ndarray,
    index: Any,
    mode: str = "moran",
    transformation: bool = True,
    n_perms: int | None = None,
    two_tailed: bool = False,
    corr_method: str | None = "fdr_bh",
    seed: int | None = None,
    perm_idx: np.ndarray | None = None,
) -> pd.DataFrame:
    """gr/_ppatterns.py:196-255 on already-extracted ``vals`` (G, N)."""
    from sklearn.preprocessing import normalize

    n = g.shape[0]
    stat, expected, ascending = ("I", -1.0 / (n - 1), False) if mode == "moran" else ("C", 1.0, True)
    g = sparse.csr_matrix(g).copy()
    if transformation:
        normalize(g, norm="l1", axis=1, copy=False)
    func = morans_i if mode == "moran" else gearys_c
    score = func(g, vals)
    sims = None
    if n_perms is not None:
        if perm_idx is None:
            perm_idx = autocorr_perm_indices(n, seed, n_perms)
        sims = score_perms(mode, g, vals, perm_idx)
    with np.errstate(divide="ignore", invalid="ignore"):
        pv = p_value_calc(score, sims, g, mode, expected, two_tailed)
    df = pd.DataFrame({stat: score, **pv}, index=index)
    if corr_method is not None:
        for c in [c for c in df.columns if "pval" in c]:
            df[f"{c}_{corr_method}"] = fdr_bh(df[c].values)
    df.sort_values(by=stat, ascending=ascending, inplace=True)
    return df


# =========================================================================== co-occurrence


def find_min_max(spatial: np.ndarray) -> tuple[np.float32, np.float32]:
    """gr/_ppatterns.py:431-440 (sklearn.pairwise_distances on the float32 coordinates)."""
    from sklearn.metrics import pairwise_distances

    coord_sum = np.sum(spatial, axis=1)
    min_idx, min_idx2 = np.argpartition(coord_sum, 2)[:2]
    max_idx = np.argmax(coord_sum)
    tmax = pairwise_distances(spatial[min_idx, :].reshape(1, -1), spatial[max_idx, :].reshape(1, -1))[0, 0] / 2.0
    tmin = pairwise_distances(spatial[min_idx, :].reshape(1, -1), spatial[min_idx2, :].reshape(1, -1))[0, 0]
    return tmin.astype(np.float32), tmax.astype(np.float32)


def occur_count(
    x: np.ndarray, y: np.ndarray, thresholds: np.ndarray, labs: np.ndarray, k: int, chunk: int = 512
) -> np.ndarray:
    """gr/_ppatterns.py:283-310: counts[a, b, r] = #{i != j: lab_i=a, lab_j=b, d2_ij <= thr_r}; int64.

    float32 arithmetic exactly as the literal source executes it: ``dx*dx + dy*dy`` with each
    product and the sum rounded to float32 (no fused multiply-add).
    """
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    thr = np.asarray(thresholds, dtype=np.float32)
    labs = np.asarray(labs).astype(np.int64)
    n, L = len(x), len(thr)
    out = np.zeros((k * k, L), dtype=np.int64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        dx = x[s:e, None] - x[None, :]
        dy = y[s:e, None] - y[None, :]
        d2 = dx * dx + dy * dy
        pair = labs[s:e, None] * k + labs[None, :]
        notself = np.arange(s, e)[:, None] != np.arange(n)[None, :]
        pf = pair[notself]
        d2f = d2[notself]
        for r in range(L):
            out[:, r] += np.bincount(pf[d2f <= thr[r]], minlength=k * k)
    return out.reshape(k, k, L)


def co_occurrence_probs(counts: np.ndarray) -> np.ndarray:
    """gr/_ppatterns.py:343-358: counts (K, K, L) -> occ (K, K, L) float64."""
    k, _, L = counts.shape
    occ = np.zeros((k, k, L), dtype=np.float64)
    row_sums = counts.sum(axis=0)
    totals = row_sums.sum(axis=0)
    for r in range(L):
        with np.errstate(divide="ignore", invalid="ignore"):
            probs = row_sums[:, r] / totals[r]
        for c in range(k):
            for i in range(k):
                if probs[i] != 0.0 and row_sums[c, r] != 0.0:
                    occ[i, c, r] = (counts[c, i, r] / row_sums[c, r]) / probs[i]
    return occ


def co_occurrence(spatial: np.ndarray, labs: np.ndarray, interval: Any = 50) -> tuple[np.ndarray, np.ndarray]:
    """gr/_ppatterns.py:401-419 + 313-358 on extracted arrays.  Returns (occ, interval)."""
    spatial = np.asarray(spatial).astype(np.float32)
    labs = np.asarray(labs).astype(np.int32)
    if isinstance(interval, int):
        tmin, tmax = find_min_max(spatial)
        interval = np.linspace(tmin, tmax, num=interval, dtype=np.float32)
    else:
        interval = np.array(sorted(interval), dtype=np.float32, copy=True)
    if len(interval) <= 1:
        raise ValueError(f"Expected interval to be of length `>= 2`, found `{len(interval)}`.")
    k = len(np.unique(labs))
    thr = (interval[1:]) ** 2
    counts = occur_count(spatial[:, 0], spatial[:, 1], thr, labs, k)
    return co_occurrence_probs(counts), interval


# =========================================================================== Ripley


def pair_counts_bruteforce(points: np.ndarray, support: np.ndarray, chunk: int = 1024) -> np.ndarray:
    """#ordered pairs i != j with ``sqrt(dx^2 + dy^2) <= r`` in float64 — what
    ``KDTree.two_point_correlation(points, support) - m`` returns (gr/_ripley.py:220-222)."""
    p = np.asarray(points, dtype=np.float64)
    out = np.zeros(len(support), dtype=np.int64)
    for s in range(0, len(p), chunk):
        d = p[s : s + chunk, None, :] - p[None, :, :]
        dist = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
        for r, rr in enumerate(support):
            out[r] += (dist <= rr).sum()
    return out - len(p)


def l_function(points: np.ndarray, support: np.ndarray, n: int, area: float) -> tuple[np.ndarray, np.ndarray]:
    """gr/_ripley.py:212-227 with sklearn's KDTree exactly as the reference calls it."""
    from sklearn.neighbors import KDTree

    tree = KDTree(points, metric="euclidean")
    npairs = tree.two_point_correlation(points, support, dualtree=True) - points.shape[0]
    k_est = (npairs / n) / (n / area)
    return support, np.sqrt(k_est / np.pi)


def f_g_function(distances: np.ndarray, support: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """gr/_ripley.py:206-209."""
    counts, bins = np.histogram(distances, bins=support)
    with np.errstate(divide="ignore", invalid="ignore"):
        fracs = np.cumsum(counts) / counts.sum()
    return bins, np.concatenate((np.zeros((1,), dtype=float), fracs))


def ppp(hull: Any, n_simulations: int, n_observations: int, rng: np.random.Generator) -> np.ndarray:
    """gr/_ripley.py:230-271."""
    from scipy.spatial import Delaunay

    vxs = hull.points[hull.vertices]
    deln = Delaunay(vxs)
    bbox = np.array([*vxs.min(0), *vxs.max(0)])
    result = np.empty((n_simulations, n_observations, 2))
    for i_sim in range(n_simulations):
        i_obs = 0
        while i_obs < n_observations:
            x, y = rng.uniform(bbox[0], bbox[2]), rng.uniform(bbox[1], bbox[3])
            if deln.find_simplex((x, y)) >= 0:
                result[i_sim, i_obs] = (x, y)
                i_obs += 1
    return result.squeeze()


def ripley(
    coordinates: np.ndarray,
    clusters: np.ndarray,
    mode: str = "F",
    metric: str = "euclidean",
    n_neigh: int = 2,
    n_simulations: int = 100,
    n_observations: int = 1000,
    max_dist: float | None = None,
    n_steps: int = 50,
    seed: int | None = None,
) -> dict[str, Any]:
    """gr/_ripley.py:112-185 on extracted arrays; returns obs (K, n_steps), sims, bins, pvalues."""
    from scipy.spatial import ConvexHull
    from sklearn.neighbors import NearestNeighbors
    from sklearn.preprocessing import LabelEncoder

    N = coordinates.shape[0]
    hull = ConvexHull(coordinates)
    area = hull.volume
    if max_dist is None:
        max_dist = (area / 2) ** 0.5
    support = np.linspace(0, max_dist, n_steps)
    le = LabelEncoder().fit(clusters)
    cluster_idx = le.transform(clusters)
    obs_arr = np.empty((le.classes_.shape[0], n_steps))
    obs_rng, *sim_rngs = spawn_generators(seed, n_simulations + 1)
    random = None
    for i in np.arange(np.max(cluster_idx) + 1):
        coord_c = coordinates[cluster_idx == i, :]
        if mode == "F":
            random = ppp(hull, 1, n_observations, obs_rng)
            tree_c = NearestNeighbors(metric=metric, n_neighbors=n_neigh).fit(coord_c)
            distances, _ = tree_c.kneighbors(random, n_neighbors=n_neigh)
            bins, obs_stats = f_g_function(distances.squeeze(), support)
        elif mode == "G":
            tree_c = NearestNeighbors(metric=metric, n_neighbors=n_neigh).fit(coord_c)
            distances, _ = tree_c.kneighbors(coordinates[cluster_idx != i, :], n_neighbors=n_neigh)
            bins, obs_stats = f_g_function(distances.squeeze(), support)
        elif mode == "L":
            bins, obs_stats = l_function(coord_c, support, N, area)
        else:
            raise NotImplementedError(mode)
        obs_arr[i] = obs_stats
    sims = np.empty((n_simulations, len(bins)))
    pvalues = np.ones((le.classes_.shape[0], len(bins)))
    for i in range(n_simulations):
        random_i = ppp(hull, 1, n_observations, sim_rngs[i])
        if mode == "F":
            tree_i = NearestNeighbors(metric=metric, n_neighbors=n_neigh).fit(random_i)
            distances_i, _ = tree_i.kneighbors(random, n_neighbors=1)
            _, stats_i = f_g_function(distances_i.squeeze(), support)
        elif mode == "G":
            tree_i = NearestNeighbors(metric=metric, n_neighbors=n_neigh).fit(random_i)
            distances_i, _ = tree_i.kneighbors(coordinates, n_neighbors=1)
            _, stats_i = f_g_function(distances_i.squeeze(), support)
        else:
            _, stats_i = l_function(random_i, support, N, area)
        for j in range(obs_arr.shape[0]):
            pvalues[j] += stats_i >= obs_arr[j]
        sims[i] = stats_i
    pvalues /= n_simulations + 1
    pvalues = np.minimum(pvalues, 1 - pvalues)
    return {"obs": obs_arr, "sims": sims, "bins": bins, "pvalues": pvalues, "classes": le.classes_, "area": area}


# =========================================================================== synthetic inputs


def hex_grid(rows: int, cols: int, scale: float = 100.0) -> np.ndarray:
    """SURVEY.md §8(d) hex lattice: x = c + 0.5 (r mod 2), y = r sqrt(3)/2, scaled; float64 (N, 2)."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    x = (c + 0.5 * (r % 2)).ravel() * scale
    y = (r * (np.sqrt(3.0) / 2.0)).ravel() * scale
    return np.stack([x, y], axis=1)


def hex_grid_graph(rows: int, cols: int) -> sparse.csr_matrix:
    """Closed-form 6-neighbour CSR of :func:`hex_grid` (float32 ones, int32 indices, sorted rows);
    equals the reference's ``GridBuilder(n_neighs=6)`` logic (gr/neighbors.py:335-419: kNN(6) then
    ``dist < 1.3 median``) — checked in tests at small sizes."""
    r, c = np.meshgrid(np.arange(rows), np.arange(cols), indexing="ij")
    r, c = r.ravel(), c.ravel()
    odd = r % 2
    cand = [
        (r, c - 1),
        (r, c + 1),
        (r - 1, c - 1 + odd),
        (r - 1, c + odd),
        (r + 1, c - 1 + odd),
        (r + 1, c + odd),
    ]
    src, dst = [], []
    me = r * cols + c
    for rr, cc in cand:
        ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
        src.append(me[ok])
        dst.append((rr * cols + cc)[ok])
    src, dst = np.concatenate(src), np.concatenate(dst)
    n = rows * cols
    g = sparse.csr_matrix((np.ones(len(src), dtype=np.float32), (src, dst)), shape=(n, n))
    g.sort_indices()
    g.indices = g.indices.astype(np.int32)
    g.indptr = g.indptr.astype(np.int32)
    return g


# =========================================================================== spatial graph construction (§8f-3)


def spatial_graph(
    coords: np.ndarray,
    kind: str,
    n_neighs: int = 6,
    radius: Any = None,
    n_rings: int = 1,
    set_diag: bool = False,
    percentile: float | None = None,
    transform: str | None = None,
    delaunay: bool = False,
) -> tuple[sparse.csr_matrix, sparse.csr_matrix]:
    """The reference's builders with sklearn's KD-tree exactly as they call it: KNNBuilder gr/neighbors.py:194-209,
    RadiusBuilder :247-269, DelaunayBuilder :319-331 (kind "delaunay"; a scalar radius means (0, r), :296-300),
    GridBuilder :366-419 (``delaunay``: base connectivity from the triangulation, :395-398), post-processors :425-476,
    spectral transform :514-548."""
    import warnings

    from sklearn.neighbors import NearestNeighbors

    N = coords.shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", sparse.SparseEfficiencyWarning)

        def base_grid(diag: bool) -> sparse.csr_matrix:
            if delaunay:
                from scipy.spatial import Delaunay

                indptr, indices = Delaunay(coords).vertex_neighbor_vertices
                a = sparse.csr_matrix((np.ones_like(indices, dtype=np.float32), indices, indptr), shape=(N, N))
                a.setdiag(1.0 if diag else a.diagonal())
                return a
            tree = NearestNeighbors(n_neighbors=n_neighs, radius=1, metric="euclidean").fit(coords)
            dists, cols = tree.kneighbors()
            dists, cols = dists.reshape(-1), cols.reshape(-1)
            rows = np.repeat(np.arange(N), n_neighs)
            mask = dists < np.median(dists) * 1.3
            a = sparse.csr_matrix((np.ones(mask.sum(), dtype=np.float32), (rows[mask], cols[mask])), shape=(N, N))
            a.setdiag(1.0 if diag else a.diagonal())
            return a

        if kind == "grid":
            if n_rings > 1:
                adj = base_grid(True)
                res, walk = adj, adj
                for i in range(n_rings - 1):
                    walk = walk @ adj
                    walk[res.nonzero()] = 0.0
                    walk.eliminate_zeros()
                    walk.data[:] = i + 2.0
                    res = res + walk
                adj = res
                adj.setdiag(float(set_diag))
                adj.eliminate_zeros()
                dst = adj.copy()
                adj.data[:] = 1.0
            else:
                adj = base_grid(set_diag)
                dst = adj.copy()
            dst.setdiag(0.0)
        else:
            if kind == "knn":
                tree = NearestNeighbors(n_neighbors=n_neighs, radius=1, metric="euclidean").fit(coords)
                dists, cols = tree.kneighbors()
                dists, cols = dists.reshape(-1), cols.reshape(-1)
                rows = np.repeat(np.arange(N), n_neighs)
            elif kind == "delaunay":
                from scipy.spatial import Delaunay

                indptr, cols = Delaunay(coords).vertex_neighbor_vertices
                rows = np.repeat(np.arange(N), np.diff(indptr))
                dists = np.linalg.norm(coords[rows] - coords[cols], axis=1)
                if isinstance(radius, (int, float)):
                    radius = (0.0, float(radius))
            else:
                r = radius if isinstance(radius, (int, float)) else max(radius)
                tree = NearestNeighbors(radius=r, metric="euclidean").fit(coords)
                dists, cols = tree.radius_neighbors()
                rows = np.repeat(np.arange(N), [len(x) for x in cols])
                dists, cols = np.concatenate(dists), np.concatenate(cols)
            adj = sparse.csr_matrix((np.ones_like(rows, dtype=np.float32), (rows, cols)), shape=(N, N))
            dst = sparse.csr_matrix((dists, (rows, cols)), shape=(N, N))
            adj.setdiag(1.0 if set_diag else adj.diagonal())
            dst.setdiag(0.0)
            if kind in ("radius", "delaunay") and isinstance(radius, tuple):
                minn, maxx = sorted(radius)
                mask = (dst.data < minn) | (dst.data > maxx)
                a_diag = adj.diagonal()
                dst.data[mask] = 0.0
                adj.data[mask] = 0.0
                adj.setdiag(a_diag)
            if percentile is not None:
                threshold = np.percentile(dst.data, percentile)
                adj[dst > threshold] = 0.0
                dst[dst > threshold] = 0.0
        adj.eliminate_zeros()
        dst.eliminate_zeros()
        if transform == "spectral" and adj.nnz:
            with np.errstate(divide="ignore"):
                deg = np.squeeze(np.array(np.sqrt(1.0 / adj.sum(axis=0))))
            rows = np.repeat(np.arange(N), np.diff(adj.indptr))
            adj = sparse.csr_matrix(((deg[rows] * deg[adj.indices] * adj.data).astype(np.float32), adj.indices, adj.indptr), shape=adj.shape)
        elif transform == "cosine":
            from sklearn.metrics.pairwise import cosine_similarity

            adj = cosine_similarity(adj, dense_output=False)
    return sparse.csr_matrix(adj), sparse.csr_matrix(dst)


# ----------------------------------------------------------------------------------------------- ligrec (§8(f) row 4)
def ligrec_group_means(data: np.ndarray, perm: np.ndarray, inv_counts: np.ndarray) -> np.ndarray:
    """`groups` of one permutation (gr/_ligrec.py:647-655): per cluster the sum of the rows in cell order, then
    multiplied by the reciprocal cluster size.  ``np.add.at`` is unbuffered and visits the rows in index order, i.e.
    every (cluster, gene) cell is accumulated in exactly the sequence of the reference's double loop."""
    groups = np.zeros((len(inv_counts), data.shape[1]), dtype=np.float64)
    np.add.at(groups, np.asarray(perm, dtype=np.int64), data)
    return groups * np.asarray(inv_counts, dtype=np.float64)[:, None]


def ligrec_perm_labels_numpy(clustering: np.ndarray, seed: int | None, n_perms: int) -> np.ndarray:
    """Label vectors of all permutations under numpy streams: one generator per permutation, each shuffling a fresh
    copy of the clustering (gr/_ligrec.py:643-646, 748)."""
    out = np.empty((n_perms, len(clustering)), dtype=np.int32)
    for p, rs in enumerate(spawn_generators(seed, n_perms)):
        perm = np.array(clustering, dtype=np.int32, copy=True)
        rs.shuffle(perm)
        out[p] = perm
    return out


def ligrec_perm_labels_philox(clustering: np.ndarray, seed: int, perm_begin: int, perm_end: int) -> np.ndarray:
    """Label vectors of the device generator (oracle/devrng.py), permutations [perm_begin, perm_end)."""
    from oracle import devrng

    return np.stack([devrng.shuffled_labels(np.asarray(clustering), seed, p) for p in range(perm_begin, perm_end)]).astype(np.int32)


def ligrec_score_permutations(
    data: np.ndarray,
    perm_labels: np.ndarray,
    inv_counts: np.ndarray,
    mean_obs: np.ndarray,
    interactions: np.ndarray,
    interaction_clusters: np.ndarray,
    valid: np.ndarray,
) -> np.ndarray:
    """``_score_permutations`` (gr/_ligrec.py:616-673) for the given shuffled label vectors (n_perms, n_cells):
    counts[i, j] = #{p : valid[i, j] and groups_p[a_j, rec_i] + groups_p[b_j, lig_i] > mean_obs[a_j, rec_i] + mean_obs[b_j, lig_i]}."""
    rec, lig = interactions[:, 0], interactions[:, 1]
    a, b = interaction_clusters[:, 0], interaction_clusters[:, 1]
    obs = mean_obs[a][:, rec].T + mean_obs[b][:, lig].T  # (n_inter, n_cpairs), float64 add as in :665
    counts = np.zeros(obs.shape, dtype=np.int64)
    for perm in perm_labels:
        groups = ligrec_group_means(data, perm, inv_counts)
        shuf = groups[a][:, rec].T + groups[b][:, lig].T
        counts += (valid & (shuf > obs)).astype(np.int64)
    return counts


def ligrec_prepare(
    data: np.ndarray, clustering: np.ndarray, interactions: np.ndarray, interaction_clusters: np.ndarray, threshold: float
) -> dict[str, np.ndarray]:
    """Host part of ``_analysis`` before the kernel call (gr/_ligrec.py:712-745): observed cluster means (pandas
    groupby mean, as the reference), expression-fraction mask, reciprocal cluster sizes, `valid` and `means`."""
    import pandas as pd

    df = pd.DataFrame(np.asarray(data, dtype=np.float64))
    groups = df.groupby(np.asarray(clustering), observed=True)
    mean_obs = groups.mean().values
    n_cls = mean_obs.shape[0]
    sizes = np.bincount(np.asarray(clustering), minlength=n_cls).astype(np.float64)
    frac = np.stack([(np.asarray(data)[np.asarray(clustering) == k] > 0).astype(np.int64).sum(axis=0) / sizes[k] for k in range(n_cls)])
    mask = frac >= threshold
    inv_counts = 1.0 / np.maximum(sizes, 1)
    rec, lig = interactions[:, 0], interactions[:, 1]
    c1, c2 = interaction_clusters[:, 0], interaction_clusters[:, 1]
    m_rec = mean_obs[c1, :][:, rec].T
    m_lig = mean_obs[c2, :][:, lig].T
    nonzero = (m_rec > 0) & (m_lig > 0)
    valid = nonzero & mask[c1, :][:, rec].T & mask[c2, :][:, lig].T
    means = np.where(nonzero, (m_rec + m_lig) / 2.0, 0.0)
    return {"mean_obs": mean_obs, "mask": mask, "inv_counts": inv_counts, "valid": valid, "means": means, "obs": m_rec + m_lig}


def ligrec_analysis(
    data: np.ndarray,
    clustering: np.ndarray,
    interactions: np.ndarray,
    interaction_clusters: np.ndarray,
    threshold: float = 0.1,
    n_perms: int = 1000,
    seed: int | None = None,
    perm_labels: np.ndarray | None = None,
) -> tuple[np.ndarray, np.ndarray]:
    """``_analysis`` (gr/_ligrec.py:677-775): (means, pvalues).  ``perm_labels`` overrides the numpy streams."""
    interactions = np.asarray(interactions, dtype=np.int32)
    interaction_clusters = np.asarray(interaction_clusters, dtype=np.int32)
    pre = ligrec_prepare(data, clustering, interactions, interaction_clusters, threshold)
    if perm_labels is None:
        perm_labels = ligrec_perm_labels_numpy(clustering, seed, n_perms)
    counts = ligrec_score_permutations(
        np.asarray(data, dtype=np.float64), perm_labels, pre["inv_counts"], pre["mean_obs"], interactions, interaction_clusters, pre["valid"]
    )
    pvalues = counts.astype(np.float64) / len(perm_labels)
    pvalues[~pre["valid"]] = np.nan
    return pre["means"], pvalues
