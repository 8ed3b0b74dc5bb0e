"""``ripley`` (F / G / L statistics) with the reference's signature on the MI355X path.

Reference: /root/reference/src/squidpy/gr/_ripley.py:27-271.  The pair counting of ``_l_function`` and the
nearest-neighbour queries of the F/G modes run in ``libsqgr.so``; hull/area, the Poisson-process simulations (numpy
RNG parity) and the result frames stay on the host exactly as in the reference."""

from __future__ import annotations

from typing import Any

import numpy as np
import pandas as pd

from .._constants import Key, RipleyStat
from .._lib import METRICS, Context, default_context, knn_dist, pair_counts
from .._utils import _assert_categorical_obs, _assert_spatial_basis, _save_data, extract_adata_if_sdata, spawn_generators

__all__ = ["ripley"]

# sklearn.neighbors.KDTree.valid_metrics (the reference's check in `_l_function`, gr/_ripley.py:213-214)
KDTREE_VALID_METRICS = ["euclidean", "l2", "minkowski", "p", "manhattan", "cityblock", "l1", "chebyshev", "infinity"]


def _reshape_res(results: np.ndarray, columns: Any, index: np.ndarray, var_name: str) -> pd.DataFrame:
    """gr/_ripley.py:197-203."""
    df = pd.DataFrame(results, columns=columns, index=index)
    df.index.set_names(["bins"], inplace=True)
    df = df.melt(var_name=var_name, value_name="stats", ignore_index=False)
    df[var_name] = df[var_name].astype("category", copy=True)
    df.reset_index(inplace=True)
    return df


def _f_g_function(distances: np.ndarray, support: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """gr/_ripley.py:206-209."""
    counts, bins = np.histogram(distances, bins=support)
    with np.errstate(divide="ignore", invalid="ignore"):
        fracs = np.cumsum(counts) / counts.sum()
    return bins, np.concatenate((np.zeros((1,), dtype=float), fracs))


def _l_function(ctx: Context, points: np.ndarray, support: np.ndarray, n: int, area: float, metric: str) -> tuple[np.ndarray, np.ndarray]:
    """gr/_ripley.py:212-227 with the dual-tree pair count replaced by the GPU sweep."""
    if metric not in KDTREE_VALID_METRICS:
        raise ValueError(f"Unsupported metric '{metric}'. Ripley's L supports {KDTREE_VALID_METRICS}")
    n_pairs = pair_counts(ctx, points, support, metric)
    intensity = n / area
    k_estimate = (n_pairs / n) / intensity
    l_estimate = np.sqrt(k_estimate / np.pi)
    return support, l_estimate


def _ppp(hull: Any, n_simulations: int, n_observations: int, rng: np.random.Generator) -> np.ndarray:
    """gr/_ripley.py:230-271: rejection sampling of uniform points inside the convex hull.

    Draw-for-draw equivalent to the reference's scalar loop (two ``rng.uniform`` per candidate, accepted iff
    ``Delaunay.find_simplex >= 0``): candidates are generated in vectorised blocks from the same stream of doubles,
    and the generator is left in exactly the state the scalar loop would leave it in."""
    from scipy.spatial import Delaunay

    vxs = hull.points[hull.vertices]
    deln = Delaunay(vxs)
    bbox = np.array([*vxs.min(0), *vxs.max(0)])
    result = np.empty((n_simulations, n_observations, 2))
    for i_sim in range(n_simulations):
        i_obs = 0
        while i_obs < n_observations:
            need = n_observations - i_obs
            block = max(64, int(need * 1.5) + 16)
            state = rng.bit_generator.state
            u = rng.random(2 * block)
            x = bbox[0] + (bbox[2] - bbox[0]) * u[0::2]
            y = bbox[1] + (bbox[3] - bbox[1]) * u[1::2]
            pts = np.stack([x, y], axis=1)
            inside = deln.find_simplex(pts) >= 0
            n_in = int(inside.sum())
            if n_in >= need:  # consume exactly the candidates the scalar loop would have drawn
                last = int(np.flatnonzero(inside)[need - 1])
                rng.bit_generator.state = state
                rng.random(2 * (last + 1))
                sel = pts[: last + 1][inside[: last + 1]]
            else:
                sel = pts[inside]
            result[i_sim, i_obs : i_obs + len(sel)] = sel
            i_obs += len(sel)
    return result.squeeze()


def ripley(
    adata: Any,
    cluster_key: str,
    mode: str = "F",
    spatial_key: str = Key.obsm.spatial,
    metric: str = "euclidean",
    n_neigh: int = 2,
    n_simulations: int = 100,
    n_observations: int = 1000,
    max_dist: float | None = None,
    n_steps: int = 50,
    seed: int | None = None,
    copy: bool = False,
    *,
    table_key: str | None = None,
    device: int | None = None,
) -> dict[str, Any] | None:
    """Calculate various Ripley's statistics for point processes (drop-in for ``squidpy.gr.ripley``).

    Same parameters, numpy random streams (``spawn_generators(seed, n_simulations + 1)``), result keys
    (``'{mode}_stat'``, ``'sims_stat'``, ``'bins'``, ``'pvalues'``) and ``adata.uns['{cluster_key}_ripley_{mode}']``
    slot as the reference.  Supported metrics on the GPU: euclidean / manhattan / chebyshev (and their aliases).
    """
    from scipy.spatial import ConvexHull
    from sklearn.preprocessing import LabelEncoder

    adata = extract_adata_if_sdata(adata, table_key=table_key)
    _assert_categorical_obs(adata, key=cluster_key)
    _assert_spatial_basis(adata, key=spatial_key)
    coordinates = np.asarray(adata.obsm[spatial_key])
    clusters = adata.obs[cluster_key].values

    mode = RipleyStat(mode)
    if mode == RipleyStat.L and metric not in KDTREE_VALID_METRICS:
        raise ValueError(f"Unsupported metric '{metric}'. Ripley's L supports {KDTREE_VALID_METRICS}")
    if metric not in METRICS:
        raise NotImplementedError(f"Metric `{metric}` is not implemented on the GPU path; use one of {sorted(METRICS)}.")
    ctx = default_context(device)

    # prepare support
    N = coordinates.shape[0]
    hull = ConvexHull(coordinates)
    area = hull.volume
    if max_dist is None:
        max_dist = (area / 2) ** 0.5
    support = np.linspace(0, max_dist, n_steps)

    # prepare labels
    le = LabelEncoder().fit(clusters)
    cluster_idx = le.transform(clusters)
    obs_arr = np.empty((le.classes_.shape[0], n_steps))
    obs_rng, *sim_rngs = spawn_generators(seed, n_simulations + 1)
    coords64 = coordinates.astype(np.float64)

    random = None
    bins = support
    for i in np.arange(np.max(cluster_idx) + 1):
        coord_c = coords64[cluster_idx == i, :]
        if mode == RipleyStat.F:
            random = _ppp(hull, n_simulations=1, n_observations=n_observations, rng=obs_rng)
            distances = knn_dist(ctx, random, coord_c, n_neigh, metric)
            bins, obs_stats = _f_g_function(distances.squeeze(), support)
        elif mode == RipleyStat.G:
            distances = knn_dist(ctx, coords64[cluster_idx != i, :], coord_c, n_neigh, metric)
            bins, obs_stats = _f_g_function(distances.squeeze(), support)
        else:
            bins, obs_stats = _l_function(ctx, coord_c, support, N, area, metric)
        obs_arr[i] = obs_stats

    sims = np.empty((n_simulations, len(bins)))
    pvalues = np.ones((le.classes_.shape[0], len(bins)))
    for i in range(n_simulations):
        random_i = _ppp(hull, n_simulations=1, n_observations=n_observations, rng=sim_rngs[i])
        if mode == RipleyStat.F:
            distances_i = knn_dist(ctx, random, random_i, 1, metric)
            _, stats_i = _f_g_function(distances_i.squeeze(), support)
        elif mode == RipleyStat.G:
            distances_i = knn_dist(ctx, coords64, random_i, 1, metric)
            _, stats_i = _f_g_function(distances_i.squeeze(), support)
        else:
            _, stats_i = _l_function(ctx, random_i, support, N, area, metric)
        for j in range(obs_arr.shape[0]):
            pvalues[j] += stats_i >= obs_arr[j]
        sims[i] = stats_i

    pvalues /= n_simulations + 1
    pvalues = np.minimum(pvalues, 1 - pvalues)

    obs_df = _reshape_res(obs_arr.T, columns=le.classes_, index=bins, var_name=cluster_key)
    sims_df = _reshape_res(sims.T, columns=np.arange(n_simulations), index=bins, var_name="simulations")
    res = {f"{mode}_stat": obs_df, "sims_stat": sims_df, "bins": bins, "pvalues": pvalues}

    if copy:
        return res
    _save_data(adata, attr="uns", key=Key.uns.ripley(cluster_key, mode.s), data=res)
    return None
