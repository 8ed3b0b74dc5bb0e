"""``ripley`` (F / G / L statistics) with the reference's signature on the MI355X path.

Reference: /root/reference/src/squidpy/gr/_ripley.py:27-271.  The pair counting of ``_l_function`` and the
nearest-neighbour queries of the F/G modes run in ``libsqgr.so``; hull/area, the Poisson-process simulations (numpy
RNG parity) and the result frames stay on the host exactly as in the reference."""

from __future__ import annotations

from typing import Any

import numpy as np
import pandas as pd

from .._constants import Key, RipleyStat
from .. import _dist
from .._lib import METRICS, Context, DevicePoints, default_context, knn_dist, pair_counts, pair_counts_batch
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


class _Engine:
    """Evaluates one Ripley statistic (F, G or L) for a point set on the GPU.

    ``mode`` decides what "points" means: L -> pair counts inside the set; G -> nearest-neighbour distances from the
    query set to the set; F -> the same with simulated query points (gr/_ripley.py:140-155, 160-177)."""

    def __init__(self, ctx: Context, mode: RipleyStat, metric: str, support: np.ndarray, n_total: int, area: float):
        self.ctx, self.mode, self.metric, self.support = ctx, mode, metric, support
        self.n_total, self.area = n_total, area

    def l_stat(self, points: np.ndarray) -> np.ndarray:
        return _l_function(self.ctx, points, self.support, self.n_total, self.area, self.metric)[1]

    def l_stat_many(self, point_sets: "list[np.ndarray]") -> np.ndarray:
        """`_l_function` (gr/_ripley.py:212-227) of several point sets — one per cluster, or one per simulation — with their
        pair counts taken in ONE launch (`sqgr_pair_counts_batch`); row k = `l_stat(point_sets[k])`."""
        if self.metric not in KDTREE_VALID_METRICS:
            raise ValueError(f"Unsupported metric '{self.metric}'. Ripley's L supports {KDTREE_VALID_METRICS}")
        n_pairs = pair_counts_batch(self.ctx, point_sets, self.support, self.metric)
        k_estimate = (n_pairs / self.n_total) / (self.n_total / self.area)
        return np.sqrt(k_estimate / np.pi)

    def nn_stat(self, queries: np.ndarray, refs: np.ndarray, k: int) -> np.ndarray:
        dist = knn_dist(self.ctx, queries, refs, k, self.metric)
        return _f_g_function(dist.squeeze(), self.support)[1]

    def nn_stat_resident(self, queries: DevicePoints, refs: np.ndarray, k: int, exclude_label: int = -1) -> np.ndarray:
        """The same statistic with the query set resident on the device and the histogram of `_f_g_function` formed
        there (Ripley's G queries ~all points per cluster and per simulation)."""
        counts = queries.knn_hist(refs, k, self.support, self.metric, exclude_label)
        with np.errstate(divide="ignore", invalid="ignore"):
            fracs = np.cumsum(counts) / counts.sum()
        return np.concatenate((np.zeros((1,), dtype=float), fracs))


def _assign_by_cost(cost: np.ndarray, world: int) -> np.ndarray:
    """Longest-processing-time greedy: item -> rank, deterministic (ties by index), balanced on ``cost``."""
    owner = np.zeros(len(cost), dtype=np.int64)
    if world <= 1:
        return owner
    load = np.zeros(world)
    for i in sorted(range(len(cost)), key=lambda j: (-cost[j], j)):
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += cost[i]
    return owner


def _tail_pvalues(obs: np.ndarray, sims: np.ndarray) -> np.ndarray:
    """gr/_ripley.py:175-180: ``(1 + #{sim >= obs}) / (n_sim + 1)`` folded to the smaller tail."""
    exceed = (sims[:, None, :] >= obs[None, :, :]).sum(axis=0)
    p = (1.0 + exceed) / (sims.shape[0] + 1)
    return np.minimum(p, 1 - p)


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

    Same parameters, numpy random streams (``spawn_generators(seed, n_simulations + 1)``: the first generator draws the
    observed-mode Poisson patterns, the others one simulation each), result keys (``'{mode}_stat'``, ``'sims_stat'``,
    ``'bins'``, ``'pvalues'``) and ``adata.uns['{cluster_key}_ripley_{mode}']`` slot as the reference.  Metrics on the GPU:
    euclidean / manhattan / chebyshev and their sklearn aliases (``l2``, ``minkowski`` / ``p`` at sklearn's default p = 2,
    ``cityblock``, ``l1``, ``infinity``) for every mode; ``canberra`` for F and G (Ripley's L takes
    :class:`sklearn.neighbors.KDTree` metrics only, as in the reference).  Metrics that need parameters a metric string
    cannot carry (``seuclidean``, ``mahalanobis``, ``minkowski`` with another p), libm transcendentals (``haversine``) or
    that are not metrics, so that sklearn's ball tree itself returns inexact neighbours for them (``braycurtis``), raise
    ``NotImplementedError`` naming the supported set.
    """
    from scipy.spatial import ConvexHull
    from sklearn.preprocessing import LabelEncoder

    adata = extract_adata_if_sdata(adata, table_key=table_key)
    _assert_categorical_obs(adata, key=cluster_key)
    _assert_spatial_basis(adata, key=spatial_key)
    stat = RipleyStat(mode)
    if stat == RipleyStat.L and metric not in KDTREE_VALID_METRICS:
        raise ValueError(f"Unsupported metric '{metric}'. Ripley's L supports {KDTREE_VALID_METRICS}")
    if metric in ("seuclidean", "mahalanobis"):
        # the reference fails here too: `NearestNeighbors(metric=metric)` needs V / VI, which `ripley` has no argument for
        # (gr/_ripley.py:144,148; sklearn: TypeError for seuclidean, "Must provide either V or VI" for mahalanobis)
        # same exception TYPES as the reference's sklearn call raises for them: TypeError / ValueError
        exc = TypeError if metric == "seuclidean" else ValueError
        raise exc(f"Metric `{metric}` needs parameters (V / VI) that `ripley` cannot pass on: the reference's NearestNeighbors call fails as well.")
    if metric not in METRICS:
        raise NotImplementedError(f"Metric `{metric}` is not implemented on the GPU path; use one of {sorted(METRICS)}.")

    xy = np.asarray(adata.obsm[spatial_key])
    xy64 = xy.astype(np.float64)
    hull = ConvexHull(xy)
    area = hull.volume
    radius_max = (area / 2) ** 0.5 if max_dist is None else max_dist
    support = np.linspace(0, radius_max, n_steps)
    encoder = LabelEncoder().fit(adata.obs[cluster_key].values)
    codes = encoder.transform(adata.obs[cluster_key].values)
    n_groups = encoder.classes_.shape[0]
    engine = _Engine(default_context(device), stat, metric, support, xy.shape[0], area)
    first_rng, *other_rngs = spawn_generators(seed, n_simulations + 1)

    # Multi-GPU: clusters are independent point sets -> spread over the ranks by cost (longest-processing-time
    # greedy on m_c^2 pair tests for L, m_c reference points for F/G); simulations round-robin.  Every rank draws the host-side
    # random streams it needs in the reference's order, so the result does not depend on the number of ranks.
    rank, world = _dist.world()
    sizes = np.bincount(codes, minlength=n_groups).astype(np.float64)
    owner = _assign_by_cost(sizes * sizes if stat == RipleyStat.L else sizes, world)

    # the clusters' points, each in obs order (= `xy[codes == gidx]`), from ONE stable sort instead of a mask per cluster
    by_cluster = xy64[np.argsort(codes, kind="stable")]
    starts = np.concatenate([[0], np.cumsum(np.bincount(codes, minlength=n_groups))])

    def members_of(gidx: int) -> np.ndarray:
        return by_cluster[starts[gidx] : starts[gidx + 1]]

    # observed statistic per cluster (F: each cluster is probed with its own Poisson pattern drawn from `first_rng`)
    observed = np.full((n_groups, n_steps), np.nan)
    probe = None
    everyone = DevicePoints(engine.ctx, xy64, codes) if stat == RipleyStat.G else None  # G queries (nearly) all points
    for gidx in range(int(codes.max()) + 1):
        if stat == RipleyStat.F:  # the stream of `first_rng` runs through all clusters: drawn on every rank
            probe = _ppp(hull, n_simulations=1, n_observations=n_observations, rng=first_rng)
        if owner[gidx] != rank:
            continue
        members = members_of(gidx)
        if stat == RipleyStat.L:
            pass  # the clusters this rank owns are counted together below
        elif stat == RipleyStat.G:
            observed[gidx] = engine.nn_stat_resident(everyone, members, n_neigh, exclude_label=gidx)
        else:
            observed[gidx] = engine.nn_stat(probe, members, n_neigh)

    if stat == RipleyStat.L:
        mine = [gidx for gidx in range(int(codes.max()) + 1) if owner[gidx] == rank]
        if mine:
            observed[mine] = engine.l_stat_many([members_of(gidx) for gidx in mine])

    # null distribution: complete spatial randomness inside the hull, one pattern per generator
    simulated = np.full((n_simulations, n_steps), np.nan)
    l_patterns: list[tuple[int, np.ndarray]] = []
    for s_idx, rng in enumerate(other_rngs):
        if s_idx % world != rank:
            continue
        pattern = _ppp(hull, n_simulations=1, n_observations=n_observations, rng=rng)
        if stat == RipleyStat.L:
            l_patterns.append((s_idx, pattern))  # all of this rank's simulations in one launch below
        elif stat == RipleyStat.G:
            simulated[s_idx] = engine.nn_stat_resident(everyone, pattern, 1)
        else:  # the reference reuses the probe pattern of the LAST cluster here (gr/_ripley.py:163-165)
            simulated[s_idx] = engine.nn_stat(probe, pattern, 1)
    if l_patterns:
        simulated[[k for k, _ in l_patterns]] = engine.l_stat_many([pat for _, pat in l_patterns])
    if world > 1:  # rows are owned by exactly one rank: gather and take each from its owner (int64[n_steps]-sized rows)
        parts = _dist.allgather_object((observed, simulated))
        for gidx in range(n_groups):
            observed[gidx] = parts[owner[gidx]][0][gidx]
        for s_idx in range(n_simulations):
            simulated[s_idx] = parts[s_idx % world][1][s_idx]

    if everyone is not None:
        everyone.close()
    res = {
        f"{stat}_stat": _reshape_res(observed.T, columns=encoder.classes_, index=support, var_name=cluster_key),
        "sims_stat": _reshape_res(simulated.T, columns=np.arange(n_simulations), index=support, var_name="simulations"),
        "bins": support,
        "pvalues": _tail_pvalues(observed, simulated),
    }
    if copy:
        return res
    _save_data(adata, attr="uns", key=Key.uns.ripley(cluster_key, stat.s), data=res)
    return None
