"""``nhood_enrichment`` / ``interaction_matrix`` with the reference's signatures on the MI355X path.

Reference: /root/reference/src/squidpy/gr/_nhood.py:146-242 (nhood_enrichment), :349-429
(interaction_matrix).  All counting and all label shuffling run in ``libsqgr.so`` (HIP); the host only
validates, uploads and turns exact integer moments into z-scores."""

from __future__ import annotations

import math
import warnings
from typing import Any, NamedTuple

import numpy as np
import pandas as pd

from .. import _dist
from .._constants import Key
from .._lib import Context, Graph, NhoodPlan, cached_graph, default_context, interaction_matrix as _intmat, nhood_counts, nhood_counts_batch
from .._utils import (
    _assert_categorical_obs,
    _assert_connectivity_key,
    _save_data,
    assert_positive,
    category_codes,
    extract_adata_if_sdata,
    get_n_processes,
    logg,
    pcg64_states,
    progress,
    resolve_seed,
    spawn_generators,
)

__all__ = ["nhood_enrichment", "interaction_matrix", "NhoodEnrichmentResult"]


class NhoodEnrichmentResult(NamedTuple):
    """Result of nhood_enrichment (gr/_nhood.py:44-48)."""

    zscore: np.ndarray
    counts: np.ndarray


def expected_counts(labels: np.ndarray, n_cls: int, nnz: int) -> np.ndarray:
    """Integer shift ~ E[count] under shuffling, nnz * p_a * p_b: keeps the accumulated d = count - shift small.
    Any integer shift gives the same exact result; this one just keeps sum d^2 far from 2**64."""
    n = len(labels)
    freq = np.bincount(labels, minlength=n_cls).astype(np.float64) / max(n, 1)
    return np.rint(nnz * np.outer(freq, freq)).astype(np.int64)


def zscore_from_moments(count: np.ndarray, shift: np.ndarray, sum_d: np.ndarray, sum_d2: np.ndarray, n_perms: int) -> np.ndarray:
    """(count - mean) / std with population std (gr/_nhood.py:231), from exact integer moments.

    mean = shift + S1/P and var = (P*S2 - S1^2)/P^2 are evaluated in exact integer arithmetic and rounded
    once, so the result equals numpy's float64 ``perms.mean/std`` up to its own rounding (~1e-13 rel.)."""
    k = count.shape[0]
    P = int(n_perms)
    mean = np.empty((k, k), dtype=np.float64)
    std = np.empty((k, k), dtype=np.float64)
    sh, s1, s2 = shift.reshape(-1), sum_d.reshape(-1), sum_d2.reshape(-1)
    for i in range(k * k):
        a, b = int(s1[i]), int(s2[i])
        mean.flat[i] = (int(sh[i]) * P + a) / P
        std.flat[i] = math.sqrt((P * b - a * a) / (P * P))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (count.astype(np.float64) - mean) / std


def nhood_enrichment(
    adata: Any,
    cluster_key: str,
    library_key: str | None = None,
    connectivity_key: str | None = None,
    n_perms: int = 1000,
    numba_parallel: bool = False,
    seed: int | None = None,
    copy: bool = False,
    n_jobs: int | None = None,
    backend: str = "loky",
    show_progress_bar: bool = True,
    *,
    table_key: str | None = None,
    rng: str | None = None,
    device: int | None = None,
) -> NhoodEnrichmentResult | None:
    """Compute neighborhood enrichment by permutation test (drop-in for ``squidpy.gr.nhood_enrichment``).

    Same positional parameters, defaults, validation errors and ``adata.uns`` slots as the reference.
    ``numba_parallel``, ``n_jobs`` and ``backend`` are accepted (and ``n_jobs`` validated) but do not influence the GPU path;
    ``show_progress_bar`` drives a tqdm bar over permutation batches (on a terminal, rank 0 only).

    Extra keyword-only parameters
    -----------------------------
    rng
        ``None`` = ``"numpy"`` (the default since round 5 — a BREAKING change against rounds 1-4, whose default was ``"philox"``:
        the z-scores of a given ``seed`` are now Squidpy's, not the device generator's; a default call that is large enough for the
        11x slower stream to matter — ``n_obs * n_perms >= 5e9`` — says so once, and so does a multi-rank call that has to gather
        the per-permutation counts through the host).  ``"numpy"``: the reference's own streams (``SeedSequence(seed).spawn(n_perms)`` -> PCG64 ->
        ``Generator.shuffle``, gr/_nhood.py:213, 530-539) are reproduced bit for bit *on the GPU* (LCG jump-ahead draws; long
        arrays replay the swaps phase by phase through LDS, ``csrc/sqgr_pcg.hip``), and the z-score is formed with the
        reference's float64 ``perms.mean/std``: a default call returns **Squidpy's z-scores for that ``seed``, exactly**
        (1e6 spots: tens of thousands of permutations/s — a default call of 1000 permutations takes ~30 ms; the CPU does ~20/s
        per core).
        ``"philox"``: the throughput mode (~1 M permutations/s at 1e6 spots x 30 clusters): label shuffles by the counter-based
        generator of ``csrc/sqgr_rng.h`` keyed by ``(seed, permutation index, library)``; reproducible for a given ``seed``
        and independent of the number of GPUs, but another stream than numpy's — the z-scores agree with Squidpy's
        statistically (same null distribution, ``tests/test_null_moments_gpu.py``), not digit for digit.
        ``"numpy-host"``: numpy's streams drawn by numpy on the host and injected (cross-check path).
    device
        HIP device index (default: ``LOCAL_RANK`` or 0).

    With a process group (``squidpy_amd.init_distributed()`` under any one-process-per-GPU launcher, or an already
    initialised ``torch.distributed`` group) the permutation range is split across ranks and the exact integer moments
    are all-reduced on the device by RCCL inside libsqgr; every rank returns the full result.
    """
    adata = extract_adata_if_sdata(adata, table_key=table_key)
    connectivity_key = Key.obsp.spatial_conn(connectivity_key)
    _assert_categorical_obs(adata, cluster_key)
    _assert_connectivity_key(adata, connectivity_key)
    assert_positive(n_perms, name="n_perms")
    rng_defaulted = rng is None
    rng = "numpy" if rng is None else rng
    if rng not in ("philox", "numpy", "numpy-host"):
        raise ValueError(f"Invalid option `{rng}` for `rng`. Valid options are: `['philox', 'numpy', 'numpy-host']`.")
    if rng_defaulted and adata.n_obs * n_perms >= DEFAULT_STREAM_NOTICE_WORK:
        warnings.warn(
            f"nhood_enrichment: {n_perms} permutations of {adata.n_obs} observations with the default `rng='numpy'` (Squidpy's own PCG64 streams, "
            "reproduced on the GPU: ~90 k permutations/s at 1e6 spots).  `rng='philox'` runs the same test ~11x faster with the device "
            "generator (same null distribution, other digits); pass `rng='numpy'` explicitly to keep Squidpy's numbers without this note.",
            UserWarning, stacklevel=2)

    adj = adata.obsp[connectivity_key]
    int_clust, n_cls = category_codes(adata.obs[cluster_key])
    if library_key is not None:
        _assert_categorical_obs(adata, key=library_key)
        lib_codes, n_libs = category_codes(adata.obs[library_key])
    else:
        lib_codes, n_libs = None, 0
    if n_cls <= 1:
        raise ValueError(f"Expected at least `2` clusters, found `{n_cls}`.")  # gr/_nhood.py:107-108
    get_n_processes(n_jobs)

    ctx = default_context(device)
    graph = cached_graph(ctx, adj, with_data=False)  # stays resident for the next statistic on the same matrix
    count = nhood_counts(ctx, graph, int_clust, n_cls)
    if n_cls > MAX_DEVICE_SHUFFLE_CLUSTERS or (n_cls > 256 and (rng == "numpy-host" or (rng == "numpy" and lib_codes is not None))):
        # (numpy streams + libraries + more than 256 clusters: per-library sub-shuffles of 16-bit labels are not on the device;
        #  the injected-permutation entry point `sqgr_nhood_counts_batch` takes uint8 labels, so host-drawn streams with more
        #  than 256 clusters are counted one permutation at a time by the any-K kernel)
        zscore = _zscore_many_clusters(ctx, graph, int_clust, n_cls, lib_codes, n_libs, seed, n_perms, count)
    elif rng == "numpy-host":
        zscore = _zscore_numpy_streams(ctx, graph, int_clust, n_cls, lib_codes, n_libs, seed, n_perms, count)
    elif rng == "numpy":
        rank, world = _dist.world()
        lo, hi = _dist.shard_range(n_perms, rank, world)
        if seed is None:
            seed = _broadcast_seed(resolve_seed(None))
        comm = _device_comm(ctx)
        plan = NhoodPlan(ctx, graph, int_clust, n_cls, lib_codes, n_libs)
        try:
            if world == 1 or comm is not None:
                # the float64 mean/std of gr/_nhood.py:231 formed on the device, bit for bit; with several ranks each
                # runs its contiguous chunk of the streams and the per-permutation counts are all-gathered by RCCL
                plan.set_comm(comm)
                mean, std = plan.run_pcg64_stats(pcg64_states(seed, n_perms))
                perms = None
            else:
                if n_perms * n_cls * n_cls >= HOST_GATHER_NOTICE_ENTRIES:
                    warnings.warn(
                        f"nhood_enrichment(rng='numpy') on {world} ranks without a device communicator: the {n_perms} x {n_cls} x {n_cls} per-permutation counts "
                        "are gathered through the host side channel on every rank (RCCL between the ranks' GPUs — one process per GPU — keeps them on the "
                        "device; `rng='philox'` all-reduces 2 K^2 integers instead).", UserWarning, stacklevel=2)
                _, _, perms = plan.run_pcg64(pcg64_states(seed, n_perms, lo, hi), return_perms=True)
        finally:
            plan.close()
        if perms is not None:  # host side channel only (ranks sharing a GPU): numpy reduces the gathered counts
            perms = np.concatenate(_dist.allgather_object(perms), axis=0).astype(np.float64)
            mean, std = perms.mean(axis=0), perms.std(axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            zscore = (count - mean) / std  # gr/_nhood.py:231
    else:
        rank, world = _dist.world()
        lo, hi = _dist.shard_range(n_perms, rank, world)
        shift = expected_counts(int_clust, n_cls, graph.nnz)
        comm = _device_comm(ctx)
        # observations in no spatial order: the plan runs on a renumbered twin of the graph (built on the device, kept with the
        # graph) and the generator permutes the ranks of the CALLER's observations — the same moments, bit for bit
        order = _internal_order(ctx, adata, adj, n_cls, lib_codes, n_perms)
        if order is not None:
            plan = NhoodPlan(ctx, graph.renumbered(order), int_clust[order], n_cls)
            plan.set_spot_map(order)
        else:
            plan = NhoodPlan(ctx, graph, int_clust, n_cls, lib_codes, n_libs)
        try:
            key = _broadcast_seed(resolve_seed(seed))
            plan.set_comm(comm)  # the exact integer moments are all-reduced on the device (RCCL inside libsqgr)
            # the permutation range in a few pieces (whole launch groups each): the exact integer moments add up, the generator is
            # keyed by the global index, so the result does not depend on the split — it only lets the progress bar move
            longest = -(-n_perms // world)  # every rank enters the same number of collectives: pieces are cut from the longest range
            step = max(PROGRESS_STEP, -(-longest // 20) // PROGRESS_STEP * PROGRESS_STEP) if show_progress_bar else max(longest, 1)
            s1 = np.zeros((n_cls, n_cls), dtype=np.int64)
            s2 = np.zeros((n_cls, n_cls), dtype=np.uint64)
            n_pieces = -(-longest // step)
            with progress(hi - lo, "perm", show_progress_bar) as bar:
                for q in range(max(n_pieces, 1)):
                    a, b = min(hi, lo + q * step), min(hi, lo + (q + 1) * step)
                    p1, p2, _ = plan.run(key, a, b, shift)
                    s1 += p1
                    s2 += p2
                    bar.update(b - a)
        finally:
            plan.close()
        if comm is None:
            s1, s2 = _dist.allreduce_sum_([s1, s2])
        zscore = zscore_from_moments(count, shift, s1, s2, n_perms)

    if copy:
        return NhoodEnrichmentResult(zscore=zscore, counts=count)
    _save_data(adata, attr="uns", key=Key.uns.nhood_enrichment(cluster_key), data={"zscore": zscore, "count": count})
    return None


def _device_comm(ctx: Context):
    """libsqgr's RCCL communicator of the process group, if it lives on ``ctx`` (else the host side channel is used)."""
    if not _dist.is_distributed():
        return None
    comm = _dist.device_comm()
    return comm if comm is not None and comm.ctx is ctx else None


def _broadcast_seed(key: int) -> int:
    """All ranks must use rank 0's key when ``seed=None`` drew fresh entropy."""
    if not _dist.is_distributed():
        return key
    return int(_dist.broadcast_object(int(key), src=0))


RENUMBER_MIN_OBS = 32_768      # observations from which rng="philox" looks at their order (squidpy_amd/_order.py) ...
RENUMBER_MIN_PERMS = 256       # ... and permutations: the twin graph costs a few ms
RENUMBER_NEAR = 0.2            # fraction of the edges whose endpoints lie within 8 positions of each other below which ...
RENUMBER_SPAN = 0.02           # ... or mean |row - col| / n above which the order counts as "not spatial"
RENUMBER_RCM_PERMS = 50_000    # without coordinates: permutations from which a reverse Cuthill-McKee order on the host (~0.1 s per 1e6) pays


def _internal_order(ctx: Context, adata: Any, adj: Any, n_cls: int, lib_codes: Any, n_perms: int) -> np.ndarray | None:
    """``order[new] = old`` for the plan-internal renumbering of ``rng="philox"``, or ``None``: the observations are in a spatial
    order already (grids in scan order, cells listed tile by tile), the call is small, or the plan cannot carry a spot map
    (libraries, more than 256 clusters).  ``SQGR_NHOOD_RENUMBER=0`` switches it off, ``=1`` forces it where admissible.
    tools/spot_order_time.py: random order 240 k permutations/s, fields of view of 100 x 100 cells in random order inside 448 k,
    renumbered 875 k (scan order: 900 k) at 1e6 spots."""
    import os

    from .._lib import spatial_order_device
    from .._order import edge_locality, spatial_order

    mode = os.environ.get("SQGR_NHOOD_RENUMBER", "auto")
    if mode == "0" or lib_codes is not None or n_cls > 256:
        return None
    n = adata.n_obs
    if mode != "1":
        if n < RENUMBER_MIN_OBS or n_perms < RENUMBER_MIN_PERMS:
            return None
        near, span = edge_locality(adj)
        if near >= RENUMBER_NEAR and span <= RENUMBER_SPAN:
            return None
    xy = None
    try:
        xy = np.asarray(adata.obsm[Key.obsm.spatial])
        if xy.ndim != 2 or xy.shape[0] != n or xy.shape[1] < 2 or not np.issubdtype(xy.dtype, np.number):
            xy = None
    except Exception:
        xy = None
    if xy is not None:
        try:
            return spatial_order_device(ctx, xy)
        except Exception:  # coordinates that are not finite, ...: the graph alone decides below
            pass
    if mode == "1" or n_perms >= RENUMBER_RCM_PERMS:
        return spatial_order(adj).astype(np.int32)
    return None


DEFAULT_STREAM_NOTICE_WORK = 5_000_000_000   # n_obs * n_perms from which a DEFAULTED `rng` (numpy's streams, ~11x slower than "philox") is pointed out
HOST_GATHER_NOTICE_ENTRIES = 64_000_000      # n_perms * K * K from which the host gather of several ranks' per-permutation counts is pointed out
PROGRESS_STEP = 40_960  # permutations per progress update: 16 launch groups of 2560
MAX_DEVICE_SHUFFLE_CLUSTERS = 4096  # batched permutation kernels: uint8 labels + LDS counters up to 256 clusters, uint16 labels + device-scope
# counters up to 4096 (K*K*16 counters per batch: 1 GiB); beyond that the any-K edge-pair kernel counts host-drawn numpy shuffles


def _zscore_many_clusters(
    ctx: Context,
    graph: Graph,
    int_clust: np.ndarray,
    n_cls: int,
    lib_codes: np.ndarray | None,
    n_libs: int,
    seed: int | None,
    n_perms: int,
    count: np.ndarray,
) -> np.ndarray:
    """More than 4096 clusters: the batched device shuffle does not apply (K*K*16 counters per pass), so each
    permutation is drawn with the reference's own numpy stream on the host (gr/_nhood.py:213, 530-539) and counted by the
    general edge-pair kernel (`sqgr_nhood_counts`, any K) — Squidpy's z-scores for the seed, at ~1 ms per permutation
    plus the shuffle.  Permutation ranges are split over ranks like the other paths."""
    logg.info("`%s` clusters > %s: label shuffles are drawn with numpy on the host for this call", n_cls, MAX_DEVICE_SHUFFLE_CLUSTERS)
    if seed is None:
        seed = _broadcast_seed(resolve_seed(None))
    rank, world = _dist.world()
    lo, hi = _dist.shard_range(n_perms, rank, world)
    gens = spawn_generators(seed, n_perms)[lo:hi]
    lib_idx = [np.where(lib_codes == c)[0] for c in range(n_libs)] if lib_codes is not None else None
    perms = np.empty((hi - lo, n_cls, n_cls), dtype=np.float64)
    for q, rs in enumerate(gens):
        lab = int_clust.copy()
        if lib_idx is None:
            rs.shuffle(lab)
        else:  # gr/_utils.py:185-213
            for idx in lib_idx:
                grp = int_clust[idx].copy()
                rs.shuffle(grp)
                lab[idx] = grp
        perms[q] = nhood_counts(ctx, graph, lab, n_cls)
    perms = np.concatenate(_dist.allgather_object(perms), axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (count - perms.mean(axis=0)) / perms.std(axis=0)


def _zscore_numpy_streams(
    ctx: Context,
    graph: Graph,
    int_clust: np.ndarray,
    n_cls: int,
    lib_codes: np.ndarray | None,
    n_libs: int,
    seed: int | None,
    n_perms: int,
    count: np.ndarray,
    chunk_bytes: int = 1 << 28,
) -> np.ndarray:
    """``rng="numpy"``: the reference's streams (gr/_nhood.py:213, 530-539) drawn on the host, counted on
    the GPU, and reduced with the reference's own float64 ``mean``/``std`` (gr/_nhood.py:231)."""
    gens = spawn_generators(seed, n_perms)
    n = len(int_clust)
    base = int_clust.astype(np.uint8)
    per = max(1, min(n_perms, chunk_bytes // max(n, 1)))
    perms = np.empty((n_perms, n_cls, n_cls), dtype=np.float64)
    lib_idx = [np.where(lib_codes == c)[0] for c in range(n_libs)] if lib_codes is not None else None
    for p0 in range(0, n_perms, per):
        p1 = min(n_perms, p0 + per)
        lab = np.empty((p1 - p0, n), dtype=np.uint8)
        for k, ix in enumerate(range(p0, p1)):
            r = gens[ix]
            if lib_idx is None:
                lab[k] = base
                r.shuffle(lab[k])
            else:  # gr/_utils.py:185-213
                for idx in lib_idx:
                    grp = base[idx].copy()
                    r.shuffle(grp)
                    lab[k, idx] = grp
        perms[p0:p1] = nhood_counts_batch(ctx, graph, lab, n_cls)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (count - perms.mean(axis=0)) / perms.std(axis=0)


def interaction_matrix(
    adata: Any,
    cluster_key: str,
    connectivity_key: str | None = None,
    normalized: bool = False,
    copy: bool = False,
    weights: bool = False,
    *,
    table_key: str | None = None,
    device: int | None = None,
) -> np.ndarray | None:
    """Compute interaction matrix for clusters (drop-in for ``squidpy.gr.interaction_matrix``,
    gr/_nhood.py:349-429): spots with a NaN category are masked out, edges summed by label pair."""
    adata = extract_adata_if_sdata(adata, table_key=table_key)
    connectivity_key = Key.obsp.spatial_conn(connectivity_key)
    _assert_categorical_obs(adata, cluster_key)
    _assert_connectivity_key(adata, connectivity_key)

    cats = adata.obs[cluster_key]
    codes = cats.cat.codes.to_numpy().astype(np.int32)  # -1 for NaN: masked on the device
    if not (codes >= 0).any():
        raise RuntimeError(f"After removing NaNs in `adata.obs[{cluster_key!r}]`, none remain.")
    g = adata.obsp[connectivity_key]
    n_cats = len(cats.cat.categories)
    is_int = pd.api.types.is_bool_dtype(g.dtype) or pd.api.types.is_integer_dtype(g.dtype)

    ctx = default_context(device)
    out = _intmat(ctx, cached_graph(ctx, g, with_data=weights), codes, n_cats, weights)
    output = out.astype(int) if is_int else out
    if normalized:
        with np.errstate(divide="ignore", invalid="ignore"):
            output = output / output.sum(axis=1).reshape((-1, 1))
    if copy:
        return output
    _save_data(adata, attr="uns", key=Key.uns.interaction_matrix(cluster_key), data=output)
    return None
