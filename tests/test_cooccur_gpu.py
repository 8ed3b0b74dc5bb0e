"""GPU parity tests of co_occurrence: pair counts bit-exact vs the oracle / the reference kernel's golden output."""

from __future__ import annotations

import numpy as np
import pandas as pd
import pytest

from oracle import restate as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


@pytest.mark.parametrize("name", ["lattice", "jitter"])
def test_counts_match_reference_kernel_golden(L, ctx, golden, name):
    """Output of the reference's literal `_occur_count` (lattice data puts many pairs exactly on thresholds)."""
    xy = golden[f"cooc_{name}_xy"].astype(np.float32)
    labs = golden[f"cooc_{name}_labs"]
    interval = golden[f"cooc_{name}_interval"]
    counts = L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], labs, 4, interval[1:] ** 2)
    np.testing.assert_array_equal(counts, golden[f"cooc_{name}_counts"])


@pytest.mark.parametrize("n,k,l", [(1, 2, 3), (2, 2, 1), (255, 3, 7), (256, 1, 5), (257, 5, 49), (1500, 7, 20), (3000, 30, 49)])
def test_counts_bit_exact_vs_oracle(L, ctx, n, k, l):
    rng = np.random.default_rng(n + k)
    x = (rng.random(n) * 1000).astype(np.float32)
    y = (rng.random(n) * 700).astype(np.float32)
    labs = rng.integers(0, k, n).astype(np.int32)
    thr = (np.linspace(5, 600, l, dtype=np.float32)) ** 2
    got = L.cooccur_counts(ctx, x, y, labs, k, thr)
    np.testing.assert_array_equal(got, O.occur_count(x, y, thr, labs, k))
    # sharded sweeps (multi-GPU partition) sum to the same counts
    parts = sum(L.cooccur_counts(ctx, x, y, labs, k, thr, shard_index=s, shard_count=3) for s in range(3))
    np.testing.assert_array_equal(parts, got)


def test_edge_cases(L, ctx):
    rng = np.random.default_rng(5)
    n, k = 700, 6
    x = rng.integers(0, 12, n).astype(np.float32)  # heavy duplication: many d2 == 0 and exact ties
    y = rng.integers(0, 12, n).astype(np.float32)
    labs = rng.integers(0, k - 2, n).astype(np.int32)  # two empty categories
    thr = np.array([25.0, 0.0, 4.0, 4.0, 1e9, 2.0, np.inf], dtype=np.float32)  # unsorted, duplicates, zero, inf
    got = L.cooccur_counts(ctx, x, y, labs, k, thr)
    np.testing.assert_array_equal(got, O.occur_count(x, y, thr, labs, k))
    assert got[k - 1].sum() == 0 and got[:, k - 1].sum() == 0
    assert got[..., 6].sum() == n * (n - 1)  # every ordered pair is within an infinite radius
    # fma only changes decisions for pairs sitting on a threshold; on generic data it agrees
    xr, yr = (rng.random(n) * 50).astype(np.float32), (rng.random(n) * 50).astype(np.float32)
    t2 = np.linspace(1, 60, 9, dtype=np.float32) ** 2
    a = L.cooccur_counts(ctx, xr, yr, labs, k, t2)
    b = L.cooccur_counts(ctx, xr, yr, labs, k, t2, fma=True)
    assert np.abs(a - b).sum() <= 4


def test_frontend_matches_reference_pipeline(L, golden):
    import squidpy_amd as sq

    for name in ("lattice", "jitter"):
        xy = golden[f"cooc_{name}_xy"]
        labs = golden[f"cooc_{name}_labs"]
        adata = sq.AnnDataLite(
            obs=pd.DataFrame({"cl": pd.Categorical.from_codes(labs, ["a", "b", "c", "d"])}), obsm={"spatial": xy}
        )
        occ, interval = sq.gr.co_occurrence(adata, "cl", interval=12, copy=True)
        np.testing.assert_array_equal(interval, golden[f"cooc_{name}_interval"])
        np.testing.assert_allclose(occ, golden[f"cooc_{name}_occ"], rtol=1e-12, atol=0)  # float64 ratios of exact counts
        assert occ.dtype == np.float64 and occ.shape == (4, 4, 11)
    # reference tests/graph/test_ppatterns.py:169-207 ported
    assert sq.gr.co_occurrence(adata, "cl") is None
    slot = adata.uns["cl_co_occurrence"]
    assert set(slot) == {"occ", "interval"} and slot["occ"].ndim == 3 and slot["occ"].shape[2] == 49
    assert slot["interval"].dtype == np.float32 and len(slot["interval"]) == 50
    occ2, iv2 = sq.gr.co_occurrence(adata, "cl", interval=[30.0, 10.0, 20.0], copy=True)
    np.testing.assert_array_equal(iv2, np.array([10, 20, 30], dtype=np.float32))
    assert occ2.shape == (4, 4, 2)
    with pytest.raises(ValueError, match="Expected interval to be of length"):
        sq.gr.co_occurrence(adata, "cl", interval=[5.0])
    with pytest.warns(FutureWarning, match="deprecated"):
        sq.gr.co_occurrence(adata, "cl", n_jobs=2, copy=True)


def test_medium_size_invariants(L, ctx):
    """2e4 points (oracle too slow for a full compare): total ordered pairs within an infinite radius = N(N-1);
    counts are symmetric under (a,b) transposition; a sampled sub-block equals the oracle."""
    rng = np.random.default_rng(1)
    n, k = 20000, 8
    xy = O.hex_grid(100, 200) + rng.normal(0, 5, (n, 2))
    x, y = xy[:, 0].astype(np.float32), xy[:, 1].astype(np.float32)
    labs = rng.integers(0, k, n).astype(np.int32)
    thr = np.append(np.linspace(100, 9000, 30, dtype=np.float32) ** 2, np.float32(np.inf))
    got = L.cooccur_counts(ctx, x, y, labs, k, thr)
    assert got[..., -1].sum() == n * (n - 1)
    np.testing.assert_array_equal(got, np.transpose(got, (1, 0, 2)))
    assert (np.diff(got, axis=2) >= 0).all()
    sel = np.where(labs < 2)[0][:1500]
    sub = L.cooccur_counts(ctx, x[sel], y[sel], labs[sel], 2, thr)
    np.testing.assert_array_equal(sub, O.occur_count(x[sel], y[sel], thr, labs[sel], 2))


def test_many_thresholds_are_swept_in_chunks(L, ctx):
    """L = 300 radii exceed one LDS histogram: the library sweeps threshold chunks; counts stay bit-exact."""
    rng = np.random.default_rng(8)
    n, k = 600, 3
    x, y = (rng.random(n) * 90).astype(np.float32), (rng.random(n) * 90).astype(np.float32)
    labs = rng.integers(0, k, n).astype(np.int32)
    thr = rng.permutation(np.linspace(0.5, 120, 300, dtype=np.float32) ** 2)  # also unsorted
    np.testing.assert_array_equal(L.cooccur_counts(ctx, x, y, labs, k, thr), O.occur_count(x, y, thr, labs, k))


def test_interval_batches_reproduce_the_full_counts(L, ctx):
    """The other shard axis of config 4 (radius-interval batches, all pairs per batch): a cumulative count at threshold r
    needs only the thresholds of its own batch, so the slices of `sqgr_cooccur_counts` over threshold ranges ARE the
    corresponding slices of the full result — what `co_occurrence(..., shard="intervals")` relies on."""
    rng = np.random.default_rng(3)
    n, k = 5000, 7
    xy = (rng.random((n, 2)) * 300).astype(np.float32)
    xy[:1000] = np.round(xy[:1000])  # lattice part: exact ties with thresholds
    lab = rng.integers(0, k, n).astype(np.int32)
    thr = (np.linspace(1, 150, 23, dtype=np.float32)) ** 2
    thr[5] = np.float32(25.0)
    thr.sort()
    full = L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], lab, k, thr)
    for cuts in ((0, 8, 23), (0, 1, 2, 11, 22, 23)):
        parts = [L.cooccur_counts(ctx, xy[:, 0], xy[:, 1], lab, k, thr[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        np.testing.assert_array_equal(np.concatenate(parts, axis=2), full)


@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("n,k,l,rmax,near", [(30000, 4, 12, 40.0, True), (200000, 30, 49, 25.0, True), (40000, 3, 150, 60.0, True), (300000, 200, 9, 10.0, True),
                                              (20000, 30, 49, 25.0, False)])
def test_short_radii_skip_tile_pairs_and_change_no_count(L, ctx, n, k, l, rmax, near, fma, monkeypatch):
    """Explicit short radii (what users pass: a few spot diameters): clusters are tiled along a Hilbert curve and tile pairs whose
    bounding boxes lie beyond the largest threshold are skipped (csrc/sqgr_cooccur.hip: k_co_candidates) — exact.  The route the
    library picks by itself `==` the dense sweep (SQGR_COOCCUR_SPARSE=0) `==` the oracle; lattice coordinates put many pairs
    exactly on a threshold and exactly on a box edge.  `near` = False: so few points per cluster that a tile spans the cloud and
    most tile pairs stay — the library keeps the dense sweep.  /root/reference/src/squidpy/gr/_ppatterns.py:283-310, 407-413."""
    rng = np.random.default_rng(n + k)
    x = (rng.integers(0, 2000, n) * 0.5).astype(np.float32)   # half-integer lattice: exact ties with thresholds
    y = (rng.integers(0, 1400, n) * 0.5).astype(np.float32)
    labs = rng.integers(0, k, n).astype(np.int32)
    thr = np.linspace(0.5, rmax, l, dtype=np.float32) ** 2
    ctx.timer_enable(True)
    ctx.timer_reset()
    got = L.cooccur_counts(ctx, x, y, labs, k, thr, fma=fma)
    kern = ctx.timer_report()
    ctx.timer_enable(False)
    assert any(name.startswith("cooccur_pairs_near") and v[0] > 0 for name, v in kern.items()) == near, kern   # which route ran
    monkeypatch.setenv("SQGR_COOCCUR_SPARSE", "0")
    dense = L.cooccur_counts(ctx, x, y, labs, k, thr, fma=fma)
    np.testing.assert_array_equal(got, dense)
    if n <= 30000:
        np.testing.assert_array_equal(got, O.occur_count(x, y, thr, labs, k))   # (half-integer coordinates: every product is exact, fused or not)
    monkeypatch.delenv("SQGR_COOCCUR_SPARSE")
    parts = sum(L.cooccur_counts(ctx, x, y, labs, k, thr, fma=fma, shard_index=s, shard_count=3) for s in range(3))
    np.testing.assert_array_equal(parts, got)


def test_short_radii_forced_on_a_long_interval_and_degenerate_clouds(L, ctx, monkeypatch):
    """SQGR_COOCCUR_SPARSE=1 forces the candidate lists whatever the radius (every tile pair stays: the lists are then complete);
    a cloud on a line (zero extent in y) and one of identical points go through it too."""
    monkeypatch.setenv("SQGR_COOCCUR_SPARSE", "1")
    rng = np.random.default_rng(1)
    n, k = 5000, 5
    labs = rng.integers(0, k, n).astype(np.int32)
    for x, y in ((rng.random(n).astype(np.float32) * 100, rng.random(n).astype(np.float32) * 100),
                 (rng.random(n).astype(np.float32) * 100, np.zeros(n, np.float32)),
                 (np.full(n, 3.0, np.float32), np.full(n, 4.0, np.float32))):
        thr = np.linspace(1, 150, 20, dtype=np.float32) ** 2
        got = L.cooccur_counts(ctx, x, y, labs, k, thr)
        np.testing.assert_array_equal(got, O.occur_count(x, y, thr, labs, k))
