"""GPU soak tests of the count kernels that use inline-asm LDS atomics (`k_count` at 16 permutations per pass, `k_count_pass`
at every width): the same plan, the same seed and the same permutation range, launched thousands of times — the moments must be
`array_equal` on EVERY launch and the counts of every launch must sum to nnz x permutations.

Why: the `ds_add_u32` of these kernels are inline asm, invisible to the compiler's wait-count pass; for
three rounds the flush barrier of the headline kernel was reached with atomics still in flight and lost one increment in ~5e5
cells now and then — every parity test ran each kernel a handful of times and stayed green.  The reference's counterpart is its
determinism tests (same seed -> same result, /root/reference/tests/graph/test_nhood.py:41-70).

Negative control (tools/soak_negative.sh, round 6; profiles/r06_soak_negative.txt): built with -DSQGR_DEBUG_NO_FLUSH_WAIT — the
explicit `s_waitcnt lgkmcnt(0)` in front of the flush barrier removed — the piled-up cases below FAIL for the
one-permutation-per-pass instantiations (6 of 27 tests; in most `k_count` instantiations the compiler's own wait for a scalar
load happens to sit in front of the barrier, so removing ours does not change their ISA), and all 27 pass on the product build."""

from __future__ import annotations

import os

import numpy as np
import pytest

from squidpy_amd._synthetic import hex_grid_graph

pytestmark = pytest.mark.gpu

LAUNCH_SCALE = float(os.environ.get("SQGR_SOAK_SCALE", "1"))  # tools/soak_negative.sh shortens the runs it expects to fail early


@pytest.fixture(scope="module")
def L():
    from squidpy_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


@pytest.fixture(scope="module")
def c5_graph(L, ctx):
    """BASELINE config 5's graph: 1e6 spots on the hex grid (nnz = 5 992 002), resident once for the module."""
    adj = hex_grid_graph(1000, 1000)
    return adj, L.Graph(ctx, adj, with_data=False)


def _soak(L, ctx, adj, g, k: int, width: int, perms: int, launches: int, labels: np.ndarray | None = None) -> dict:
    n, nnz = adj.shape[0], int(adj.nnz)
    if labels is None:
        labels = np.random.default_rng(k).integers(0, k, n).astype(np.int32)
    plan = L.NhoodPlan(ctx, g, labels, k)
    if width:
        plan.tune(width, 0, 0)
    info = plan.info()
    ref1, ref2, _ = plan.run(2024, 0, perms, None)
    assert int(ref1.sum()) == nnz * perms, "the first launch already lost (or invented) increments"
    bad = []
    launches = max(3, int(launches * LAUNCH_SCALE))
    for it in range(launches):
        s1, s2, _ = plan.run(2024, 0, perms, None)
        if not (np.array_equal(s1, ref1) and np.array_equal(s2, ref2)) or int(s1.sum()) != nnz * perms:
            bad.append((it, int(np.abs(s1 - ref1).sum()), int(s1.sum()) - nnz * perms))
            if len(bad) >= 5:
                break
    plan.close()
    assert not bad, f"K={k} width={width or 'auto'}: launches that differ from the first one (launch, |d sum|, lost increments): {bad} of {launches}; {info}"
    return info


def test_soak_headline_kernel_k30_at_config5_shape(L, ctx, c5_graph):
    """`k_count<16, ..., DOT2>` on the half list: 2 000 launch groups of 2 560 permutations at 1e6 spots x 30 clusters (~5 s)."""
    adj, g = c5_graph
    info = _soak(L, ctx, adj, g, 30, 0, 2560, 2000)
    assert info["perms_per_pass"] == 16 and info["symmetric"]


@pytest.mark.parametrize("k,width", [(64, 0), (100, 0), (130, 0), (200, 0), (230, 0), (30, 8), (30, 4), (30, 2), (30, 1)])
def test_soak_pass_kernel_every_width_at_config5_shape(L, ctx, c5_graph, k, width):
    """`k_count_pass` at the width K selects (K = 64, 100, 130, 200; 230: split rows) and at every width forced onto K = 30:
    400 launch groups of 640 permutations each on the 1e6-spot half list."""
    adj, g = c5_graph
    _soak(L, ctx, adj, g, k, width, 640, 400)


def test_soak_full_list_and_self_loops(L, ctx):
    """The same kernels on a directed kNN-like full list and on a half list with self loops (weights 2 and 1, halved sums):
    2.5e5 spots, 600 launch groups of 640 permutations at K = 30 and K = 100."""
    import scipy.sparse as sp

    rng = np.random.default_rng(3)
    adj = hex_grid_graph(500, 500)
    n = adj.shape[0]
    with_self = sp.csr_matrix(adj + sp.identity(n, format="csr", dtype=np.float32))
    with_self.sort_indices()
    rows = np.repeat(np.arange(n), 6)
    directed = sp.csr_matrix((np.ones(rows.size, np.float32), (rows, (rows + rng.integers(1, 400, rows.size)) % n)), shape=(n, n))
    directed.sum_duplicates()
    directed.sort_indices()
    for adj_v in (with_self, directed):
        g = L.Graph(ctx, adj_v, with_data=False)
        for k in (30, 100):
            _soak(L, ctx, adj_v, g, k, 0, 640, 600)
        g.close()


@pytest.mark.parametrize("self_loops", [False, True])
@pytest.mark.parametrize("k,width", [(30, 0), (30, 8), (30, 4), (30, 2), (30, 1), (100, 0), (200, 0), (230, 0)])
def test_soak_piled_up_counters(L, ctx, k, width, self_loops):
    """The input that keeps the LDS pipe busiest at the flush barrier: EVERY spot in one cluster, so that all 64 lanes of every
    `ds_add_u32` hit the same few counters and serialise — the atomics of the last iterations are still queued when the first
    wavefronts arrive at the barrier, and the counters they are queued for are the first ones the flush reads.  These are the
    cases the negative control (tools/soak_negative.sh) fails on without the explicit wait."""
    import scipy.sparse as sp

    adj = hex_grid_graph(300, 300)
    n = adj.shape[0]
    if self_loops:
        adj = sp.csr_matrix(adj + sp.identity(n, format="csr", dtype=np.float32))
        adj.sort_indices()
    g = L.Graph(ctx, adj, with_data=False)
    _soak(L, ctx, adj, g, k, width, 320, 150, labels=np.zeros(n, dtype=np.int32))
    g.close()


def test_numpy_stream_replay_is_the_same_on_every_launch(L, ctx, c5_graph, monkeypatch):
    """Round 6's numpy-stream kernels (k_pcg_draws_bucketed2 + k_pcg_apply_claims) resolve contested swaps through LDS atomics whose
    arrival order differs from launch to launch (which record of a chunk claims a position first; which of two deferred records
    reaches the queue first): the rows must not.  1024 permutations of config 5's 1e6 positions, six launches — the moments of
    the counts `array_equal` on every launch and equal to those of rounds 4-5's kernels (hashed-tag replay, 64-draw generator),
    which the stream tests pin to numpy's own shuffles."""
    from squidpy_amd._utils import pcg64_states

    adj, g = c5_graph
    n, k, P = adj.shape[0], 30, 1024
    labels = np.random.default_rng(5).integers(0, k, n).astype(np.int32)
    states = pcg64_states(77, P)
    monkeypatch.setenv("SQGR_PCG_APPLY", "tags")
    monkeypatch.setenv("SQGR_PCG_DRAWS", "64")
    plan = L.NhoodPlan(ctx, g, labels, k)
    ref1, ref2, _ = plan.run_pcg64(states)
    plan.close()
    assert int(ref1.sum()) == int(adj.nnz) * P
    monkeypatch.delenv("SQGR_PCG_APPLY")
    monkeypatch.delenv("SQGR_PCG_DRAWS")
    plan = L.NhoodPlan(ctx, g, labels, k)
    for it in range(6):
        s1, s2, _ = plan.run_pcg64(states)
        np.testing.assert_array_equal(s1, ref1, err_msg=f"launch {it}")
        np.testing.assert_array_equal(s2, ref2, err_msg=f"launch {it}")
    plan.close()
