"""Worker of tests/test_dist_gloo.py: runs under ``python -m torch.distributed.run`` with the gloo backend (CPU), or —
``SQGR_TEST_GROUP=socket`` — as plain subprocesses with RANK / WORLD_SIZE / MASTER_PORT in the environment and NO torch:
the product's own socket rendezvous (squidpy_amd._dist.SocketGroup).

Exercises the multi-GPU plumbing of the product (squidpy_amd._dist: permutation-range sharding, exact integer
all-reduce, feature-block merge) with per-rank partial results computed by the CPU oracle standing in for the HIP
kernels (this is test code: the product itself never touches oracle/)."""

from __future__ import annotations

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    socket_mode = os.environ.get("SQGR_TEST_GROUP") == "socket"
    from squidpy_amd import _dist

    if socket_mode:
        os.environ["SQGR_DIST_COLLECTIVE"] = "host"  # no GPU here: the integer all-reduce goes through the hub
        _dist.init("socket")
    else:
        import torch.distributed as dist

        dist.init_process_group(backend="gloo")
    from oracle import restate as O
    from squidpy_amd.gr._nhood import _broadcast_seed, expected_counts, zscore_from_moments
    from squidpy_amd.gr._ppatterns import _merge_blocks

    rank, world = _dist.world()
    assert world == int(os.environ["WORLD_SIZE"]) and _dist.is_distributed()
    out = {"rank": rank, "world": world}

    # ---- nhood: permutation ranges + all-reduce of exact moments == single-rank result
    rows, cols, k, P, seed = 20, 25, 4, 37, 5
    adj = O.hex_grid_graph(rows, cols)
    labels = np.random.default_rng(0).integers(0, k, rows * cols).astype(np.int32)
    shift = expected_counts(labels, k, adj.nnz)
    lo, hi = _dist.shard_range(P, rank, world)
    mine = O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, seed, lo, hi).astype(np.int64)
    d = mine - shift
    s1, s2 = d.sum(0), (d * d).sum(0).astype(np.uint64)
    s1, s2 = _dist.allreduce_sum_([s1, s2])
    full = O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, seed, 0, P)
    dfull = full.astype(np.int64) - shift
    assert np.array_equal(s1, dfull.sum(0)) and np.array_equal(s2, (dfull * dfull).sum(0).astype(np.uint64))
    count = O.nhood_counts(adj.indices, adj.indptr, labels, k)
    z = zscore_from_moments(count, shift, s1, s2, P)
    np.testing.assert_allclose(z, O.nhood_zscore(count, full), rtol=1e-10)
    out["nhood_z00"] = float(z[0, 0])

    # ---- BASELINE config 5's strong-scaling arithmetic: 100 000 permutations over the ranks — contiguous, disjoint, complete,
    # sizes within one of each other; config 4's 49 radius intervals likewise
    for total in (100_000, 49, 3):
        ranges = [_dist.shard_range(total, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    (tot_p,) = _dist.allreduce_sum_([np.array([_dist.shard_range(100_000, rank, world)[1] - _dist.shard_range(100_000, rank, world)[0]], dtype=np.int64)])
    assert int(tot_p[0]) == 100_000

    # ---- seed=None: every rank ends up with rank 0's key
    key = _broadcast_seed(1000 + rank)
    assert key == 1000
    # uint64 wrap-around is preserved by the int64 view
    big = np.array([2**63 + 5], dtype=np.uint64)
    (tot,) = _dist.allreduce_sum_([big])
    assert int(tot[0]) == (world * (2**63 + 5)) % 2**64

    # ---- co-occurrence: row-tile shards sum to the full counts (counts of disjoint point subsets as stand-in)
    rng = np.random.default_rng(1)
    x, y = (rng.random(300) * 50).astype(np.float32), (rng.random(300) * 50).astype(np.float32)
    labs = rng.integers(0, 3, 300).astype(np.int32)
    thr = np.linspace(2, 40, 6, dtype=np.float32) ** 2
    fullc = O.occur_count(x, y, thr, labs, 3)
    # ordered pairs (i, j) with i in this rank's slice
    sl = slice(*_dist.shard_range(300, rank, world))
    part = np.zeros_like(fullc)
    dx = x[sl, None] - x[None, :]
    dy = y[sl, None] - y[None, :]
    d2 = dx * dx + dy * dy
    ii = np.arange(300)[sl]
    for r, t in enumerate(thr):
        m = (d2 <= t) & (ii[:, None] != np.arange(300)[None, :])
        np.add.at(part[:, :, r], (labs[sl][:, None].repeat(300, 1)[m], labs[None, :].repeat(len(ii), 0)[m]), 1)
    (summed,) = _dist.allreduce_sum_([part])
    assert np.array_equal(summed, fullc)

    # ---- co-occurrence, the other shard axis (radius-interval batches per rank, all pairs): a rank's cumulative counts need
    # its own thresholds only; zero elsewhere, the all-reduce assembles the full array
    lo, hi = _dist.shard_range(len(thr), rank, world)
    part_iv = np.zeros_like(fullc)
    if hi > lo:
        part_iv[:, :, lo:hi] = O.occur_count(x, y, thr[lo:hi], labs, 3)
    (summed_iv,) = _dist.allreduce_sum_([part_iv])
    assert np.array_equal(summed_iv, fullc)

    # ---- autocorr: contiguous runs of feature blocks per rank (a rank's features are one column range), merged by gather
    from squidpy_amd.gr._ppatterns import _block_owner

    G = 10
    blocks = [(b0, min(G, b0 + 3)) for b0 in range(0, G, 3)]
    owners = [_block_owner(bi, len(blocks), world) for bi in range(len(blocks))]
    assert owners == sorted(owners) and set(owners) <= set(range(world)) and (world > len(blocks) or set(owners) == set(range(world)))
    score = np.full(G, np.nan)
    sims = np.full((4, G), np.nan)
    for bi, (b0, b1) in enumerate(blocks):
        if owners[bi] == rank:
            score[b0:b1] = np.arange(b0, b1)
            sims[:, b0:b1] = np.arange(b0, b1)[None, :] + 100 * np.arange(4)[:, None]
    score = _merge_blocks(score, blocks, world, axis=0)
    sims = _merge_blocks(sims, blocks, world, axis=1)
    assert np.array_equal(score, np.arange(G)) and np.array_equal(sims[2], np.arange(G) + 200)

    out["collective"] = _dist.collective_kind()
    assert out["collective"] == ("socket-hub" if socket_mode else "torch.distributed")
    if socket_mode:
        # without a GPU the RCCL communicator cannot be created: every rank must learn that and agree on the host side channel
        os.environ.pop("SQGR_DIST_COLLECTIVE", None)
        _dist._comm_tried = False
        assert _dist.device_comm() is None
        (again,) = _dist.allreduce_sum_([np.array([rank + 1], dtype=np.int64)])
        assert int(again[0]) == world * (world + 1) // 2 and _dist.collective_kind() == "socket-hub"
    assert _dist.broadcast_object({"r": rank}, src=world - 1) == {"r": world - 1}
    _dist.barrier()
    if socket_mode:
        assert "torch" not in sys.modules, "the socket side channel must not import torch"
    if rank == 0:
        print("DIST_OK " + json.dumps(out))
    if socket_mode:
        _dist.shutdown()
    else:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
