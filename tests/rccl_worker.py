"""Worker of tests/test_rccl_gpu.py: one rank, backend nccl (= RCCL): the product's collective path on real hardware."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    socket_mode = os.environ.get("SQGR_TEST_GROUP") == "socket"  # no torch at all: the product's own rendezvous
    os.environ["SQGR_DIST_FORCE"] = "1"
    import squidpy_amd as sq
    from oracle import restate as O
    from squidpy_amd import _dist, _lib
    from tests.helpers import codes, hex_adata

    if socket_mode:
        _dist._group = _dist.SocketGroup(0, 1)  # init() is a no-op for one process; a 1-rank group exercises the path
    else:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
    assert _dist.is_distributed()
    comm = _dist.device_comm()
    assert comm is not None and (comm.rank, comm.world) == (0, 1), "libsqgr's RCCL communicator was not created"
    a = np.array([[1, -2], [3, 2**40]], dtype=np.int64)
    b = np.array([2**63 + 7, 5], dtype=np.uint64)
    ra, rb = _dist.allreduce_sum_([a, b])
    assert np.array_equal(ra, a) and np.array_equal(rb, b) and rb.dtype == np.uint64
    assert _dist.collective_kind() == "rccl-in-library"
    comm.barrier()
    m = np.array([5, -3], dtype=np.int64)
    assert np.array_equal(comm.allreduce_i64(m.copy(), op=_lib.Comm.MAX), m)
    adata = hex_adata(30, 40, 5, seed=2)
    adj = adata.obsp["spatial_connectivities"]
    lab = codes(adata, "cluster")
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=70, seed=3, copy=True, rng="philox")
    ref = O.nhood_perm_counts_philox(adj.indices, adj.indptr, lab, 5, 3, 0, 70)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(res.counts, ref), rtol=1e-9)
    occ, _ = sq.gr.co_occurrence(adata, "cluster", interval=8, copy=True)
    occ_ref, _ = O.co_occurrence(adata.obsm["spatial"], lab, interval=8)
    np.testing.assert_allclose(occ, occ_ref, rtol=1e-12)
    res_np = sq.gr.nhood_enrichment(adata, "cluster", n_perms=20, seed=None, copy=True, rng="numpy")  # seed broadcast path
    assert np.isfinite(res_np.zscore).all()
    # numpy streams with the communicator attached: counts all-gathered on the device, Squidpy's z-scores exactly
    res_np = sq.gr.nhood_enrichment(adata, "cluster", n_perms=33, seed=4, copy=True)
    ref_np = O.nhood_perm_counts_numpy(adj.indices, adj.indptr, lab, 5, 4, 33)
    np.testing.assert_array_equal(res_np.zscore, O.nhood_zscore(res_np.counts, ref_np))
    # the plan-level device all-reduce returns what the plain run returns (one rank: identity)
    ctx = _lib.default_context()
    g = _lib.Graph(ctx, adj, with_data=False)
    plan = _lib.NhoodPlan(ctx, g, lab, 5)
    s1, s2, _ = plan.run(9, 3, 60)
    plan.set_comm(comm)
    t1, t2, _ = plan.run(9, 3, 60)
    assert np.array_equal(s1, t1) and np.array_equal(s2, t2)
    plan.close(); g.close()
    _dist.barrier()
    if socket_mode:
        assert "torch" not in sys.modules
        _dist.shutdown()
    else:
        dist.destroy_process_group()
    print("RCCL_OK")


if __name__ == "__main__":
    main()
