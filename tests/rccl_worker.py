"""Worker of tests/test_rccl_gpu.py: one rank, backend nccl (= RCCL): the product's collective path on real hardware."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    os.environ["SQGR_DIST_FORCE"] = "1"
    import squidpy_amd as sq
    from oracle import restate as O
    from squidpy_amd import _dist
    from tests.helpers import codes, hex_adata

    assert _dist.is_distributed() and dist.get_backend() == "nccl"
    a = np.array([[1, -2], [3, 2**40]], dtype=np.int64)
    b = np.array([2**63 + 7, 5], dtype=np.uint64)
    ra, rb = _dist.allreduce_sum_([a, b])
    assert np.array_equal(ra, a) and np.array_equal(rb, b) and rb.dtype == np.uint64
    adata = hex_adata(30, 40, 5, seed=2)
    adj = adata.obsp["spatial_connectivities"]
    lab = codes(adata, "cluster")
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=70, seed=3, copy=True)
    ref = O.nhood_perm_counts_philox(adj.indices, adj.indptr, lab, 5, 3, 0, 70)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(res.counts, ref), rtol=1e-9)
    occ, _ = sq.gr.co_occurrence(adata, "cluster", interval=8, copy=True)
    occ_ref, _ = O.co_occurrence(adata.obsm["spatial"], lab, interval=8)
    np.testing.assert_allclose(occ, occ_ref, rtol=1e-12)
    res_np = sq.gr.nhood_enrichment(adata, "cluster", n_perms=20, seed=None, copy=True, rng="numpy")  # seed broadcast path
    assert np.isfinite(res_np.zscore).all()
    dist.barrier()
    print("RCCL_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
