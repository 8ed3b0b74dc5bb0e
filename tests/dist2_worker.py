"""Worker of tests/test_dist2_gpu.py: TWO ranks sharing one GPU (backend gloo for the collectives — RCCL refuses two
ranks on one device): every sharded front end must return, on every rank, exactly what a single process returns."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    import pandas as pd
    import scipy.sparse as sp
    import squidpy_amd as sq
    from oracle import restate as O
    from squidpy_amd import _dist
    from tests.helpers import codes, hex_adata

    assert _dist.is_distributed() and _dist.world() == (rank, 2)
    adata = hex_adata(30, 40, 5, seed=2, n_genes=150)
    adj = adata.obsp["spatial_connectivities"]
    lab = codes(adata, "cluster")
    # nhood: permutation ranges + all-reduce of the integer moments; the device generator is keyed by the global index
    res = sq.gr.nhood_enrichment(adata, "cluster", n_perms=75, seed=3, copy=True, rng="philox")
    ref = O.nhood_perm_counts_philox(adj.indices, adj.indptr, lab, 5, 3, 0, 75)
    np.testing.assert_allclose(res.zscore, O.nhood_zscore(res.counts, ref), rtol=1e-9)
    res_np = sq.gr.nhood_enrichment(adata, "cluster", n_perms=31, seed=4, copy=True)
    ref_np = O.nhood_perm_counts_numpy(adj.indices, adj.indptr, lab, 5, 4, 31)
    np.testing.assert_array_equal(res_np.zscore, O.nhood_zscore(res_np.counts, ref_np))
    # co-occurrence: row tiles t % world == rank, all-reduce of the pair counts
    occ, _ = sq.gr.co_occurrence(adata, "cluster", interval=8, copy=True)
    occ_ref, _ = O.co_occurrence(adata.obsm["spatial"], lab, interval=8)
    np.testing.assert_allclose(occ, occ_ref, rtol=1e-12)
    # the other shard axis (BASELINE north star: radius-interval batches per rank): identical result
    occ_iv, _ = sq.gr.co_occurrence(adata, "cluster", interval=8, copy=True, shard="intervals")
    np.testing.assert_array_equal(occ_iv, occ)
    # autocorr: contiguous runs of feature blocks per rank (3 blocks over 2 ranks), each rank uploads its columns only; gathered
    df = sq.gr.spatial_autocorr(adata, mode="geary", n_perms=12, seed=5, copy=True, gene_block=64)
    want = O.spatial_autocorr(adj, adata.X.T, adata.var_names, mode="geary", n_perms=12, seed=5)
    for c in df.columns:
        np.testing.assert_allclose(df[c].to_numpy(), want[c].to_numpy(), rtol=1e-6, atol=1e-12, err_msg=c)
    ad_sp = adata.copy()
    ad_sp.X = sp.csr_matrix(np.asarray(adata.X, dtype=np.float32))  # sparse float32 expression, resident per rank
    df_sp = sq.gr.spatial_autocorr(ad_sp, mode="geary", n_perms=12, seed=5, copy=True, rng="numpy", gene_block=64)
    want32 = O.spatial_autocorr(adj, np.asarray(adata.X, dtype=np.float32).astype(np.float64).T, adata.var_names, mode="geary", n_perms=12, seed=5)
    for c in df_sp.columns:
        np.testing.assert_allclose(df_sp[c].to_numpy(), want32[c].to_numpy(), rtol=1e-6, atol=1e-12, err_msg=c)
    # ligrec: permutation ranges, all-reduce of the indicator counts
    genes = list(adata.var_names[:6])
    inter = [(a, b) for a in genes for b in genes if a != b]
    lr = sq.gr.ligrec(adata, "cluster", interactions=inter, n_perms=33, seed=6, use_raw=False, copy=True, threshold=0.1)
    gi = {g: i for i, g in enumerate(genes)}
    _, pv = O.ligrec_analysis(np.asarray(adata.X)[:, :6], lab, np.array([(gi[a], gi[b]) for a, b in inter]),
                              np.array([(a, b) for a in range(5) for b in range(5)]), threshold=0.1, n_perms=33, seed=6)
    assert np.array_equal(lr["pvalues"].to_numpy(dtype=np.float64), pv, equal_nan=True)
    # ripley: clusters spread over the ranks by cost, simulations round-robin, rows gathered from their owners
    for mode in ("L", "F", "G"):
        rp = sq.gr.ripley(adata, "cluster", mode=mode, n_simulations=5, n_observations=60, n_steps=10, seed=2, copy=True)
        rref = O.ripley(adata.obsm["spatial"], adata.obs["cluster"].values, mode=mode, n_simulations=5, n_observations=60, n_steps=10, seed=2)
        np.testing.assert_allclose(rp[f"{mode}_stat"]["stats"].to_numpy().reshape(5, 10), rref["obs"], rtol=1e-12)
        np.testing.assert_allclose(rp["sims_stat"]["stats"].to_numpy().reshape(5, 10), rref["sims"], rtol=1e-12)
        assert np.array_equal(rp["pvalues"], rref["pvalues"]), mode
    dist.barrier()
    print(f"DIST2_OK rank {rank}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
