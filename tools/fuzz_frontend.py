"""Randomised differential test of the `squidpy_amd.gr` FRONT ENDS (options, value extraction, formats, result frames) against
the oracle's restatement of the reference pipelines, in the reference's own random streams (rng="numpy").  Run on the GPU box:

    python tools/fuzz_frontend.py [seconds] [seed]
    FUZZ_ITERS=6 python tools/fuzz_frontend.py 0 3
    FUZZ_BIG=1 python tools/fuzz_frontend.py 600 5       # fewer, larger cases (the CPU oracle needs ~1 min for each)
"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, scipy.sparse as sp
from oracle import restate as O
import squidpy_amd as sq

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed0)
print("fuzz_frontend seed", seed0, flush=True)
VERBOSE = os.environ.get("FUZZ_VERBOSE") == "1"
ITERS = int(os.environ.get("FUZZ_ITERS", "0"))
def note(*a):
    if VERBOSE: print(*a, flush=True)
warnings.simplefilter("ignore")

def knn(xy, k):
    from sklearn.neighbors import NearestNeighbors
    k = min(k, len(xy) - 1)
    idx = NearestNeighbors(n_neighbors=k + 1).fit(xy).kneighbors(xy, return_distance=False)[:, 1:]
    n = len(xy)
    a = sp.csr_matrix((np.ones(n * k), (np.repeat(np.arange(n), k), idx.ravel())), shape=(n, n))
    a = ((a + a.T) > 0).astype(np.float64 if rng.random() < 0.5 else np.float32)
    a = sp.csr_matrix(a); a.sort_indices()
    return a

t0 = time.time(); it = 0
while (it < ITERS) if ITERS else (time.time() - t0 < budget):
    it += 1
    if os.environ.get("FUZZ_BIG") == "1":  # fewer, larger cases: several feature blocks, tiles, launch groups
        n = int(rng.choice([3000, 9000, 20000])); G = int(rng.choice([3, 70, 300])); K = int(rng.choice([2, 9, 31]))
    else:
        n = int(rng.choice([12, 60, 300, 1200])); G = int(rng.choice([1, 2, 9, 40])); K = int(rng.choice([2, 3, 7]))
    xy = rng.random((n, 2)) * rng.choice([1.0, 100.0, 5000.0])
    if rng.random() < 0.3: xy = np.round(xy, 1)
    X = np.where(rng.random((n, G)) < rng.choice([0.2, 1.0]), np.rint(rng.gamma(2.0, 3.0, (n, G))), 0.0) + (rng.random((n, G)) < 0.02)
    cl = rng.integers(0, K, n); cl[:K] = np.arange(K)
    obs = pd.DataFrame({"cl": pd.Categorical([f"c{v}" for v in cl], categories=[f"c{v}" for v in range(K)]),
                        "lib": pd.Categorical([f"l{v}" for v in rng.integers(0, 2, n)]), "num": rng.random(n), "cnt": rng.integers(0, 9, n)})
    var = pd.DataFrame({"highly_variable": rng.random(G) < 0.6}, index=[f"g{i}" for i in range(G)])
    if not var["highly_variable"].any(): var.iloc[0, 0] = True
    fmt = str(rng.choice(["dense64", "dense32", "csr32", "csc64"]))
    Xs = {"dense64": X, "dense32": X.astype(np.float32), "csr32": sp.csr_matrix(X.astype(np.float32)), "csc64": sp.csc_matrix(X)}[fmt]
    conn = knn(xy, int(rng.choice([3, 6])))
    adata = sq.AnnDataLite(X=Xs, obs=obs, var=var, obsm={"spatial": xy, "feat": rng.random((n, 3))}, obsp={"spatial_connectivities": conn},
                           layers={"lay": X * 2.0})
    labels = cl.astype(np.int32)

    # ---- nhood_enrichment in numpy's streams: the z-scores are the reference's for the seed
    P = int(rng.integers(2, 60)); sd = int(rng.integers(0, 1 << 31)); use_lib = rng.random() < 0.4
    note("iter", it, "n", n, "G", G, "K", K, fmt, "| nhood P", P, "lib", use_lib)
    res = sq.gr.nhood_enrichment(adata, "cl", library_key="lib" if use_lib else None, n_perms=P, seed=sd, copy=True, rng="numpy", show_progress_bar=False)
    libs = obs["lib"].cat.codes.to_numpy().astype(np.int32) if use_lib else None
    cnt = O.nhood_counts(conn.indices, conn.indptr, labels, K)
    perms = O.nhood_perm_counts_numpy(conn.indices, conn.indptr, labels, K, sd, P, libs, 2 if use_lib else 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.array_equal(res.counts, cnt) and np.array_equal(res.zscore, O.nhood_zscore(cnt, perms), equal_nan=True), ("nhood", n, K, P)

    # ---- nhood_enrichment with the device generator: against its restatement, on the caller's graph and (observations of a random
    # cloud come in no spatial order) on the renumbered twin — the same moments, bit for bit; also the directed kNN graph
    if rng.random() < 0.5:
        from squidpy_amd._synthetic import knn_directed_graph
        gconn = conn if rng.random() < 0.5 else knn_directed_graph(xy, min(int(rng.choice([3, 6])), n - 1))
        ad2 = sq.AnnDataLite(obs=obs, obsm={"spatial": xy}, obsp={"spatial_connectivities": gconn})
        zs = []
        for ren in ("0", "1"):
            os.environ["SQGR_NHOOD_RENUMBER"] = ren
            zs.append(sq.gr.nhood_enrichment(ad2, "cl", n_perms=P, seed=sd, copy=True, rng="philox", show_progress_bar=False))
        os.environ.pop("SQGR_NHOOD_RENUMBER", None)
        cnt2 = O.nhood_counts(gconn.indices, gconn.indptr, labels, K)
        want = O.nhood_zscore(cnt2, O.nhood_perm_counts_philox(gconn.indices, gconn.indptr, labels, K, sd, 0, P))
        ok = np.isfinite(want)
        assert np.array_equal(zs[0].counts, cnt2) and np.array_equal(zs[0].zscore, zs[1].zscore, equal_nan=True), ("nhood philox twin", n, K, P)
        assert np.allclose(zs[0].zscore[ok], want[ok], rtol=1e-9), ("nhood philox", n, K, P)

    # ---- spatial_autocorr
    mode = str(rng.choice(["moran", "geary"])); two = bool(rng.random() < 0.5); trans = bool(rng.random() < 0.7)
    corr = "fdr_bh" if rng.random() < 0.7 else None; Pa = None if rng.random() < 0.3 else int(rng.integers(1, 50))
    pick = rng.random(); attr = "X"; layer = None; genes = None
    if pick < 0.25: genes = None; vals = X[:, var["highly_variable"].to_numpy()].T; index = var.index[var["highly_variable"].to_numpy()]
    elif pick < 0.5:
        genes = [str(v) for v in rng.choice(var.index, int(rng.integers(1, G + 1)), replace=False)]; vals = X[:, [int(v[1:]) for v in genes]].T; index = genes
        if rng.random() < 0.5: layer = "lay"; vals = vals * 2.0
    elif pick < 0.6: genes = str(rng.choice(var.index)); vals = X[:, [int(genes[1:])]].T; index = [genes]
    elif pick < 0.8: attr = "obs"; genes = ["num", "cnt"][: int(rng.integers(1, 3))]; vals = obs[genes].to_numpy(dtype=np.float64).T; index = genes
    else: attr = "obsm"; layer = "feat"; genes = [int(v) for v in rng.choice(3, int(rng.integers(1, 4)), replace=False)]; vals = adata.obsm["feat"][:, genes].T; index = genes
    note("  autocorr", mode, attr, genes if genes is None or len(str(genes)) < 60 else "...", "layer", layer, "P", Pa, "two", two, "trans", trans, corr)
    df = sq.gr.spatial_autocorr(adata, mode=mode, genes=genes, attr=attr, layer=layer, transformation=trans, n_perms=Pa, two_tailed=two, corr_method=corr,
                                seed=sd, copy=True, rng="numpy", show_progress_bar=False, gene_block=int(rng.choice([1, 3, 2048])))
    ref = O.spatial_autocorr(conn, np.asarray(vals, dtype=np.float64), index, mode=mode, transformation=trans, n_perms=Pa, two_tailed=two, corr_method=corr, seed=sd)
    assert list(df.columns) == list(ref.columns), (list(df.columns), list(ref.columns))
    assert sorted(map(str, df.index)) == sorted(map(str, ref.index))
    ref = ref.loc[df.index]
    stat = "I" if mode == "moran" else "C"
    np.testing.assert_allclose(df[stat].to_numpy(), ref[stat].to_numpy(), rtol=1e-9, atol=1e-12, err_msg=stat)
    assert (np.diff(df[stat].to_numpy()[~np.isnan(df[stat].to_numpy())]) * (1 if mode == "geary" else -1) >= 0).all()  # sorted like the reference
    near = np.zeros(len(df))  # permutation scores within rounding of the observed one (frequent on tiny graphs): `>=` may go either way
    if Pa is not None:
        from sklearn.preprocessing import normalize
        gn = sp.csr_matrix(conn).copy()
        if trans: normalize(gn, norm="l1", axis=1, copy=False)
        sims = pd.DataFrame(O.score_perms(mode, gn, np.asarray(vals, dtype=np.float64), O.autocorr_perm_indices(n, sd, Pa)), columns=list(index))
        sc = ref[stat].to_numpy()
        with np.errstate(invalid="ignore"):
            near = (np.abs(sims[list(df.index)].to_numpy() - sc) <= 1e-9 * np.maximum(1.0, np.abs(sc))).sum(0)
    for c in df.columns:
        a, b = df[c].to_numpy(dtype=np.float64), ref[c].to_numpy(dtype=np.float64)
        if c == "pval_sim":
            assert (np.abs(a - b) <= (near + 1e-9) / (Pa + 1))[~np.isnan(b)].all() and np.array_equal(np.isnan(a), np.isnan(b)), c
        elif c.startswith("pval_sim"):
            if not near.any(): np.testing.assert_allclose(a, b, rtol=1e-12, atol=0, err_msg=c)
        elif c.startswith("pval_z_sim"):
            # z = 0/0 (every permutation score equal to the observed one, e.g. a single permutation that ties): the reference leaves
            # the entry of its `np.empty` array unwritten (gr/_ppatterns.py:481-484) — undefined there, NaN here
            # (and a spread of the permutation scores that is rounding noise — all of them tie — gives z = +-inf or 0/0 by chance)
            undefined = (ref["var_sim"].to_numpy() <= 1e-18 * np.maximum(1.0, ref[stat].to_numpy() ** 2)) & ~np.isnan(df[stat].to_numpy())
            if not undefined.any(): np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9, err_msg=c)
            elif "fdr" not in c: np.testing.assert_allclose(a[~undefined], b[~undefined], rtol=2e-6, atol=1e-9, err_msg=c)
        else:
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-9, err_msg=c)

    # ---- co_occurrence
    interval = int(rng.integers(2, 30)) if rng.random() < 0.6 else sorted(set(np.round(rng.random(int(rng.integers(2, 12))) * xy.max(), 3)))
    if not isinstance(interval, int) and len(interval) < 2: interval = 5
    note("  cooc interval", interval if isinstance(interval, int) else len(interval))
    occ, iv = sq.gr.co_occurrence(adata, "cl", interval=interval, copy=True)
    rocc, riv = O.co_occurrence(xy, labels, interval)
    assert np.array_equal(iv, riv)
    np.testing.assert_allclose(occ, rocc, rtol=1e-6, atol=1e-7)

    # ---- ripley
    rmode = str(rng.choice(["F", "G", "L"])); metric = str(rng.choice(["euclidean", "manhattan", "chebyshev"]))
    kw = dict(n_neigh=int(rng.integers(1, 4)), n_simulations=int(rng.integers(1, 8)), n_observations=int(rng.choice([5, 40, 300])),
              max_dist=None if rng.random() < 0.5 else float(rng.random() * xy.max()), n_steps=int(rng.choice([2, 11, 50])), seed=sd)
    if min(np.bincount(cl, minlength=K)) <= kw["n_neigh"] or kw["n_observations"] <= kw["n_neigh"]: kw["n_neigh"] = 1
    do_ripley = min(np.bincount(cl, minlength=K)) >= 2
    note("  ripley", rmode, metric, kw, do_ripley)
    res = sq.gr.ripley(adata, "cl", mode=rmode, metric=metric, copy=True, **kw) if do_ripley else None
    if do_ripley and (metric == "euclidean" or rmode != "L"):  # (the oracle's L is written for the euclidean KDTree)
        ref = O.ripley(xy, obs["cl"].to_numpy(), mode=rmode, metric=metric, **kw)
        assert np.array_equal(res["bins"], ref["bins"])
        ns = kw["n_steps"]
        got_obs = res[f"{rmode}_stat"]["stats"].to_numpy().reshape(K, ns); got_sims = res["sims_stat"]["stats"].to_numpy().reshape(kw["n_simulations"], ns)
        np.testing.assert_allclose(got_obs, ref["obs"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(got_sims, ref["sims"], rtol=1e-12, atol=1e-12)
        assert np.array_equal(res["pvalues"], ref["pvalues"])
    # ---- graph builders ("next" row f-3) against the restated reference builders (sklearn's KD tree as the reference calls it)
    kind = str(rng.choice(["knn", "radius", "grid", "delaunay"]))
    # (grid: a lattice whose coordination number is the n_neighs asked for — with ties at the k-th distance sklearn's choice depends on
    #  its tree traversal; libsqgr's documented policy is the smaller index)
    grid_k = int(rng.choice([4, 6])); gr_, gc_ = int(rng.integers(3, 9)), int(rng.integers(3, 12))
    lattice = O.hex_grid(gr_, gc_) if grid_k == 6 else np.stack(np.meshgrid(np.arange(float(gc_)), np.arange(float(gr_))), -1).reshape(-1, 2) * 10.0
    pts = rng.random((n, 2)) * rng.choice([1.0, 300.0]) if kind != "grid" else lattice + rng.normal(0, 0.5, (1, 2))  # (continuous: no ties)
    span = float(np.ptp(pts, axis=0).max()) + 1e-9
    bkw = dict(set_diag=bool(rng.random() < 0.4), transform=[None, "spectral", "cosine"][int(rng.integers(0, 3))])
    # (k < n/2: from there on sklearn answers by brute force with inexact distances, and a percentile threshold can land on either side)
    if kind == "knn": bkw.update(n_neighs=int(rng.integers(1, max(1, min((len(pts) - 1) // 2, 9)) + 1)), percentile=None if rng.random() < 0.6 else float(rng.choice([50.0, 90.0, 99.0])))
    elif kind == "radius":
        r1 = span * float(rng.choice([0.05, 0.15, 0.4]))
        bkw.update(radius=r1 if rng.random() < 0.5 else (r1 * 0.3, r1), percentile=None if rng.random() < 0.7 else 80.0)
    elif kind == "grid": bkw.update(n_neighs=grid_k, n_rings=int(rng.integers(1, 4)), delaunay=bool(rng.random() < 0.3))
    else: bkw.update(radius=None if rng.random() < 0.5 else (0.0, span * 0.3))
    note("  graph", kind, bkw)
    gdata = sq.AnnDataLite(X=np.ones((len(pts), 2)), obsm={"spatial": pts})
    fn = {"knn": sq.gr.spatial_neighbors_knn, "radius": sq.gr.spatial_neighbors_radius, "grid": sq.gr.spatial_neighbors_grid, "delaunay": sq.gr.spatial_neighbors_delaunay}[kind]
    got = fn(gdata, copy=True, **bkw)
    radj, rdst = O.spatial_graph(pts, kind, **bkw)
    assert got.connectivities.dtype == radj.dtype and got.distances.dtype == rdst.dtype, (kind, got.connectivities.dtype, radj.dtype)
    np.testing.assert_allclose(got.connectivities.toarray(), radj.toarray(), rtol=1e-6, atol=1e-7, err_msg=f"{kind} adj {bkw}")
    # (sklearn switches to brute force for k >= n/2 and forms those distances through the expanded |x|^2 - 2xy + |y|^2: ~1e-12 off)
    np.testing.assert_allclose(got.distances.toarray(), rdst.toarray(), rtol=1e-9, atol=0, err_msg=f"{kind} dst {bkw}")

    # ---- interaction_matrix ("next" row f-2)
    wts = bool(rng.random() < 0.5); nrm = bool(rng.random() < 0.5)
    wconn = conn.copy(); wconn.data = (rng.random(conn.nnz) + 0.5).astype(conn.dtype)
    adata.obsp["spatial_connectivities"] = wconn
    got = sq.gr.interaction_matrix(adata, "cl", weights=wts, normalized=nrm, copy=True)
    want = O.interaction_matrix(wconn.data, wconn.indices, wconn.indptr, labels, K, wts).astype(np.float64)
    if nrm:
        with np.errstate(divide="ignore", invalid="ignore"):
            want = want / want.sum(axis=1).reshape((-1, 1))
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), want, rtol=1e-12, atol=0, equal_nan=True)
    adata.obsp["spatial_connectivities"] = conn

    # ---- ligrec ("next" row f-4) in numpy's streams: p-values equal the reference's for the seed
    if G >= 2:
        ng = min(G, 6); Pl = int(rng.integers(1, 40))
        pairs_l = list(dict.fromkeys((int(a), int(b)) for a, b in rng.integers(0, ng, (int(rng.integers(1, 12)), 2))))
        thr_l = float(rng.choice([0.0, 0.1, 0.5]))
        given = [(f"G{a}", f"G{b}") for a, b in pairs_l]
        if len(pairs_l) == 2 or rng.random() < 0.3:  # (exactly two items would be read as (sources, targets), gr/_ligrec.py:184-185)
            given = {"source": [g[0] for g in given], "target": [g[1] for g in given]}
        Xl = X[:, :ng]  # the front end keeps the genes of the interactions; hand it exactly those columns' universe
        ldata = sq.AnnDataLite(X=sp.csr_matrix(Xl) if rng.random() < 0.5 else Xl, obs=obs, var=pd.DataFrame(index=[f"G{i}" for i in range(ng)]))
        note("  ligrec", len(pairs_l), "pairs P", Pl, "thr", thr_l)
        res = sq.gr.ligrec(ldata, "cl", interactions=given, threshold=thr_l, n_perms=Pl, seed=sd, use_raw=False, copy=True,
                           rng="numpy", show_progress_bar=False)
        # the reference sorts the cluster PAIRS as tuples of strings (gr/_ligrec.py: `clusters = sorted(...)`) and codes the clusters
        # in category order
        names = [f"c{v}" for v in range(K)]
        cpairs = np.array([(names.index(a), names.index(b)) for a, b in sorted((a, b) for a in names for b in names)], dtype=np.int32)
        assert res["pvalues"].columns.to_list() == sorted((a, b) for a in names for b in names)
        means, pv = O.ligrec_analysis(Xl, labels, np.array(pairs_l, dtype=np.int32), cpairs, threshold=thr_l, n_perms=Pl, seed=sd)
        got_pv = res["pvalues"].to_numpy(dtype=np.float64); got_means = res["means"].to_numpy(dtype=np.float64)
        assert got_pv.shape == pv.shape, (got_pv.shape, pv.shape)
        if not np.array_equal(got_pv, pv, equal_nan=True):
            print("ligrec mismatch", n, ng, K, Pl, thr_l, pairs_l, "\ngot", got_pv, "\nwant", pv, "\nmeans got", got_means, "\nmeans want", means, "\nindex", res["pvalues"].index.to_list(), "\ncols", res["pvalues"].columns.to_list(), "\nX", Xl.tolist(), "\nlabels", labels.tolist(), flush=True)
            raise AssertionError(("ligrec pvalues", n, ng, K, Pl))
        np.testing.assert_allclose(got_means, means, rtol=1e-12, atol=0)
print(f"fuzz_frontend ok: {it} iterations in {time.time()-t0:.0f}s")
