"""Randomised differential test of libsqgr against the CPU oracle (run on the GPU box):

    python tools/fuzz_gpu.py [seconds] [seed]            # time budget
    FUZZ_ITERS=12 python tools/fuzz_gpu.py 0 7           # fixed number of iterations (tests/test_fuzz_gpu.py)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from oracle import devrng, restate as O
from squidpy_amd import _lib as L
from squidpy_amd._utils import pcg64_states

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = L.default_context()
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed0)
print('fuzz seed', seed0, flush=True)
VERBOSE = os.environ.get('FUZZ_VERBOSE') == '1'
def note(*a):
    if VERBOSE: print(*a, flush=True)
ITERS = int(os.environ.get('FUZZ_ITERS', '0'))
t0 = time.time(); it = 0
while (it < ITERS) if ITERS else (time.time() - t0 < budget):
    it += 1
    n = int(rng.choice([3, 17, 64, 255, 257, 1000, 4097, 20000]))
    k = int(rng.choice([2, 3, 7, 30, 50, 51, 64, 100, 127, 203, 256, 300, 1000]))
    dens = rng.choice([0.0, 2.0, 6.0, 20.0]) / max(n, 1)
    A = sp.random(n, n, density=min(1.0, dens), format="csr", random_state=int(rng.integers(1 << 31)))
    shape = rng.random()
    if shape < 0.45: A = A + A.T                                   # structurally symmetric: the half-list path of the count kernel
    if rng.random() < 0.3: A = A + sp.identity(n, format="csr")     # ... with self loops (weights 2 / 1, halved sum)
    A = sp.csr_matrix(A); A.sort_indices()
    if shape > 0.9 and A.nnz > 4: A.indices[A.indptr[1]:A.indptr[2]] = A.indices[A.indptr[1]:A.indptr[2]][::-1]  # unsorted row: not canonical
    A.data = (rng.random(A.nnz) + 0.5).astype(np.float32 if rng.random() < 0.5 else np.float64)
    labels = rng.integers(0, k, n).astype(np.int32)
    if rng.random() < 0.3: labels[:] = rng.integers(0, max(1, k // 3), n)  # empty categories
    note('iter', it, 'nhood n', n, 'k', k, 'nnz', A.nnz)
    g = L.Graph(ctx, A)
    assert np.array_equal(L.nhood_counts(ctx, g, labels, k), O.nhood_counts(A.indices, A.indptr, labels, k)), ("counts", n, k)
    use_libs = rng.random() < 0.4
    nl = int(rng.integers(1, 5)); libs = rng.integers(0, nl, n).astype(np.int32) if use_libs else None
    note('  plan libs', use_libs, nl)
    cats = labels.copy(); cats[rng.random(n) < 0.1] = -1
    if (cats >= 0).any():  # weighted / unweighted interaction matrix with masked spots (float64 accumulation on the device)
        keep = cats >= 0; sub = A[keep][:, keep].tocsr()
        np.testing.assert_allclose(L.interaction_matrix(ctx, g, cats, k, True), O.interaction_matrix(sub.data.astype(np.float64), sub.indices, sub.indptr, cats[keep], k, True), rtol=1e-12, atol=1e-300)
        assert np.array_equal(L.interaction_matrix(ctx, g, cats, k, False), O.interaction_matrix(sub.data, sub.indices, sub.indptr, cats[keep], k, False))
    plan = L.NhoodPlan(ctx, g, labels, k, libs, nl if use_libs else 0)
    plan.tune(int(rng.choice([16, 32])) if k <= 256 else 16, int(rng.choice([0, 8, 256])), int(rng.choice([1, 3, 32])))
    P = int(rng.integers(1, 70)); seed = int(rng.integers(1 << 62)); lo = int(rng.integers(0, 1 << 40))
    note('  run P', P, 'seed', seed, 'lo', lo)
    s1, s2, perms = plan.run(seed, lo, lo + P, None, return_perms=True)
    ref = O.nhood_perm_counts_philox(A.indices, A.indptr, labels, k, seed, lo, lo + P, libs, nl if use_libs else 0)
    assert np.array_equal(perms, ref.astype(np.uint32)), ("philox", n, k, use_libs)
    assert np.array_equal(s1, ref.astype(np.int64).sum(0))
    note('  pcg')
    if n <= 4097 and not (k > 256 and use_libs):  # (numpy streams + libraries + 16-bit labels: host route in the front end)
        _, _, pp = plan.run_pcg64(pcg64_states(seed % 1000, P), return_perms=True)
        refn = O.nhood_perm_counts_numpy(A.indices, A.indptr, labels, k, seed % 1000, P, libs, nl if use_libs else 0)
        assert np.array_equal(pp, refn.astype(np.uint32)), ("pcg64", n, k, use_libs)
    plan.close(); g.close()
    # co-occurrence + ripley on small clouds
    m = int(rng.choice([2, 50, 300, 700])); kk = int(rng.choice([1, 2, 5]))
    note('  cooc m', m, 'kk', kk)
    x = np.round(rng.random(m) * 50, int(rng.integers(0, 3))).astype(np.float32); y = np.round(rng.random(m) * 50, 1).astype(np.float32)
    labs = rng.integers(0, kk, m).astype(np.int32)
    thr = np.sort(rng.random(int(rng.integers(1, 30))) * 60).astype(np.float32) ** 2
    assert np.array_equal(L.cooccur_counts(ctx, x, y, labs, kk, thr), O.occur_count(x, y, thr, labs, kk)), ("cooc", m, kk)
    # ... and short radii on a cloud large enough for the candidate-list route (round 6): near route == forced dense sweep, three shards
    mm = int(rng.choice([17000, 40000, 90000])); kq = int(rng.choice([1, 3, 12, 80])); ext = float(rng.choice([100.0, 3000.0]))
    xq = (rng.random(mm) * ext).astype(np.float32); yq = (rng.random(mm) * ext * rng.choice([0.02, 0.7, 1.0])).astype(np.float32)
    if rng.random() < 0.3: xq = np.round(xq)                          # ties with thresholds and box edges
    lq = rng.integers(0, kq, mm).astype(np.int32)
    tq = (np.sort(rng.random(int(rng.integers(1, 60)))) * ext * rng.choice([0.005, 0.03, 0.15])).astype(np.float32) ** 2
    fq = bool(rng.random() < 0.3)
    note('  cooc short mm', mm, 'kq', kq, 'L', len(tq), 'fma', fq)
    near = L.cooccur_counts(ctx, xq, yq, lq, kq, tq, fma=fq)
    os.environ["SQGR_COOCCUR_SPARSE"] = "0"
    dense = L.cooccur_counts(ctx, xq, yq, lq, kq, tq, fma=fq)
    os.environ.pop("SQGR_COOCCUR_SPARSE")
    assert np.array_equal(near, dense), ("cooc short radii", mm, kq, len(tq), fq)
    ns = int(rng.integers(2, 5))
    assert np.array_equal(sum(L.cooccur_counts(ctx, xq, yq, lq, kq, tq, fma=fq, shard_index=r, shard_count=ns) for r in range(ns)), near), ("cooc short shards", mm, ns)
    note('  pairs')
    pts = np.stack([x, y], 1).astype(np.float64); sup = np.linspace(0, 40, int(rng.integers(2, 40)))
    assert np.array_equal(L.pair_counts(ctx, pts, sup), O.pair_counts_bruteforce(pts, sup)), ("pairs", m)
    # numpy permutation streams at a random size
    nn = int(rng.integers(2, 9000)); Pn = int(rng.integers(1, 40)); sd = int(rng.integers(1 << 30))
    note('  perm nn', nn, 'Pn', Pn, 'sd', sd)
    assert np.array_equal(L.pcg64_permutations(ctx, nn, pcg64_states(sd, Pn)), O.autocorr_perm_indices(nn, sd, Pn)), ("perm", nn, sd)
    # ligrec: random sparse expression, both generators
    nc = int(rng.choice([40, 333, 2000])); ng = int(rng.integers(2, 30)); kc = int(rng.choice([2, 3, 9, 40, 90]))
    data = (rng.random((nc, ng)) < rng.choice([0.05, 0.3, 1.0])) * (np.rint(rng.gamma(2, 2, (nc, ng))) if rng.random() < 0.5 else rng.gamma(2, 1, (nc, ng)))
    kc = min(kc, nc)
    cl = rng.integers(0, kc, nc).astype(np.int32); cl[:kc] = np.arange(kc)  # every cluster populated
    inter = rng.integers(0, ng, (int(rng.integers(1, 60)), 2)).astype(np.int32)
    cp = rng.integers(0, kc, (int(rng.integers(1, 300)), 2)).astype(np.int32)
    note('  ligrec nc', nc, 'ng', ng, 'kc', kc, 'inter', len(inter), 'cp', len(cp))
    pre = O.ligrec_prepare(data, cl, inter, cp, float(rng.choice([0.0, 0.1, 0.5])))
    Pl = int(rng.integers(1, 80)); sd = int(rng.integers(1 << 30))
    got = L.ligrec_counts(ctx, sp.csc_matrix(data), cl, kc, pre["inv_counts"], inter, cp, pre["obs"], pre["valid"].astype(np.uint8),
                          pcg_states=pcg64_states(sd, Pl), perm_begin=0, perm_end=Pl)
    want = O.ligrec_score_permutations(data, O.ligrec_perm_labels_numpy(cl, sd, Pl), pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    assert np.array_equal(got, want), ("ligrec numpy", nc, ng, kc)
    lo = int(rng.integers(0, 1000))
    got = L.ligrec_counts(ctx, sp.csc_matrix(data), cl, kc, pre["inv_counts"], inter, cp, pre["obs"], pre["valid"].astype(np.uint8),
                          seed=sd, perm_begin=lo, perm_end=lo + Pl)
    want = O.ligrec_score_permutations(data, O.ligrec_perm_labels_philox(cl, sd, lo, lo + Pl), pre["inv_counts"], pre["mean_obs"], inter, cp, pre["valid"])
    assert np.array_equal(got, want), ("ligrec philox", nc, ng, kc)
    # ---- round-3 surfaces
    # pair counts of several point sets in one launch (empty and single-point sets included), the three KDTree metrics
    metric = str(rng.choice(["euclidean", "manhattan", "chebyshev"]))
    sets = [np.round(rng.random((int(rng.choice([0, 1, 2, 65, 300, 1500])), 2)) * rng.choice([1.0, 50.0]), int(rng.integers(0, 4))) for _ in range(int(rng.integers(1, 6)))]
    note('  pair batch', [len(a) for a in sets], metric)
    got = L.pair_counts_batch(ctx, sets, sup, metric)
    for a, row in zip(sets, got):
        if len(a) >= 2:
            from sklearn.neighbors import KDTree
            assert np.array_equal(row, KDTree(a, metric=metric).two_point_correlation(a, sup) - len(a)), ("pair batch", len(a), metric)  # gr/_ripley.py:220-222
        else:
            assert np.array_equal(row, np.zeros(len(sup), dtype=np.int64)), ("pair batch tiny", len(a))
    # k nearest neighbours: brute force below 512 reference points, the cell list above; ties from rounded coordinates
    from sklearn.neighbors import NearestNeighbors
    nr = int(rng.choice([1, 5, 511, 512, 3000, 20000])); nq = int(rng.choice([1, 64, 1000])); kq = int(rng.integers(1, min(nr, 16) + 1))
    metric = str(rng.choice(["euclidean", "manhattan", "chebyshev", "canberra"]))
    refs = rng.random((nr, 2)) * np.array([rng.choice([1.0, 1000.0]), rng.choice([1.0, 1000.0])])
    if rng.random() < 0.3: refs = np.round(refs, 1)
    if rng.random() < 0.3: refs[:, 0] = refs[0, 0]  # degenerate: all on a line
    qs = rng.random((nq, 2)) * refs.max(0) * 1.2 - 0.1
    note('  knn nr', nr, 'nq', nq, 'k', kq, metric)
    want = NearestNeighbors(n_neighbors=kq, metric=metric, algorithm="brute" if metric == "canberra" else "kd_tree").fit(refs).kneighbors(qs)[0]
    got = L.knn_dist(ctx, qs, refs, kq, metric)
    if metric == "euclidean":
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=1e-300)
    edges = np.linspace(0, float(want.max()) * 1.1 + 1e-9, int(rng.integers(3, 40)))
    qlab = rng.integers(0, 3, nq).astype(np.int32); ex = int(rng.integers(-1, 3))
    dp = L.DevicePoints(ctx, qs, qlab)
    hist = dp.knn_hist(refs, kq, edges, metric, ex)
    keep = qlab != ex
    assert hist.sum() <= keep.sum() * kq
    if metric != "euclidean":  # (euclidean distances are compared with sklearn's at 1e-12: a histogram could differ at an edge)
        assert np.array_equal(hist, np.histogram(got[keep], bins=edges)[0]), ("knn hist", nr, nq, kq, metric)
    dp.close()
    # co-occurrence: a batch of the radius thresholds on its own gives the same (cumulative) counts; row-tile shards add up
    if len(thr) >= 2:
        full = L.cooccur_counts(ctx, x, y, labs, kk, thr)
        cut = int(rng.integers(1, len(thr)))
        assert np.array_equal(L.cooccur_counts(ctx, x, y, labs, kk, thr[cut:]), full[:, :, cut:]), ("cooc interval batch", m, cut)
        ns = int(rng.integers(2, 5))
        assert np.array_equal(sum(L.cooccur_counts(ctx, x, y, labs, kk, thr, shard_index=r, shard_count=ns) for r in range(ns)), full), ("cooc tiles", m, ns)
    # spatial autocorrelation: expression formats x column lists, scores and permutation statistics
    na = int(rng.choice([5, 64, 300, 1500])); Ga = int(rng.choice([1, 2, 63, 64, 65, 130]))
    W = sp.random(na, na, density=min(1.0, 6.0 / na), format="csr", random_state=int(rng.integers(1 << 31))); W = sp.csr_matrix(W + W.T); W.sort_indices()
    if W.nnz == 0: W = sp.csr_matrix(np.eye(na, k=1) + np.eye(na, k=-1))
    W.data = rng.random(W.nnz) + 0.1
    Xa = np.where(rng.random((na, Ga)) < rng.choice([0.1, 0.6, 1.0]), np.rint(rng.gamma(2, 3, (na, Ga))), 0.0)
    fmt = str(rng.choice(["dense64", "dense32", "csr32", "csr64", "csc32", "csc64", "pitch"])); ity = rng.choice([np.int32, np.int64])
    if fmt.startswith("dense"): Xd = Xa.astype(np.float64 if fmt == "dense64" else np.float32)
    elif fmt == "pitch": Xd = np.hstack([Xa, Xa])[:, :Ga]  # a column range of a wider row-major array
    else:
        Xd = (sp.csr_matrix if fmt.startswith("csr") else sp.csc_matrix)(Xa.astype(np.float32 if fmt.endswith("32") else np.float64))
        Xd = type(Xd)((Xd.data, Xd.indices.astype(ity), Xd.indptr.astype(ity)), shape=Xd.shape)
    cols = rng.integers(0, Ga, int(rng.integers(1, 2 * Ga + 1))).astype(np.int32) if rng.random() < 0.6 else None
    note('  autocorr n', na, 'G', Ga, fmt, 'cols', None if cols is None else len(cols))
    ga = L.Graph(ctx, W, with_data=True); dm = L.DeviceMatrix(ctx, Xd)
    plan = L.AutocorrPlan.from_column_list(ctx, ga, dm, cols) if cols is not None else L.AutocorrPlan.from_columns(ctx, ga, dm, 0, Ga)
    sel = Xa[:, cols] if cols is not None else Xa
    ref_plan = L.AutocorrPlan(ctx, ga, np.ascontiguousarray(sel.T))
    for mode in ("moran", "geary"):
        sc = plan.scores(mode)
        assert np.array_equal(sc, ref_plan.scores(mode), equal_nan=True), ("autocorr formats", fmt, mode)
        want = (O.morans_i if mode == "moran" else O.gearys_c)(W, sel.T)
        ok = ~np.isnan(want)
        np.testing.assert_allclose(sc[ok], want[ok], rtol=1e-9, atol=1e-12)
        Pa = int(rng.integers(1, 40)); sd = int(rng.integers(1 << 30))
        sims = plan.perms_pcg64(mode, pcg64_states(sd, Pa))
        red = plan.perm_stats(mode, sc, pcg_states=pcg64_states(sd, Pa), only_feature=sims.shape[1] == 1)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(red["n_ge"], (sims >= sc).sum(0)), ("perm_stats n_ge", mode)
            for key, val in (("sum", sims.sum(0)), ("std", sims.std(0)), ("var", np.var(sims, 0))):
                assert np.array_equal(red[key], val, equal_nan=True), ("perm_stats", key, mode)
        wantp = O.score_perms(mode, W, sel.T, O.autocorr_perm_indices(na, sd, Pa))
        np.testing.assert_allclose(sims[:, ok], wantp[:, ok], rtol=1e-8, atol=1e-11)
    plan.close(); ref_plan.close(); dm.close(); ga.close()
print(f"fuzz ok: {it} iterations in {time.time()-t0:.0f}s")
