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
print(f"fuzz ok: {it} iterations in {time.time()-t0:.0f}s")
