"""Generate the golden vectors in this directory FROM THE REFERENCE'S OWN SOURCE.

Run once in the build container (``/root/reference`` must exist):

    python tests/golden/make_golden.py                      # needs only python3.10 + numpy/scipy/sklearn
    /opt/conda/bin/python3.9 tests/golden/make_golden.py --export-h5ad   # needs h5py (conda python)

The reference's kernel functions are executed literally through ``oracle/ref_shim.py`` (AST
extraction + numba stub); nothing here is computed by ``oracle/restate.py`` except the two
third-party pieces the reference does not contain (scanpy's Moran/Geary, flagged
"parity unpinned") — those outputs are stored under ``unpinned_*`` keys.

The committed ``*.npz`` files are what travels to the GPU box; this script does not.
"""

from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def export_h5ad() -> None:
    """tests/_data/test_data.h5ad (reference fixture ``adata``; tests/conftest.py:40-41,85-87)."""
    import h5py

    f = h5py.File("/root/reference/tests/_data/test_data.h5ad", "r")
    n, g = f["X"].attrs["shape"]
    data, indices, indptr = f["X/data"][:], f["X/indices"][:], f["X/indptr"][:]
    X = np.zeros((n, g), dtype=np.float32)
    for i in range(n):
        X[i, indices[indptr[i] : indptr[i + 1]]] = data[indptr[i] : indptr[i + 1]]
    hv = f["var/highly_variable"][:]
    np.savez_compressed(
        os.path.join(HERE, "visium49.npz"),
        spatial=f["obsm/spatial"][:],
        leiden_codes=f["obs/leiden/codes"][:].astype(np.int8),
        leiden_categories=np.array([c.decode() if isinstance(c, bytes) else c for c in f["obs/leiden/categories"][:]]),
        X40=X[:, :40],
        highly_variable40=hv[:40],
        var_names40=np.array([v.decode() if isinstance(v, bytes) else v for v in f["var/_index"][:40]]),
    )
    print("wrote visium49.npz")
    export_ligrec_reference()


def export_ligrec_reference() -> None:
    """tests/_data/ligrec_pvalues_reference.h5ad — the ONE pinned numeric result the reference's tests hold next to the hot
    path (fixture tests/conftest.py:316-327, used by tests/graph/test_ligrec.py:346-360: ``ligrec(adata, "leiden",
    interactions=product(raw.var_names[:5], raw.var_names[:5]), n_perms=25, seed=42)`` on tests/_data/test_data.h5ad):
    X = p-values, layers/means = means, rows = (source, target) gene pairs, columns = (cluster_1, cluster_2)."""
    import h5py

    f = h5py.File("/root/reference/tests/_data/ligrec_pvalues_reference.h5ad", "r")

    def cat(group: str) -> np.ndarray:
        cats = np.array([c.decode() if isinstance(c, bytes) else c for c in f[f"{group}/categories"][:]])
        return cats[f[f"{group}/codes"][:]]

    np.savez_compressed(
        os.path.join(HERE, "ligrec_pvalues_reference.npz"),
        pvalues=f["X"][:],
        means=f["layers/means"][:],
        source=cat("obs/source"),
        target=cat("obs/target"),
        cluster_1=cat("var/cluster_1"),
        cluster_2=cat("var/cluster_2"),
    )
    print("wrote ligrec_pvalues_reference.npz")


def main() -> None:
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp
    from sklearn.neighbors import NearestNeighbors

    from oracle import ref_shim as R
    from oracle import restate as O

    assert R.available(), "reference tree not mounted"
    out: dict[str, np.ndarray] = {}

    # ------------------------------------------------------------------ nhood (gr/_nhood.py)
    ns = R.nhood()
    ut = R.utils()
    rng = np.random.default_rng(20260924)
    n = 300
    pts = rng.random((n, 2)) * 100
    nbr = NearestNeighbors(n_neighbors=7).fit(pts).kneighbors(pts, return_distance=False)[:, 1:]
    g = sp.csr_matrix((np.ones(n * 6, np.float32), nbr.ravel(), np.arange(0, n * 6 + 1, 6)), shape=(n, n))
    k = 5
    labels = rng.integers(0, k, n).astype(np.uint32)
    indices, indptr = g.indices.astype(np.uint32), g.indptr.astype(np.uint32)
    fn = ns["create_function"](k)
    count = fn(indices, indptr, labels)
    n_perms, seed = 25, 42
    gens = ut["spawn_generators"](seed, n_perms)
    perms = ns["_nhood_enrichment_helper"](
        list(range(n_perms)), fn, indices, indptr, labels, None, k, gens
    )
    out.update(
        nhood_indices=g.indices.astype(np.int32),
        nhood_indptr=g.indptr.astype(np.int32),
        nhood_labels=labels,
        nhood_k=np.int64(k),
        nhood_count=count,
        nhood_seed=np.int64(seed),
        nhood_perms=perms,
        nhood_zscore=(count - perms.mean(axis=0)) / perms.std(axis=0),
    )
    # per-library shuffles (gr/_utils.py:185-213)
    import pandas as pd

    libs = pd.Series(pd.Categorical.from_codes(rng.integers(0, 3, n), ["l0", "l1", "l2"]))
    gens = ut["spawn_generators"](seed, n_perms)
    perms_lib = ns["_nhood_enrichment_helper"](list(range(n_perms)), fn, indices, indptr, labels, libs, k, gens)
    out.update(nhood_lib_codes=libs.cat.codes.to_numpy().astype(np.int32), nhood_perms_lib=perms_lib)

    # interaction matrix known answers: reference tests/graph/test_nhood.py:153-173 + conftest.py:177-194
    dense = np.array([[0, 1, 1, 0, 0], [0, 0, 0, 0, 1], [1, 2, 0, 0, 0], [0, 1, 0, 0, 1], [0, 0, 1, 2, 0]])
    gi = sp.csr_matrix(dense)
    cats = np.array([0, 0, 0, 1, 1])
    w = np.zeros((2, 2), dtype=int)
    ns["_interaction_matrix"](gi.data, gi.indices, gi.indptr, cats, w)
    u = np.zeros((2, 2), dtype=int)
    ns["_interaction_matrix"](np.broadcast_to(1, len(gi.data)), gi.indices, gi.indptr, cats, u)
    assert (w == [[5, 1], [2, 3]]).all() and (u == [[4, 1], [2, 2]]).all()  # the reference's KATs
    out.update(
        intmat_data=gi.data.astype(np.int64),
        intmat_indices=gi.indices.astype(np.int32),
        intmat_indptr=gi.indptr.astype(np.int32),
        intmat_cats=cats.astype(np.int32),
        intmat_weighted=w,
        intmat_unweighted=u,
    )

    # ------------------------------------------------------------------ co-occurrence (gr/_ppatterns.py)
    pp = R.ppatterns(O.morans_i, O.gearys_c)
    n2 = 260
    hexpts = O.hex_grid(13, 20, 100.0)  # lattice: many pairs exactly on thresholds
    jit = hexpts + rng.normal(0, 5, hexpts.shape)
    for name, xy in (("lattice", hexpts), ("jitter", jit)):
        xy32 = xy.astype(np.float32)
        labs = rng.integers(0, 4, n2).astype(np.int32)
        tmin, tmax = pp["_find_min_max"](xy32)
        interval = np.linspace(tmin, tmax, num=12, dtype=np.float32)
        thr = interval[1:] ** 2
        counts = pp["_occur_count"](xy32[:, 0], xy32[:, 1], thr, labs, n2, 4, len(thr))
        occ = pp["_co_occurrence_helper"](xy32[:, 0], xy32[:, 1], interval, labs)
        out.update(
            {
                f"cooc_{name}_xy": xy,
                f"cooc_{name}_labs": labs,
                f"cooc_{name}_interval": interval,
                f"cooc_{name}_counts": counts.astype(np.int64),
                f"cooc_{name}_occ": occ,
            }
        )

    # ------------------------------------------------------------------ autocorr p-values (gr/_ppatterns.py:443-559)
    ng, npm = 7, 30
    gw = sp.csr_matrix((np.ones(n * 6, np.float32), nbr.ravel(), np.arange(0, n * 6 + 1, 6)), shape=(n, n))
    from sklearn.preprocessing import normalize

    gw = normalize(gw, norm="l1", axis=1)
    vals = rng.gamma(2.0, 1.0, size=(ng, n))
    vals[1] += np.sin(pts[:, 0] / 10.0) * 2
    gens = ut["spawn_generators"](7, npm)
    perm_idx = np.stack([np.random.default_rng(s).permutation(n) for s in np.random.SeedSequence(7).spawn(npm)])
    for mode in ("moran", "geary"):
        func = O.morans_i if mode == "moran" else O.gearys_c
        score = func(gw, vals)
        gens = ut["spawn_generators"](7, npm)
        modeobj = pp["SpatialAutocorr"](mode)
        sims = pp["_score_helper"](list(range(npm)), modeobj, gw, vals, gens)
        params = {
            "mode": mode,
            "two_tailed": False,
            "expected": -1.0 / (n - 1) if mode == "moran" else 1.0,
        }
        with np.errstate(divide="ignore"):
            res = pp["_p_value_calc"](score, sims, gw, params)
        out.update(
            {
                f"unpinned_{mode}_score": score,
                f"unpinned_{mode}_sims": sims,
                f"autocorr_{mode}_pval_norm": res["pval_norm"],
                f"autocorr_{mode}_var_norm": np.float64(res["var_norm"]),
                f"autocorr_{mode}_pval_z_sim": res["pval_z_sim"],
                f"autocorr_{mode}_pval_sim": res["pval_sim"],
                f"autocorr_{mode}_var_sim": res["var_sim"],
            }
        )
    s0, s1, s2 = pp["_g_moments"](gw)
    out.update(
        autocorr_g_data=gw.data.astype(np.float32),
        autocorr_g_indices=gw.indices.astype(np.int32),
        autocorr_g_indptr=gw.indptr.astype(np.int32),
        autocorr_vals=vals,
        autocorr_perm_idx=perm_idx.astype(np.int64),
        autocorr_seed=np.int64(7),
        autocorr_moments=np.array([s0, s1, s2], dtype=np.float64),
    )

    # ------------------------------------------------------------------ Ripley helpers (gr/_ripley.py:197-271)
    rp = R.ripley()
    from scipy.spatial import ConvexHull

    cpts = rng.random((180, 2)) * 50
    support = np.linspace(0, 20, 15)
    _, lstat = rp["_l_function"](cpts, support, 400, 2500.0, "euclidean")
    hull = ConvexHull(cpts)
    sim = rp["_ppp"](hull, 1, 50, np.random.default_rng(5))
    d = np.sort(rng.random((60, 2)) * 25, axis=1)
    _, fg = rp["_f_g_function"](d.squeeze(), support)
    out.update(
        ripley_points=cpts, ripley_support=support, ripley_l=lstat, ripley_ppp=sim, ripley_fg_dist=d, ripley_fg=fg
    )

    np.savez_compressed(os.path.join(HERE, "reference_kernels.npz"), **out)
    print("wrote reference_kernels.npz with", len(out), "arrays")


if __name__ == "__main__":
    if "--export-h5ad" in sys.argv:
        export_h5ad()
    else:
        main()
