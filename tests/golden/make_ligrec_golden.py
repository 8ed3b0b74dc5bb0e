"""Generates tests/golden/ligrec_reference.npz by executing the reference's OWN source (gr/_ligrec.py:616-775,
`_score_permutations` + `_analysis`) through oracle/ref_shim.py.  Needs /root/reference (build container only).

    python tests/golden/make_ligrec_golden.py

Cases (all inputs and outputs are stored, so the GPU tests need neither the reference nor pandas-side logic):
  A  240 cells x 14 genes, 4 clusters, sparse non-integer expression, all 16 cluster pairs, threshold 0.1, 64 perms
  B  150 cells x 9 genes, 3 clusters, small-integer counts (exact ties between permuted and observed sums occur),
     7 selected cluster pairs, threshold 0.2, 100 perms
  C  the reference's own `test_ligrec_nan_counts` layout (tests/graph/test_ligrec.py:446-): 6 cells, 3 genes,
     2 clusters, threshold 0.8, 5 perms
"""

from __future__ import annotations

import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402


def run_case(ns, data, cl, inter, cp, threshold, n_perms, seed):
    df = pd.DataFrame(data, columns=list(range(data.shape[1])))
    df["clusters"] = pd.Categorical(cl)
    res = ns["_analysis"](df, inter, cp, threshold=threshold, n_perms=n_perms, seed=seed, n_jobs=1, show_progress_bar=False)
    return np.asarray(res.means), np.asarray(res.pvalues)


def main() -> None:
    ns = ref_shim.ligrec()
    out: dict[str, np.ndarray] = {}
    rng = np.random.default_rng(20240924)

    def store(tag, data, cl, inter, cp, threshold, n_perms, seed):
        means, pvals = run_case(ns, data, cl, inter, cp, threshold, n_perms, seed)
        out.update({
            f"{tag}_data": data, f"{tag}_clusters": cl, f"{tag}_interactions": inter, f"{tag}_cpairs": cp,
            f"{tag}_threshold": np.float64(threshold), f"{tag}_n_perms": np.int64(n_perms), f"{tag}_seed": np.int64(seed),
            f"{tag}_means": means, f"{tag}_pvalues": pvals,
        })
        print(tag, means.shape, "tested:", int((~np.isnan(pvals)).sum()), "distinct p:", len(np.unique(pvals[~np.isnan(pvals)])))

    # A
    n, g, k = 240, 14, 4
    data = rng.poisson(0.5, size=(n, g)).astype(np.float64) * rng.gamma(2.0, 1.0, size=(n, g))
    cl = rng.integers(0, k, n).astype(np.int32)
    data[cl == 1, :4] += rng.gamma(2.0, 1.0, size=(int((cl == 1).sum()), 4))  # some real signal
    inter = np.array([(i, j) for i in range(g) for j in range(g) if i != j], dtype=np.int32)[::3]
    cp = np.array([(a, b) for a in range(k) for b in range(k)], dtype=np.int32)
    store("A", data, cl, inter, cp, 0.1, 64, 42)

    # B
    n, g, k = 150, 9, 3
    data = rng.poisson(0.3, size=(n, g)).astype(np.float64)
    cl = rng.integers(0, k, n).astype(np.int32)
    inter = np.array([(i, (i + 1) % g) for i in range(g)] + [(i, i) for i in range(0, g, 2)], dtype=np.int32)
    cp = np.array([(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 1), (2, 2)], dtype=np.int32)
    store("B", data, cl, inter, cp, 0.2, 100, 7)

    # C
    data = np.array([[1, 1, 0], [0, 1, 0], [0, 1, 0], [1, 0, 1], [0, 0, 1], [0, 0, 1]], dtype=np.float64)
    cl = np.array([0, 0, 0, 1, 1, 1], dtype=np.int32)
    inter = np.array([(0, 1), (1, 2), (2, 0)], dtype=np.int32)
    cp = np.array([(0, 0), (0, 1), (1, 0), (1, 1)], dtype=np.int32)
    store("C", data, cl, inter, cp, 0.8, 5, 0)

    path = os.path.join(ROOT, "tests", "golden", "ligrec_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
