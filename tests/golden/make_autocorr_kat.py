"""Exact-rational known answers for Moran's I and Geary's C  ->  tests/golden/autocorr_kat.json

The arithmetic of the statistic lives in `scanpy.metrics.morans_i / gearys_c` (third party, not in /root/reference, not
installable here): SURVEY.md §8c calls it "parity unpinned".  What CAN be pinned is the definition the reference documents
(gr/_ppatterns.py:79-99 cites pysal's global Moran / Geary; scanpy implements the same closed forms):

    I = (N / W) * sum_ij w_ij z_i z_j / sum_i z_i^2                       z = x - mean(x),  W = sum_ij w_ij
    C = (N - 1) * sum_ij w_ij (x_i - x_j)^2 / (2 W sum_i z_i^2)

evaluated here in EXACT rational arithmetic (fractions.Fraction — every float input is converted exactly), then rounded
once to float64.  oracle/restate.py and the HIP kernels must hit these to the last few ulps; a formula error, a wrong
normalisation or a float32 detour shows.  Inputs (stored in the JSON, so the tests need nothing else):

  intmat5   the reference's 5-node fixture graph `adata_intmat` (tests/conftest.py:177-194), its integer weights as they
            are AND row-normalised the way `spatial_autocorr(transformation=True)` does (sklearn normalize, float64), three
            small integer-valued features;
  visium49  the 49 spots of the reference's tests/_data/test_data.h5ad (tests/golden/visium49.npz), 6-nearest-neighbour
            graph of their coordinates (sklearn; indices stored), row-normalised in float64, genes 0..2 of the matrix.

    python tests/golden/make_autocorr_kat.py
"""

from __future__ import annotations

import json
import os
from fractions import Fraction

import numpy as np
import scipy.sparse as sp
from sklearn.neighbors import NearestNeighbors
from sklearn.preprocessing import normalize

HERE = os.path.dirname(os.path.abspath(__file__))


def exact(indptr, indices, data, x):
    n = len(x)
    xs = [Fraction(float(v)) for v in x]
    mean = sum(xs, Fraction(0)) / n
    z = [v - mean for v in xs]
    ss = sum((v * v for v in z), Fraction(0))
    if ss == 0:
        return None, None
    W = sum((Fraction(float(w)) for w in data), Fraction(0))
    num_i = Fraction(0)
    num_c = Fraction(0)
    for i in range(n):
        for e in range(indptr[i], indptr[i + 1]):
            j, w = int(indices[e]), Fraction(float(data[e]))
            num_i += w * z[i] * z[j]
            num_c += w * (xs[i] - xs[j]) ** 2
    return Fraction(n) / W * num_i / ss, Fraction(n - 1) * num_c / (2 * W * ss)


def case(name, g, X):
    g = sp.csr_matrix(g)
    g.sort_indices()
    rec = {"name": name, "n": g.shape[0], "indptr": g.indptr.tolist(), "indices": g.indices.tolist(),
           "data_hex": [float(w).hex() for w in g.data], "X_hex": [[float(v).hex() for v in row] for row in X], "I": [], "C": [],
           "I_fraction": [], "C_fraction": []}
    for row in X:
        i, c = exact(g.indptr, g.indices, g.data, row)
        rec["I"].append(None if i is None else float(i).hex())
        rec["C"].append(None if c is None else float(c).hex())
        rec["I_fraction"].append(None if i is None else f"{i.numerator}/{i.denominator}" if len(str(i.denominator)) < 60 else "(long)")
        rec["C_fraction"].append(None if c is None else f"{c.numerator}/{c.denominator}" if len(str(c.denominator)) < 60 else "(long)")
    return rec


def main() -> None:
    out = []
    dense = np.array([[0, 1, 1, 0, 0], [0, 0, 0, 0, 1], [1, 2, 0, 0, 0], [0, 1, 0, 0, 1], [0, 0, 1, 2, 0]], dtype=np.float64)
    X5 = np.array([[1, 2, 3, 4, 5], [3, 1, 4, 1, 5], [2, 2, 2, 2, 2], [0, 0, 1, 0, 7]], dtype=np.float64)  # row 2 constant -> NaN
    out.append(case("intmat5_raw", sp.csr_matrix(dense), X5))
    out.append(case("intmat5_rownorm", normalize(sp.csr_matrix(dense), norm="l1", axis=1), X5))
    v = np.load(os.path.join(HERE, "visium49.npz"))
    xy = v["spatial"].astype(np.float64)
    n = len(xy)
    idx = NearestNeighbors(n_neighbors=7).fit(xy).kneighbors(xy, return_distance=False)[:, 1:]
    g = sp.csr_matrix((np.ones(n * 6), idx.ravel(), np.arange(0, n * 6 + 1, 6)), shape=(n, n))
    out.append(case("visium49_knn6_rownorm", normalize(g, norm="l1", axis=1), v["X40"][:, :3].T.astype(np.float64)))
    path = os.path.join(HERE, "autocorr_kat.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    for rec in out:
        print(rec["name"], [None if h is None else float.fromhex(h) for h in rec["I"]], [None if h is None else float.fromhex(h) for h in rec["C"]])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
