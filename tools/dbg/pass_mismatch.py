import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as O
from squidpy_amd import _lib as L
from tests.helpers import knn_graph

ctx = L.default_context()
k, width = 102, 0
rng = np.random.default_rng(k * 7 + width)
n = 21000
adj = knn_graph(rng.random((n, 2)), 6)
labels = rng.integers(0, k, n).astype(np.int32)
g = L.Graph(ctx, adj)
ref = O.nhood_perm_counts_philox(adj.indices, adj.indptr, labels, k, 11, 5, 55).astype(np.uint32)
for tune in (None, None, (2, 0, 3), (1, 0, 3), (2, 8, 0), (2, 64, 0), (2, 31, 0), (4, 0, 0)):
    plan = L.NhoodPlan(ctx, g, labels, k)
    if tune:
        plan.tune(*tune)
    _, _, perms = plan.run(11, 5, 55, None, return_perms=True)
    bad = np.argwhere(perms != ref)
    print(tune, plan.info()["blocks_per_batch"], "mismatches", len(bad), [(tuple(b), int(perms[tuple(b)]), int(ref[tuple(b)])) for b in bad[:6]], "sum", int(perms[0].sum()), adj.nnz, flush=True)
# which labels does the device produce for the first bad permutation?
if len(bad):
    p = int(bad[0][0]) + 5
    dev = plan.shuffled_labels(11, p)
    from oracle import devrng
    host = devrng.shuffled_labels(labels, 11, p, None, 0)
    print("labels differ at", np.flatnonzero(dev != host)[:10], "of", n)
