"""Developer tool (GPU box): what the ORDER of the spots costs nhood_enrichment's count kernel — the hex grid in scan order (what the
bench legs use), in random order (cells of a real data set come in no spatial order), and the random order brought back by a
reverse Cuthill-McKee / coordinate sort (what an internal renumbering could do)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph
from squidpy_amd.gr._nhood import expected_counts
ctx = L.default_context()
rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
adj = hex_grid_graph(rows, cols).tocsr(); n = adj.shape[0]
rng = np.random.default_rng(0)
labels = rng.integers(0, 30, n).astype(np.int32)
def run(name, A, lab):
    A = sp.csr_matrix(A); A.sort_indices()
    g = L.Graph(ctx, A, with_data=False)
    plan = L.NhoodPlan(ctx, g, lab, 30)
    shift = expected_counts(lab, 30, int(A.nnz))
    plan.run(3, 0, 2560, shift); ctx.sync()
    ctx.timer_enable(True); ctx.timer_reset()
    t = time.perf_counter()
    for i in range(3): plan.run(3, 2560 * i, 2560 * (i + 1), shift)
    ctx.sync(); dt = (time.perf_counter() - t) / 3
    rep = {k: round(v[1] / 3, 2) for k, v in ctx.timer_report().items() if v[0]}
    ctx.timer_enable(False)
    print(f"{name}: {2560 / dt:.0f} perms/s {rep}", flush=True)
    plan.close(); g.close()
run("scan order", adj, labels)
perm = rng.permutation(n)                 # new index -> old index
inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
coo = adj.tocoo()
shuf = sp.csr_matrix((coo.data, (inv[coo.row], inv[coo.col])), shape=(n, n))
run("random order", shuf, labels[perm])
def renumber(A, lab, order):   # order: new position -> old index
    invo = np.empty(n, np.int64); invo[order] = np.arange(n)
    c = A.tocoo()
    return sp.csr_matrix((c.data, (invo[c.row], invo[c.col])), shape=(n, n)), lab[order]
# cells listed field of view by field of view (tiles of T x T spots), in random order inside a field of view
for T in (250, 100, 32):
    r, c = np.divmod(np.arange(n), cols)
    key = ((r // T) * (cols // T + 1) + (c // T)).astype(np.int64) * n + rng.permutation(n)
    order = np.argsort(key)
    A2, l2 = renumber(adj, labels, order)
    run(f"fields of view of {T} x {T} spots, random order inside", A2, l2)
t = time.perf_counter()
from scipy.sparse.csgraph import reverse_cuthill_mckee
rcm = reverse_cuthill_mckee(shuf, symmetric_mode=True)
print("reverse Cuthill-McKee on the host: %.2f s" % (time.perf_counter() - t))
inv2 = np.empty(n, np.int64); inv2[rcm] = np.arange(n)
c2 = shuf.tocoo()
back = sp.csr_matrix((c2.data, (inv2[c2.row], inv2[c2.col])), shape=(n, n))
run("random order, renumbered by RCM", back, labels[perm][rcm])
import squidpy_amd as sq
from squidpy_amd._synthetic import hex_grid
xy = hex_grid(rows, cols)[perm]           # coordinates in the random order
t = time.perf_counter(); mo = sq.spatial_order(coords=xy); print("Morton order on the host: %.2f s" % (time.perf_counter() - t))
A3, l3 = renumber(shuf, labels[perm], mo)
run("random order, renumbered along the Morton curve", A3, l3)
