"""Developer tool (GPU box): the nhood permutation test over the number of clusters — which count kernel K selects, what it
costs per permutation, and the tuning space of the pass kernel (pass width, edge chunks per batch, batches per launch).

    python tools/nhood_k_sweep.py [rows] [perms] [--graph hex|knn] [--sweep]
Writes one JSON line per configuration (gpurun_out/nhood_k_sweep.jsonl when run through tools/r05_*.sh)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from squidpy_amd import _lib as L
from squidpy_amd._synthetic import hex_grid_graph

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows = cols = int(args[0]) if len(args) > 0 else 1000
P = int(args[1]) if len(args) > 1 else 2560
graph_kind = "knn" if "--graph=knn" in sys.argv else "hex"
ctx = L.default_context()
if graph_kind == "hex":
    adj = hex_grid_graph(rows, cols)
else:  # the directed 6-nearest-neighbour graph of the same lattice + jitter (full edge list: no symmetry to halve)
    from sklearn.neighbors import NearestNeighbors

    from squidpy_amd._synthetic import hex_grid

    xy = hex_grid(rows, cols) + np.random.default_rng(1).normal(0, 5, (rows * cols, 2))
    adj = NearestNeighbors(n_neighbors=7).fit(xy).kneighbors_graph(xy, mode="connectivity").tolil()
    adj.setdiag(0)
    adj = adj.tocsr()
    adj.eliminate_zeros()
n = adj.shape[0]
g = L.Graph(ctx, adj, with_data=False)


def measure(K, tune=None, reps=2):
    labels = np.random.default_rng(0).integers(0, K, n).astype(np.int32)
    plan = L.NhoodPlan(ctx, g, labels, K)
    if tune:
        plan.tune(*tune)
    info = plan.info()
    plan.run(1, 0, 64)
    ctx.sync()
    best = None
    for _ in range(reps):
        ctx.timer_enable(True)
        ctx.timer_reset()
        t = time.perf_counter()
        plan.run(1, 0, P)
        ctx.sync()
        dt = time.perf_counter() - t
        rep = ctx.timer_report()
        ctx.timer_enable(False)
        if best is None or dt < best[0]:
            best = (dt, rep)
    dt, rep = best
    ks = {k: [v[0], round(v[1], 3)] for k, v in rep.items() if k.startswith("nhood") and v[0] > 0}
    cnt_ms = sum(v[1] for k, v in ks.items() if k.startswith("nhood_count"))
    rec = {"K": K, "tune": tune, "perms": P, "perms_per_s": round(P / dt), "count_us_per_perm": round(cnt_ms * 1e3 / P, 4),
           "perms_per_pass": info["perms_per_pass"], "counter_mode": info["counter_mode"], "blocks_per_batch": info["blocks_per_batch"], "batches_per_launch": info["batches_per_launch"], "list_edges": info["list_edges"], "kernels": ks}
    print(json.dumps(rec), flush=True)
    plan.close() if hasattr(plan, "close") else None
    return rec


if "--sweep" in sys.argv:
    # the pass kernel's tuning space at the three bench sizes, and the K <= 50 kernel forced through narrower passes
    for K, widths in ((30, (0, 8, 4)), (64, (0, 4)), (100, (0, 2)), (200, (0,))):
        for w in widths:
            for blocks in (0, 8, 16, 32, 64):
                for nbatch in (0, 64):
                    measure(K, (w, blocks, nbatch) if (w or blocks or nbatch) else None, reps=1)
else:
    ks = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--K=")] or [30, 50, 51, 64, 71, 72, 100, 101, 102, 150, 200, 202, 203, 256]
    tune = [tuple(int(v) for v in a.split("=")[1].split(",")) for a in sys.argv if a.startswith("--tune=")]
    for K in ks:
        measure(K, tune[0] if tune else None)
