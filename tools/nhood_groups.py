"""Developer tool (GPU box; VERDICT r3 task 2b): does keeping the shuffle -> count label slab inside the 256 MiB Infinity Cache
move the count kernel?  One process, the C5-shaped workload of bench.py, launch groups of 8 / 12 / 16 / 32 / 160 batches of 16
permutations (slab 16 MB per batch at 1e6 spots: 128 / 192 / 256 / 512 / 2560 MB in flight), `k_shuffle` and `k_count`
alternating on one stream over ONE slab buffer.  Prints one JSON line per group size with permutations/s and per-kernel HIP-event
averages; under `rocprofv3 --pmc` the per-dispatch counter rows are told apart by the grid size (tools/summarize_groups.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from squidpy_amd import _lib as L  # noqa: E402
from squidpy_amd._synthetic import hex_grid_graph  # noqa: E402
from squidpy_amd.gr._nhood import expected_counts  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
groups = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 12, 16, 32, 160]
ctx = L.default_context()
adj = hex_grid_graph(1000, 1000)
n, nnz, K = adj.shape[0], int(adj.nnz), 30
labels = np.random.default_rng(0).integers(0, K, n).astype(np.int32)
g = L.Graph(ctx, adj, with_data=False)
plan = L.NhoodPlan(ctx, g, labels, K)
shift = expected_counts(labels, K, nnz)
for nb in groups:
    plan.tune(16, 0, nb)
    plan.run(1, 0, 16 * nb * 2, shift)  # warm: workspaces of this geometry
    ctx.sync()
    ctx.timer_enable(True)
    ctx.timer_reset()
    t0 = time.perf_counter()
    s1, s2, _ = plan.run(2, 0, P, shift)
    ctx.sync()
    dt = time.perf_counter() - t0
    rep = ctx.timer_report()
    ctx.timer_enable(False)
    info = plan.info()
    rec = {"batches_per_group": nb, "perms_per_group": 16 * nb, "slab_MB_in_flight": 16 * nb * n / 1e6, "perms": P, "perms_per_s": P / dt,
           "blocks_per_batch": info["blocks_per_batch"], "checksum": int(s1.sum()),
           "kernels": {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1), "us_per_perm": v[1] * 1e3 / P} for k, v in rep.items() if v[0] > 0}}
    print(json.dumps(rec), flush=True)
