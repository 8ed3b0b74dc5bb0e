"""Developer tool (GPU box): round-4 experiments on the LDS-bucketed dot at config 3's shape, 1000 permutations.
`--one`: one timing of both statistics under the environment given (tools/joint_order.sh drives it for the list schedules).
Without arguments: round 3's per-list order (SQGR_AUTOCORR_ORDER=single) by Z class / by Y class / none; chunks per XCD round."""
import os, sys, time, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from sklearn.preprocessing import normalize
    from squidpy_amd import _lib as L
    from squidpy_amd._synthetic import hex_grid_graph
    ctx = L.default_context()
    rows, cols, G, P = int(os.environ.get("EXP_ROWS", 250)), int(os.environ.get("EXP_COLS", 400)), 2048, int(os.environ.get("EXP_PERMS", 1000))
    n = rows * cols
    g = normalize(hex_grid_graph(rows, cols), norm="l1", axis=1)
    vals = np.random.default_rng(11).gamma(2.0, 1.0, size=(G, n))
    graph = L.Graph(ctx, g, with_data=True)
    plan = L.AutocorrPlan(ctx, graph, vals)
    out = {}
    for mode in ("moran", "geary"):
        plan.perms(mode, seed=1, perm_begin=0, perm_end=P)
        ctx.sync(); ctx.timer_enable(True); ctx.timer_reset()
        for i in range(3):
            plan.perms(mode, seed=2 + i, perm_begin=0, perm_end=P)
        ctx.sync()
        rep = ctx.timer_report(); ctx.timer_enable(False)
        out[mode] = {k: round(v[1] / 3, 2) for k, v in rep.items() if v[0]}
    print(json.dumps(out))
    sys.exit(0)
for label, env in (("defaults (Moran: Z classes, 4 chunks per XCD round; Geary: Y classes, 1 chunk)", {}),
                   ("Z classes, 1 chunk (round 3)", {"SQGR_AUTOCORR_ORDER_BY": "z", "SQGR_AUTOCORR_XCD_CHUNKS": "1"}),
                   ("Y classes, 1 chunk", {"SQGR_AUTOCORR_ORDER_BY": "y", "SQGR_AUTOCORR_XCD_CHUNKS": "1"}),
                   ("Z classes, 2 chunks", {"SQGR_AUTOCORR_ORDER_BY": "z", "SQGR_AUTOCORR_XCD_CHUNKS": "2"}),
                   ("Z classes, 4 chunks", {"SQGR_AUTOCORR_ORDER_BY": "z", "SQGR_AUTOCORR_XCD_CHUNKS": "4"}),
                   ("Y classes, 4 chunks", {"SQGR_AUTOCORR_ORDER_BY": "y", "SQGR_AUTOCORR_XCD_CHUNKS": "4"}),
                   ("Z classes, 8 chunks", {"SQGR_AUTOCORR_ORDER_BY": "z", "SQGR_AUTOCORR_XCD_CHUNKS": "8"}),
                   ("Y classes, 2 chunks", {"SQGR_AUTOCORR_ORDER_BY": "y", "SQGR_AUTOCORR_XCD_CHUNKS": "2"}),
                   ("no order, 1 chunk", {"SQGR_AUTOCORR_ORDER_LISTS": "0", "SQGR_AUTOCORR_XCD_CHUNKS": "1"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, SQGR_AUTOCORR_KERNEL="lds", SQGR_AUTOCORR_ORDER="single", **env), capture_output=True, text=True, timeout=300)
    print(json.dumps({"variant": label, "ms_per_2048_genes_x_1000_perms": json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else r.stderr[-300:]}), flush=True)
