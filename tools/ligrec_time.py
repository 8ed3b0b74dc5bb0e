"""Timing of the ligrec permutation test on the GPU box: device kernels (per-kernel timers) against the C port of the
reference's numba kernel on the host cores.  Usage: python tools/ligrec_time.py [--json out.json] [--cpu-perms 4]

Workload (synthetic, shaped like a CellPhoneDB run on a large atlas): n_cells x n_genes sparse expression with the
given density, K clusters, all K*K cluster pairs, n_inter random gene pairs, P permutations."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from squidpy_amd import _lib as L
from squidpy_amd._utils import pcg64_states

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--cpu-perms", type=int, default=4)
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--cases", default="small,medium,large")
args = ap.parse_args()

CASES = {
    # name: (n_cells, n_genes, density, K, n_inter, P)
    "small": (20_000, 200, 0.10, 12, 500, 1000),
    "medium": (100_000, 400, 0.10, 25, 1000, 1000),
    "large": (500_000, 800, 0.08, 30, 2000, 1000),
}


def make(n, g, dens, k, n_inter, seed=0):
    rng = np.random.default_rng(seed)
    x = sp.random(n, g, density=dens, format="csc", random_state=rng, data_rvs=lambda s: rng.gamma(2.0, 1.0, s)).astype(np.float64)
    x.sort_indices()
    cl = rng.integers(0, k, n).astype(np.int32)
    inter = rng.integers(0, g, (n_inter, 2)).astype(np.int32)
    cp = np.array([(a, b) for a in range(k) for b in range(k)], dtype=np.int32)
    sizes = np.bincount(cl, minlength=k).astype(np.float64)
    onehot = sp.csr_matrix((np.ones(n), (cl, np.arange(n))), shape=(k, n))
    mean_obs = np.asarray((onehot @ x).todense()) / sizes[:, None]
    inv = 1.0 / np.maximum(sizes, 1)
    obs = mean_obs[cp[:, 0]][:, inter[:, 0]].T + mean_obs[cp[:, 1]][:, inter[:, 1]].T
    valid = np.ones(obs.shape, dtype=np.uint8)
    return x, cl, inter, cp, mean_obs, inv, obs, valid


ctx = L.default_context()
out = {}
for name in args.cases.split(","):
    n, g, dens, k, n_inter, P = CASES[name]
    x, cl, inter, cp, mean_obs, inv, obs, valid = make(n, g, dens, k, n_inter)
    L.ligrec_counts(ctx, x[:, :], cl, k, inv, inter, cp, obs, valid, seed=1, perm_begin=0, perm_end=64)  # warm-up
    res = {"n_cells": n, "n_genes": g, "nnz": int(x.nnz), "K": k, "n_inter": n_inter, "n_cpairs": len(cp), "n_perms": P}
    for mode in ("philox", "numpy"):
        kw = dict(seed=1) if mode == "philox" else dict(pcg_states=pcg64_states(1, P))
        ctx.timer_enable(True); ctx.timer_reset()
        t = time.perf_counter()
        c = L.ligrec_counts(ctx, x, cl, k, inv, inter, cp, obs, valid, perm_begin=0, perm_end=P, **kw)
        wall = time.perf_counter() - t
        rep = {kname: (cnt, ms) for kname, (cnt, ms) in ctx.timer_report().items() if kname.startswith("ligrec")}
        ctx.timer_enable(False)
        kern = sum(ms for _, ms in rep.values())
        res[mode] = {"wall_s": wall, "kernels_ms": kern, "perms_per_s_wall": P / wall, "per_kernel": rep, "checksum": int(c.sum())}
        sums_ms = rep.get("ligrec_sums", (0, 0.0))[1]
        if sums_ms:
            # algorithmic traffic of the sum kernel: per permutation and stored entry one label byte + one f64 LDS add
            res[mode]["sums_adds_per_s"] = x.nnz * P / (sums_ms * 1e-3)
        print(f"{name} {mode}: wall {wall:.3f}s kernels {kern:.1f} ms -> {P / wall:.0f} perms/s; " + ", ".join(f"{a}={b[1]:.2f}ms/{b[0]}" for a, b in rep.items()), flush=True)
    if not args.no_cpu:
        from oracle import cport

        cport.build(native=True)
        q = args.cpu_perms
        dense = x.toarray()
        rng = np.random.default_rng(0)
        lab = np.stack([rng.permutation(cl) for _ in range(q)]).astype(np.int32)
        t = time.perf_counter(); c1 = cport.ligrec_score(dense, lab, inv, mean_obs, inter, cp, valid, native=True); t1 = time.perf_counter() - t
        ncore = len(os.sched_getaffinity(0))
        labp = np.stack([rng.permutation(cl) for _ in range(max(q, ncore))]).astype(np.int32)
        t = time.perf_counter(); cport.ligrec_score(dense, labp, inv, mean_obs, inter, cp, valid, parallel=True, native=True); tp = time.perf_counter() - t
        res["cpu_port"] = {"perms_per_s_1core": q / t1, "perms_per_s_allcores": len(labp) / tp, "cores": ncore, "sample_perms": q}
        print(f"{name} cpu port: {q / t1:.2f} perms/s (1 core), {len(labp) / tp:.1f} perms/s ({ncore} threads)", flush=True)
        del dense
    out[name] = res
if args.json:
    os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
    json.dump(out, open(args.json, "w"), indent=1)
