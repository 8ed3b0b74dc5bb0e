#!/usr/bin/env python
"""Benchmark of the sq.gr hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1): ``nhood_enrichment`` permutation test on the 1e6-spot hex grid x 30 clusters that BASELINE.json's
metric is quoted on; one *step* = one pass of the hot path over one batch of PERMS_PER_STEP (10 000) permutations
with graph and labels already resident in HBM.  N>1: one process per GPU (torch.distributed, backend nccl = RCCL),
every rank runs its own permutation range of each step (weak scaling, no data-path collective) followed by the
path's one real exchange: an all-reduce of the exact integer moments.

Rank 0 prints ONE JSON line with `roofline` (HIP-event timing of the CSR-gather kernel on the library's own
stream) and, at N=1, `cpu_baseline` (the oracle's C restatement of Squidpy's numba kernel driven by numpy's
PCG64 shuffles, timed on this box's host cores on a bounded sample)."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS = COLS = 1000
N_CLS = 30
PERMS_PER_STEP = 10_000
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


def _cpu_worker(args):
    path, n_cls, first, count = args
    from oracle import cport  # checker code: cpu_baseline leg only

    z = np.load(path)
    indices, indptr, base = z["indices"], z["indptr"], z["base"]
    gens = [np.random.default_rng(s) for s in np.random.SeedSequence(0).spawn(first + count)][first:]
    cport.lib(native=True)
    t0 = time.perf_counter()
    for g in gens:
        shuffled = base.copy()
        g.shuffle(shuffled)
        cport.nenrich(indices, indptr, shuffled, n_cls, parallel=False, native=True).astype(np.float64)
    return time.perf_counter() - t0


def _cpu_baseline_all_cores(indices, indptr, base, budget_s: float) -> dict:
    """One single-threaded worker *process* per host core (plain subprocesses of this file, hard timeout)."""
    import subprocess
    import tempfile

    ncores = len(os.sched_getaffinity(0))
    workers = min(ncores, 64)
    per_worker = 6
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "graph.npz")
        np.savez(path, indices=indices, indptr=indptr, base=base)
        t0 = time.perf_counter()
        procs = [
            subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(w * per_worker), str(per_worker)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            for w in range(workers)
        ]
        busy = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=120)
                busy.append(float(o.strip().splitlines()[-1]))
            except Exception:
                p.kill()
        wall = time.perf_counter() - t0
    if len(busy) != workers:
        return {"error": f"{workers - len(busy)} of {workers} workers failed"}
    # throughput while all cores are busy: workers run concurrently, each reports the time of its own loop
    return {"value": workers * per_worker / max(busy), "unit": "permutations/s", "cores": workers, "host_cores": ncores,
            "mode": "n_jobs=all-cores analogue: one single-threaded worker process per core, contiguous permutation chunks",
            "sample": f"{workers * per_worker} permutations, slowest worker loop {max(busy):.1f} s (wall incl. start-up {wall:.1f} s)"}


def cpu_baseline(adj, labels: np.ndarray, budget_s: float = 12.0) -> dict:
    """Squidpy's default CPU path (n_jobs=None, numba_parallel=False: ONE core) on a bounded sample of the same
    workload: per permutation `shuffled = int_clust.copy(); rng.shuffle(shuffled); _nenrich(...)`
    (gr/_nhood.py:530-539), numba kernel restated in C (oracle/c/sqgr_cpu.c) because numba is absent."""
    from oracle import cport  # checker code: cpu_baseline leg only

    cport.lib(native=True)
    indices, indptr = adj.indices.astype(np.uint32), adj.indptr.astype(np.uint32)
    base = labels.astype(np.uint32)
    gens = [np.random.default_rng(s) for s in np.random.SeedSequence(0).spawn(4096)]
    t0 = time.perf_counter()
    done = 0
    while done < len(gens):
        shuffled = base.copy()
        gens[done].shuffle(shuffled)
        cport.nenrich(indices, indptr, shuffled, N_CLS, parallel=False, native=True).astype(np.float64)
        done += 1
        if time.perf_counter() - t0 > budget_s and done >= 5:
            break
    dt = time.perf_counter() - t0
    out = {
        "value": done / dt,
        "unit": "permutations/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} permutations of the same {len(base)}-spot x {N_CLS}-cluster workload (numpy PCG64 shuffle + C restatement "
        f"of the numba kernel incl. its res[N,K] scratch), {dt:.1f} s on 1 core = Squidpy's default n_jobs=None",
    }
    # all host cores: Squidpy's n_jobs=-1 (joblib process fan-out, contiguous permutation chunks per worker,
    # _utils.py:223-231), each worker single-threaded like `_callback_wrapper` forces numba to be
    try:
        out["all_cores"] = _cpu_baseline_all_cores(indices, indptr, base, budget_s)
    except Exception as exc:  # pragma: no cover
        out["all_cores"] = {"error": repr(exc)}
    return out


def moran_secondary(ctx, world: int, rank: int, fence, steps: int, with_cpu: bool) -> dict:
    """Second half of BASELINE.json's metric: Moran's I genes/sec on the C3 shape (1e5 spots, k=6 CSR graph,
    n_perms=1000); one step = observed score + 1000 permuted scores for a resident block of 2048 genes per GPU."""
    from sklearn.preprocessing import normalize

    from squidpy_amd import _lib
    from squidpy_amd._synthetic import hex_grid_graph

    rows, cols, G, P = 250, 400, 2048, 1000
    n = rows * cols
    g = normalize(hex_grid_graph(rows, cols), norm="l1", axis=1)
    vals = np.random.default_rng(1 + rank).gamma(2.0, 1.0, size=(G, n))
    graph = _lib.Graph(ctx, g, with_data=True)
    plan = _lib.AutocorrPlan(ctx, graph, vals)  # resident from here on
    plan.perms("moran", seed=1, perm_begin=0, perm_end=32)
    fence()
    ctx.timer_enable(True)
    ctx.timer_reset()
    t0 = time.perf_counter()
    for i in range(steps):
        score = plan.scores("moran")
        sims = plan.perms("moran", seed=7, perm_begin=i * P, perm_end=(i + 1) * P)
    fence()
    elapsed = time.perf_counter() - t0
    kernels = ctx.timer_report()
    ctx.timer_enable(False)
    assert np.isfinite(score).all() and np.isfinite(sims).all()
    if world > 1:
        import torch
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cnt, ms = kernels.get("autocorr_perm_dot_moran", (0, 0.0))
    b_gene = (P + 1) * 8 * n
    out = {
        "metric": "Moran's I genes/sec (1e5 spots, CSR k=6, n_perms=1000)",
        "value": steps * G * world / elapsed,
        "unit": "genes/s",
        "ms_per_step": elapsed / steps * 1e3,
        "dtype": "f64",
        "config": {"workload": f"spatial_autocorr moran: {n} spots, {G} genes per GPU per step, {P} permutations, device permutations"},
        "roofline": {
            "kernel": "autocorr_perm_dot_moran",
            "bound": "hbm",
            "achieved": (b_gene * G * steps / (ms * 1e-3) / 1e9) if ms > 0 else None,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": (b_gene * G * steps / (ms * 1e-3) / HBM_PEAK) if ms > 0 else None,
            "traffic": None,
            "launches": cnt,
            "avg_launch_ms": ms / max(cnt, 1),
            "algorithmic_bytes_per_gene": b_gene,
            "note": "512-byte row gathers served mostly by the 256 MB Infinity Cache (working set per 64-gene tile = 51 MB)",
        },
    }
    if with_cpu:
        from oracle import cport  # checker code: cpu_baseline leg only

        gsub, evals = 64, 0
        gens = [np.random.default_rng(s) for s in np.random.SeedSequence(0).spawn(64)]
        g64 = g.astype(np.float64)
        t0 = time.perf_counter()
        while evals < len(gens):
            idx = gens[evals].permutation(n)
            cport.morans_i(g64[idx, :], vals[:gsub], parallel=False, native=True)  # gr/_ppatterns.py:271-272
            evals += 1
            if time.perf_counter() - t0 > 6.0 and evals >= 2:
                break
        per_eval_gene = (time.perf_counter() - t0) / (evals * gsub)
        out["cpu_baseline"] = {
            "value": 1.0 / ((P + 1) * per_eval_gene),
            "unit": "genes/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{evals} permutations x {gsub} genes (scipy row permutation g[idx,:] + C restatement of scanpy's per-gene "
            f"Moran loop), scaled to {P + 1} evaluations per gene",
        }
    plan.close()
    graph.close()
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--perms-per-step", type=int, default=PERMS_PER_STEP)
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--label-dist", choices=["uniform", "dirichlet"], default="uniform",
                    help="cluster sizes: uniform (the headline workload) or Dirichlet(0.5) proportions (SURVEY §8d: skewed variant "
                    "that concentrates the LDS-atomic traffic of the count kernel on few counters)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the Moran's I genes/sec leg")
    ap.add_argument("--no-numpy-leg", action="store_true", help="skip the bit-compatible numpy-stream leg")
    ap.add_argument("--tune", type=str, default="", help="perms_per_pass,blocks_per_batch,batches_per_launch")
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        print(_cpu_worker((sys.argv[2], N_CLS, int(sys.argv[3]), int(sys.argv[4]))))
        return
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        backend = os.environ.get("SQGR_DIST_BACKEND", "nccl")  # "gloo" lets one GPU host several ranks (testing only)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1)))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from squidpy_amd import _dist, _lib
    from squidpy_amd._synthetic import hex_grid_graph
    from squidpy_amd.gr._nhood import expected_counts, zscore_from_moments

    ctx = _lib.default_context(local_rank % max(_lib.device_count(), 1))
    adj = hex_grid_graph(args.rows, args.cols)
    n, nnz = adj.shape[0], int(adj.nnz)
    lab_rng = np.random.default_rng(0)
    if args.label_dist == "dirichlet":
        labels = lab_rng.choice(N_CLS, size=n, p=lab_rng.dirichlet(np.full(N_CLS, 0.5))).astype(np.int32)
    else:
        labels = lab_rng.integers(0, N_CLS, n).astype(np.int32)
    graph = _lib.Graph(ctx, adj, with_data=False)          # resident in HBM from here on
    plan = _lib.NhoodPlan(ctx, graph, labels, N_CLS)
    if args.tune:
        plan.tune(*[int(v) for v in args.tune.split(",")])
    count = _lib.nhood_counts(ctx, graph, labels, N_CLS)
    shift = expected_counts(labels, N_CLS, nnz)
    P = args.perms_per_step

    def step(i: int):
        lo = (i * world + rank) * P                         # disjoint global permutation indices per rank & step
        s1, s2, _ = plan.run(12345, lo, lo + P, shift)
        if world > 1:
            s1, s2 = _dist.allreduce_sum_([s1, s2])         # the path's only exchange (RCCL over xGMI)
        return s1, s2

    def fence():
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.timer_enable(True)
    ctx.timer_reset()
    t0 = time.perf_counter()
    tot1 = np.zeros((N_CLS, N_CLS), dtype=np.int64)
    tot2 = np.zeros((N_CLS, N_CLS), dtype=np.uint64)
    for i in range(args.steps):
        s1, s2 = step(args.warmup + i)
        tot1 += s1
        tot2 += s2
    fence()
    elapsed = time.perf_counter() - t0
    kernels = ctx.timer_report()
    ctx.timer_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    secondary = None
    if not args.no_secondary:
        secondary = moran_secondary(ctx, world, rank, fence, max(1, min(args.steps, 3)), world == 1 and rank == 0 and not args.no_cpu_baseline)

    if rank == 0:
        total_perms = args.steps * P * world
        z = zscore_from_moments(count, shift, tot1, tot2, total_perms)
        assert np.isfinite(z).all(), "non-finite z-score in benchmark run"
        # ---- roofline of the CSR-gather kernel (nhood_count*), HIP events on the library's stream
        cnt_name = [k for k in kernels if k.startswith("nhood_count") and kernels[k][0] > 0]
        launches = sum(kernels[k][0] for k in cnt_name)
        ms = sum(kernels[k][1] for k in cnt_name)
        perms_per_launch = args.steps * P / max(launches, 1)
        # algorithmic bytes per permutation: indices + indptr streamed once, shuffled labels read once
        # by this kernel (their write, N bytes, belongs to the shuffle kernel and is counted in the pipeline figure)
        b_gather = 4 * nnz + 4 * (n + 1) + n
        b_perm = 4 * nnz + 4 * (n + 1) + 2 * n
        achieved = b_gather * perms_per_launch / (ms / max(launches, 1) * 1e-3) if ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    traffic = json.load(fh).get("nhood_count_bytes_per_launch")
            except Exception:
                traffic = None
        kern_ms = {k: round(v[1] / max(v[0], 1), 4) for k, v in kernels.items() if v[0] > 0}
        gpu_ms = sum(v[1] for v in kernels.values())
        out = {
            "metric": "nhood_enrichment permutations/sec (1e6 spots x 30 clusters)",
            "value": total_perms / elapsed,
            "unit": "permutations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 labels / u32 counts / i64 moments",
            "data": "synthetic",
            "config": {
                "workload": f"nhood_enrichment: {n} spots ({args.rows}x{args.cols} hex grid, nnz={nnz}), {N_CLS} clusters, "
                f"{P} permutations per step per GPU, on-device Philox/Feistel shuffles"
                + ("" if args.label_dist == "uniform" else ", Dirichlet(0.5) cluster proportions"),
                "label_dist": args.label_dist,
                "perms_per_step_per_gpu": P,
                "parallelism": f"permutation ranges over {world} rank(s), all-reduce of int64 moments",
            },
            "roofline": {
                "kernel": "+".join(cnt_name) or "nhood_count",
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "traffic": traffic,
                "launches": launches,
                "avg_launch_ms": ms / max(launches, 1),
                "perms_per_launch": perms_per_launch,
                "algorithmic_bytes_per_perm": b_gather,
                "note": "algorithmic bytes / HIP-event time; >1 is possible because one pass over the CSR serves 16-32 "
                "permutations (see DESIGN.md); pipeline figure below prices the whole permutation (30.0 MB) against "
                "the sum of all kernels",
            },
            "pipeline": {
                "algorithmic_bytes_per_perm": b_perm,
                "gpu_ms_all_kernels": gpu_ms,
                "achieved_GBps": b_perm * args.steps * P / (gpu_ms * 1e-3) / 1e9 if gpu_ms > 0 else None,
                "frac_of_hbm_peak": b_perm * args.steps * P / (gpu_ms * 1e-3) / HBM_PEAK if gpu_ms > 0 else None,
                "wall_frac_of_hbm_peak": b_perm * (total_perms / world) / elapsed / HBM_PEAK,
                "avg_kernel_ms": kern_ms,
                "time_share": {k: round(v[1] / gpu_ms, 3) for k, v in kernels.items() if v[0] > 0 and gpu_ms > 0},
                "note": "nhood_shuffle (label generation, VALU-bound: ~45 packed-16 ops per spot and permutation, no HBM "
                "roofline applies) and nhood_count (the CSR gather the roofline object prices) share the step",
            },
        }
        if secondary is not None:
            out["secondary"] = secondary
        if world == 1 and not args.no_numpy_leg:  # bonus leg: the same test with numpy's own PCG64 streams reproduced bit for bit on the GPU
            from squidpy_amd._utils import pcg64_states

            legs = {}
            for n_exact in (1000, 8192):  # Squidpy's default n_perms, and a throughput-sized batch
                states = pcg64_states(0, n_exact)
                plan.run_pcg64(states, shift)  # warm-up at full size: the workspaces are allocated (and first touched) here
                t1 = time.perf_counter()
                plan.run_pcg64(states, shift)
                legs[n_exact] = n_exact / (time.perf_counter() - t1)
            out["numpy_stream_mode"] = {
                "value": legs[8192],
                "unit": "permutations/s",
                "at_n_perms_1000": legs[1000],
                "note": "rng='numpy': PCG64 + Generator.shuffle reproduced on the device, one wave per permutation (LCG "
                "jump-ahead draws, parallel swaps, exact replay of conflicting steps); z-scores equal Squidpy's for the same "
                "seed bit for bit",
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(adj, labels)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
