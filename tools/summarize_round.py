"""Condense the rocprofv3 output of tools/profile_round.sh: kernel-trace statistics -> <tag>_rocprofv3_summary.txt, PMC passes ->
<tag>_counters.json with PER-LAUNCH averages per kernel (what bench.py combines with its own HIP-event timings), the
FETCH_SIZE / WRITE_SIZE calibration -> <tag>_fetch_calibration.json.  <tag>_counters.json carries the fingerprint of the kernel
sources it was taken from (squidpy_amd._build.source_fingerprint): bench.py refuses it for any other build."""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squidpy_amd._build import source_fingerprint  # noqa: E402

out, tag, cmd, pmc_cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
legs_cmd = sys.argv[5] if len(sys.argv) > 5 else ""


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "").strip()


def stats_table(sub: str, limit: int = 24) -> list[str]:
    rows_out = []
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        rows_out.append(f"{'kernel':64s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
        for r in rows[:limit]:
            rows_out.append(f"{short(r['Name'])[-64:]:64s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
    return rows_out


def bench_line(log: str):
    """The FULL record of a profiled bench.py run (`--detail-out <out>/<name>_detail.json`; the stdout line is the compact one)."""
    name = {"stats.log": "stats", "sqa.log": "pmc", "legs_stats.log": "legs", "legs_sqa.log": "legs", "npy_fetch.log": "npy"}.get(log, log)
    try:
        with open(os.path.join(out, f"{name}_detail.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


lines = [f"== rocprofv3 --kernel-trace --stats -- {cmd} =="] + stats_table("stats")
bench = bench_line("stats.log")
if bench:
    lines.append("== the same run's JSON line (HIP events on the library's stream) ==")
    lines.append("value %.0f %s; avg_kernel_ms %s" % (bench["value"], bench["unit"], json.dumps(bench["pipeline"]["avg_kernel_ms"])))
    if bench.get("secondary"):
        lines.append("moran: %.0f genes/s; perm_dot avg launch %.3f ms" % (bench["secondary"]["value"], bench["secondary"]["roofline"]["avg_launch_ms"]))
legs_bench = bench_line("legs_stats.log")
if legs_cmd:
    lines += [f"== rocprofv3 --kernel-trace --stats -- {legs_cmd} =="] + stats_table("legs_stats", 16)
    if legs_bench and legs_bench.get("legs"):
        for k, v in legs_bench["legs"].items():
            if "value" in v:
                km, ws = v.get("kernel_ms"), v.get("wall_s")
                km = f"{km:.2f}" if isinstance(km, (int, float)) else "-"
                ws = f"{ws:.3f}" if isinstance(ws, (int, float)) else "-"
                lines.append(f"{k}: {v['value']:.4g} {v['unit']} (kernel {km} ms, wall {ws} s)")
open(os.path.join(out, f"{tag}_rocprofv3_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))


def collect(subs) -> dict:
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0, None, None]))  # kernel -> counter -> [dispatches, total, first dispatch id, its value]
    for sub in subs:
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                did = int(r["Dispatch_Id"])
                if a[2] is None or did < a[2]:
                    a[2], a[3] = did, float(r["Counter_Value"])
    return agg


def section(agg, match, after_first: bool = False) -> dict:
    """Per-launch averages per kernel.  after_first: the first dispatch of a kernel is the bench's small warm-up call, the others
    are the timed ones — report TOTALS over the others (`*_timed_total`), which is what bench.py's timed region ran."""
    ks = {}
    for k, cs in agg.items():
        if not match(k):
            continue
        rec = {"dispatches": max(v[0] for v in cs.values())}
        for c, (n, tot, _, first) in cs.items():
            scale = 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 1.0  # reported in KiB; FETCH_SIZE still to be doubled (see <tag>_fetch_calibration.json)
            name = c + "_bytes" if scale != 1.0 else c
            rec[name] = tot / max(n, 1) * scale
            if after_first:
                rec[name + "_timed_total"] = (tot - (first or 0.0)) * scale
        ks[k] = rec
    return ks


agg = collect(("fetch", "write", "sqa", "sqb", "tcc", "tcp"))
pmc_bench = bench_line("sqa.log") or bench
sec = (pmc_bench or {}).get("secondary") or {}
gea = ((pmc_bench or {}).get("legs") or {}).get("geary_c") or (pmc_bench or {}).get("geary_c") or {}
autocorr_match = lambda k: "k_perm_dot" in k or "k_spmv" in k or "k_perm_ind" in k or "k_bucket" in k or "k_perm_stats" in k
rep = {
    "note": "per-LAUNCH averages of rocprofv3 --pmc counters, separate passes (tools/profile_round.sh); FETCH_SIZE/WRITE_SIZE converted "
    "from KiB to bytes, FETCH_SIZE NOT yet doubled (bench.py applies the calibrated factor 2)",
    "source_sha16": source_fingerprint(),
    "command": pmc_cmd,
    "nhood": {"workload": (pmc_bench or {}).get("roofline", {}).get("workload_key"),
              "kernels": section(agg, lambda k: "k_count" in k or "k_shuffle" in k or "k_reduce" in k or "k_keygen" in k or "k_finalize" in k)},
    # Moran and Geary launches of the same kernel template differ by their template argument: k_perm_dot_lds<0> Moran (and Geary on a
    # graph with ONE row sum), <1> / <2> / <3> Geary (row sums in LDS / in classes / exception lists); the gather kernel k_perm_dot<false> Moran, <true> Geary
    "moran": {"workload": sec.get("roofline", {}).get("workload_key"),
              "kernels": section(agg, lambda k: autocorr_match(k) and "<true>" not in k and "<1>" not in k and "<2>" not in k and "<3>" not in k)},
    "geary": {"workload": gea.get("roofline", {}).get("workload_key"),
              "kernels": section(agg, lambda k: autocorr_match(k) and "<false>" not in k and "<0>" not in k)},
}
lagg = collect(("legs_sqa", "legs_fetch", "legs_write", "legs_tcp"))
lb = bench_line("legs_sqa.log") or legs_bench or {}
rep["legs"] = {"command": legs_cmd, "workload": (((lb.get("legs") or {}).get("co_occurrence") or {}).get("roofline") or {}).get("workload_key"),
               "kernels": section(lagg, lambda k: "k_cooccur" in k or "k_pair_hist" in k or "k_knn" in k, after_first=True),
               # the pass kernel of the nhood_K64 / K100 / K200 legs: one template instantiation each (<lanes per edge, atomics per edge and
               # lane, self loops, row split, packed list>); a leg = a 64-permutation warm-up launch + the launches of 10 000 permutations
               "nhood_pass_kernels": section(lagg, lambda k: "k_count_pass" in k, after_first=True)}
nagg = collect(("npy_fetch", "npy_write"))
nb = bench_line("npy_fetch.log") or {}
rep["numpy"] = {"workload": (((nb.get("numpy_stream_mode") or {}).get("roofline")) or {}).get("workload_key"),
                "kernels": section(nagg, lambda k: "k_pcg_" in k or "k_rows_to_" in k or "k_columns_to_slab" in k, after_first=True)}
# ---- probe variants of the count kernel (tools/count_probe.py under SQGR_COUNT_DEBUG)
probes = {}
for dbg, label in ((0, "full_kernel"), (1, "no_atomics_real_gathers (DBG=1)"), (2, "no_row_gathers_real_labels_through_coalesced_loads (DBG=2)"), (7, "valu_skeleton (DBG=7)")):
    try:
        txt = open(os.path.join(out, f"count_probe_{dbg}.log")).read()
        import ast
        probes[label] = {k: v for k, v in ast.literal_eval(txt[txt.index("{"): txt.index("}") + 1]).items() if k.startswith("nhood_count")}
    except (OSError, ValueError, SyntaxError):
        pass
if probes:
    json.dump({"tool": "tools/count_probe.py with SQGR_COUNT_DEBUG (csrc/sqgr_nhood.hip k_count<..., DBG>), 1e6-spot hex grid, 30 clusters, 4096 permutations, MI355X",
               "unit": "ms per launch of the count kernel (HIP events; 2560 permutations per full launch)", "variants": probes},
              open(os.path.join(out, f"{tag}_count_probes.json"), "w"), indent=1)
json.dump(rep, open(os.path.join(out, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps({k: (list(v["kernels"]) if isinstance(v, dict) and "kernels" in v else v) for k, v in rep.items() if k != "note"}, indent=1)[:3000])

# ---- FETCH_SIZE / WRITE_SIZE calibration
cagg = collect(("calib_fetch", "calib_write"))
if cagg:
    true_bytes = 512 << 20
    cal = {"tool": "tools/ubench_fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (tools/profile_round.sh)", "bytes_moved_per_launch": true_bytes, "kernels": {}}
    for k, cs in sorted(cagg.items()):
        for c, (n, tot, _, _) in cs.items():
            cal["kernels"].setdefault(k, {})[c + "_reported_bytes"] = tot / n * 1024
            cal["kernels"][k][c + "_ratio_to_true"] = tot / n * 1024 / true_bytes
    cal["conclusion"] = ("FETCH_SIZE reports 0.5 of the bytes read for 16 B/lane streaming reads, for the count kernel's 4 B/lane quad-per-row gathers and for its "
                         "8 B/lane list loads; WRITE_SIZE reports 1.0 for 16-byte rows and 4-byte words: traffic = 2*FETCH_SIZE + WRITE_SIZE")
    json.dump(cal, open(os.path.join(out, f"{tag}_fetch_calibration.json"), "w"), indent=1)
