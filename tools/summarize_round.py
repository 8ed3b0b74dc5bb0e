"""Condense the rocprofv3 output of tools/profile_round.sh: kernel-trace statistics -> <tag>_rocprofv3_summary.txt, PMC passes ->
<tag>_counters.json with PER-LAUNCH averages per kernel (what bench.py combines with its own HIP-event timings)."""
import csv, glob, json, os, sys
from collections import defaultdict

out, tag, cmd, pmc_cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "").strip()


lines = [f"== rocprofv3 --kernel-trace --stats -- {cmd} =="]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    lines.append(f"{'kernel':64s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:24]:
        lines.append(f"{short(r['Name'])[-64:]:64s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
# the bench's own JSON line of the traced run (HIP-event averages to compare with)
bench = None
for ln in open(os.path.join(out, "stats.log"), errors="replace"):
    if ln.startswith('{"metric"'):
        bench = json.loads(ln)
if bench:
    lines.append("== the same run's JSON line (HIP events on the library's stream) ==")
    lines.append("value %.0f %s; avg_kernel_ms %s" % (bench["value"], bench["unit"], json.dumps(bench["pipeline"]["avg_kernel_ms"])))
    if bench.get("secondary"):
        lines.append("moran: %.0f genes/s; perm_dot avg launch %.3f ms" % (bench["secondary"]["value"], bench["secondary"]["roofline"]["avg_launch_ms"]))
open(os.path.join(out, f"{tag}_rocprofv3_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))  # kernel -> counter -> [dispatches, total]
for sub in ("fetch", "write", "sqa", "sqb", "tcc", "tcp"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
pmc_bench = None
for ln in open(os.path.join(out, "sqa.log"), errors="replace"):
    if ln.startswith('{"metric"'):
        pmc_bench = json.loads(ln)
pmc_bench = pmc_bench or bench


def section(match) -> dict:
    ks = {}
    for k, cs in agg.items():
        if not match(k):
            continue
        rec = {"dispatches": max(v[0] for v in cs.values())}
        for c, (n, tot) in cs.items():
            per = tot / max(n, 1)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                rec[c + "_bytes"] = per * 1024.0  # reported in KiB; FETCH_SIZE still to be doubled (MI355X_MICROARCH.md §HBM)
            else:
                rec[c] = per
        ks[k] = rec
    return ks


rep = {
    "note": "per-LAUNCH averages of rocprofv3 --pmc counters, separate passes (tools/profile_round.sh); FETCH_SIZE/WRITE_SIZE converted "
    "from KiB to bytes, FETCH_SIZE NOT yet doubled (bench.py applies the gfx950 correction of MI355X_MICROARCH.md §HBM)",
    "command": pmc_cmd,
    "nhood": {"workload": (pmc_bench or {}).get("roofline", {}).get("workload_key"), "kernels": section(lambda k: "k_count" in k or "k_shuffle" in k or "k_reduce" in k or "k_keygen" in k or "k_finalize" in k)},
    "moran": {"workload": ((pmc_bench or {}).get("secondary") or {}).get("roofline", {}).get("workload_key"), "kernels": section(lambda k: "k_perm_dot" in k or "k_spmv" in k or "k_perm_ind" in k or "k_bucket" in k)},
}
json.dump(rep, open(os.path.join(out, f"{tag}_counters.json"), "w"), indent=1)
print(json.dumps(rep["nhood"], indent=1)[:3000])
