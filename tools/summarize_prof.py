"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON report."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
rep = {}
print("== rocprofv3 --kernel-trace --stats ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(f"{'kernel':60s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    rep["kernel_stats"] = []
    for r in rows[:15]:
        name = r["Name"].split("(")[0][-58:]
        print(f"{name:60s} {r['Calls']:>8s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
        rep["kernel_stats"].append({"name": name, "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                                    "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])})
for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    print(f"== rocprofv3 --pmc {counter} (per dispatch, KiB units as reported) ==")
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0][-58:]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    rep[counter] = {}
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} dispatches={n:6d} total={v:16.1f} per_dispatch={v/max(n,1):14.1f}")
        rep[counter][k] = {"dispatches": n, "total": v, "per_dispatch": v / max(n, 1)}
json.dump(rep, open(os.path.join(out, "summary.json"), "w"), indent=1)
# HBM traffic of the CSR-gather kernel per launch: FETCH_SIZE is reported in KiB and, on gfx950, counts half the bytes of
# wide coalesced reads (MI355X_MICROARCH.md §HBM) -> doubled; WRITE_SIZE (KiB) taken as is.
def per_dispatch(counter, key):
    for k, v in rep.get(counter, {}).items():
        if key in k:
            return v["per_dispatch"], v["dispatches"]
    return None, 0
f, nf = per_dispatch("FETCH_SIZE", "k_count")
w, nw = per_dispatch("WRITE_SIZE", "k_count")
if f is not None and w is not None:
    traffic = {"nhood_count_bytes_per_launch": (2.0 * f + w) * 1024.0, "fetch_KiB_per_launch_raw": f, "write_KiB_per_launch_raw": w,
               "dispatches": nf, "note": "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024; separate --pmc passes; bench invoked with --perms-per-step 2048"}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("traffic", traffic)
