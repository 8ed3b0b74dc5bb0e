#!/bin/bash
# final lease of a round: the whole GPU suite, the driver's smoke, the default bench line and the profiles it is priced with
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_final
mkdir -p $OUT
cd $REPO
( time timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_gpu.time; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log | cut -c1-200
cp gpurun_out/prof_r03/r03_*.json gpurun_out/prof_r03/r03_*.txt profiles/ 2>/dev/null
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "roofline", r["bound"], r["achieved"], r["frac"], "pmc:", d.get("pmc_profile"))
print("issue", {k: (r["issue_limits"].get(k)) for k in ("bound", "frac")}, "lds_atomic", (r["issue_limits"].get("lds_atomic") or {}).get("frac"))
print("moran", round(d["secondary"]["value"]), d["secondary"]["roofline"].get("frac"), d["secondary"]["roofline"].get("hbm_frac_of_peak"))
for k, v in d.get("legs", {}).items():
    rr = v.get("roofline") or {}
    print(k, v.get("value"), v.get("unit"), "kernel_ms", v.get("kernel_ms"), "frac", rr.get("frac"), "pmc", (rr.get("pmc") or {}).get("valu_per_64_unordered_pairs"), "cpu", (v.get("cpu_baseline") or {}).get("value"), v.get("moran"), v.get("geary"))
n = d["numpy_stream_mode"]; print("numpy", n["value"], n["roofline"]["frac"], n["roofline"].get("traffic_frac"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"].get("value"))
PY
