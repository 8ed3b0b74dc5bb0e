#!/bin/bash
# final lease of a round: the whole GPU suite, the driver's smoke, the default bench line and the profiles it is priced with
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_final
mkdir -p $OUT
cd $REPO
( time timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_gpu.time; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r04 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log | cut -c1-200
cp gpurun_out/prof_r04/r04_*.json gpurun_out/prof_r04/r04_*.txt profiles/ 2>/dev/null
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
cp $OUT/bench_detail.json profiles/r04_bench_detail.json 2>/dev/null; tail -1 $OUT/bench.json > profiles/r04_bench.json
python - $OUT/bench_detail.json $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
line = open(sys.argv[2]).read().strip().splitlines()[-1]
print("final line bytes:", len(line))
r = d["roofline"]
print("value", round(d["value"]), "roofline", r["bound"], r["achieved"], r["frac"], "fabric", r.get("fabric_frac"), "alg", r.get("algorithmic_frac"), "pmc:", d.get("pmc_profile"))
print("moran", round(d["secondary"]["value"]), d["secondary"]["roofline"].get("frac"))
for k, v in d.get("legs", {}).items():
    rr = v.get("roofline") or {}
    print(k, v.get("value"), v.get("unit"), "kernel_ms", v.get("kernel_ms"), "frac", rr.get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"), v.get("moran"), v.get("geary"))
n = d["numpy_stream_mode"]; print("numpy", n["value"], n["at_n_perms_1000"], n["roofline"]["frac"], n["roofline"].get("traffic_MB_per_perm"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"].get("value"), "emulated", d.get("emulated_ranks", {}).get("shard_seconds"))
PY
