#!/bin/bash
# reduced final lease (GPU budget nearly spent): the autocorr GPU tests, the few-permutation timings, then the profiles + bench record
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_final
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_autocorr_gpu.py -x -q > $OUT/pytest_autocorr.log 2>&1; rc=$?; tail -3 $OUT/pytest_autocorr.log
[ $rc -ne 0 ] && { echo "autocorr tests failed: stopping"; exit 1; }
timeout 120 python tools/autocorr_small_time.py > $OUT/small_time.log 2>&1; grep -E "P=(50|100|256) lds-split" $OUT/small_time.log | cut -c1-200
bash tools/profile_round.sh r04 > $OUT/profile_round.log 2>&1; tail -2 $OUT/profile_round.log | cut -c1-160
( time timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
python - $OUT/bench_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"]), "moran", round(d["secondary"]["value"]), "pmc:", d.get("pmc_profile"))
for k, v in d.get("legs", {}).items():
    print(k, v.get("value"), v.get("unit"), v.get("moran"), v.get("geary"))
PY
