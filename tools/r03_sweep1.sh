#!/bin/bash
# round-3 lease 1: parity of the deferred exact route of the label shuffle, stream/LDS-claim sweep, micro-benchmarks
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_sweep1
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_nhood_gpu.py tests/test_ligrec_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-legs --no-numpy-leg"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - "$name" $OUT/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d["value"]), d["pipeline"]["avg_kernel_ms"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run default SQGR_X=0
run nodefer SQGR_SHUFFLE_DEFER=0
run streams2 SQGR_NHOOD_STREAMS=2
run streams2_lds81 SQGR_NHOOD_STREAMS=2 SQGR_COUNT_LDS_KB=81
run lds81 SQGR_COUNT_LDS_KB=81
run streams2_lds81_nodefer SQGR_NHOOD_STREAMS=2 SQGR_COUNT_LDS_KB=81 SQGR_SHUFFLE_DEFER=0
timeout 300 tools/ubench_count_shape.bin > $OUT/ubench_count_shape.json 2> $OUT/ubench_count_shape.err
cat $OUT/ubench_count_shape.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $REPO/tools/ubench_fetch_calib.bin > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $REPO/tools/ubench_fetch_calib.bin > $OUT/calib_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/calib_stats -- $REPO/tools/ubench_fetch_calib.bin > $OUT/calib_stats.log 2>&1
python - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: [0, 0.0])
for sub in ("calib_fetch", "calib_write"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, tot) in sorted(agg.items()):
    print(k, "launches", n, "KiB per launch", tot / n, "ratio to 512 MiB", tot / n * 1024 / (512 << 20))
for f in glob.glob(os.path.join(out, "calib_stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, "us", (512 << 20) / float(r["AverageNs"]), "GB/s")
PY
