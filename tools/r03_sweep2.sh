#!/bin/bash
# round-3 lease 2: pair-per-row count kernel (parity + timing), shuffle exact-route ceiling probe, address-form micro-benchmark
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_sweep2
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_nhood_gpu.py tests/test_full_size_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "nhood" > $OUT/pytest.log 2>&1
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
run pair SQGR_X=0
run quad SQGR_COUNT_PAIR=0
run pair_probe_noatomics SQGR_COUNT_DEBUG=1
run pair_probe_nogathers SQGR_COUNT_DEBUG=2
run shuffle_probe_noexact SQGR_SHUFFLE_DEFER=2
run pair_dirichlet SQGR_X=0 
timeout 300 tools/ubench_count_shape.bin > $OUT/ubench_count_shape.json 2> $OUT/ubench_count_shape.err
cat $OUT/ubench_count_shape.json
