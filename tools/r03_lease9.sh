#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_lease9
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_ripley_gpu.py tests/test_dist2_gpu.py tests/test_full_size_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "ripley or Ripley or pair_counts or two_ranks or config4" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python bench.py --no-secondary --no-numpy-leg --no-cpu-baseline --steps 3 > $OUT/bench_legs.json 2> $OUT/bench_legs.err
python - $OUT/bench_legs.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d.get("legs", {}).items():
    r = v.get("roofline") or {}
    print(k, v.get("value"), "wall", v.get("wall_s"), "kernel_ms", v.get("kernel_ms"), "frac", r.get("frac"))
PY
tail -3 $OUT/bench_legs.err | cut -c1-300
