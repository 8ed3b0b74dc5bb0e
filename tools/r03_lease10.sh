#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_lease10
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_cooccur_gpu.py tests/test_full_size_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "cooccur or occurrence or config4" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 tools/ubench_ds_mix.bin $OUT/r03_ubench_ds_mix.json
timeout 600 python bench.py --no-secondary --no-numpy-leg --no-cpu-baseline --steps 3 > $OUT/bench_legs.json 2> $OUT/bench_legs.err
python - $OUT/bench_legs.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d.get("legs", {}).items():
    r = v.get("roofline") or {}
    print(k, v.get("value"), "wall", v.get("wall_s"), "kernel_ms", v.get("kernel_ms"), "frac", r.get("frac"))
PY
