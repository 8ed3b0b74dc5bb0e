#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for T in 0,0,0 0,8,256 0,8,320 0,16,256 0,8,128 0,24,160 0,0,0; do
echo -n "tune $T: "
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-legs --no-secondary --no-numpy-leg --emulate-ranks 0 --tune $T 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernels']['nhood_shuffle']['ms'])"
done
