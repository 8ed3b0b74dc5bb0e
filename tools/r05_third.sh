#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for PACK in 1 0; do
echo "== PACK=$PACK"
SQGR_COUNT_PASS_PACK=$PACK timeout 900 python -m pytest tests/test_nhood_gpu.py -m gpu -x -q -k "lds_pass_kernel or all_cluster_count_regimes or skewed or numpy_streams" 2>&1 | tail -3
SQGR_COUNT_PASS_PACK=$PACK timeout 600 python tools/nhood_k_sweep.py 1000 2560 --K=30 --K=64 --K=100 --K=150 --K=200 --K=256 2>&1 | cut -c1-330
done
