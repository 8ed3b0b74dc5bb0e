#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for LOGS in 16 15 14; do
echo "== LOGS=$LOGS"
SQGR_PCG_BUCKET_LOGS=$LOGS timeout 600 python tools/numpy_call_breakdown.py 2>&1 | tail -2 | cut -c1-420
done
