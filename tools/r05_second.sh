#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_nhood_gpu.py -m gpu -x -q -k "lds_pass_kernel or all_cluster_count_regimes" 2>&1 | tail -2
done
python tools/dbg/pass_mismatch.py 2>&1 | tail -9
bash tools/pmc_pass.sh "30 64 100 200" > /dev/null 2>&1
cat gpurun_out/pmc_pass.txt
