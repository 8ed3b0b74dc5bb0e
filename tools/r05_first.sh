#!/bin/bash
# round 5, first lease: the pass kernel's parity tests, the two-rank bench through its own launcher, the K sweep
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nhood_gpu.py -m gpu -x -q -k "lds_pass_kernel or all_cluster_count_regimes or skewed or more_than_256" > gpurun_out/r05_first_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05_first_pytest.log
tail -5 gpurun_out/r05_first_pytest.log
timeout 600 python -m pytest tests/test_bench_launcher_gpu.py -m gpu -x -q > gpurun_out/r05_first_launcher.log 2>&1
echo "launcher rc=$?" >> gpurun_out/r05_first_launcher.log
tail -5 gpurun_out/r05_first_launcher.log
timeout 600 python tools/nhood_k_sweep.py 1000 2560 > gpurun_out/nhood_k_sweep.jsonl 2> gpurun_out/nhood_k_sweep.err
tail -20 gpurun_out/nhood_k_sweep.jsonl | cut -c1-400
