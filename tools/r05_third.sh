#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nhood_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "not full_size" 2>&1 | tail -5
timeout 600 python tools/nhood_k_sweep.py 1000 2560 --K=30 --K=64 --K=100 --K=150 --K=200 --K=203 --K=256 2>&1 | cut -c1-330
