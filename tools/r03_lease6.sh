#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_lease6
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_nhood_gpu.py tests/test_ripley_gpu.py -x -q -m gpu -k "interaction or weighted or ripley or Ripley or ball_tree or knn" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python tools/sparse_time.py > $OUT/sparse_time.log 2>&1; grep -v "^$" $OUT/sparse_time.log | head -60 | cut -c1-400
