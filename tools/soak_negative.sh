#!/bin/bash
# Negative control of tests/test_soak_gpu.py: libsqgr built WITHOUT the explicit `s_waitcnt lgkmcnt(0)` in front of the
# flush barrier of k_count / k_count_pass (-DSQGR_DEBUG_NO_FLUSH_WAIT) must FAIL the soak tests.  Build the variant in the build
# container (`bash tools/soak_negative.sh build`: squidpy_amd/csrc/libsqgr_nowait.so travels with the snapshot), run on the GPU box
# (`bash tools/soak_negative.sh`): writes gpurun_out/soak_negative.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
VAR=squidpy_amd/csrc/libsqgr_nowait.so
if [ "${1:-run}" = "build" ]; then
  OBJS=""
  for f in squidpy_amd/csrc/*.hip; do
    o=/tmp/nowait_$(basename ${f%.hip}).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSQGR_DEBUG_NO_FLUSH_WAIT -c $f -o $o -Wall -Wno-unused-function &
    OBJS="$OBJS $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR $OBJS && echo built $VAR
  exit $?
fi
mkdir -p gpurun_out
{
  echo "soak tests against libsqgr built with -DSQGR_DEBUG_NO_FLUSH_WAIT (the s_waitcnt in front of the flush barrier removed): expected to FAIL"
  SQGR_LIBRARY=$REPO/$VAR timeout 900 python -m pytest tests/test_soak_gpu.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | grep -v '^$' | tail -25
  echo "---- and the product build: expected to PASS"
  timeout 900 python -m pytest tests/test_soak_gpu.py -q -m gpu --no-header -p no:cacheprovider 2>&1 | tail -4
} > gpurun_out/soak_negative.txt 2>&1
cat gpurun_out/soak_negative.txt | cut -c1-400
