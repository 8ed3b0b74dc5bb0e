#!/bin/bash
# Developer tool: where the numpy-stream replay kernel (k_pcg_apply_claims) spends its shader clocks.  `bash tools/pcg_profile.sh build`
# in the build container compiles csrc/sqgr_pcg.hip with -DSQGR_PCG_PROFILE into squidpy_amd/csrc/libsqgr_prof.so (travels with the
# snapshot; PCG_EXTRA="-DSQGR_PCG_ABLATE=<bits>" PCG_TAG=<name>: the timing experiments of the kernel, wrong results by design); `bash tools/pcg_profile.sh [n_perms]` on the GPU box prints the per-section clocks and the chunk / drain counts.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
VAR=squidpy_amd/csrc/libsqgr_${PCG_TAG:-prof}.so
if [ "${1:-run}" = "build" ]; then
  OBJS=""
  for f in squidpy_amd/csrc/*.hip; do
    if [ "$(basename $f)" = "sqgr_pcg.hip" ]; then
      o=/tmp/prof_sqgr_pcg.o
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${PCG_EXTRA:--DSQGR_PCG_PROFILE} -c $f -o $o -Wall -Wno-unused-function || exit 1
    else
      o=${f%.hip}.o
    fi
    OBJS="$OBJS $o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VAR $OBJS && echo built $VAR
  exit $?
fi
SQGR_LIBRARY=$REPO/$VAR python tools/pcg_bucket_time.py ${1:-2048} bucket 2>&1 | grep -v "^$"
