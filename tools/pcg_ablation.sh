#!/bin/bash
# Ablation of the numpy-stream replay kernel k_pcg_apply_claims: timing experiments, wrong results by design.
# `bash tools/pcg_ablation.sh build` in the build container compiles the variants (they travel with the snapshot as
# squidpy_amd/csrc/libsqgr_abl<bits>.so: -DSQGR_PCG_ABLATE bits 1 = drains without their rounds, 2 = ranges neither loaded nor stored,
# 4 = no record claims or swaps anything), plus the instrumented build; `bash tools/pcg_ablation.sh` on the GPU box times them and the
# product kernels old and new -> gpurun_out/r06_pcg_replay_ablation.txt
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
if [ "${1:-run}" = "build" ]; then
  for a in 1 2 3 4 7; do PCG_EXTRA="-DSQGR_PCG_ABLATE=$a" PCG_TAG=abl$a bash tools/pcg_profile.sh build || exit 1; done
  bash tools/pcg_profile.sh build
  exit $?
fi
mkdir -p gpurun_out
{
  echo "numpy-stream kernels at 1e6 positions: round 6 (generator: 128 draws per trip; replay: exact claims + deferred queue) against rounds 4-5 (bucket-tags)"
  for P in 8192 4096 1000; do python tools/pcg_bucket_time.py $P bucket,bucket-tags 2>&1 | grep -v "^$" | cut -c1-330; done
  echo
  echo "occupancy of the generators (SQGR_PCG_LDS_PAD bytes of LDS more per wavefront), 8192 permutations"
  for pad in 0 4096 8192 16384; do echo "pad $pad"; SQGR_PCG_LDS_PAD=$pad python tools/pcg_bucket_time.py 8192 bucket,bucket-tags 2>&1 | grep "perms/s" | cut -c1-140; done
  echo
  echo "ablation builds of k_pcg_apply_claims, 4096 permutations (apply time = nhood_pcg64_shuffle_apply; WRONG RESULTS BY DESIGN)"
  echo "  abl1 = drains without their rounds | abl2 = ranges neither loaded nor stored | abl3 = both | abl4 = no record claims or swaps (and so no drains) | abl7 = all"
  for t in abl1 abl2 abl3 abl4 abl7; do echo $t; PCG_TAG=$t bash tools/pcg_profile.sh 4096 2>&1 | grep "perms/s" | cut -c1-200; done
  echo product; python tools/pcg_bucket_time.py 4096 bucket 2>&1 | grep "perms/s" | cut -c1-200
  echo
  echo "instrumented build (shader clocks of wave 0 per permutation; the instrumentation itself slows the kernel ~3.5x: read the COUNTS, not the clocks)"
  bash tools/pcg_profile.sh 2048 2>&1 | grep "pcgq profile" | tail -1
  echo
  echo "wave kernel against the pipeline at small and mid-size arrays (pcg_use_bucket switches at 65 536 positions)"
  python tools/pcg_threshold_time.py 2>&1 | grep -v "^$"
} > gpurun_out/r06_pcg_replay_ablation.txt 2>&1
tail -5 gpurun_out/r06_pcg_replay_ablation.txt
