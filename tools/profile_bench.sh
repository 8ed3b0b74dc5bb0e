#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the bench command.
# Summaries land in gpurun_out/ (merged back); copy the ones to be judged into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-numpy-leg ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
SMALL="python $REPO/bench.py --steps 1 --warmup 1 --perms-per-step 2048 --no-cpu-baseline --no-secondary --no-numpy-leg ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $SMALL > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $SMALL > $OUT/write.log 2>&1
python $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
