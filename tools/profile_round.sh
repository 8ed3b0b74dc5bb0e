#!/bin/bash
# rocprofv3 evidence for bench.py's JSON line, taken on the GPU box in ONE lease:
#   1. --kernel-trace --stats                      -> profiles/<tag>_rocprofv3_summary.txt
#   2. separate --pmc passes (never combined with a trace): FETCH_SIZE | WRITE_SIZE | SQ instruction counts |
#      SQ LDS/busy counters | TCC hit/miss            -> profiles/<tag>_counters.json (per kernel, per launch)
# bench.py reads <tag>_counters.json for `traffic` and the VALU instruction counts and uses it only when the workload key
# (spots, list edges, permutations per launch) equals its own.
#   usage: tools/profile_round.sh [tag]          (default tag r02; outputs under gpurun_out/prof_<tag>/ and profiles/)
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-numpy-leg --no-legs"
PMC="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-numpy-leg --no-legs"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMC > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMC > $OUT/write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sqa -- $PMC > $OUT/sqa.log 2>&1
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sqb -- $PMC > $OUT/sqb.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $PMC > $OUT/tcc.log 2>&1
timeout 900 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/tcp -- $PMC > $OUT/tcp.log 2>&1
python $REPO/tools/summarize_round.py $OUT $TAG "$CMD" "$PMC" && cp $OUT/${TAG}_rocprofv3_summary.txt $OUT/${TAG}_counters.json $REPO/profiles/
tail -2 $OUT/stats.log | cut -c1-600
