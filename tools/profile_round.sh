#!/bin/bash
# rocprofv3 evidence for bench.py's JSON line, taken on the GPU box in ONE lease:
#   1. --kernel-trace --stats of the bench command       -> profiles/<tag>_rocprofv3_summary.txt
#   2. separate --pmc passes (never combined with a trace): FETCH_SIZE | WRITE_SIZE | SQ instruction counts |
#      SQ LDS/busy counters | TCC hit/miss | L1 (TCP/TD/TA)  -> profiles/<tag>_counters.json (per kernel, per launch),
#      for the nhood + Moran + Geary kernels and (FETCH/WRITE/SQ instruction counts) for the config-4 legs
#   3. FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench_fetch_calib.hip) -> profiles/<tag>_fetch_calibration.json
#   4. micro-benchmarks: float64 issue rates, gathers beside LDS atomics
# <tag>_counters.json is stamped with the kernel sources' fingerprint; bench.py uses it only for that build and for the
# workload key (spots, list edges, permutations per launch) it was taken on.
#   usage: tools/profile_round.sh [tag]          (default tag r05; outputs under gpurun_out/prof_<tag>/ and profiles/)
# bench.py's final stdout line is the compact one: the summaries read the FULL record each profiled run writes (--detail-out).
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-numpy-leg --no-legs --emulate-ranks 0 --detail-out $OUT/stats_detail.json"
PMC="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-numpy-leg --no-legs --emulate-ranks 0 --detail-out $OUT/pmc_detail.json"
LEGS="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-numpy-leg --no-secondary --no-short-radii --emulate-ranks 0 --detail-out $OUT/legs_detail.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMC > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMC > $OUT/write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sqa -- $PMC > $OUT/sqa.log 2>&1
timeout 900 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sqb -- $PMC > $OUT/sqb.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- $PMC > $OUT/tcc.log 2>&1
timeout 900 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/tcp -- $PMC > $OUT/tcp.log 2>&1
# the config-4 legs (co-occurrence, Ripley L / G): kernel trace + instruction counts + traffic
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/legs_stats -- $LEGS > $OUT/legs_stats.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/legs_sqa -- $LEGS > $OUT/legs_sqa.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/legs_fetch -- $LEGS > $OUT/legs_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/legs_write -- $LEGS > $OUT/legs_write.log 2>&1
timeout 900 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/legs_tcp -- $LEGS > $OUT/legs_tcp.log 2>&1
# the numpy-stream shuffle (rng="numpy"): traffic of k_pcg_shuffle_wave
NPY="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-legs --emulate-ranks 0 --detail-out $OUT/npy_detail.json"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/npy_fetch -- $NPY > $OUT/npy_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/npy_write -- $NPY > $OUT/npy_write.log 2>&1
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts, in the nhood kernels' access patterns
if [ -x $REPO/tools/ubench_fetch_calib.bin ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $REPO/tools/ubench_fetch_calib.bin > $OUT/calib_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- $REPO/tools/ubench_fetch_calib.bin > $OUT/calib_write.log 2>&1
fi
# issue-rate / LDS / L2-gather ceilings of THIS lease (bench.py prices with the current tag's files)
[ -x $REPO/tools/ubench_ops.bin ] && timeout 300 $REPO/tools/ubench_ops.bin $OUT/${TAG}_ubench_ops.json > $OUT/ubench_ops.log 2>&1
[ -x $REPO/tools/ubench_lds_read.bin ] && timeout 300 $REPO/tools/ubench_lds_read.bin > $OUT/${TAG}_ubench_lds_read.json 2> $OUT/ubench_lds_read.err
[ -x $REPO/tools/ubench_gather.bin ] && timeout 300 $REPO/tools/ubench_gather.bin > $OUT/${TAG}_ubench_gather.json 2> $OUT/ubench_gather.err
# probe variants of the count kernel (what bounds it): full / no atomics / no row gathers / VALU skeleton
for dbg in 0 1 2 7; do SQGR_COUNT_DEBUG=$dbg timeout 300 python $REPO/tools/count_probe.py > $OUT/count_probe_$dbg.log 2>&1; done
[ -x $REPO/tools/ubench_f64.bin ] && timeout 300 $REPO/tools/ubench_f64.bin $OUT/${TAG}_ubench_f64.json > $OUT/ubench_f64.log 2>&1
[ -x $REPO/tools/ubench_ds_mix.bin ] && timeout 300 $REPO/tools/ubench_ds_mix.bin $OUT/${TAG}_ubench_ds_mix.json > $OUT/ubench_ds_mix.log 2>&1
[ -x $REPO/tools/ubench_count_shape.bin ] && timeout 300 $REPO/tools/ubench_count_shape.bin > $OUT/${TAG}_ubench_count_shape.json 2> $OUT/ubench_count_shape.err
python $REPO/tools/summarize_round.py $OUT $TAG "$CMD" "$PMC" "$LEGS" && cp $OUT/${TAG}_*.txt $OUT/${TAG}_*.json $REPO/profiles/
tail -2 $OUT/stats.log | cut -c1-600
