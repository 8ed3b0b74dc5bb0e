#!/bin/bash
# Round 4, lease 1 (VERDICT r3 task 2): (a) do FETCH_SIZE / TCC_EA0_RDREQ_DRAM tell Infinity-Cache hits from DRAM reads?
# (b) launch groups sized to the Infinity Cache: perms/s, DRAM-side requests and L2-miss latency per group size.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04_mall
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EA="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"
CAL=$REPO/tools/ubench_fetch_calib.bin
timeout 120 $CAL > $OUT/calib.log 2>&1; cat $OUT/calib.log
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $CAL > $OUT/cal_fetch.log 2>&1
timeout 200 rocprofv3 --pmc $EA --output-format csv -d $OUT/cal_ea -- $CAL > $OUT/cal_ea.log 2>&1
GRP="python $REPO/tools/nhood_groups.py"
timeout 300 $GRP 40000 8,12,16,32,64,160 > $OUT/groups.log 2> $OUT/groups.err; cat $OUT/groups.log | cut -c1-400
timeout 300 rocprofv3 --pmc $EA --output-format csv -d $OUT/grp_ea -- $GRP 5120 8,12,16,160 > $OUT/grp_ea.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/grp_fetch -- $GRP 5120 8,12,16,160 > $OUT/grp_fetch.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/grp_lat -- $GRP 5120 8,12,16,160 > $OUT/grp_lat.log 2>&1
python $REPO/tools/summarize_groups.py $OUT > $OUT/summary.log 2>&1; tail -c 3000 $OUT/summary.log
# the compact bench line on real hardware (and its detail file)
cd $REPO
( time timeout 900 python bench.py --emulate-ranks 8 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
tail -1 $OUT/bench.json | wc -c; tail -1 $OUT/bench.json | cut -c1-1500; tail -3 $OUT/bench.time
cp gpurun_out/bench_detail.json $OUT/ 2>/dev/null
rm -rf $OUT/*/*/*.db 2>/dev/null
