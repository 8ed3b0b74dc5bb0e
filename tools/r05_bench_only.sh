#!/bin/bash
# short lease: the default bench line priced with the committed profiles/r05_counters.json (same kernel sources), the two-rank line
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --share-devices --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-secondary --no-numpy-leg --emulate-ranks 0 \
  --detail-out $OUT/bench_gpus2_detail.json > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err
tail -c 4200 $OUT/bench.json
timeout 600 python tools/streams_table.py > gpurun_out/r05_streams_table.json 2> $OUT/streams_table.err
timeout 600 python tools/nhood_k_sweep.py 1000 2560 > gpurun_out/r05_nhood_k_sweep.jsonl 2> $OUT/k_sweep.err
timeout 300 python tools/numpy_call_breakdown.py > gpurun_out/r05_numpy_call_breakdown.jsonl 2> $OUT/numpy_breakdown.err
