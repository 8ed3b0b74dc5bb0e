#!/bin/bash
# in the build container, after a lease: what the lease wrote under gpurun_out/ -> profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r05_final
tail -1 $F/bench.json > profiles/r05_bench.json
cp $F/bench_detail.json profiles/r05_bench_detail.json
tail -1 $F/bench_gpus2.json > profiles/r05_bench_gpus2_shared_device.json
cp $F/pytest_gpu.log profiles/r05_pytest_gpu.log
for f in r05_streams_table.json r05_nhood_k_sweep.jsonl r05_numpy_call_breakdown.jsonl r05_pmc_pass_kernel.txt; do [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f; done
ls -la profiles/r05_*
