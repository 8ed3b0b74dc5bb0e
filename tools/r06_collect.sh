#!/bin/bash
# in the build container, after a lease: what the lease wrote under gpurun_out/ -> profiles/ (tracked)
cd "$(dirname "$0")/.."
F=gpurun_out/r06_final
tail -1 $F/bench.json > profiles/r06_bench.json
cp $F/bench_detail.json profiles/r06_bench_detail.json
tail -1 $F/bench_gpus2.json > profiles/r06_bench_gpus2_shared_device.json
cp $F/pytest_gpu.log profiles/r06_pytest_gpu.log
cp gpurun_out/prof_r06/r06_*.json gpurun_out/prof_r06/r06_*.txt profiles/ 2>/dev/null
for f in r06_streams_table.json r06_nhood_k_sweep.jsonl r06_nhood_k_sweep_c16_off.jsonl r06_nhood_k_sweep_knn.jsonl r06_numpy_call_breakdown.jsonl r06_pmc_pass_kernel.txt r06_pcg_replay_ablation.txt r06_soak_negative_run.txt r06_bench_gpus8_shared_device.json; do [ -s gpurun_out/$f ] && cp gpurun_out/$f profiles/$f; done
ls -la profiles/r06_*
