#!/bin/bash
# round 5: the full GPU suite with the new default stream, the stream table, and one bench run
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu.log 2>&1
tail -15 gpurun_out/r05_pytest_gpu.log
timeout 600 python tools/streams_table.py > gpurun_out/streams_table.json 2> gpurun_out/streams_table.err
cat gpurun_out/streams_table.err | cut -c1-400
timeout 900 python bench.py > gpurun_out/r05_bench_stdout.log 2> gpurun_out/r05_bench_stderr.log
tail -c 4500 gpurun_out/r05_bench_stdout.log; tail -5 gpurun_out/r05_bench_stderr.log
