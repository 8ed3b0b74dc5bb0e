#!/bin/bash
# (what it writes under profiles/ stays on the box: tools/r06_collect.sh copies the lease's gpurun_out/ files into profiles/ afterwards)
# final lease of round 6: the whole GPU suite, the driver's smoke, the profiles the bench line is priced with, the default bench
# line, the two-rank bench through its own launcher, the stream table and the cluster-count sweep — ONE lease, one build
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_gpu.time; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log | cut -c1-200
cp gpurun_out/prof_r06/r06_*.json gpurun_out/prof_r06/r06_*.txt profiles/ 2>/dev/null
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
cp $OUT/bench_detail.json profiles/r06_bench_detail.json 2>/dev/null; tail -1 $OUT/bench.json > profiles/r06_bench.json
# the driver's multi-GPU invocation, as far as one GPU can show it: `python bench.py --gpus 2` starts two ranks by itself
timeout 600 python bench.py --gpus 2 --share-devices --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-secondary --no-numpy-leg --emulate-ranks 0 \
  --detail-out $OUT/bench_gpus2_detail.json > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; tail -1 $OUT/bench_gpus2.json > profiles/r06_bench_gpus2_shared_device.json
timeout 600 python tools/streams_table.py > gpurun_out/r06_streams_table.json 2> $OUT/streams_table.err
timeout 600 python tools/nhood_k_sweep.py 1000 2560 > gpurun_out/r06_nhood_k_sweep.jsonl 2> $OUT/k_sweep.err
timeout 300 python tools/numpy_call_breakdown.py > gpurun_out/r06_numpy_call_breakdown.jsonl 2> $OUT/numpy_breakdown.err
# round 6: the counter layouts above 50 clusters switched off (rounds 1-5's 32-bit K*K counters) on the same box, the replay kernel's
# geometry sweep, the soak tests' negative control, the driver's launcher invocation at the node's size on this one GPU
SQGR_COUNT_C16=0 timeout 600 python tools/nhood_k_sweep.py 1000 2560 --K=51 --K=64 --K=100 --K=150 --K=200 --K=256 > gpurun_out/r06_nhood_k_sweep_c16_off.jsonl 2> $OUT/k_sweep_off.err
timeout 600 python tools/nhood_k_sweep.py 1000 2560 --graph=knn --K=30 --K=64 --K=100 --K=200 > gpurun_out/r06_nhood_k_sweep_knn.jsonl 2> $OUT/k_sweep_knn.err
timeout 900 bash tools/pcg_ablation.sh > /dev/null 2>&1  # numpy-stream kernels old against new, occupancy, the replay kernel's ablation -> gpurun_out/r06_pcg_replay_ablation.txt
bash tools/soak_negative.sh > /dev/null 2>&1; cp gpurun_out/soak_negative.txt gpurun_out/r06_soak_negative_run.txt
timeout 900 python bench.py --gpus 8 --share-devices --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-secondary --no-numpy-leg --emulate-ranks 0 --scaling strong --total-perms 100000 \
  --detail-out $OUT/bench_gpus8_detail.json > $OUT/bench_gpus8.json 2> $OUT/bench_gpus8.err; tail -1 $OUT/bench_gpus8.json > gpurun_out/r06_bench_gpus8_shared_device.json
bash tools/pmc_pass.sh "30 64 100 200" > /dev/null 2>&1; cp gpurun_out/pmc_pass.txt gpurun_out/r06_pmc_pass_kernel.txt
cp $OUT/pytest_gpu.log profiles/r06_pytest_gpu.log
python - $OUT/bench_detail.json $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
line = open(sys.argv[2]).read().strip().splitlines()[-1]
print("final line bytes:", len(line))
r = d["roofline"]
print("value", round(d["value"]), "roofline", r["bound"], r["achieved"], r["frac"], "fabric", r.get("fabric_frac"), "alg", r.get("algorithmic_frac"), "pmc:", d.get("pmc_profile"))
print("moran", round(d["secondary"]["value"]), d["secondary"]["roofline"].get("frac"))
for k, v in d.get("legs", {}).items():
    rr = v.get("roofline") or {}
    print(k, v.get("value"), v.get("unit"), "kernel_ms", v.get("kernel_ms") if not isinstance(v.get("kernel_ms"), dict) else "", "frac", rr.get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"), v.get("moran"), v.get("geary"))
n = d["numpy_stream_mode"]; print("numpy", n["value"], n["at_n_perms_1000"], n["roofline"]["frac"], n["roofline"].get("traffic_MB_per_perm"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"].get("value"), "emulated", d.get("emulated_ranks", {}).get("shard_seconds"))
PY
tail -1 profiles/r06_bench_gpus2_shared_device.json | cut -c1-600
