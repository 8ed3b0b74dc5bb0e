#!/bin/bash
# Developer tool (GPU box): memory-side PMC counters of the nhood count kernels at several K (separate --pmc passes, no tracing).
#   usage: tools/pmc_pass.sh "30 64 100 200"   -> gpurun_out/pmc_pass.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_pass
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in ${1:-30 64 100 200}; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/K$K/p$i -- python $REPO/tools/nhood_k_sweep.py 1000 2560 --K=$K $2 > $OUT/K$K.p$i.log 2>&1
  done
done
python - <<'PY' > $REPO/gpurun_out/pmc_pass.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_pass"
for kd in sorted(glob.glob(out + "/K*/"), key=lambda p: int(os.path.basename(p.rstrip("/"))[1:])):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(kd + "/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][-46:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("=====", os.path.basename(kd.rstrip("/")))
    for k in acc:
        if "k_count" in k:
            print(k)
            for c in sorted(acc[k]):
                v = acc[k][c]
                print(f"   {c:40s} last dispatch {v[-1]:18.1f}   ({len(v)} dispatches, max {max(v):.1f})")
PY
cat $REPO/gpurun_out/pmc_pass.txt
