#!/bin/bash
# Developer tool (GPU box): memory-pipeline PMC counters of the nhood count kernel (separate --pmc passes, no tracing).
#   usage: tools/pmc_count.sh   -> gpurun_out/pmc_count.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_count
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TD_TD_BUSY_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum TCP_TOTAL_READ_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/count_probe.py > $OUT/p$i.log 2>&1
done
python - <<'PY' > $REPO/gpurun_out/pmc_count.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_count"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in acc:
    if "k_count" in k or "k_shuffle" in k:
        print(k)
        for c in sorted(acc[k]):
            print(f"   {c:45s} {acc[k][c] / max(n[k][c],1):16.1f} per dispatch ({n[k][c]} dispatches)")
PY
cat $REPO/gpurun_out/pmc_count.txt
