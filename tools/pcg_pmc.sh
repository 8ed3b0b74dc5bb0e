#!/bin/bash
# Developer tool (GPU box): SQ counters of the bucketed numpy-stream shuffle kernels (separate --pmc passes, no tracing).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pcg_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=${1:-1024}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $REPO/tools/pcg_bucket_time.py $P bucket > $OUT/p$i.log 2>&1
done
python - <<'PY' > $REPO/gpurun_out/pcg_pmc.txt
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pcg_pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:44] + " grid=" + row["Grid_Size"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    if "k_pcg" in k or "rows_to" in k or "columns_to" in k:
        print(k)
        for c in sorted(acc[k]):
            print(f"   {c:34s} {acc[k][c] / max(n[k][c],1):18.1f} per dispatch ({n[k][c]} dispatches)")
PY
cat $REPO/gpurun_out/pcg_pmc.txt
