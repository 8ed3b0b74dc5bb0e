#!/bin/bash
# SQ-side PMC pass for the nhood kernels (separate from kernel-trace / FETCH / WRITE passes)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 1 --warmup 1 --perms-per-step 2048 --no-cpu-baseline --no-secondary --no-numpy-leg"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("a","b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "k_count" in k or "k_shuffle" in k or "k_reduce" in k:
            print(tag, k, {c: f"{x:.3e}" for c, x in sorted(v.items())})
PY
tail -3 $OUT/a.log
