#!/bin/bash
# Developer tool (GPU box): SQ-side PMC pass for the list schedule kernel (k_bucket_order_steps) — where do its waves wait
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_order
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/autocorr_order_exp.py --one"
export SQGR_AUTOCORR_KERNEL=lds EXP_ONLY=moran
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-include-regex "k_bucket_order" --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --kernel-include-regex "k_bucket_order" --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("a","b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-30:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(tag, k, {c: f"{x / n[(k, c)]:.3e}" for c, x in sorted(v.items())}, "launches", max(n[(k, c)] for c in v))
PY
tail -2 $OUT/a.log
