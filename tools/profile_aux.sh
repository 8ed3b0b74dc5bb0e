#!/bin/bash
# rocprofv3 kernel-trace stats of the paths bench.py does not exercise: co_occurrence, ripley, ligrec, graph builders,
# numpy-stream shuffles.  Runs on the GPU box; summary -> gpurun_out/prof_aux/summary.txt (copy into profiles/).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_aux
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/aux_workload.py <<PY
import os, sys, runpy
sys.path.insert(0, "$REPO")
os.environ["SIDE"] = "1000"
for tool, argv in (("frontend_big.py", []), ("ligrec_time.py", ["--no-cpu", "--cases", "medium,large"]), ("pcg_time.py", []), ("neighbors_time.py", [])):
    sys.argv = [tool] + argv
    try:
        runpy.run_path(os.path.join("$REPO", "tools", tool), run_name="__main__")
    except SystemExit:
        pass
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python /tmp/aux_workload.py > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open("$OUT/summary.txt", "w") as fh:
    fh.write("== rocprofv3 --kernel-trace --stats: tools/frontend_big.py (co_occurrence, ripley L/F/G at 1e6 points), ligrec_time.py medium+large, pcg_time.py, neighbors_time.py ==\n")
    fh.write(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}\n")
    for r in rows[:28]:
        fh.write(f"{r['Name'].split('(')[0][:70]:70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}\n")
print(open("$OUT/summary.txt").read())
PY
