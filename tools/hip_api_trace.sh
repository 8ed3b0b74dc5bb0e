#!/bin/bash
# HIP API trace of the default bench command (no CPU baselines, no numpy-stream leg): which runtime calls the wall time outside the
# kernels goes to.  Round 3 found the ~5 s hipMalloc stall of spatial_autocorr's third call with it ("parked").
#   bash tools/hip_api_trace.sh [tag]      -> gpurun_out/hiptrace/<tag>_hip_api_summary.txt
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/hiptrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 900 rocprofv3 --hip-runtime-trace -d $OUT -o $TAG -- python bench.py --no-cpu-baseline --no-numpy-leg > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_err.log
python - $OUT $TAG <<'PY'
import json, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/{tag}_bench.json").read().strip().splitlines()[-1])
c = sqlite3.connect(f"{out}/{tag}_results.db")
t0 = c.execute("select min(start) from regions").fetchone()[0]
lines = [f"HIP API trace of `python bench.py --no-cpu-baseline --no-numpy-leg` (rocprofv3 --hip-runtime-trace), env SQGR_POOL_GB={__import__('os').environ.get('SQGR_POOL_GB', 'default')}",
         f"config 3 end to end (s): {d['legs'].get('config3_full')}; secondary: {d.get('secondary', {}).get('value')} genes/s, {d.get('secondary', {}).get('ms_per_step')} ms/step", "",
         "longest single calls:"]
for name, start, dur in c.execute("select name, start, end - start from regions order by 3 desc limit 24"):
    args = dict((r[0], r[1]) for r in c.execute("select name, value from region_args where id = (select id from regions where start = ? and name = ?)", (start, name)))
    lines.append(f"  {dur / 1e6:9.1f} ms  {name:28s} at t = {(start - t0) / 1e9:6.2f} s  {('size ' + args['size']) if 'size' in args else ''}")
lines += ["", "totals per function:"]
for name, cnt, tot in c.execute("select name, count(*), sum(end - start) from regions group by name order by 3 desc limit 8"):
    lines.append(f"  {tot / 1e6:9.1f} ms  {cnt:6d} calls  {name}")
open(f"{out}/{tag}_hip_api_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -f $OUT/${TAG}_results.db
