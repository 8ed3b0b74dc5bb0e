#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_lease4
mkdir -p $OUT
cd $REPO
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
tail -3 $OUT/bench.time; tail -5 $OUT/bench.err | cut -c1-400
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", round(d["value"]), "roofline.frac", d["roofline"]["frac"], "pmc:", d.get("pmc_profile"))
    print("moran", round(d["secondary"]["value"]), d["secondary"]["roofline"].get("frac"))
    for k, v in d.get("legs", {}).items():
        print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "unit", "wall_s", "kernel_ms", "moran", "geary", "synthetic_input_build_s")}, "frac", (v.get("roofline") or {}).get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
    print("numpy", d["numpy_stream_mode"]["value"], d["numpy_stream_mode"]["roofline"]["clk_per_swap_step_per_wave_at_32_waves_per_cu"])
except Exception as e:
    print("FAILED", e)
PY
timeout 200 python bench.py --no-cpu-baseline --no-secondary --no-legs --no-numpy-leg --emulate-ranks 8 --steps 2 --warmup 1 > $OUT/emulate.json 2> $OUT/emulate.err
python -c "
import json;d=json.loads(open('$OUT/emulate.json').read().strip().splitlines()[-1]);print(json.dumps(d['emulated_ranks']))"
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1
tail -40 $OUT/profile_round.log | cut -c1-300
