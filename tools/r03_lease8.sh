#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03_lease8
mkdir -p $OUT
cd $REPO
# two ranks on the one GPU: (a) host side channel forced, (b) RCCL attempted (refuses two ranks on one device -> agreed fall-back)
SQGR_DIST_COLLECTIVE=host timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench2_host.json 2> $OUT/bench2_host.err
echo "rc=$?"; tail -c 600 $OUT/bench2_host.json | head -c 600; echo; tail -3 $OUT/bench2_host.err | cut -c1-300
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --scaling strong > $OUT/bench2_rccl.json 2> $OUT/bench2_rccl.err
echo "rc=$?"; python - $OUT/bench2_host.json $OUT/bench2_rccl.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["n_gpus"], round(d["value"]), d["scaling"], d["config"]["collective"], d["secondary"]["value"] if d.get("secondary") else None)
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -4 $OUT/bench2_rccl.err | cut -c1-300
