#!/bin/bash
# bench.py over the (perms_per_pass, blocks_per_batch, batches_per_launch) space (run on the GPU box)
for t in 16,128,64 16,96,64 16,64,64 16,48,64 16,32,64 16,64,128 16,96,128 16,128,128 16,64,32; do
  echo -n "tune=$t: "; timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-numpy-leg --steps 5 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['pipeline']['avg_kernel_ms'])"; done
