#!/bin/bash
# bench.py over the (perms_per_pass, blocks_per_batch, batches_per_launch) space (run on the GPU box)
for t in 16,0,0 32,0,32 32,0,16 32,64,32 32,128,32 32,32,32; do
  echo -n "tune=$t: "; timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-numpy-leg --steps 5 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['pipeline']['avg_kernel_ms'])"; done
