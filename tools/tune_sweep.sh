#!/bin/bash
# bench.py over the (perms_per_pass, blocks_per_batch, batches_per_launch) space (run on the GPU box).  0 = automatic.
# Late round 2 (count kernel with the dot2 address path): blocks per batch 32 (auto) > 40 > 48 > 64 >> 128 >> 256;
# batches per launch 64: 970k, 128: 1016k, 160: 1020k (default), 208: 1020k, 256: 1014k, 320: 996k, 640: 999k permutations/s.
for t in 16,0,0 16,40,0 16,48,0 16,64,0 16,128,0 16,0,64 16,0,128 16,0,160 16,0,208 16,0,256 16,0,320 32,0,32; do
  echo -n "tune=$t: "; timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-numpy-leg --no-legs --steps 6 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['perms_per_launch'], d['pipeline']['avg_kernel_ms'])"; done
