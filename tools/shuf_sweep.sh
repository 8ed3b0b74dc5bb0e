for b in 24 32 48 64 96 128; do echo "BLOCKS_PER_CU=$b"; SQGR_SHUFFLE_BLOCKS_PER_CU=$b timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-numpy-leg --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['pipeline']['avg_kernel_ms'])"; done
