#!/bin/bash
# Developer tool (GPU box): the list schedules (k_bucket_order_steps / _joint) against the round-3 one, Geary's row-sum shortcuts, and the
# premise of the schedule (tools/ubench_lds_read: the ds_read_b128 service groups).
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/joint; mkdir -p $out
tools/ubench_lds_read.bin > $out/ubench_lds_read.json 2>$out/ubench.err
timeout 900 python -m pytest tests/test_autocorr_gpu.py -x -q 2>&1 | tail -5 | tee $out/pytest_autocorr.log
one() { echo "== $1"; shift; env SQGR_AUTOCORR_KERNEL=lds "$@" timeout 300 python tools/autocorr_order_exp.py --one; }
{
one "step schedule (default; Geary: constant + exception lists)"
one "rotation schedule" SQGR_AUTOCORR_ORDER=rotation
one "single (round 3: every list on its own)" SQGR_AUTOCORR_ORDER=single
one "lists as built" SQGR_AUTOCORR_ORDER_LISTS=0
one "steps, Geary through the class table" SQGR_AUTOCORR_ROWSUM_EXCEPTIONS=0
one "steps, general Geary kernel" SQGR_AUTOCORR_ROWSUM_CLASSES=0
one "single, general Geary kernel" SQGR_AUTOCORR_ROWSUM_CLASSES=0 SQGR_AUTOCORR_ORDER=single
one "steps, 30 000 spots (lists of ~830 pairs: 3 segments)" EXP_ROWS=150 EXP_COLS=200
one "rotation, 30 000 spots" EXP_ROWS=150 EXP_COLS=200 SQGR_AUTOCORR_ORDER=rotation
one "single, 30 000 spots (lists left as built)" EXP_ROWS=150 EXP_COLS=200 SQGR_AUTOCORR_ORDER=single
one "steps, 5 000 spots (one bucket)" EXP_ROWS=50 EXP_COLS=100
one "single, 5 000 spots" EXP_ROWS=50 EXP_COLS=100 SQGR_AUTOCORR_ORDER=single
} 2>&1 | tee $out/times.log
