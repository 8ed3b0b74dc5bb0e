# (round 4, before the joint schedules: the per-list order of round 3 is selected explicitly)
export SQGR_AUTOCORR_ORDER=single SQGR_AUTOCORR_ROWSUM_EXCEPTIONS=0
echo "== default (class table, Y-class order)"; SQGR_AUTOCORR_KERNEL=lds timeout 300 python tools/autocorr_order_exp.py --one
echo "== class table, Z-class order"; SQGR_AUTOCORR_KERNEL=lds SQGR_AUTOCORR_ORDER_BY=z timeout 300 python tools/autocorr_order_exp.py --one
echo "== class table, Z order, 1 chunk per round"; SQGR_AUTOCORR_KERNEL=lds SQGR_AUTOCORR_ORDER_BY=z SQGR_AUTOCORR_XCD_CHUNKS=1 timeout 300 python tools/autocorr_order_exp.py --one
echo "== class table, Y order, 2 chunks per round"; SQGR_AUTOCORR_KERNEL=lds SQGR_AUTOCORR_XCD_CHUNKS=2 timeout 300 python tools/autocorr_order_exp.py --one
echo "== general kernel (classes off)"; SQGR_AUTOCORR_KERNEL=lds SQGR_AUTOCORR_ROWSUM_CLASSES=0 timeout 300 python tools/autocorr_order_exp.py --one
