for a in 1 2 4 8; do echo "== XCD_CHUNKS=$a"; SQGR_AUTOCORR_XCD_CHUNKS=$a timeout 300 python tools/autocorr_small_time.py 2>&1 | grep "moran P=100\|moran P=256\|geary P=100\|moran P=50 " | grep -v gather | cut -c1-200; done
echo "== default"; timeout 300 python tools/autocorr_small_time.py 2>&1 | grep "P=100\|P=50" | cut -c1-200
