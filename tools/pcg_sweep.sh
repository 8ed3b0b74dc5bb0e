#!/bin/bash
python tools/pcg_check.py | tail -1
SQGR_PCG_FORCE_SLOW=1 python tools/pcg_check.py | tail -1
python -m pytest tests/test_nhood_gpu.py -x -q -k "variants or numpy" 2>&1 | tail -2
python tools/pcg_time.py | cut -c1-110
python tools/pcg_time.py | cut -c1-110
