#!/bin/bash
for rep in 1 2; do
echo "== new"; python tools/pcg_time.py | cut -c1-130
echo "== prev"; SQGR_LIBRARY=$PWD/build_ab/libsqgr_prev.so python tools/pcg_time.py | cut -c1-130
done
