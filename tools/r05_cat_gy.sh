#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_cat_gy; mkdir -p $O
for gy in 4 8 16 32 64; do
echo "gy $gy"; CAT_CONV_GY=$gy timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 | cut -c380-560
CAT_CONV_GY=$gy timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1 | cut -c380-560
done
