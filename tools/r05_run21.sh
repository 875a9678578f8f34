#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run21; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_hoc.py tests/test_gpu_categorical.py tests/test_gpu_nmx.py tests/test_gpu_missing.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python tools/hoc_bench.py > $O/hoc.json 2> $O/hoc.err; cat $O/hoc.json; tail -3 $O/hoc.err
timeout 600 python tools/categorical_bench.py 2>&1 | tail -1 | cut -c1-600
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 | cut -c1-600
