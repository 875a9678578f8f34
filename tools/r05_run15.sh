#!/bin/bash
# round 5, call 15: wave step for 9 .. 16 categories per item -- categorical + HOC tests, HOC bench, categorical bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_categorical.py tests/test_gpu_hoc.py -x -q -m gpu > $O/tests.txt 2>&1; tail -12 $O/tests.txt
timeout 600 python tools/hoc_bench.py 2>&1 | tail -1 > $O/hoc_bench.json; cat $O/hoc_bench.json
timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 | cut -c100-130,380-600
