#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_categorical.py tests/test_gpu_hoc.py -x -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
for v in 1 0 1 0; do
CAT_NM_EMIT=$v timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('emit $v', d['replicates_per_s'], d['ms_per_step'], d['kernel_ms_per_step'], d['all_ok'], d['replicate_iterations'])"
CAT_NM_EMIT=$v timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('emit $v', d['replicates_per_s'], d['ms_per_step'], d['kernel_ms_per_step'], d['all_ok'], d['replicate_iterations'])"
done
