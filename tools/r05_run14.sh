#!/bin/bash
# round 5, call 14: 4-bit resample counters -- tests of the large-N routes, size bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run14; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gram_i8.py tests/test_gpu_parity.py -x -q -m gpu -k "65535 or four_bit or beyond" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
timeout 600 python tools/size_bench.py 2>&1 | grep "^{" > $O/size_bench.jsonl; cat $O/size_bench.jsonl
