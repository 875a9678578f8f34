#!/bin/bash
O=gpurun_out/p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.txt
timeout 300 python tools/i8_bench.py 5000 4 > $O/i8_bench.txt 2>&1
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_n1.json
cat $O/pytest_gpu.txt; grep "^{" $O/i8_bench.txt | sed -n 1,3p; cat $O/bench_n1.json
