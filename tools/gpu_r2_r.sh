#!/bin/bash
O=gpurun_out/r; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 600 python tools/aux_ab.py resample_aux=0,1 > $O/aux_ab.txt 2>&1
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_n1.json
cat $O/pytest_gpu.txt $O/aux_ab.txt; python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['kernels_ms_per_step'], d['api_inclusive'], d['pcie_inclusive']['value'])"
