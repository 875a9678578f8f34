#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d.get('pcie_inclusive'))"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_n1_dist.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
