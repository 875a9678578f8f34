#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_n1.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -5 | tee gpurun_out/bench_n1_dist.txt
