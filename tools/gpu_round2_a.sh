#!/bin/bash
# GPU call A of round 2: new dist/group tests first (fast feedback), then the whole GPU suite, bench (plain, group/RCCL, launcher), fit bench.
mkdir -p gpurun_out/a
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q 2>&1 | tail -30 > gpurun_out/a/pytest_dist.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/a/pytest_gpu.txt
timeout 600 python bench.py 2>&1 | tail -3 > gpurun_out/a/bench_n1.txt
timeout 300 python bench.py --group --no-cpu-baseline --no-api 2>&1 | tail -3 > gpurun_out/a/bench_n1_group.txt
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>&1 | tail -3 > gpurun_out/a/bench_n1_launcher.txt
timeout 600 python tools/fit_bench.py 2>&1 | tail -4 > gpurun_out/a/fit_bench.txt
tail -5 gpurun_out/a/*.txt
