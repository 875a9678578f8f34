#!/bin/bash
# round-5 GPU pass 1: the GPU suite, the bench in its four forms, the host-buffer sub-batch sweep
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run1; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
timeout 300 python bench.py --group --no-cpu-baseline --no-api 2>$O/bench_group.err | tail -1 > $O/bench_n1_group_rccl.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>$O/bench_launcher.err | grep '^{' | tail -1 > $O/bench_n1_launcher_rccl.json
PLSPM_BENCH_SHARED_DEVICE=1 timeout 300 python bench.py --gpus 2 --no-cpu-baseline --no-api 2>$O/bench_seam2.err | tail -1 > $O/bench_seam2.json
timeout 300 python tools/pcie_chunks.py 5000 > $O/pcie_chunks.jsonl 2>$O/pcie_chunks.err
tail -3 $O/pytest.txt
