#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run2; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
timeout 300 python tools/group_ab.py > $O/group_ab.jsonl 2>$O/group_ab.err
timeout 300 python tools/pcie_chunks.py 5000 > $O/pcie_chunks.jsonl 2>$O/pcie_chunks.err
PLSPM_BENCH_SHARED_DEVICE=1 timeout 300 python bench.py --gpus 2 --no-cpu-baseline --no-api 2>$O/bench_seam2.err | tail -1 > $O/bench_seam2.json
tail -3 $O/pytest.txt
