#!/bin/bash
# first GPU contact: parity tests, verbose
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.txt
