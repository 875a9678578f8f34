#!/bin/bash
# GPU-box script (round-end refresh): full GPU test suite, headline bench (plain + torch.distributed N=1), rocprofv3 kernel stats and
# HBM counters of the headline, non-metric bench + its kernel stats, single-fit bench.  Outputs under gpurun_out/.
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_n1.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_n1_dist.json
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 | tee gpurun_out/nonmetric_bench.json
timeout 900 python tools/fit_bench.py c2 c5 2>&1 | grep "^{" | tee gpurun_out/fit_bench.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_nm
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats -- python $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 > $R/gpurun_out/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o fetch -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o write -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
NM_BENCH_STEPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nm -o nm -- python $R/tools/nonmetric_bench.py > $R/gpurun_out/prof_nm.log 2>&1
cd $R
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_nm -type f | head -30
du -sh gpurun_out
