#!/bin/bash
# GPU-box script: the headline part of profiles/<tag>_*: the bench line, `rocprofv3 --kernel-trace --stats` of the bench command, the HBM counters
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes, no trace domain) and their summary -- the files tests/test_host_api.py holds against each other.
# usage: bash tools/gpu_profiles_headline.sh r06   (tools/gpu_profiles.sh runs it first)
TAG=${1:-r06}
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
timeout 900 python bench.py 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python $R/bench.py --no-cpu-baseline --no-api --no-next-rows --no-single-fit --steps 50 --warmup 5 > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --no-cpu-baseline --no-api --no-next-rows --no-single-fit --steps 3 --warmup 1 > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o write -- python $R/bench.py --no-cpu-baseline --no-api --no-next-rows --no-single-fit --steps 3 --warmup 1 > $O/prof_write.log 2>&1
cd $R
python tools/rocprof_summary.py ${TAG}_tmp $(ls $O/prof_stats/*/*.db $O/prof_stats/*.db 2>/dev/null | head -1) $(ls $O/prof_fetch/*/*.db $O/prof_fetch/*.db 2>/dev/null | head -1) $(ls $O/prof_write/*/*.db $O/prof_write/*.db 2>/dev/null | head -1) $O/prof_stats.log > $O/rocprof_summary_stdout.txt 2>&1
mv profiles/${TAG}_tmp_rocprof_summary.md $O/rocprof_summary.md 2>/dev/null; mv profiles/${TAG}_tmp_rocprof_summary.json $O/rocprof_summary.json 2>/dev/null; mv profiles/${TAG}_tmp_gram_traffic.json $O/gram_traffic.json 2>/dev/null; mv profiles/${TAG}_tmp_gram_i8_traffic.json $O/gram_i8_traffic.json 2>/dev/null
