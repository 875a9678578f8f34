#!/bin/bash
# round 5, call 13: Philox products by v_mad_u64_u32 -- full GPU suite, size bench, headline bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run13; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python tools/size_bench.py 2>&1 | grep "^{" > $O/size_bench.jsonl; cat $O/size_bench.jsonl
timeout 300 python bench.py --no-cpu-baseline --no-api 2>/dev/null | tail -1 > $O/bench.json; python -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['avg_launch_ms'])"
