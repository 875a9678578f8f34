#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run27; mkdir -p $O
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
(AB_MODES=BBBBBB python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{"; AB_MODES=ABABAB python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{") > $O/ab_solver_wave_mode_b.jsonl
python -c "import __graft_entry__ as g; g.smoke()"
timeout 600 python bench.py 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json; python tools/show_bench.py $O/bench_n1.json | head -8
