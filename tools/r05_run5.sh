#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run5; rm -rf $O; mkdir -p $O; cd $R
PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py 5000 > $O/persist_ab_exp.jsonl 2>$O/persist_ab.err
PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py 2500 >> $O/persist_ab_exp.jsonl 2>>$O/persist_ab.err
timeout 300 python tools/group_ab.py > $O/group_ab.jsonl 2>$O/group_ab.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
cat $O/persist_ab_exp.jsonl; tail -8 $O/pytest.txt
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open('gpurun_out/r05_run5/group_ab.jsonl'):
    r=json.loads(l); d[r['kind']].append(r['ms_per_step'])
for k,v in d.items(): print(k, v, min(v))
PY
