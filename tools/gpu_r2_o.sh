#!/bin/bash
mkdir -p gpurun_out/o
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/o/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/o/prof -o stats -- python $GRAFT_REPO_ROOT/tools/i8_bench.py > $GRAFT_REPO_ROOT/gpurun_out/o/i8_bench.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/o/prof/*.db") + glob.glob("gpurun_out/o/prof/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration)/1e3 from kernels group by name order by 3 desc"):
    print("%-60s %5d %10.2f us" % (r[0][:60], r[1], r[2]))
PY
find gpurun_out/o -name "*.db" -delete
cat gpurun_out/o/pytest_gpu.txt; grep "^{" gpurun_out/o/i8_bench.txt | sed -n 2,2p
