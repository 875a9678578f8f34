#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sk -o sk -- python $R/tools/sk_check.py 16384 5000 4864 10240 > $O/prof_sk.log 2>&1
cd $R
python - <<PY
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/q/prof_sk/*.db") + glob.glob("gpurun_out/q/prof_sk/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
seq = {}
for name, dur in cur.execute("select name, duration from kernels where name like '%gram_i8%' order by start"):
    seq.setdefault(name.split("(")[0], []).append(dur / 1e3)
for name, d in seq.items():
    for k in range(0, len(d), 14):
        part = d[k:k + 14][4:]          # (skip the parity call and the warm-up launches of each batch size)
        print("%-34s batch %d: launches %2d  avg %8.2f us  min %8.2f us" % (name, k // 14, len(part), sum(part) / len(part), min(part)))
PY
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/q/prof_sk/*.db") + glob.glob("gpurun_out/q/prof_sk/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name || ' grid ' || grid_x, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels where name like '%gram_i8%' group by name, grid_x order by name, grid_x"):
    print("%-50s %5d avg %9.2f us  min %9.2f us" % (r[0][:50], r[1], r[2], r[3]))
PY
find $O -name "*.db" -delete
tail -2 $O/prof_sk.log
