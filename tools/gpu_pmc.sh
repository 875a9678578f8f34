#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pmc1 $R/gpurun_out/prof_pmc2
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 -d $R/gpurun_out/prof_pmc1 -o pmc1 -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/prof_pmc1.log 2>&1
tail -3 $R/gpurun_out/prof_pmc1.log
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/prof_pmc2 -o pmc2 -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/prof_pmc2.log 2>&1
tail -3 $R/gpurun_out/prof_pmc2.log
cd $R
python - <<'PY'
import sqlite3, glob
for db in glob.glob("gpurun_out/prof_pmc*/*.db"):
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
    for r in cur.execute(q):
        if "gram" in r[0] or "solver" in r[0] or "resample" in r[0]:
            print(r[0][:40], r[1], r[2], round(r[3], 1), round(r[4], 1))
PY
