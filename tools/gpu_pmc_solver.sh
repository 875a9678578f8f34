#!/bin/bash
# PMC passes over the rows solver (i8_bench, one round)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc_a -o a -- python $R/tools/i8_bench.py 5000 1 > $O/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/pmc_b -o b -- python $R/tools/i8_bench.py 5000 1 > $O/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/pmc_c -o c -- python $R/tools/i8_bench.py 5000 1 > $O/pmc_c.log 2>&1
cd $R
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/q/pmc_?/*.db") + glob.glob("gpurun_out/q/pmc_?/*/*.db")):
    cur = sqlite3.connect(db).cursor()
    for r in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%solver_rows%' group by kernel_name, counter_name"):
        print(r[0][:20], r[1], r[2], round(r[3], 1), round(r[4], 1))
PY
find gpurun_out/q -name "*.db" -delete
