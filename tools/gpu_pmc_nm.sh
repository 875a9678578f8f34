#!/bin/bash
# PMC evidence for the non-metric bootstrap kernels (dense stop-rule pass, step kernels): separate --pmc passes, no trace domains.
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_nm_pmc1 $R/gpurun_out/prof_nm_pmc2
NM_BENCH_STEPS=3 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $R/gpurun_out/prof_nm_pmc1 -o pmc1 -- python $R/tools/nonmetric_bench.py > $R/gpurun_out/prof_nm_pmc1.log 2>&1
tail -2 $R/gpurun_out/prof_nm_pmc1.log
NM_BENCH_STEPS=3 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_nm_pmc2 -o pmc2 -- python $R/tools/nonmetric_bench.py > $R/gpurun_out/prof_nm_pmc2.log 2>&1
tail -2 $R/gpurun_out/prof_nm_pmc2.log
cd $R
python - <<'PY'
import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/prof_nm_pmc*/*.db")):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection c where grid_size = "
         "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name, counter_name")
    for r in cur.execute(q):
        if "nm_conv_dense" in r[0] or "nm_kernel" in r[0] or "coef_table" in r[0]:
            print(r[0].split("(")[0][:44], r[1], r[2], round(r[3], 1), round(r[4], 1))
PY
