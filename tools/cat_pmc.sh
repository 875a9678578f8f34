cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/cat_pmc; rm -rf $O; mkdir -p $O
CAT_BENCH_STEPS=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/p1 -o p1 -- python $R/tools/categorical_bench.py 2000 > $O/p1.log 2>&1
CAT_BENCH_STEPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $O/p2 -o p2 -- python $R/tools/categorical_bench.py 2000 > $O/p2.log 2>&1
CAT_BENCH_STEPS=1 timeout 600 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/p3 -o p3 -- python $R/tools/categorical_bench.py 2000 > $O/p3.log 2>&1
python - "$O" <<'PY'
import sqlite3, glob, sys
O=sys.argv[1]
for db in sorted(glob.glob(O+"/p*/*.db")+glob.glob(O+"/p*/*/*.db")):
    cur=sqlite3.connect(db).cursor()
    q=("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection c where grid_size = (select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name, counter_name")
    for r in cur.execute(q):
        if any(k in r[0] for k in ("nmg_kernel","nm_conv_codes","nmw_step","conv_mfma","planes_kernel")):
            print(r[0].split("(")[0].replace("void ","")[:32], r[1], r[2], "%.4g"%r[3], "%.0f ns"%r[4])
PY
