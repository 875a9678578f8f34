#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/l; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python $R/bench.py --no-cpu-baseline --no-api --steps 50 --warmup 5 > $O/prof_stats.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 -d $O/prof_pmc1 -o pmc1 -- python $R/tools/i8_bench.py > $O/prof_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM -d $O/prof_pmc2 -o pmc2 -- python $R/tools/i8_bench.py > $O/prof_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o fetch -- python $R/bench.py --no-cpu-baseline --no-api --steps 3 --warmup 1 > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o write -- python $R/bench.py --no-cpu-baseline --no-api --steps 3 --warmup 1 > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum -d $O/prof_pmc3 -o pmc3 -- python $R/tools/i8_bench.py > $O/prof_pmc3.log 2>&1
cd $R
python - "$O" <<'PY'
import sqlite3, glob, sys, json
O = sys.argv[1]
out = []
for db in sorted(glob.glob(O + "/prof_*/*.db") + glob.glob(O + "/prof_*/*/*.db")):
    if "stats" in db: continue
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection c where grid_size = "
         "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name, counter_name")
    try:
        for r in cur.execute(q):
            if any(k in r[0] for k in ("gram", "solver_kernel", "resample")):
                out.append({"run": db.split("/")[-2] if "prof_" in db.split("/")[-2] else db.split("/")[-3], "kernel": r[0].split("(")[0].replace("void ", ""), "counter": r[1], "dispatches": r[2], "avg": round(r[3], 1), "avg_duration_ns": round(r[4], 1)})
    except Exception as e:
        print("db", db, e)
json.dump(out, open(O + "/pmc_rows.json", "w"), indent=0)
for r in out: print(r["run"], r["kernel"][:36], r["counter"], r["dispatches"], r["avg"], r["avg_duration_ns"])
PY
python tools/rocprof_summary.py r02l_tmp $(ls $O/prof_stats/*/*.db $O/prof_stats/*.db 2>/dev/null | head -1) $(ls $O/prof_fetch/*/*.db $O/prof_fetch/*.db 2>/dev/null | head -1) $(ls $O/prof_write/*/*.db $O/prof_write/*.db 2>/dev/null | head -1) > $O/rocprof_summary_stdout.txt 2>&1
mv profiles/r02l_tmp_* $O/ 2>/dev/null
find $O -name "*.db" -size +20M -delete
cat $O/bench_n1.json; tail -5 $O/prof_pmc1.log; ls $O
