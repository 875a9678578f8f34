#!/bin/bash
# GPU-box script: everything that goes into profiles/ for one round.  usage: bash tools/gpu_profiles.sh r02
# bench lines (plain, in-library group/RCCL, launcher), fit bench, non-metric bench, rocprofv3 kernel stats of the headline command,
# HBM counters (separate --pmc passes, no trace domains), SQ counters of the headline Gram and of the configs[4] Gram.
TAG=${1:-r05}
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/profiles_$TAG
rm -rf $O; mkdir -p $O
cd $R
bash tools/gpu_profiles_headline.sh $TAG      # bench line, kernel trace, HBM counters, summary (profiles/<tag>_bench_n1.json, _rocprof_summary.*, _gram_i8_traffic.json)
timeout 300 python bench.py --group --no-cpu-baseline --no-api --no-next-rows --no-single-fit 2>/dev/null | tail -1 > $O/bench_n1_group_rccl.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_n1_launcher_rccl.json
PLSPM_BENCH_SHARED_DEVICE=1 timeout 300 python bench.py --gpus 2 --no-cpu-baseline --no-api --no-next-rows --no-single-fit 2>/dev/null | tail -1 > $O/bench_seam_2ranks_one_device.json
# round 5: the persistent Gram against the tiled launch (with the experiments build: the same launches without epilogue stores), what the group path
# costs a one-rank step, the host-buffer entry point under the sub-batch options
[ -f plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so ] && (for a in "5000" "2500" "5000 7"; do PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py $a; done) 2>/dev/null | grep "^{" > $O/persist_ab.jsonl
timeout 300 python tools/group_ab.py 2>/dev/null | grep "^{" > $O/group_ab.jsonl
timeout 300 python tools/pcie_chunks.py 5000 2>/dev/null | grep "^{" > $O/pcie_chunks.jsonl
timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1 > $O/categorical_bench_5000.json
CAT_NM_WAVE=0 timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 > $O/categorical_bench_workgroup_step.json
# round 5, second half: A/B lines of the categorical path (stop rule on category codes instead of the int8 matrix product; packed fp64 matrices + scatter pass instead of
# uint16 counts from the Gram), the stand-alone resample kernel with its generator variants at 10,000 and 100,000 rows, issue cost of the epilogue's instructions
(for v in 0 1; do CAT_NM_MFMA=$v timeout 300 python tools/categorical_bench.py 2>&1 | tail -1; CAT_NM_MFMA=$v timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1; done) > $O/categorical_ab_nm_mfma.jsonl
(for v in 0 1; do CAT_NM_DIRECT16=$v timeout 300 python tools/categorical_bench.py 2>&1 | tail -1; CAT_NM_DIRECT16=$v timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1; done) > $O/categorical_ab_nm_direct16.jsonl
(./tools/ubench/resample_rng 10000 5000 256 x | head -9; ./tools/ubench/resample_rng 100000 5000 1024) 2>&1 | grep "^{" > $O/ubench_resample_rng.jsonl
./tools/ubench/mul_issue > $O/ubench_mul_issue.txt 2>&1
(bash tools/cat_pmc.sh > $O/categorical_pmc.txt 2>&1; rm -rf $R/gpurun_out/cat_pmc)
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 > $O/nonmetric_bench.json
(NM_BENCH_N=100000 NM_BENCH_SPINUP=3 NM_BENCH_STEPS=5 timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1; NM_BENCH_N=100000 NM_BENCH_GRAM_PATH=1 NM_BENCH_SPINUP=1 NM_BENCH_STEPS=2 timeout 600 python tools/nonmetric_bench.py 1000 2>&1 | tail -1) > $O/nonmetric_100k.jsonl
timeout 900 python tools/fit_bench.py c2 c5 2>&1 | grep "^{" > $O/fit_bench.jsonl
# round 6: the one-launch NUM / RAW solver against the per-iteration launches (A/B), its verification pass with the lower bound forced to fail / to one row block
# (kernel tables), the dense wide Gram of configs[4] on the ring of row buffers against the two-stage ping-pong
(for v in 1 0 1 0; do NM_BENCH_WAVE16=$v timeout 300 python tools/nonmetric_bench.py 2>&1 | tail -1; done) > $O/nonmetric_ab_wave16.jsonl
bash tools/experiments/nm_verify_prof.sh > $O/nonmetric_verify_kernels.txt 2>&1
# round 6: the categorical bootstrap launch by launch with every pass over all rows (round 5) / with the step's own bound + rows on request / as one launch + verification
(for o in "nm_cat_one=0,nm_subset=0" "nm_cat_one=0" "nm_cat_one=1"; do for B in 1000 5000; do CAT_BENCH_OPTS=$o timeout 300 python tools/categorical_bench.py $B 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'options': '$o', 'replicates_per_step': $B, 'replicates_per_s': d['replicates_per_s'], 'ms_per_step': d['ms_per_step'], 'kernel_ms_per_step': d['kernel_ms_per_step'], 'replicate_iterations': d['replicate_iterations']}))"; done; done) > $O/categorical_ab_one_launch.jsonl
bash tools/experiments/cat_one_prof.sh 1000 > $O/categorical_one_launch_kernels.txt 2>&1
timeout 120 python tools/experiments/cat_bound_probe.py > $O/categorical_bound_probe.txt 2>&1
(for o in wide_ring=0 wide_ring=4 wide_ring=6 wide_ring=0 wide_ring=4 wide_ring=6; do FIT_BENCH_OPTS=$o timeout 300 python tools/fit_bench.py c5 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'option': '$o', 'kernel_ms': d['kernel_ms'], 'gram_kernel': d['roofline']['gram_kernel'], 'iterations': d['iterations']}))"; done) > $O/c5_ring_ab.jsonl
timeout 300 python tools/api_phase_times.py 2>&1 | grep "^{" > $O/api_phase_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 -d $O/prof_pmc1 -o pmc1 -- python $R/bench.py --no-cpu-baseline --no-api --no-next-rows --no-single-fit --steps 3 --warmup 1 > $O/prof_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 -d $O/prof_c5_pmc1 -o pmc1 -- python $R/tools/fit_bench.py c5 > $O/prof_c5_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/prof_c5_pmc2 -o pmc2 -- python $R/tools/fit_bench.py c5 > $O/prof_c5_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/prof_c5_fetch -o fetch -- python $R/tools/fit_bench.py c5 > $O/prof_c5_fetch.log 2>&1
# the int8 digit-plane Gram and the rows solver of the headline step (interleaved A/B tool: every Gram variant runs full-size launches)
# (round 4: tools/i8p_counters.py runs full-size launches of BOTH Gram kernels -- gram_i8p_kernel, the default, and the round-3 gram_i8_kernel -- so
#  that one counter pass yields the A/B table of profiles/<tag>_pmc.md)
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 -d $O/prof_i8_pmc1 -o pmc1 -- python $R/tools/i8p_counters.py > $O/prof_i8_pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH -d $O/prof_i8_pmc2 -o pmc2 -- python $R/tools/i8p_counters.py > $O/prof_i8_pmc2.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQC_ICACHE_REQ SQC_ICACHE_MISSES -d $O/prof_i8_pmc3 -o pmc3 -- python $R/tools/i8p_counters.py > $O/prof_i8_pmc3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $O/prof_i8_pmc4 -o pmc4 -- python $R/tools/i8p_counters.py > $O/prof_i8_pmc4.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_i8_fetch -o f -- python $R/tools/i8p_counters.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_i8_write -o w -- python $R/tools/i8p_counters.py > /dev/null 2>&1
cd $R
timeout 300 python tools/i8p_bench.py 5000 5120 2>&1 | grep "^{" > $O/i8p_bench.jsonl
I8P_SLICES=7 timeout 300 python tools/i8p_bench.py 5000 2>&1 | grep "^{" > $O/i8p_bench_s7.jsonl
[ -f plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so ] && PLSPM_HIP_LIB=plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so I8P_ABLATE=1 timeout 300 python tools/i8p_bench.py 5120 2>&1 | grep "^{" > $O/i8p_ablate.jsonl
timeout 120 python tools/group_enqueue.py 2>/dev/null | grep "^{" > $O/group_enqueue.jsonl
./tools/ubench/mfma_i8_fillers > $O/ubench_mfma_i8_fillers.jsonl 2>&1
# round 3: the wave solver (A/B against the rows solver, kernel time against the batch size, SQ counters of both), sizes next to the headline,
# co-scheduling experiments (narrow Gram tiles + two pipelines), ablation probes of the Gram (experiments build, if present)
timeout 300 python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{" > $O/ab_solver_wave.jsonl
(AB_MODES=BBBBBB timeout 300 python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{"; AB_MODES=ABABAB timeout 300 python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{") > $O/ab_solver_wave_mode_b.jsonl
timeout 300 python tools/solver_rounds.py 2>&1 | grep "^{" > $O/solver_rounds.jsonl
timeout 600 python tools/size_bench.py 2>&1 | grep "^{" > $O/size_bench.jsonl
timeout 600 python tools/solver_quad_ab.py 2>/dev/null | grep "^{" > $O/solver_quad_ab.jsonl
timeout 600 python tools/solver_lv_sweep.py 2>/dev/null | grep "^{" > $O/solver_lv_sweep.jsonl
[ -f plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so ] && PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/experiments/solver_marks_120.py > $O/solver_quad_marks.txt 2>&1
timeout 300 python tools/i8_mix_calib.py 2>&1 | grep "^{" > $O/i8_mix_calib.jsonl
[ -f plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so ] && PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/experiments/solver_marks.py > $O/solver_marks.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $O/prof_solver_pmc1 -o p1 -- python $R/tools/solver_pmc_run.py > /dev/null 2>&1; timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC -d $O/prof_solver_pmc2 -o p2 -- python $R/tools/solver_pmc_run.py > /dev/null 2>&1; timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_solver_fetch -o f -- python $R/tools/solver_pmc_run.py > /dev/null 2>&1; timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_solver_write -o w -- python $R/tools/solver_pmc_run.py > /dev/null 2>&1)
python tools/pmc_rows.py $O solver > $O/solver_pmc.txt 2>&1
# round 5, second half: the same counters for the solvers of the models next to the headline (quad, wave16 <16>, wave16 <8, true>: tools/size_rows.py)
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $O/prof_sizes_pmc1 -o p1 -- python $R/tools/size_rows.py > /dev/null 2>&1; timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/prof_sizes_pmc2 -o p2 -- python $R/tools/size_rows.py > /dev/null 2>&1; timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/prof_sizes_fetch -o p3 -- python $R/tools/size_rows.py > /dev/null 2>&1)
python tools/pmc_rows.py $O solver_quad solver_wave16_kernel\<16 "solver_wave16_kernel<8, true" > $O/solver_sizes_pmc.txt 2>&1
timeout 300 python tools/aux_ab.py i8_priv=0,1 2>&1 | grep "^{" > $O/ab_i8_priv.jsonl
timeout 300 python tools/aux_ab.py i8_slices=0,7 2>&1 | grep "^{" > $O/ab_i8_slices.jsonl
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d /tmp/tl -o tl -- python $R/tools/api_timeline.py run > $O/api_timeline.txt 2>/dev/null; python $R/tools/api_timeline.py read /tmp/tl >> $O/api_timeline.txt 2>&1)
timeout 600 python tools/categorical_bench.py 2>&1 | tail -1 > $O/categorical_bench.json
[ -f plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so ] && CAT_BENCH_STEPS=1 PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/categorical_bench.py 1000 2>&1 | grep clocks | tail -4 > $O/categorical_marks.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $R/tools/categorical_bench.py > /dev/null 2>&1; python $R/tools/kernel_table.py /tmp/cp > $O/categorical_kernels.txt 2>&1; rm -rf /tmp/cp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $R/tools/nonmetric_bench.py > /dev/null 2>&1; python $R/tools/kernel_table.py /tmp/cp > $O/nonmetric_kernels.txt 2>&1)
timeout 600 python tools/hoc_bench.py 2>&1 | tail -1 > $O/hoc_bench.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $R/tools/hoc_bench.py > /dev/null 2>&1; python $R/tools/kernel_table.py /tmp/cp > $O/hoc_kernels.txt 2>&1)
python - "$O" <<'PY'
import sqlite3, glob, sys, json
O = sys.argv[1]
out = []
for db in sorted(glob.glob(O + "/prof_*pmc*/*.db") + glob.glob(O + "/prof_c5_fetch/*.db")):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection c where grid_size = "
         "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name, counter_name")
    for r in cur.execute(q):
        if any(k in r[0] for k in ("gram", "solver_kernel", "solver_rows", "resample", "scores_kernel", "summary")):
            out.append({"run": db.split("/")[-2], "kernel": r[0].split("(")[0].replace("void ", ""), "counter": r[1], "dispatches": r[2], "avg": round(r[3], 1), "avg_duration_ns": round(r[4], 1)})
json.dump(out, open(O + "/pmc_rows.json", "w"), indent=0)
for r in out: print(r["run"], r["kernel"][:36], r["counter"], r["dispatches"], r["avg"], r["avg_duration_ns"])
PY
find $O -name "*.db" -size +20M -delete
du -sh $O; ls $O
