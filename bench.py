#!/usr/bin/env python3
"""Headline benchmark: bootstrap replicates / second, 6-LV satisfaction model, N = 10,000 obs x 60 MVs
(BASELINE.json configs[2]: Mode A, Scheme.PATH, scaled, 5,000 replicates per GPU -- weak scaling).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: every rank resamples + solves its shard of the replicate
range on its own GPU (inputs already resident in HBM), then ONE ncclAllGather (RCCL over xGMI, issued by
libplspm_hip.so itself: plspm_group_bootstrap) merges the B x 156 result rows -- the whole job, gather included, is
inside the timed region.  Rank 0 prints one JSON line.  No torch: under `python -m torch.distributed.run` the launcher
only spawns the ranks (RANK / LOCAL_RANK / WORLD_SIZE); without a launcher `--gpus N` drives N GPUs from this process.

Extra objects on the line:
  roofline      dominant kernel, duration measured with HIP events on the library's own stream during the timed steps.  Default path:
                gram_i8p_kernel<S, MTW, 64> (round 4: S digit planes -- 6 on this data, `digit_planes` -- four waves per workgroup, each
                MTW count tiles x 32 pairs x S planes of 16x16x64 int8 MFMA accumulators, count fragments straight from global memory) --
                the batch's moment matrices as ONE exact int8 MFMA product of the dense resample multiplicities with the S base-256 digit
                planes of the pair products x_p x_q (csrc/kernels_gram_i8p.h); its algorithmic work is 2 N (P+1)(P+2)/2 S int8 ops per
                replicate.  --gram-path 1: the fp64 MFMA Gram of round 1 (gram_rows_kernel<4,false>, SURVEY.md 8(d) flops).
                `fp64_mfma_path` on the line = the same workload on that path; `round3_gram_kernel` = the same steps on the round-3 kernel.
  api_inclusive replicates/s a user of the drop-in API sees: wall of Plspm(data, config, Scheme.PATH, bootstrap=True,
                bootstrap_iterations=5000) minus the wall of the same call without the bootstrap (N = 1 only)
  cpu_baseline  the NumPy oracle (oracle/plspm_oracle.py, a port of the reference arithmetic) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only)
  config.gram_tile_plan_cus  (more than one rank, or --group) the all-gather of step k runs beside the kernels of step k + 1 and a Gram workgroup
                needs a whole CU: before the warm-up steps a few CU counts are tried for the tile-row cut (set_option "i8_cus"; results do not
                depend on it), every rank making the same calls; `chosen` 0 = the device's count stayed
PLSPM_BENCH_SHARED_DEVICE=1 is a test seam: --gpus N with every rank on device 0 (tests/test_gpu_dist.py) -- such a line is not a multi-GPU figure.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))

import numpy as np  # noqa: E402

N_OBS, MVS_PER_LV, N_LV = 10000, 10, 6
REPS_PER_GPU = 5000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6       # MI355X datasheet fp64 matrix; 77.9 TF measured with v_mfma_f64_16x16x4_f64 (tools/ubench)
I8_MFMA_MEASURED_CEILING_TOPS = 3944.0   # MI355X_MICROARCH.md, matrix-core table: a pure v_mfma_i32_16x16x64_i8 stream on this chip (power-limited clock)
I8_MFMA_PEAK_TOPS = 5033.0     # dense int8 matrix peak: 1,024 SIMDs x 2,048 ops/clk x 2.4 GHz = 2 x the ~2.5 PF bf16 dense peak of
                               # MI355X_MICROARCH.md (its table has no int8 spec entry; measured ceilings there: 3,944 TOP/s with
                               # 16x16x64, 4,404 with 32x32x32)


def synth_inputs():
    """The workload: synthetic generator of SURVEY.md 8(d) (tools/synthetic.py -- data only, no PLS arithmetic)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthetic
    X, blocks = synthetic.synth(N_OBS, synthetic.satisfaction_C(), MVS_PER_LV, seed=0)
    return synthetic, X, blocks


SPINUP_STEPS = 200      # untimed launches of the same step before the W warm-up steps (device clocks / power state; ~0.13 s)
PROF_EVERY = 4          # a HIP-event pair around the dominant kernel of every 4th timed step (every step when fewer than 8 are timed)


def cpu_worker(args):
    seed, count = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)                      # one BLAS thread per worker: the pool, not BLAS, spreads over the cores
    except ImportError:
        pass
    synthetic, X, blocks = synth_inputs()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import plspm_oracle as orc                    # the cpu_baseline leg is the ONLY place bench.py touches the oracle
    model = orc.Model(blocks, synthetic.satisfaction_C(), "A" * N_LV, "path", True)
    corr = orc.correction(N_OBS)
    rs = np.random.RandomState(seed)
    t0 = time.perf_counter()
    for _ in range(count):
        orc.bootstrap_replicate(X, model, rs.randint(N_OBS, size=N_OBS), corr)
    return time.perf_counter() - t0


def cpu_baseline(budget_s=20.0):
    """Oracle replicates/s on the host cores: a process pool of min(cpu_count, 32) single-threaded workers."""
    import multiprocessing as mp
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    per_rep = cpu_worker((1, 8)) / 8                # single-core seconds per replicate (loop only; data synthesis excluded)
    cores = max(1, min(os.cpu_count() or 1, 32))
    count = min(64, max(4, int(budget_s / max(per_rep, 1e-3))))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        busy = pool.map(cpu_worker, [(100 + i, count) for i in range(cores)])
        wall = time.perf_counter() - t0
    # throughput of the replicate loops themselves (workers time only their loop; data synthesis excluded)
    value = cores * count / max(busy)
    return {"value": round(value, 2), "unit": "replicates/s", "cores": cores, "kind": "port",
            "sample": "%d oracle replicates per core on %d cores (10k x 60 x 6, Mode A, PATH), single-core %.2f rep/s, pool wall %.1f s"
                      % (count, cores, 1.0 / per_rep, wall)}


def kernel_key(name):
    """Kernel name as rocprofv3 prints it -> comparison key: no `void `, no argument list, no blanks."""
    return name.replace("void ", "").split("(")[0].replace(" ", "")


def live_traffic(kernel_name, gram_path, grid_hint=None):
    """HBM bytes per launch of the dominant kernel -- the EXACT instantiation `kernel_name` the timed steps launched (a run also launches other
    instantiations of the same template: sub-batches of the host-buffer leg, the seven-plane leg), measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes, counters only -- no trace domain), read back from the rocpd databases.  FETCH_SIZE is doubled
    (gfx950 reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section; checked in tools/rocprof_summary.py on a
    kernel of known traffic).  None when rocprofv3 is missing or a pass fails -- the caller then falls back to the committed figure."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    kib = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="plspm_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-api", "--no-traffic", "--no-next-rows", "--no-single-fit", "--gram-path", str(gram_path)]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            dbs = glob.glob(os.path.join(d, "*.db")) + glob.glob(os.path.join(d, "*", "*.db"))
            cur = sqlite3.connect(dbs[0]).cursor()
            q = ("select kernel_name, count(*), avg(value) from counters_collection c where counter_name=? and grid_size = "
                 "(select max(grid_size) from counters_collection c2 where c2.kernel_name = c.kernel_name) group by kernel_name")
            hit = [(n, avg) for name, n, avg in cur.execute(q, (ctr,)) if kernel_key(name) == kernel_key(kernel_name)]
            if len(hit) != 1:                                  # the timed instantiation was not launched by the child (or the name is ambiguous): no figure rather than a wrong one
                return None
            kib[ctr] = hit[0][1]
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024)


def moment_errors_vs_80bit(auto_model, seven_model, f64_model, X, reps=3):
    """Worst entry of the replicates' moment matrices against 80-bit (np.longdouble) sums over the same resampled rows, relative to
    sqrt(M_pp M_qq): automatic plane count, seven planes, fp64 MFMA route -- on THIS run's data (explicit index lists, `reps` replicates)."""
    rng = np.random.default_rng(3)
    idx = rng.integers(0, N_OBS, size=(reps, N_OBS)).astype(np.int32)
    f64_model.set_option("gram_path", 1)
    shift = auto_model.fit(want_scores=False)["mean"]                       # the device's own column means (its mean-shifted columns are what is multiplied)
    Xa = np.concatenate((X - shift[None, :], np.ones((N_OBS, 1))), axis=1).astype(np.longdouble)
    refs = []
    for b in range(reps):
        c = np.bincount(idx[b], minlength=N_OBS).astype(np.longdouble)
        refs.append((Xa * c[:, None]).T @ Xa)
    out = {}
    for name, mdl in (("automatic", auto_model), ("seven_planes", seven_model), ("fp64_mfma_route", f64_model)):
        M = mdl.bootstrap_moments(reps, idx=idx)
        worst = 0.0
        for b in range(reps):
            scale = np.sqrt(np.outer(np.diag(refs[b]), np.diag(refs[b])))
            worst = max(worst, float(np.max(np.abs(M[b].astype(np.longdouble) - refs[b]) / scale)))
        out[name] = worst
        out[name + "_route"] = {"gram_path": mdl.get_option("last_gram_path"), "planes": mdl.get_option("last_i8_slices") if mdl.get_option("last_gram_path") == 2 else None}
    out["definition"] = ("max over %d replicates (explicit index lists) and all 61 x 61 entries of |M - M_80bit| / sqrt(M_pp M_qq); M_80bit = the same "
                         "mean-shifted fp64 columns multiplied and summed in np.longdouble" % reps)
    return out


def api_inclusive(X, reps, pairs=7):
    """Replicates/s through the drop-in API: Plspm(..., bootstrap=True) minus the same call with bootstrap=False (median of paired
    runs).  The summaries are computed (on the device) inside the call; the reference-shaped frames are built on access."""
    import pandas as pd
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    import synthetic
    lvs = synthetic.SAT_LVS
    cols = ["%s%d" % (lv.lower(), k) for lv in lvs for k in range(MVS_PER_LV)]
    frame = pd.DataFrame(X, columns=cols)
    structure = c.Structure()
    for frm, to in synthetic.SAT_EDGES:
        structure.add_path([frm], [to])

    def config():
        cfg = c.Config(structure.path(), scaled=True)
        for lv in lvs:
            cfg.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
        return cfg
    diffs, fits, boots, inner, tails = [], [], [], [], []
    for k in range(pairs + 1):
        t0 = time.perf_counter()
        Plspm(frame, config(), Scheme.PATH)
        t1 = time.perf_counter()
        m = Plspm(frame, config(), Scheme.PATH, bootstrap=True, bootstrap_iterations=reps, processes=1, seed=1)
        t2 = time.perf_counter()
        if k:                                        # the first pair warms the code objects
            fits.append(t1 - t0); boots.append(t2 - t1); diffs.append((t2 - t1) - (t1 - t0)); inner.append(m.timings()["bootstrap_latency_s"]); tails.append(m.timings()["bootstrap_s"])
        used = m.bootstrap().used()
    # the same work with nothing to hide under: a fresh handle per call (create, upload, fit -- as Plspm does), then enqueue -> sync ->
    # device summary, each host-synchronised
    from plspm import _native
    C = synthetic.satisfaction_C()
    boff = np.arange(0, 61, 10).astype(np.int32)
    alone, keep, prep = [], None, []
    for k in range(pairs + 2):
        h = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(N_LV, dtype=np.int32), 2, True, 100, 1e-6, 0)
        h.upload(X)
        tp = time.perf_counter()
        h.prepare_bootstrap()                        # as Plspm(bootstrap=True) does behind its upload: the digit planes are built beside the fit
        prep.append(time.perf_counter() - tp)
        h.fit(want_scores=True, want_cov=True)
        t0 = time.perf_counter()
        h.bootstrap_device(reps, seed=1)
        h.summary(reps, np.ones(h.row_width))
        alone.append(time.perf_counter() - t0)
        keep = h                                     # the previous handle stays alive while the next one is built, as in the loop above
    standalone = float(np.median(alone[2:]))
    diff = float(np.median(diffs))
    latency = float(np.median(inner))
    bound = max(latency, standalone)
    return {"value": round(reps / bound, 1), "unit": "replicates/s",
            "value_r02_definition": round(reps / max(diff, standalone), 1),
            "prepare_ms": round(float(np.median(prep[2:])) * 1e3, 3),
            "standalone_ms": round(standalone * 1e3, 3), "paired_difference_ms": round(diff * 1e3, 3),
            "bootstrap_tail_ms": round(float(np.median(tails)) * 1e3, 3), "bootstrap_latency_ms": round(float(np.median(inner)) * 1e3, 3),
            "plspm_fit_wall_ms": round(float(np.median(fits)) * 1e3, 3), "plspm_fit_plus_bootstrap_wall_ms": round(float(np.median(boots)) * 1e3, 3),
            "replicates_used": int(used),
            "note": "paired_difference_ms = median over %d pairs of [wall of Plspm(bootstrap=True, bootstrap_iterations=%d)] - [wall of Plspm()] (the replicates "
                    "are enqueued right after the fit; the pandas report frames are built on access); bootstrap_tail_ms / bootstrap_latency_ms = Plspm.timings(); "
                    "standalone_ms = the same bootstrap on a fresh handle behind upload + plspm_bootstrap_prepare + fit, nothing overlapped (enqueue -> "
                    "kernels -> record transpose + device summaries -> %d x 6 table on the host, one stream synchronise); bootstrap_latency_ms = the same span "
                    "measured inside Plspm(bootstrap=True) (enqueue of the replicates -> summaries on the host; Plspm.timings()); value = replicates / "
                    "max(bootstrap_latency_ms, standalone_ms) -- the bootstrap's own critical path, nothing credited for what the host does meanwhile; "
                    "value_r02_definition = replicates / max(paired_difference_ms, standalone_ms), the figure of rounds 1-2 (comparable across rounds); "
                    "prepare_ms = host time of plspm_bootstrap_prepare (enqueue only since round 4: column statistics; the planes are cut in the tail "
                    "of the fit), outside standalone_ms like the upload and the fit; rows stay in HBM" % (pairs, reps, 156)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps-per-gpu", type=int, default=REPS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the categorical / Scale.NUM / HOC bootstraps and the metric models beside the headline (child runs of tools/categorical_bench.py, nonmetric_bench.py, hoc_bench.py, size_rows.py)")
    ap.add_argument("--no-single-fit", action="store_true", help="skip the single-fit rows (configs[1] and configs[4]: a child run of tools/fit_bench.py)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure the dominant kernel's HBM bytes")
    ap.add_argument("--group", action="store_true", help="N = 1 without a launcher: still go through the group / RCCL path")
    ap.add_argument("--gather", choices=["all", "root"], default="all", help="all (default): ONE ncclAllGather per step -- every rank receives every shard; root: the gather to rank 0 "
                                                                               "(group option gather_root: ncclSend / ncclRecv, 1 / N of the bytes; only rank 0 checks the merged records)")
    ap.add_argument("--no-transport-calibration", action="store_true", help="keep RCCL as created (skip the untimed comparison with channel-capped RCCL / the copy-engine exchange)")
    ap.add_argument("--gram-path", type=int, default=0, choices=[0, 1, 2], help="0 library default (int8 digit planes), 1 fp64 MFMA Gram, 2 int8 digit planes")
    args = ap.parse_args()

    from plspm import _native, parallel
    launched = "RANK" in os.environ            # one process per GPU (a launcher set RANK / LOCAL_RANK / WORLD_SIZE)
    if launched:
        t_init = time.perf_counter()
        ctx = parallel.init_process_group()    # file rendezvous of the ncclUniqueId + ncclCommInitRank, inside libplspm_hip.so
        ctx.comm.create_s = time.perf_counter() - t_init
        rank, world, devices = ctx.rank, ctx.world, [ctx.local_rank]
        if world != args.gpus:
            raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
        comm = ctx.comm
    else:
        rank, world, devices = 0, args.gpus, list(range(args.gpus))
        shared_device_test = os.environ.get("PLSPM_BENCH_SHARED_DEVICE") == "1"      # test seam: the multi-rank code path with every rank on device 0
        if shared_device_test:
            devices = [0] * args.gpus
        elif args.gpus > _native.device_count():
            raise SystemExit("--gpus %d but %d HIP devices are visible" % (args.gpus, _native.device_count()))
        comm = parallel.local_comm(devices) if (world > 1 or args.group) else None
    comm_create_s = round(getattr(comm, "create_s", 0.0), 3) if comm is not None else None      # librccl load + ncclCommInit* (+ the id rendezvous under a launcher): once per process

    synthetic, X, blocks = synth_inputs()
    C = synthetic.satisfaction_C()
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)

    def make_model(device):
        mdl = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(N_LV, dtype=np.int32), 2, True, 100, 1e-6, device)
        mdl.upload(X)                                          # X resident in HBM before the timed region
        if args.gram_path:
            mdl.set_option("gram_path", args.gram_path)
        return mdl

    upload_s = []
    models = []
    for d in devices:
        t_up = time.perf_counter()
        models.append(make_model(d))
        models[-1].sync()
        upload_s.append(round(time.perf_counter() - t_up, 4))      # handle + descriptors + X over PCIe + mean shift, per device (the first one also pays the HIP context)
    model = models[0]

    def make_group(on_comm):
        g = _native.NativeGroup(on_comm, models)
        if args.gather == "root":
            g.set_option("gather_root", 1)
        g.set_option("chunks", 1)      # the step loop issues calls back to back: the gather of call k overlaps the kernels of call k + 1 as it is (sub-batches are for ONE call: single_call below)
        return g
    group = make_group(comm) if comm is not None else None
    # a multi-GPU run validates itself: the communicator really spans --gpus ranks, on distinct devices the records travel through
    # RCCL (never the same-device copy route of the 1-GPU test box), and the shards partition the replicate range
    transport, ranks_seen, shards = "none", 1, [[0, args.reps_per_gpu * world]]
    if group is not None:
        ranks_seen = int(comm.nranks)
        assert ranks_seen == world == group.nranks, "communicator spans %d ranks, group %d, --gpus %d" % (ranks_seen, group.nranks, world)
        transport = comm.transport
        if world > 1 and not comm.uses_rccl and os.environ.get("PLSPM_BENCH_SHARED_DEVICE") != "1":
            raise SystemExit("bench: %d ranks but the records would travel by device-to-device copies (ranks share a device): not a multi-GPU run" % world)
        shards = [list(group.shard(args.reps_per_gpu * world, r)) for r in range(world)]
        assert shards[0][0] == 0 and all(shards[r][0] + shards[r][1] == (shards[r + 1][0] if r + 1 < world else args.reps_per_gpu * world) for r in range(world)), shards
    B_total = args.reps_per_gpu * world
    width = model.row_width
    state = {"k": 0}
    live = {"group": group, "comm": comm}

    def step():
        """One batch: a fresh replicate-id range every step (ids k*B .. (k+1)*B of the seeded stream).  Enqueue only -- with a
        group the all-gather of step k runs on its own stream beside the kernels of step k+1 (double-buffered records)."""
        offset = state["k"] * B_total
        state["k"] += 1
        if group is None:
            model.bootstrap_device(B_total, seed=1, rep_offset=offset)
        else:
            live["group"].bootstrap(B_total, seed=1, rep_offset=offset)

    def fence():
        if group is None:
            model.sync()
        else:
            live["group"].sync()                               # this process's kernel and gather streams
            live["group"].barrier()                            # every rank of the job (all-reduce of one word over RCCL)

    # cold figure: the W warm-up steps and K timed steps straight after the upload, BEFORE the spin-up below -- what the driver's
    # --warmup alone buys (reported beside `value`; every rank makes the same calls)
    for _ in range(args.warmup):
        step()
    fence()
    cold_t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    cold_elapsed = time.perf_counter() - cold_t0
    if group is not None:
        cold_elapsed = group.max(cold_elapsed)
    # spin-up: steps straight after an idle period run 10-16 % slower than steps in a stream (`cold` on the line against `value`: 0.55-0.59
    # against 0.48-0.51 ms per step in the round-3 / round-4 driver runs: clocks / power state), and W is the driver's choice -- so the device
    # is brought to its working state with SPINUP_STEPS of the same launches before the W warm-up steps; nothing of it is reused by the
    # timed steps (fresh replicate ids)
    spin_t0, spin_steps = time.perf_counter(), SPINUP_STEPS          # (a fixed count: every rank of a job must make the same collective calls)
    for i in range(spin_steps):
        step()
        if i % 10 == 9:
            fence()
    fence()
    spin_s = time.perf_counter() - spin_t0
    # More than one rank: the all-gather of step k runs beside the kernels of step k + 1, and a Gram workgroup needs a whole CU -- every CU an
    # RCCL channel occupies is missing from the launch's last round.  The tile-row cut can be planned for fewer CUs ("i8_cus"; results do not
    # depend on the cut): a few candidates are tried here, untimed, every rank making the same calls and deciding on the max-over-ranks time.
    # How the records travel (round 5): RCCL's default channel count is sized for large all-reduces, while this job issues ONE
    # all-gather of a few MB per step beside a Gram that needs whole CUs.  Candidates, each timed untimed-region-style on a group of its own over the
    # SAME handles (every rank makes the same calls, decisions on the max-over-ranks time): RCCL as created (the default -- it stays unless a
    # candidate is 1.5 % faster), RCCL communicators split off it with at most 8 / 2 channels (ncclCommSplit + ncclConfig_t.maxCTAs), and -- one
    # process driving distinct devices -- the copy-engine exchange on peer-mapped buffers (no kernel at all).
    transport_cal = None
    if group is not None and comm.uses_rccl and not args.no_transport_calibration:
        def time_steps(n_warm=4, n_timed=16):
            for _ in range(n_warm):
                step()
            fence()
            c0 = time.perf_counter()
            for _ in range(n_timed):
                step()
            fence()
            return live["group"].max(time.perf_counter() - c0) / n_timed * 1e3
        cands = {"rccl": comm}
        errors = {}
        for cap in (8, 2):
            try:
                cands["rccl(max_channels=%d)" % cap] = comm.split(cap)
            except Exception as exc:                           # (an RCCL without ncclCommSplit: every rank fails alike, the default stays)
                errors["rccl(max_channels=%d)" % cap] = str(exc)[:200]
        if not launched and world > 1:
            try:
                cands["copy-engines"] = _native.NativeComm(devices, transport="copy")
            except Exception as exc:
                errors["copy-engines"] = str(exc)[:200]
        tried = {}
        for rnd in range(2):
            for name, cc in cands.items():
                live["group"].close()
                live["group"] = make_group(cc)
                ms = round(time_steps(), 4)
                tried[name] = min(ms, tried.get(name, ms))
        best = min(tried, key=lambda k: tried[k])
        chosen = best if tried[best] < 0.985 * tried["rccl"] else "rccl"
        live["group"].close()
        live["group"] = make_group(cands[chosen])
        live["comm"] = cands[chosen]
        for name, cc in cands.items():
            if name != chosen and cc is not comm:
                cc.close()
        group = live["group"]
        transport = chosen
        transport_cal = {"tried_ms_per_step": tried, "chosen": chosen, "errors": errors,
                         "split_s": {name: round(getattr(cc, "create_s", 0.0), 3) for name, cc in cands.items() if cc is not comm}}
    plan_cus = {"chosen": 0, "tried_ms_per_step": {}}
    if group is not None:                                      # (also the one-rank --group run: RCCL's kernel is there all the same)
        for rnd in range(3):                                   # (three passes over the candidates, the best of each kept: single passes scatter)
            for cand in (0, 248, 240, 232, 224, 208):
                for mdl in models:
                    mdl.set_option("i8_cus", cand)
                for _ in range(4):
                    step()
                fence()
                c0 = time.perf_counter()
                for _ in range(16):
                    step()
                fence()
                ms = round(live["group"].max(time.perf_counter() - c0) / 16 * 1e3, 4)
                plan_cus["tried_ms_per_step"][str(cand)] = min(ms, plan_cus["tried_ms_per_step"].get(str(cand), ms))
        tried = plan_cus["tried_ms_per_step"]
        best = min(tried, key=lambda k: tried[k])
        if tried[best] < 0.985 * tried["0"]:                   # (a default that is within 1.5 % stays)
            plan_cus["chosen"] = int(best)
        for mdl in models:
            mdl.set_option("i8_cus", plan_cus["chosen"])
    for _ in range(args.warmup):
        step()
    fence()
    profiled = group is None                                   # single stream: the HIP events of the timed region see each kernel alone
    if profiled:
        model.profile(True)
        model.profile_reset()
    # HIP events bracket the dominant kernel of every PROF_EVERY-th step of the timed region (an event pair costs a few microseconds of
    # stream time: on every step they would slow down the very thing they measure)
    prof_every = PROF_EVERY if args.steps >= 2 * PROF_EVERY else 1
    t0 = time.perf_counter()
    for i in range(args.steps):
        if profiled:
            model.profile(i % prof_every == 0, only="gram")    # the dominant kernel only: the roofline's launch duration
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # the launch the timed region ran, described NOW: the legs below (host-buffer sub-batches, one-row probes) launch other instantiations and
    # overwrite the handle's last_* words
    timed_launch = {k: model.get_option(k) for k in ("last_gram_path", "last_i8_slices", "last_i8_mt", "last_i8_short", "last_i8_rt", "last_i8_priv", "last_i8_dma", "last_solver")}
    if profiled:
        model.profile(False)
        gram_timed = model.profile_read("gram")
        # the other kernels of a step (kernels_ms_per_step): ten more steps with every kernel bracketed, outside the timed region
        model.profile(True)
        model.profile_reset()
        for i in range(10):
            step()
        fence()
        model.profile(False)
    if group is not None:
        elapsed = group.max(elapsed)                           # max over ranks

    # what was timed is correct: every replicate of the last step converged, the gathered records are complete and in order
    last_offset = (state["k"] - 1) * B_total
    holds_records = not (group is not None and args.gather == "root" and rank != 0)      # (--gather root: the merged records exist on rank 0 only)
    if group is not None and holds_records:
        rows_l, status_l, iters_l = group.rows()
    elif group is None:
        rows_l, status_l, iters_l = model.fetch(0, B_total)
    probe = B_total - 3                                        # a row owned by the last rank
    one, _, _ = model.bootstrap(1, seed=1, rep_offset=last_offset + probe)
    if holds_records:
        assert rows_l.shape == (B_total, width) and np.all(status_l == 0), "a replicate of the timed batch failed"
        assert np.array_equal(rows_l[probe], one[0]), "sharded stream differs from the single-GPU stream"
    # ... and equals the reference arithmetic: replicate 0 of the seeded stream against the committed oracle row
    # (tests/golden/bench_guard.npz, made by tests/golden/make_bench_guard.py with the oracle)
    rows, status, iters = model.bootstrap(8, seed=1, rep_offset=0)
    assert np.all(status == 0), status
    guard = np.load(os.path.join(ROOT, "tests", "golden", "bench_guard.npz"))
    idx0 = _native.bootstrap_indices(1, 0, N_OBS)
    assert int(idx0.astype(np.int64).sum()) == int(guard["idx_sum"]) and np.array_equal(idx0[:16], guard["idx_head"]), "resampling stream changed"
    assert int(guard["iterations"]) == iters[0] and np.allclose(rows[0], guard["row"], rtol=1e-8, atol=1e-11), "timed path disagrees with the oracle"

    if not profiled:
        # N > 1: the kernels of the timed region overlap the previous step's collective, so the dominant kernel is timed on a
        # calibration pass of the same launches on the handle's stream alone (same B per GPU, same data)
        model.sync()
        model.profile(True)
        model.profile_reset()
        for k in range(min(args.steps, 10)):
            model.bootstrap_device(args.reps_per_gpu, seed=1, rep_offset=k * args.reps_per_gpu)
        model.sync()
        model.profile(False)

    single_call = None
    if group is not None:
        # ONE call of the whole job (configs[3] as a user issues it: Plspm(bootstrap_iterations = N x 5,000, devices = ...)): enqueue -> every rank's
        # records gathered on every rank, nothing of a neighbouring call to hide under.  With one sub-batch the whole gather is exposed; with the
        # automatic sub-batches (plspm_group_bootstrap, round 5) only the last, smallest one.  Median of 7 calls each, max over ranks.
        def one_call(chunks):
            group.set_option("chunks", chunks)
            ts = []
            for _ in range(8):
                fence()
                c0 = time.perf_counter()
                step()
                group.sync()
                ts.append(time.perf_counter() - c0)
            return group.max(float(np.median(ts[1:]))) * 1e3
        single_call = {"one_sub_batch_ms": round(one_call(1), 4), "sub_batches_ms": round(one_call(0), 4),
                       "sub_batch_plan": [[a, n] for a, n in group.plan(B_total)],
                       "note": "wall of ONE plspm_group_bootstrap call of %d replicates + plspm_group_sync on an idle job (median of 7, max over ranks): one sub-batch "
                               "(the whole gather exposed) / automatic sub-batches (the gather of sub-batch k beside the kernels of k + 1); the timed steps "
                               "above run back to back with one sub-batch per call" % B_total}
        group.set_option("chunks", 1)
        fence()

    pcie = None
    if world == 1 and group is None:
        # the same batch through the host-buffer entry point (records copied back over PCIe every step); never `value`
        host = (np.empty((B_total, width)), np.empty(B_total, dtype=np.int32), np.empty(B_total, dtype=np.int32))     # caller-owned, re-used
        for _ in range(3):
            model.bootstrap(B_total, seed=1, out=host)
        t1 = time.perf_counter()
        for _ in range(10):
            model.bootstrap(B_total, seed=1, out=host)
        pcie = B_total * 10 / (time.perf_counter() - t1)

    if rank == 0:
        gram_ms, gram_n = gram_timed if profiled else model.profile_read("gram")
        res_ms, res_n = model.profile_read("resample")
        sol_ms, sol_n = model.profile_read("solver")
        used_path = timed_launch["last_gram_path"]
        reps_per_launch = args.reps_per_gpu
        a_rep = 8.0 * N_OBS * 60 + 4.0 * N_OBS                 # one gathered read of X + the index vector
        f_rep = float(N_OBS) * 60 * 61                         # symmetric Gram flops
        gram_avg_ms = gram_ms / max(gram_n, 1)
        hbm_achieved = a_rep * reps_per_launch / (gram_avg_ms * 1e-3) / 1e9
        f64_equiv = f_rep * reps_per_launch / (gram_avg_ms * 1e-3) / 1e12

        def static_traffic(names):
            for name in names:
                prof_json = os.path.join(ROOT, "profiles", name)
                if os.path.exists(prof_json):
                    try:
                        return (json.load(open(prof_json)).get("hbm_bytes_per_launch"),
                                "static: profiles/%s (rocprofv3 --pmc passes of this command; PMC counters cannot be read from inside the bench)" % name)
                    except Exception:
                        pass
            return None, None
        timing_note = ("HIP events around the kernel's launches in every %d-th step of the timed region (single stream: the kernel alone)" % prof_every if profiled else
                       "HIP events on the handle's stream over a calibration pass of the same launches without the overlapping collective")
        if used_path == 2:
            slices, priv, rt = timed_launch["last_i8_slices"], timed_launch["last_i8_priv"], timed_launch["last_i8_rt"]
            kname = ("gram_i8p_kernel<%d, %d, 64, %s>" % (slices, rt // 4, "true" if timed_launch["last_i8_short"] else "false") if priv else
                     "gram_i8_kernel<%d, %d, %d, %d, %d, false>" % (slices, model.get_option("i8_waves") // 2, 803 if timed_launch["last_i8_dma"] == 2 else 3, model.get_option("i8_shape"), rt))
        else:
            kname = "gram_rows_kernel<4, false>"
        live = None
        if world == 1 and group is None and not launched and not args.no_traffic:
            live = live_traffic(kname, used_path)
        live_src = ("live: two child runs of this script (3 steps) under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters "
                    "only), average over the kernel's full-size launches, FETCH_SIZE x 2 (gfx950)")
        if used_path == 2:
            npair = 61 * 62 // 2                               # unordered pairs of the 60 columns + the ones column
            ops_rep = 2.0 * N_OBS * npair * slices             # int8 multiply-adds x 2, unpadded
            achieved = ops_rep * reps_per_launch / (gram_avg_ms * 1e-3) / 1e12
            traffic, traffic_src = (live, live_src) if live else static_traffic(("r04_gram_i8_traffic.json", "r03_gram_i8_traffic.json", "r02c_gram_i8_traffic.json"))
            # what the launch executes: whole tiles of 320 / 256 replicates x 64 rows x 32 pairs (= SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 of the PMC passes
            # in profiles/)
            k_rows = ((N_OBS + 127) // 128) * 128
            # (replicate slots: the launch's tile rows -- 320-replicate and, cut by plspm_gram_i8.hip i8_mix_plan, 256-replicate ones -- x their heights)
            rep_slots = 16 * timed_launch["last_i8_mt"]
            executed = 2.0 * rep_slots * k_rows * (((npair + 31) // 32) * 32) * slices
            roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                        "frac": round(achieved / I8_MFMA_PEAK_TOPS, 4),
                        "frac_of_measured_ceiling": round(achieved / I8_MFMA_MEASURED_CEILING_TOPS, 4),
                        "measured_ceiling": {"value": I8_MFMA_MEASURED_CEILING_TOPS, "unit": "TOP/s",
                                             "source": "MI355X_MICROARCH.md matrix-core table (I8, 16x16x64 micro-benchmark); the nominal peak is 2 x the bf16 dense spec"},
                        "executed_ops": executed, "executed_over_algorithmic": round(executed / (ops_rep * reps_per_launch), 4),
                        "tile_rows": {"workgroup_tile_replicates": 16 * rt, "short_rows": timed_launch["last_i8_short"], "short_row_replicates": 16 * (rt - 4), "replicate_slots": rep_slots},
                        "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": kname, "avg_launch_ms": round(gram_avg_ms, 4), "launches": gram_n, "note": timing_note,
                        "ops": "int8 multiply-add = 2 ops (TOP/s; v_mfma_i32_16x16x64_i8, exact int32 accumulation); peak = dense int8 matrix peak",
                        "algorithmic_ops_per_replicate": ops_rep,
                        "algorithmic_ops_derivation": "2 x N rows x %d pair columns x %d digit planes (SURVEY 8(d)'s N P (P+1) fp64 flops = %.4g per replicate, "
                                                      "each fp64 multiply-add carried by %d exact int8 ones)" % (npair, slices, f_rep, slices),
                        "survey_8d_model": {"not_a_roofline": "SURVEY 8(d) priced a replicate as one gathered read of X (A_rep bytes) feeding an fp64 Gram (F_rep flops); this "
                                                              "kernel neither gathers X per replicate nor multiplies in fp64 (one int8 product of the batch's count matrix with digit "
                                                              "planes shared by all replicates), so both rates below exceed the respective peaks -- they restate the kernel's time in "
                                                              "8(d)'s units and bound nothing",
                                            "algorithmic_bytes_per_replicate": a_rep, "algorithmic_flops_per_replicate": f_rep,
                                            "A_rep_rate_GBps": round(hbm_achieved, 1), "A_rep_rate_over_hbm_peak": round(hbm_achieved / HBM_PEAK_GBS, 3),
                                            "F_rep_rate_TFLOPs": round(f64_equiv, 1), "F_rep_rate_over_fp64_mfma_peak": round(f64_equiv / FP64_MFMA_PEAK_TF, 3)}}
        else:
            traffic, traffic_src = (live, live_src) if live else static_traffic(("r02_gram_traffic.json", "r01_gram_traffic.json"))
            roofline = {"bound": "mfma", "achieved": round(f64_equiv, 2), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(f64_equiv / FP64_MFMA_PEAK_TF, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": kname, "avg_launch_ms": round(gram_avg_ms, 4), "launches": gram_n, "note": timing_note,
                        "algorithmic_flops_per_replicate": f_rep, "algorithmic_bytes_per_replicate": a_rep,
                        "A_rep_rate_GBps": round(hbm_achieved, 1), "A_rep_rate_over_hbm_peak": round(hbm_achieved / HBM_PEAK_GBS, 4)}
        other = None
        if world == 1 and group is None and used_path == 2:
            # the same workload on the fp64 MFMA Gram (round 1's dominant kernel), for the record
            alt = make_model(devices[0])
            alt.set_option("gram_path", 1)
            for k in range(3):
                alt.bootstrap_device(B_total, seed=1, rep_offset=k * B_total)
            alt.sync()
            alt.profile(True); alt.profile_reset()
            t1 = time.perf_counter()
            for k in range(10):
                alt.bootstrap_device(B_total, seed=1, rep_offset=(3 + k) * B_total)
            alt.sync()
            dt = (time.perf_counter() - t1) / 10
            alt.profile(False)
            g_ms, g_n = alt.profile_read("gram")
            tf = f_rep * reps_per_launch / (g_ms / max(g_n, 1) * 1e-3) / 1e12
            other = {"value": round(B_total / dt, 1), "unit": "replicates/s", "ms_per_step": round(dt * 1e3, 4), "kernel": "gram_rows_kernel<4,false>",
                     "gram_avg_launch_ms": round(g_ms / max(g_n, 1), 4), "roofline_frac": round(tf / FP64_MFMA_PEAK_TF, 4),
                     "note": "10 steps of the same batch with set_option('gram_path', 1): fp64 MFMA Gram over (row,count) lists; frac = SURVEY 8(d) flops / 78.6 TFLOP/s"}
        planes = None
        if world == 1 and group is None and used_path == 2:
            # the same workload with seven digit planes (correctly rounded sums) beside the automatic plane count, and how far the records differ
            alt7 = make_model(devices[0])
            alt7.set_option("i8_slices", 7)
            for k in range(100):
                alt7.bootstrap_device(B_total, seed=1, rep_offset=k * B_total)
            alt7.sync()
            alt7.profile(True); alt7.profile_reset()
            t1 = time.perf_counter()
            for k in range(20):
                alt7.bootstrap_device(B_total, seed=1, rep_offset=(3 + k) * B_total)
            alt7.sync()
            dt7 = (time.perf_counter() - t1) / 20
            alt7.profile(False)
            g7_ms, g7_n = alt7.profile_read("gram")
            r_auto = model.bootstrap(256, seed=1, rep_offset=0)[0]
            r_7 = alt7.bootstrap(256, seed=1, rep_offset=0)[0]
            dev = float(np.max(np.abs(r_auto - r_7) / np.maximum(np.abs(r_7), 1e-3)))
            merr = moment_errors_vs_80bit(model, alt7, make_model(devices[0]), X)
            planes = {"S": int(slices), "min_sum_over_max": int(model.get_option("last_i8_ratio")),
                      "rule": "fewest planes whose worst-case error of a replicate's sum, N 2^-(8S-1) max|z|, stays below a quarter of the a-priori bound N 2^-53 sum|z| of an "
                              "fp64 accumulation of the same terms, in every pair column of the uploaded data (S = 6 needs sum|z| >= 256 max|z|; else 7)",
                      "seven_planes": {"value": round(B_total / dt7, 1), "unit": "replicates/s", "ms_per_step": round(dt7 * 1e3, 4),
                                       "gram_avg_launch_ms": round(g7_ms / max(g7_n, 1), 4),
                                       "note": "20 steps with set_option('i8_slices', 7): sums correctly rounded (>= 53 bits of every column maximum)"},
                      "moment_error_vs_80bit": merr,
                      "max_rel_record_difference_vs_seven_planes": dev,
                      "note": "records of 256 replicates (same seed) on both plane counts, |a - b| / max(|b|, 1e-3); the parity bar is 1e-6 (BASELINE.json north_star), "
                              "the tests hold both against the oracle at 1e-8 and the moment matrices against 80-bit sums (tests/test_gpu_gram_i8.py)"}
        round3 = None
        if world == 1 and group is None and used_path == 2 and model.get_option("last_i8_priv"):
            # the same steps on the round-3 Gram kernel (both operands through LDS), same box, same clocks regime: the A/B of round 4's kernel
            alt3 = make_model(devices[0])
            alt3.set_option("i8_priv", 0)
            for k in range(100):
                alt3.bootstrap_device(B_total, seed=1, rep_offset=k * B_total)
            alt3.sync()
            alt3.profile(True, only="gram"); alt3.profile_reset()
            t1 = time.perf_counter()
            for k in range(20):
                alt3.bootstrap_device(B_total, seed=1, rep_offset=(3 + k) * B_total)
            alt3.sync()
            dt3 = (time.perf_counter() - t1) / 20
            alt3.profile(False)
            g3_ms, g3_n = alt3.profile_read("gram")
            k3 = "gram_i8_kernel<%d, 4, 3, 16, %d, false>" % (alt3.get_option("last_i8_slices"), alt3.get_option("last_i8_rt"))
            same = bool(np.array_equal(alt3.bootstrap(64, seed=1, rep_offset=0)[0], model.bootstrap(64, seed=1, rep_offset=0)[0]))
            round3 = {"kernel": k3,
                      "gram_avg_launch_ms": round(g3_ms / max(g3_n, 1), 4), "ms_per_step": round(dt3 * 1e3, 4), "records_bit_identical": same,
                      "note": "20 steps with set_option('i8_priv', 0) behind 100 spin-up steps, every Gram launch between HIP events"}
        parallelism = ("one process, one GPU, no collective" if group is None else
                       "replicate-sharded x%d (%s), ONE ncclAllGather per step issued by libplspm_hip.so on a gather stream "
                       "(overlaps the next step's kernels; records double-buffered)" % (world, "one process per GPU" if launched else "one process, %d GPUs" % world))
        line = {
            "metric": "bootstrap replicates/sec (6-LV satisfaction model, N=10k)",
            "value": round(B_total * args.steps / elapsed, 1),
            "unit": "replicates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "dtype_note": ("fp64 data, moments and solver; the batch Gram is evaluated as an exact int8 x int8 -> int32 product on %d base-256 digit "
                           "planes of the fp64 products (the integer sums are exact; the plane count is the fewest whose representation error stays "
                           "below a quarter of the a-priori rounding bound of an fp64 accumulation of the same terms -- `digit_planes`).  Measured on this "
                           "data against 80-bit sums (`digit_planes.moment_error_vs_80bit`): seven planes sit below the fp64 MFMA route's error, the "
                           "automatic %d planes within a small multiple of it -- all nine orders below the 1e-6 the records are held to; the seven-plane "
                           "rate is on the line beside `value`" % (slices, slices)) if used_path == 2 else "fp64 throughout",
            "config": {"workload": "synthetic 10,000 obs x 60 MVs x 6 LVs, Mode A, Scheme.PATH, scaled, %d bootstrap replicates per GPU "
                                   "(BASELINE.json configs[2]; %d GPUs x %d = configs[3] at 8); on-device Philox resampling, a fresh replicate-id "
                                   "range every step; X resident in HBM" % (args.reps_per_gpu, world, args.reps_per_gpu),
                       "replicates_per_step": B_total, "iterations_per_replicate": [int(iters_l.min()), int(iters_l.max())],
                       "parallelism": parallelism, "transport": transport, "ranks_seen_by_rccl": ranks_seen if transport == "rccl" else 0,
                       "replicate_ranges": [[a, a + n] for a, n in shards], "solver_kernel": {1: "solver_kernel", 2: "solver_rows_kernel", 3: "solver_wave_kernel<8>", 4: "solver_rows_split_kernel", 5: "solver_quad_kernel<16>", 6: "solver_wave16_kernel<16>", 7: "solver_wave16_kernel<8>", 8: "solver_wave16_kernel<32>"}.get(timed_launch["last_solver"], "?"),
                       "gram_tile_plan_cus": plan_cus, "transport_calibration": transport_cal,
                       "comm_create_s": comm_create_s, "upload_s_per_device": upload_s, "single_call_latency_ms": single_call},
            "roofline": roofline,
            "spinup": "%d untimed steps (%.2f s) of the same launches before the %d warm-up steps: brings the device to its working clocks" % (spin_steps, spin_s, args.warmup),
            "cold": {"value": round(B_total * args.steps / cold_elapsed, 1), "unit": "replicates/s", "ms_per_step": round(cold_elapsed / args.steps * 1e3, 4),
                     "note": "the same %d timed steps behind %d warm-up steps only, measured before the spin-up (device coming out of idle)" % (args.steps, args.warmup)},
            "kernels_ms_per_step": {"resample": round(res_ms / max(res_n, 1), 4), "gram": round(gram_avg_ms, 4),
                                    "solver": round(sol_ms / max(sol_n, 1), 4)},
        }
        if planes is not None:
            line["digit_planes"] = planes
            # the rate that is no worse than fp64 BY CONSTRUCTION (seven planes: correctly rounded sums; what Plspm(..., precision="strict") runs and
            # what data below the six-plane bar -- heavy tails, a few hundred rows -- get anyway), as a top-level field beside `value`
            line["value_strict"] = planes["seven_planes"]["value"]
            line["value_strict_note"] = ("replicates/s of the same workload on seven digit planes (Plspm(precision='strict') / set_option i8_slices 7): moment sums correctly "
                                         "rounded by construction; `value` runs the automatic %d planes, whose error on this data is measured on this line (digit_planes)" % slices)
        if round3 is not None:
            line["round3_gram_kernel"] = round3
        if other is not None:
            line["fp64_mfma_path"] = other
        if pcie is not None:
            line["pcie_inclusive"] = {"value": round(pcie, 1), "unit": "replicates/s",
                                      "sub_batches": _native.chunk_plan(B_total, 8 * model.row_stride, model.get_option("boot_chunks"), model.get_option("boot_ratio"),
                                                                           model.get_option("boot_align") or model.get_option("boot_round_units")),
                                      "note": "plspm_bootstrap(): the B x 158 records copied to the caller's (pageable, re-used) host buffers through pinned staging every step; "
                                              "round 5: the call runs as sub-batches, the download + unpacking of sub-batch k beside the kernels of k + 1"}
        if world == 1 and group is None and not args.no_api:
            line["api_inclusive"] = api_inclusive(X, args.reps_per_gpu)
            line["api_inclusive"]["frac_of_value"] = round(line["api_inclusive"]["value"] / line["value"], 3)
            line["api_inclusive"]["frac_of_cold_value"] = round(line["api_inclusive"]["value"] / line["cold"]["value"], 3)      # (an API call starts on an idle device, as `cold` does)
        if world == 1 and group is None and not args.no_next_rows:
            # SURVEY 8(f) "next" rows beside the headline, measured by this run (child processes: their handles, their clocks): the categorical
            # (Scale.ORD, five-point) counterpart of the headline workload at 1,000 and 5,000 replicates per step -- tools/categorical_bench.py
            import subprocess
            cat = {}
            for reps in (1000, 5000):
                try:
                    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "categorical_bench.py"), str(reps)], capture_output=True, text=True, timeout=240).stdout
                    cat["replicates_per_step_%d" % reps] = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
                except Exception as e:                                    # noqa: BLE001 -- an extra of the line, never its failure
                    cat["replicates_per_step_%d" % reps] = {"error": repr(e)[:200]}
            try:                                                           # metric models next to the headline: the solvers of round 5 (tools/size_rows.py)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "size_rows.py")], capture_output=True, text=True, timeout=240).stdout
                sizes = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            except Exception as e:                                        # noqa: BLE001
                sizes = {"error": repr(e)[:200]}
            try:                                                           # the Scale.NUM counterpart of the headline workload: tools/nonmetric_bench.py
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nonmetric_bench.py")], capture_output=True, text=True, timeout=240).stdout
                num = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            except Exception as e:                                        # noqa: BLE001
                num = {"error": repr(e)[:200]}
            try:                                                           # both stages of a higher order construct per replicate, on the reference's mobi data (tools/hoc_bench.py)
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hoc_bench.py")], capture_output=True, text=True, timeout=240).stdout
                hoc = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
            except Exception as e:                                        # noqa: BLE001
                hoc = {"error": repr(e)[:200]}
            line["next_rows"] = {"nonmetric_num_bootstrap": num, "categorical_bootstrap": cat, "metric_models_next_to_the_headline": sizes, "hoc_two_stage_bootstrap": hoc,
                                 "note": "not part of `value`: ORD / NOM optimal scaling on 300 indicator columns (10k x 60 five-point items x 6 LVs), one wave per problem, "
                                         "count matrices written by the int8 product as uint16, stop rule as an int8 matrix product; DESIGN 5c"}
        if world == 1 and group is None and not args.no_single_fit:
            # SURVEY 8(d) "plus single-fit latency for C2 and C5": BASELINE.json configs[1] and configs[4] through plspm_upload / plspm_fit, in a child
            # process (tools/fit_bench.py: its own handle and 1.6 GB matrix; HIP-event kernel times, walls with and without upload / scores download,
            # A_fit / F_fit rooflines)
            import subprocess
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_bench.py"), "c2", "c5"], capture_output=True, text=True, timeout=600).stdout
                fits = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
                line["single_fit"] = {"configs[1]": fits[0], "configs[4]": fits[1],
                                      "note": "device_ms_total = sum of the fit's kernels (HIP events); fit_wall_ms_* = host wall of plspm_fit with X resident; "
                                              "upload_ms = plspm_upload of the fp64 matrix from pageable host memory (PCIe); never part of `value`"}
            except Exception as e:                                        # noqa: BLE001 -- an extra of the line, never its failure
                line["single_fit"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            # the real reference cannot travel to this box: its rate measured in the build container, and the factor between the oracle and
            # the reference measured there back to back (oracle/time_calibration.py -> profiles/r05_cpu_calibration.json), convert this
            # box's oracle figure into reference-equivalent replicates/s
            try:
                cal = json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_calibration.json")))
                ref8 = [r for r in cal["reference"]["runs"] if r["processes"] == 8][0]
                ref1 = [r for r in cal["reference"]["runs"] if r["processes"] == 1][0]
                ratio = cal["oracle_over_reference"]["all_cores"]
                line["cpu_baseline"]["reference"] = {
                    "replicates_per_s_8_processes": ref8["replicates_per_s"], "replicates_per_s_1_process": ref1["replicates_per_s"],
                    "host": "%s, %d cpus (build container)" % (cal["host"]["model"], cal["host"]["cpus"]),
                    "oracle_on_that_host": {"all_cores": cal["oracle"]["pool"]["value"], "single_process": cal["oracle"]["single_process_replicates_per_s"]},
                    "oracle_over_reference": ratio,
                    "reference_equivalent_on_this_box": round(line["cpu_baseline"]["value"] / ratio, 2),
                    "gpu_over_reference_equivalent": round(line["value"] / (line["cpu_baseline"]["value"] / ratio), 0),
                    "source": "profiles/r05_cpu_calibration.json (oracle/time_calibration.py: plspm 0.5.6 through its public API and the oracle "
                              "pool of this bench, back to back on the same cores)"}
            except Exception:
                pass
        print(json.dumps(line), flush=True)
    if group is not None:
        group.barrier()
        group.close()
    if launched:
        parallel.destroy_process_group()


if __name__ == "__main__":
    main()
