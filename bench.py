#!/usr/bin/env python3
"""Headline benchmark: bootstrap replicates / second, 6-LV satisfaction model, N = 10,000 obs x 60 MVs
(BASELINE.json configs[2]: Mode A, Scheme.PATH, scaled, 5,000 replicates per GPU -- weak scaling).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: every rank resamples + solves its shard of the replicate
range on its own GPU (inputs already resident in HBM), then ONE all_gather (RCCL over xGMI) merges the
B x 156 result rows -- the whole job, gather included, is inside the timed region.  Rank 0 prints one JSON line.

Extra objects on the line:
  roofline      dominant kernel (fp64-MFMA weighted Gram), duration measured with HIP events on the library's own
                stream during the timed steps; algorithmic work per replicate per SURVEY.md 8(d)
  cpu_baseline  the NumPy oracle (oracle/plspm_oracle.py, a port of the reference arithmetic) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N = 1 only)
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


def synth_inputs():
    """The workload: synthetic generator of SURVEY.md 8(d) (tools/synthetic.py -- data only, no PLS arithmetic)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthetic
    X, blocks = synthetic.synth(N_OBS, synthetic.satisfaction_C(), MVS_PER_LV, seed=0)
    return synthetic, X, blocks


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps-per-gpu", type=int, default=REPS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))

    from plspm import _native, parallel
    dist = None
    use_dist = "RANK" in os.environ            # launched by torch.distributed.run (also exercises RCCL at N = 1)
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device_id = local_rank if use_dist else 0

    synthetic, X, blocks = synth_inputs()
    C = synthetic.satisfaction_C()
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)

    def make_model():
        mdl = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(N_LV, dtype=np.int32), 2, True, 100, 1e-6, device_id)
        mdl.upload(X)                                          # X resident in HBM before the timed region
        return mdl

    model = make_model()
    B_total = args.reps_per_gpu * world
    width = model.row_width
    # N > 1: two handles (each with its own result buffer) alternate, so the all_gather of step k runs on RCCL's stream
    # under the kernels of step k+1; every step still does all of its work (shard compute + one collective) and the
    # closing fence waits for the last collective.
    models = [model, make_model()] if use_dist else [model]
    pending = [None, None]
    state = {"k": 0, "last": None}

    ext_streams = [torch.cuda.ExternalStream(mdl.stream_ptr()) for mdl in models] if use_dist else []

    def step():
        if not use_dist:
            model.bootstrap_device(B_total, seed=1, rep_offset=0)          # enqueue only: the closing fence synchronises
            return None
        k = state["k"] % 2
        state["k"] += 1
        if pending[k] is not None:
            with torch.cuda.stream(ext_streams[k]):
                pending[k][1].wait()                           # stream-side wait: this handle's buffer is free again before its next kernels
        mdl = models[k]
        start, stop = parallel.shard_range(B_total, rank, world)
        d_rows, _, _ = mdl.bootstrap_device(stop - start, seed=1, rep_offset=start)
        send = parallel.device_rows(d_rows, stop - start, width + 2)
        with torch.cuda.stream(ext_streams[k]):                # RCCL orders itself behind the handle's stream: no host sync
            pending[k] = parallel.gather_records(send, B_total, async_op=True, slot=k)
        state["last"] = pending[k]
        return pending[k]

    def drain():
        for pk in pending:
            if pk is not None:
                pk[1].wait()

    def fence():
        for mdl in models:
            mdl.sync()
        if use_dist:
            drain()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for mdl in models:
        mdl.profile(True)
        mdl.profile_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    for mdl in models:
        mdl.profile(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if use_dist:
        # the gathered records of the last timed step: complete, in replicate-id order, identical on every rank
        rec, work = step()
        work.wait()
        torch.cuda.synchronize()
        rec = rec.cpu().numpy()
        assert rec.shape == (B_total, width + 2) and np.all(rec[:, width] == 0), "gather lost replicates"
        probe = min(B_total - 1, (B_total // world) * (world - 1) + 3)          # a row owned by the last rank
        one, _, _ = model.bootstrap(1, seed=1, rep_offset=probe)
        assert np.array_equal(rec[probe, :width], one[0]), "sharded stream differs from the single-GPU stream"

    # correctness guard on what was timed: every replicate converged, and replicate 0 of the seeded stream equals the committed
    # reference-arithmetic row (tests/golden/bench_guard.npz, made by tests/golden/make_bench_guard.py with the oracle)
    rows, status, iters = model.bootstrap(8, seed=1, rep_offset=0)
    assert np.all(status == 0), status
    guard = np.load(os.path.join(ROOT, "tests", "golden", "bench_guard.npz"))
    idx0 = _native.bootstrap_indices(1, 0, N_OBS)
    assert int(idx0.astype(np.int64).sum()) == int(guard["idx_sum"]) and np.array_equal(idx0[:16], guard["idx_head"]), "resampling stream changed"
    assert int(guard["iterations"]) == iters[0] and np.allclose(rows[0], guard["row"], rtol=1e-8, atol=1e-11), "timed path disagrees with the oracle"

    pcie = None
    if world == 1 and not use_dist:
        # the same batch through the host-buffer entry point (results copied back over PCIe every step); never `value`
        model.bootstrap(B_total, seed=1)
        t1 = time.perf_counter()
        for _ in range(5):
            model.bootstrap(B_total, seed=1)
        pcie = B_total * 5 / (time.perf_counter() - t1)

    if rank == 0:
        def prof(name):
            parts = [mdl.profile_read(name) for mdl in models]
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        gram_ms, gram_n = prof("gram")
        res_ms, res_n = prof("resample")
        sol_ms, sol_n = prof("solver")
        reps_per_launch = args.reps_per_gpu
        a_rep = 8.0 * N_OBS * 60 + 4.0 * N_OBS                 # one gathered read of X + the index vector
        f_rep = float(N_OBS) * 60 * 61                         # symmetric Gram flops
        gram_avg_ms = gram_ms / max(gram_n, 1)
        hbm_achieved = a_rep * reps_per_launch / (gram_avg_ms * 1e-3) / 1e9
        mfma_achieved = f_rep * reps_per_launch / (gram_avg_ms * 1e-3) / 1e12
        traffic = None
        prof_json = os.path.join(ROOT, "profiles", "r01_gram_traffic.json")
        if os.path.exists(prof_json):
            try:
                traffic = json.load(open(prof_json)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "bootstrap replicates/sec (6-LV satisfaction model, N=10k)",
            "value": round(B_total * args.steps / elapsed, 1),
            "unit": "replicates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic 10,000 obs x 60 MVs x 6 LVs, Mode A, Scheme.PATH, scaled, %d bootstrap replicates per GPU "
                                   "(BASELINE.json configs[2]); on-device Philox resampling; X resident in HBM" % args.reps_per_gpu,
                       "replicates_per_step": B_total, "iterations_per_replicate": [int(iters.min()), int(iters.max())],
                       "parallelism": "replicate-sharded x%d, one all_gather per step%s" % (world, " (overlapped with the next step's kernels)" if use_dist else "")},
            "roofline": {"bound": "mfma", "achieved": round(mfma_achieved, 2), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(mfma_achieved / FP64_MFMA_PEAK_TF, 4), "traffic": traffic,
                         "kernel": "gram_rows_kernel<4,false>", "avg_launch_ms": round(gram_avg_ms, 4), "launches": gram_n,
                         "note": ("two alternating handles: HIP-event durations include kernels co-scheduled from the other stream"
                                  if use_dist else "single stream: HIP-event duration of the kernel alone"),
                         "algorithmic_flops_per_replicate": f_rep, "algorithmic_bytes_per_replicate": a_rep,
                         "hbm_equivalent": {"achieved": round(hbm_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(hbm_achieved / HBM_PEAK_GBS, 4)}},
            "kernels_ms_per_step": {"resample": round(res_ms / max(res_n, 1), 4), "gram": round(gram_avg_ms, 4),
                                    "solver": round(sol_ms / max(sol_n, 1), 4)},
        }
        if pcie is not None:
            line["pcie_inclusive"] = {"value": round(pcie, 1), "unit": "replicates/s",
                                      "note": "plspm_bootstrap(): device -> pageable host copy of the B x 156 rows every step"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
