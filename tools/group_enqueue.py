"""Host side of a multi-GPU bootstrap step, measured on ONE device: wall time until plspm_group_bootstrap RETURNS (shard
kernels of every local handle + event waits + the exchange enqueued) for 1 / 2 / 4 / 8 local handles of one process, 5,000 replicates per
handle (the weak-scaling step of bench.py --gpus N driven by one process).  Handles share device 0 here, so the exchange is the
device-copy route (one copy per peer instead of one ncclAllGather call per handle -- at least as many host calls as the RCCL route).
The step's device time on a real node is ~0.48 ms per GPU: the enqueue has to stay well below that for one host thread to feed 8 GPUs.
usage: group_enqueue.py [reps_per_handle]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
per = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
for G in (1, 2, 4, 8):
    models = []
    for g in range(G):
        nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
        nm.upload(X); models.append(nm)
    comm = _native.NativeComm([0] * G)
    group = _native.NativeGroup(comm, models)
    B = per * G
    for w in range(5): group.bootstrap(B, seed=1, rep_offset=w * B)
    group.sync()
    enq, steps, sh, ex = [], [], [], []
    for rnd in range(5):
        t0 = time.perf_counter()
        ts = []
        for k in range(10):
            a = time.perf_counter()
            group.bootstrap(B, seed=1, rep_offset=(5 + rnd * 10 + k) * B)
            ts.append(time.perf_counter() - a)
            a2, b2 = group.enqueue_times(); sh.append(a2); ex.append(b2)
        group.sync()
        steps.append((time.perf_counter() - t0) / 10)
        enq.append(float(np.median(ts)))
    print(json.dumps({"local_handles": G, "replicates_per_handle": per, "transport": "rccl" if comm.uses_rccl else "device-copies",
                      "enqueue_ms_median": round(float(np.median(enq)) * 1e3, 4), "enqueue_ms_min": round(min(enq) * 1e3, 4),
                      "shards_enqueue_ms": round(float(np.median(sh)), 4), "exchange_enqueue_ms": round(float(np.median(ex)), 4),
                      "step_ms_on_one_device": round(float(np.median(steps)) * 1e3, 4),
                      "note": "enqueue = wall of one plspm_group_bootstrap call (returns before any kernel has run); step = the same G shards sharing ONE GPU"}), flush=True)
    group.close(); comm.close()
    for nm in models: nm.close()
