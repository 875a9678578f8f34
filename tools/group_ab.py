#!/usr/bin/env python3
"""What the group path costs a ONE-rank step (RCCL at one rank; VERDICT r4 item 1: +3-4 % over the plain step): the same steps plain, through the
group as it is, with the exchange skipped (events and kernels only), and with device-scope events.  Alternating rounds on one box.
   python tools/group_ab.py > gpurun_out/group_ab.jsonl"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic  # noqa: E402
from plspm import _native  # noqa: E402

B = 5000
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
m.upload(X)
comm = _native.NativeComm([0])
group = _native.NativeGroup(comm, [m])
group.set_option("chunks", 1)
state = {"k": 0}


def run(kind, steps):
    for _ in range(steps):
        off = state["k"] * B
        state["k"] += 1
        if kind == "plain":
            m.bootstrap_device(B, seed=1, rep_offset=off)
        else:
            group.bootstrap(B, seed=1, rep_offset=off)
    if kind == "plain":
        m.sync()
    else:
        group.sync()


run("plain", 200)
for rnd in range(4):
    for kind, opts in (("plain", {}), ("group", {"skip_exchange": 0, "lean_events": 1}), ("group-round4-events", {"skip_exchange": 0, "lean_events": 0}),
                       ("group-no-exchange", {"skip_exchange": 1, "lean_events": 1}), ("group-no-exchange-round4-events", {"skip_exchange": 1, "lean_events": 0})):
        if kind != "plain":
            for k, v in opts.items():
                group.set_option(k, v)
        run("plain" if kind == "plain" else "group", 10)
        t0 = time.perf_counter()
        run("plain" if kind == "plain" else "group", 40)
        ms = (time.perf_counter() - t0) / 40 * 1e3
        print(json.dumps({"kind": kind, "round": rnd, "ms_per_step": round(ms, 4), "transport": comm.transport}), flush=True)
group.set_option("skip_exchange", 0); group.set_option("lean_events", 1)
