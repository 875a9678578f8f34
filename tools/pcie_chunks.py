#!/usr/bin/env python3
"""plspm_bootstrap() (host buffers, PCIe-inclusive) on the headline workload under the sub-batch options: replicates/s for boot_chunks x boot_ratio.
   python tools/pcie_chunks.py [B] > gpurun_out/pcie_chunks.jsonl"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
m.upload(X)
host = (np.empty((B, m.row_width)), np.empty(B, dtype=np.int32), np.empty(B, dtype=np.int32))
for k in range(100):
    m.bootstrap_device(B, seed=1, rep_offset=k * B)
m.sync()
t0 = time.perf_counter()
for k in range(20):
    m.bootstrap_device(B, seed=1, rep_offset=k * B)
m.sync()
dev_ms = (time.perf_counter() - t0) / 20 * 1e3
ref = None
for rnd in range(2):
    for chunks, ratio in ((1, 60), (2, 60), (2, 80), (2, 100), (3, 50), (3, 60), (3, 75), (3, 100), (4, 60), (4, 80), (5, 70), (0, 60)):
        m.set_option("boot_chunks", chunks); m.set_option("boot_ratio", ratio)
        for _ in range(3):
            got = m.bootstrap(B, seed=1, out=host)
        if ref is None:
            ref = got[0].copy()
        same = bool(np.array_equal(got[0], ref))
        t0 = time.perf_counter()
        for _ in range(15):
            m.bootstrap(B, seed=1, out=host)
        ms = (time.perf_counter() - t0) / 15 * 1e3
        print(json.dumps({"B": B, "boot_chunks": chunks, "boot_ratio": ratio, "parts": _native.chunk_plan(B, 8 * m.row_stride, chunks, ratio, m.get_option("boot_round_units")), "ms_per_call": round(ms, 4),
                          "replicates_per_s": round(B / ms * 1e3, 1), "device_only_ms": round(dev_ms, 4), "rows_identical": same, "round": rnd}), flush=True)
