#!/usr/bin/env python3
"""The persistent Gram (gram_i8pp_kernel, "i8_persist" 1) against the tiled launch (0) on the headline workload: step time and the Gram's own
launch time (HIP events), alternating rounds on one box, records compared bit for bit.   python tools/persist_ab.py [B] [slices]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 0
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
models = {}
for v in (0, 1):
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X); nm.set_option("i8_persist", v); nm.set_option("i8_slices", S)
    for w in range(100): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
    nm.sync(); models[v] = nm
step, gram = {0: [], 1: []}, {0: [], 1: []}
for rnd in range(6):
    for v in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        nm = models[v]
        for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=k * B)
        nm.sync()
        t = time.perf_counter()
        for k in range(40): nm.bootstrap_device(B, seed=1, rep_offset=(3 + rnd * 40 + k) * B)
        nm.sync()
        step[v].append((time.perf_counter() - t) / 40 * 1e3)
        nm.profile(True, only="gram"); nm.profile_reset()
        for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(900 + k) * B)
        nm.sync(); nm.profile(False)
        ms, n = nm.profile_read("gram"); gram[v].append(ms / max(n, 1))
# experiments build: the same launches with every epilogue store skipped (results are garbage, only the Gram's time is read) -- what the epilogue costs each form
nostore = {}
if models[0].get_option("build_experiments") == 1:
    for v in (0, 1):
        nm = models[v]; nm.set_option("i8_nostore", 1)
        ts = []
        for rnd in range(3):
            nm.profile(True, only="gram"); nm.profile_reset()
            for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(900 + k) * B)
            nm.sync(); nm.profile(False)
            ms, n = nm.profile_read("gram"); ts.append(ms / max(n, 1))
        nostore[v] = round(min(ts), 4)
        nm.set_option("i8_nostore", 0)
ref = models[0].bootstrap(256, seed=9)[0]
for v in (0, 1):
    rows = models[v].bootstrap(256, seed=9)[0]
    print(json.dumps({"i8_persist": v, "B": B, "planes": models[v].get_option("last_i8_slices"), "last_i8_persist": models[v].get_option("last_i8_persist"),
                      "step_ms_min": round(min(step[v]), 4), "step_ms_median": round(float(np.median(step[v])), 4),
                      "gram_ms_min": round(min(gram[v]), 4), "gram_ms_median": round(float(np.median(gram[v])), 4), "gram_ms_without_epilogue_stores": nostore.get(v), "records_bit_identical": bool(np.array_equal(rows, ref))}), flush=True)
