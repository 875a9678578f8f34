"""Workgroup tile heights of the int8 Gram (set_option i8_rt 16 / 20 / 0 = automatic cut / 8): kernel times from the library's HIP
events, rows against the first variant (bit-identical among the six-plane ones; the seven-plane narrow tile rounds differently).
usage: rt_check.py [B]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
ref = None
for rt in (16, 20, 0, 8):                      # 0 = the automatic cut into rows of 320 / 256 replicates (six planes); 8 = the narrow seven-plane tile
    m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X); m.set_option("i8_rt", rt)
    if rt == 8: m.set_option("i8_slices", 7); m.set_option("i8_waves", 4)
    rows, st, it = m.bootstrap(B, seed=1)
    if ref is None: ref = rows
    for w in range(20): m.bootstrap_device(B, seed=1, rep_offset=w * B)
    m.sync(); m.profile(True); m.profile_reset()
    t = time.perf_counter()
    for w in range(20): m.bootstrap_device(B, seed=1, rep_offset=(20 + w) * B)
    m.sync(); wall = (time.perf_counter() - t) / 20 * 1e3
    print(json.dumps({"i8_rt": rt, "B": B, "last_i8_rt": m.get_option("last_i8_rt"), "short_rows": m.get_option("last_i8_short"), "planes": m.get_option("last_i8_slices"), "rows_equal_default": bool(np.array_equal(rows, ref)), "status_ok": bool(np.all(st == 0)), "ms_per_step_profiled": round(wall, 4),
                      "gram_ms": round(m.profile_read("gram")[0] / max(1, m.profile_read("gram")[1]), 4), "solver_ms": round(m.profile_read("solver")[0] / max(1, m.profile_read("solver")[1]), 4),
                      "resample_ms": round(m.profile_read("resample")[0] / max(1, m.profile_read("resample")[1]), 4)}))
