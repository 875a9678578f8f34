"""Metric models next to the headline, for the `next_rows` of the bench line (one JSON object on stdout): 10k x 120 x 12 (the quad solver), 10k x 64 x 16 and
10k x 60 x 12 (the wave solver for 9 ... 16 LVs), the headline model with all blocks Mode B -- 5,000 replicates per step, PATH, replicates/s and kernel times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
from size_bench_models import chain_C

B = 5000
NAMES = {1: "solver_kernel", 2: "solver_rows_kernel", 3: "solver_wave_kernel<8>", 4: "solver_rows_split_kernel", 5: "solver_quad_kernel<16>", 6: "solver_wave16_kernel<16>", 7: "solver_wave16_kernel<8>", 8: "solver_wave16_kernel<32>"}
out = {}
for name, C, per, modes in (("10k x 120 x 12", chain_C(12), 10, None), ("10k x 64 x 16", chain_C(16), 4, None), ("10k x 60 x 12", chain_C(12), 5, None),
                            ("10k x 60 x 6 all Mode B", orc.satisfaction_C(), 10, "B")):
    L = C.shape[0]
    X, blocks = orc.synth(10000, C, per, seed=0)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    m = _native.NativeModel(boff, C.astype(np.uint8), np.full(L, 1 if modes == "B" else 0, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X)
    rows, st, it = m.bootstrap(64, seed=1)
    for w in range(30): m.bootstrap_device(B, seed=1, rep_offset=w * B)
    m.sync()
    t0 = time.perf_counter()
    for w in range(10): m.bootstrap_device(B, seed=1, rep_offset=(30 + w) * B)
    m.sync()
    wall = (time.perf_counter() - t0) / 10
    m.profile(True); m.profile_reset()
    for w in range(5): m.bootstrap_device(B, seed=1, rep_offset=(40 + w) * B)
    m.sync(); m.profile(False)
    k = {n: round(m.profile_read(n)[0] / max(1, m.profile_read(n)[1]), 4) for n in ("resample", "gram", "solver")}
    out[name] = {"replicates_per_s": round(B / wall, 1), "ms_per_step": round(wall * 1e3, 4), "kernels_ms": k, "solver_kernel": NAMES.get(m.get_option("last_solver"), "?"),
                 "status_ok": bool(np.all(st == 0))}
    m.close()
print(json.dumps(out))
