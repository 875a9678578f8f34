"""Trajectory of ONE replicate of a categorical fuzz case on both step forms: handles with max_iter = 1, 2, ... (a replicate that has not converged by then reports the
state it stopped in), wave step against workgroup step, trip by trip.  Usage: cat_fuzz_trace.py SEED REPLICATE"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
np.set_printoptions(linewidth=220, precision=6)
import plspm_oracle as orc
import test_gpu_fuzz as f
import test_gpu_categorical as tc
from test_solver_hostemu_ordnom import build_aug
from plspm import _native
seed, r = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3 and sys.argv[3] == "big"
wave_opts = dict(kv.split("=") for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []))      # options of the wave-step handle, e.g. nm_subset=0,nm_cpl=8
import fuzz_cases as fc
small = len(sys.argv) > 3 and sys.argv[3] == "small"
data, model = (fc.make_cat_small_case if small else fc.make_cat_big_case if big else f.make_cat_case)(seed)
n = data.shape[0]
idx = _native.bootstrap_indices(seed, r, n)[None, :].astype(np.int32)
Xaug, mv_off, mv_kind, lmv_off, boff, mv_data_col = build_aug(data, model)
modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
Pm = len(mv_kind)
print("mv_off", list(mv_off), "kinds", list(mv_kind), "data col of MV", list(mv_data_col))
Xr = data[idx[0]]
for p in range(Pm):
    col = mv_data_col[p]
    vals = np.unique(data[:, col]); present = np.isin(vals, np.unique(Xr[:, col]))
    if not present.all():
        print("MV", p, "(data column", col, ") categories", len(vals), "absent:", [int(i) for i in np.flatnonzero(~present)], "counts present", [int((Xr[:, col] == v).sum()) for v in vals])
for k in range(1, int(os.environ.get('TRACE_TRIPS', '21')) + 1):
    rows = {}
    for wave in (1, 0):
        nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, tc.SCHEME_ID[model.scheme], True, k, model.tol, 0, nonmetric=True, categorical=(mv_off, mv_kind))
        nm.upload(Xaug)
        nm.set_option("nm_wave", wave)
        nm.set_option("nm_cat_one", 0)
        if wave:
            for k_, v_ in wave_opts.items():
                nm.set_option(k_, int(v_))
        out = nm.bootstrap(1, idx=idx)
        rows[wave] = (out[0][0].copy(), int(out[1][0]), int(out[2][0]))
        nm.close()
    a, b = rows[1][0], rows[0][0]
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    print("max_iter", k, "status/iters wave", rows[1][1:], "group", rows[0][1:], "max rel diff", float(np.nanmax(d)) if np.isfinite(d).any() else "all-nan", "worst weight index", int(np.nanargmax(d[:Pm])) if np.isfinite(d[:Pm]).any() else -1, "nan", int(np.isnan(a).sum()), int(np.isnan(b).sum()))
