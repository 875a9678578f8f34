"""Stop-rule criterion of ONE explicit-index replicate of a small-sample categorical fuzz case trip by trip: device (handles with max_iter = k, plspm_nonmetric_criteria) (compare with the oracle's loop run beside it).  Usage: cat_criteria_trace.py SEED REPLICATE(0..5 of RandomState(seed)) [KMAX]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import plspm_oracle as orc
import fuzz_cases as fc
import test_gpu_categorical as tc
from test_solver_hostemu_ordnom import build_aug
from plspm import _native
seed, r = int(sys.argv[1]), int(sys.argv[2])
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 12
data, model = fc.make_cat_small_case(seed)
n = data.shape[0]
idx = np.random.RandomState(seed).randint(n, size=(6, n)).astype(np.int32)[r:r + 1]
Xaug, mv_off, mv_kind, lmv_off, boff, mv_data_col = build_aug(data, model)
modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
for k in range(1, kmax + 1):
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, tc.SCHEME_ID[model.scheme], True, k, 1e-300, 0, nonmetric=True, categorical=(mv_off, mv_kind))
    nm.upload(Xaug)
    rows, status, iters = nm.bootstrap(1, idx=idx)
    crit = nm.nonmetric_criteria(1)[0] if hasattr(nm, "nonmetric_criteria") else float("nan")
    print("max_iter", k, "device status", int(status[0]), "iterations", int(iters[0]), "criterion of the last trip %.6e" % crit, "weights[:4]", rows[0][:4])
    nm.close()
