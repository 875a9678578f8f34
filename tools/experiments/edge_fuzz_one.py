"""One seed of the edge-case fuzz in detail: statuses of the fit and of a few device-resampled replicates on every route (int8 / fp64 Gram, one-launch / per-iteration NUM solver)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
np.set_printoptions(linewidth=200, precision=6)
import plspm_oracle as orc
import fuzz_cases as fc
from test_gpu_parity import SCHEME_ID
from plspm import _native
seed = int(sys.argv[1])
X, model, nonmetric, kind = fc.make_degenerate_case(seed)
n = X.shape[0]
print("kind", kind, X.shape, model.modes, model.scheme, "NUM" if nonmetric else "metric", "blocks", [list(map(int, b)) for b in model.blocks])
print("column correlations with the next column:", [round(float(np.corrcoef(X[:, p], X[:, p + 1])[0, 1]), 6) for p in range(X.shape[1] - 1)])
boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0, nonmetric=nonmetric)
nm.upload(X)
g = nm.fit(want_scores=True)
print("fit status", g["status"], "iterations", g["iterations"])
for opts in ({}, {"gram_path": 1}, {"nm_wave16": 0}, {"gram_path": 1, "nm_wave16": 0}):
    for k, v in opts.items():
        nm.set_option(k, v)
    rows, status, iters = nm.bootstrap(6, seed=seed)
    print(opts, "status", status.tolist(), "iters", iters.tolist(), "gram path", nm.get_option("last_gram_path"), "solver", nm.get_option("last_solver"), "wave16", nm.get_option("last_nm_wave16") if nonmetric else "-")
    for k in opts:
        nm.set_option(k, {"gram_path": 0, "nm_wave16": 1}[k])
corr = orc.correction(n)
for b in range(3):
    idx = _native.bootstrap_indices(seed, b, n)
    try:
        with np.errstate(all="ignore"):
            mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        print("oracle replicate", b, "iterations", its, "finite", bool(np.all(np.isfinite(mine))))
    except Exception as e:
        print("oracle replicate", b, "raised", type(e).__name__, str(e)[:80])
