"""One seed of the rare-indicator fuzz on every metric solver form: weight / loading of the constant column per replicate against the oracle."""
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
X, model, idx = fc.make_rare_indicator_case(seed)
n, P = X.shape
col = [p for p in range(P) if set(np.unique(X[:, p])) <= {0.0, 1.0}][0]
lv = [l for l, b in enumerate(model.blocks) if col in b][0]
print("seed", seed, X.shape, model.modes, model.scheme, "scaled", model.scaled, "indicator column", col, "in block", [int(x) for x in model.blocks[lv]], "mode", model.modes[lv])
boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0)
nm.upload(X)
corr = orc.correction(n)
ne = nm.n_eff
want = []
for b in range(len(idx)):
    try:
        with np.errstate(all="ignore"):
            want.append(orc.bootstrap_replicate(X, model, idx[b], corr))
    except Exception as e:
        want.append((None, type(e).__name__))
for opts in ({}, {"solver_wave": 3}, {"solver_wave": 0}, {"solver_wave": 0, "solver_rows": 0}, {"gram_path": 1}):
    for k, v in opts.items():
        nm.set_option(k, v)
    rows, status, iters = nm.bootstrap(len(idx), idx=idx)
    line = []
    for b in range(len(idx)):
        w = want[b][0]
        d = "oracle:" + str(want[b][1]) if w is None else ("%.2e" % float(np.max(np.abs(rows[b] - w))))
        line.append("st%d it%d w=%.3g ld=%.3g maxdiff %s" % (status[b], iters[b], rows[b][col], rows[b][P + model.L + 2 * ne + col], d))
    print(opts, "solver", nm.get_option("last_solver"), "gram", nm.get_option("last_gram_path"))
    for b, l in enumerate(line):
        print("    replicate", b, l)
    for k in opts:
        nm.set_option(k, {"solver_wave": 1, "solver_rows": 1, "gram_path": 0}[k])
