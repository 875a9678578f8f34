"""One seed of the categorical fuzz in detail: the device-resampled replicates whose rows differ between the wave step and the workgroup step, each against the oracle."""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
np.set_printoptions(linewidth=220, precision=8)
import plspm_oracle as orc
import test_gpu_fuzz as f
import test_gpu_categorical as tc
from plspm import _native
seed = int(sys.argv[1])
import fuzz_cases as fc
data, model = {"small": fc.make_cat_small_case, "big": fc.make_cat_big_case}.get(sys.argv[2] if len(sys.argv) > 2 else "", fc.make_cat_case)(seed)
n = data.shape[0]
nm, g = tc.gpu_fit_cat(data, model)
a = nm.bootstrap(40, seed=seed)
nm.set_option("nm_wave", 0)
w = nm.bootstrap(40, seed=seed)
nm.set_option("nm_wave", 1)
Pm = len(model.scales)
ra = tc._rows_in_data_order(a[0], g["inv"], Pm, model.L, nm.n_eff)
rw = tc._rows_in_data_order(w[0], g["inv"], Pm, model.L, nm.n_eff)
for r in range(40):
    if a[1][r] != w[1][r] or a[2][r] != w[2][r]:
        idx = _native.bootstrap_indices(seed, r, n)
        try:
            with np.errstate(all="ignore"):
                mine, its = orc.bootstrap_replicate(data, model, idx, orc.correction(n)); o = "iterations %d" % its
        except Exception as e:
            o = "raises " + type(e).__name__
        Xr = data[idx]
        print("replicate", r, "status wave/group", a[1][r], w[1][r], "iterations", a[2][r], w[2][r], "oracle", o, "categories present per MV:", [len(np.unique(Xr[:, p])) for p in range(Pm)], "of", [len(np.unique(data[:, p])) for p in range(Pm)])
        continue
    if a[1][r] != 0 or w[1][r] != 0:
        continue
    d = np.max(np.abs(ra[r] - rw[r]) / np.maximum(np.abs(rw[r]), 1e-3))
    if d > 1e-9:
        idx = _native.bootstrap_indices(seed, r, n)
        with np.errstate(all="ignore"):
            mine, its = orc.bootstrap_replicate(data, model, idx, orc.correction(n))
        print("replicate", r, "wave vs group", d, "iterations wave/group/oracle", a[2][r], w[2][r], its)
        print("  wave  vs oracle", np.max(np.abs(ra[r] - mine) / np.maximum(np.abs(mine), 1e-3)))
        print("  group vs oracle", np.max(np.abs(rw[r] - mine) / np.maximum(np.abs(mine), 1e-3)))
        Xr = data[idx]
        print("  categories present per MV:", [len(np.unique(Xr[:, p])) for p in range(Pm)], "of", [len(np.unique(data[:, p])) for p in range(Pm)])
        nm.set_option("nm_cat_one", 0)
        b = nm.bootstrap(40, seed=seed)
        rb = tc._rows_in_data_order(b[0], g["inv"], Pm, model.L, nm.n_eff)
        print("  wave launch-by-launch vs oracle", np.max(np.abs(rb[r] - mine) / np.maximum(np.abs(mine), 1e-3)), "iterations", b[2][r])
        nm.set_option("nm_subset", 0)
        b = nm.bootstrap(40, seed=seed)
        rb = tc._rows_in_data_order(b[0], g["inv"], Pm, model.L, nm.n_eff)
        print("  ... with every pass over all rows vs oracle", np.max(np.abs(rb[r] - mine) / np.maximum(np.abs(mine), 1e-3)), "iterations", b[2][r])
        nm.set_option("nm_subset", 4); nm.set_option("nm_cat_one", 1)
        for opts in (dict(nm_c10=0), dict(nm_c10=0, nm_cpl=8), dict(nm_c10=1, nm_cpl=0)):
            for k, v in opts.items():
                nm.set_option(k, v)
            b = nm.bootstrap(40, seed=seed)
            rb = tc._rows_in_data_order(b[0], g["inv"], Pm, model.L, nm.n_eff)
            print("  ", opts, "vs oracle", np.max(np.abs(rb[r] - mine) / np.maximum(np.abs(mine), 1e-3)), "iterations", b[2][r], "last_nm_wave", nm.get_option("last_nm_wave"))
        bad = np.abs(ra[r] - mine) / np.maximum(np.abs(mine), 1e-3)
        print("   worst entries (index, wave, oracle):", [(int(i), float(ra[r][i]), float(mine[i])) for i in np.argsort(-bad)[:6]])
        print("   row layout: weights[0:%d] r2 total direct loadings" % Pm)
