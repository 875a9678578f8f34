"""One device-resampled replicate of a small-sample categorical fuzz case under the wave-step options: status / iterations.  Usage: cat_small_probe.py SEED REPLICATE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import fuzz_cases as fc
import test_gpu_categorical as tc
seed, r = int(sys.argv[1]), int(sys.argv[2])
data, model = fc.make_cat_small_case(seed)
print("n", data.shape[0], "P", data.shape[1], model.modes, model.scheme)
nm, g = tc.gpu_fit_cat(data, model)
print("fit status", g["status"], "iterations", g["iterations"])
for opts in ({}, {"nm_cat_one": 0}, {"nm_cat_one": 0, "nm_subset": 0}, {"nm_subset": 1}, {"nm_bound_shift": 0, "nm_cpl": 8}, {"nm_mfma": 0}, {"nm_wave": 0}):
    for k, v in opts.items():
        nm.set_option(k, v)
    rows, status, iters = nm.bootstrap(40, seed=seed)
    print(opts, "-> status", int(status[r]), "iterations", int(iters[r]), "| all statuses", np.bincount(status, minlength=4).tolist(), "flagged", nm.get_option("last_nm_flagged"), "replayed", nm.get_option("last_nm_replayed"), "one", nm.get_option("last_nm_one"), "wave", nm.get_option("last_nm_wave"))
    for k in opts:
        nm.set_option(k, {"nm_cat_one": 1, "nm_subset": 4, "nm_bound_shift": 0, "nm_cpl": 0, "nm_mfma": 1, "nm_wave": 1}[k])
