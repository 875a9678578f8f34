#!/usr/bin/env python3
"""How tight is the wave step's own upper bound of the categorical criterion (kernels_nmw.h, round 6)?  The value every replicate stopped on with the bound
(nm_subset 1) against the exact criterion of the same step (nm_subset 0), and how many (problem, step) pairs needed the pass over all rows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_gpu_categorical import gpu_fit_cat
import plspm_oracle as o2
C = orc.satisfaction_C()
X, blocks = orc.synth(10000, C, 10, seed=0)
Z = (X - X.mean(axis=0)) / X.std(axis=0)
likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
model = o2.Model(blocks, C, "AAAAAA", "path", True, tol=1e-6, scales=["ORD"] * 60)
nm, g = gpu_fit_cat(likert, model)
B = 1000
for pct in (1, 2, 4, 8, 16):
    nm.set_option("nm_subset", pct)
    sub = nm.bootstrap(B, seed=1)
    print("nm_subset", pct, "exact pairs", nm.get_option("last_nm_exact"), "iterations", np.bincount(sub[2]))
    ub = nm.nonmetric_criteria(B)
nm.set_option("nm_subset", 0)
ex = nm.bootstrap(B, seed=1)
cv = nm.nonmetric_criteria(B)
print("equal iterations", np.array_equal(sub[2], ex[2]))
r = ub / cv
print("ub/conv quantiles", np.quantile(r, [0, 0.01, 0.5, 0.99, 1]), "conv quantiles", np.quantile(cv, [0, 0.5, 1]))
