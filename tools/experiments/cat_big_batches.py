"""Sharding invariance of large categorical batches: B replicates of a large fuzz model in ONE call against three calls with rep_offset (the chunking of a call by its scratch --
count matrices, map store -- must not show in the records).  Usage: cat_big_batches.py SEED B"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import fuzz_cases as fc
import test_gpu_categorical as tc
seed, B = int(sys.argv[1]), int(sys.argv[2])
data, model = fc.make_cat_big_case(seed)
nm, g = tc.gpu_fit_cat(data, model)
t0 = time.time(); one = nm.bootstrap(B, seed=seed); t1 = time.time()
third = B // 3
parts = [nm.bootstrap(third if k < 2 else B - 2 * third, seed=seed, rep_offset=k * third) for k in range(3)]
rows = np.concatenate([p[0] for p in parts]); status = np.concatenate([p[1] for p in parts]); iters = np.concatenate([p[2] for p in parts])
print("seed", seed, "L", model.L, "P", data.shape[1], "wave", nm.get_option("last_nm_wave"), "one-launch", nm.get_option("last_nm_one"), "B", B, "%.2f s" % (t1 - t0),
      "ok", int((one[1] == 0).sum()), "identical", bool(np.array_equal(one[0], rows, equal_nan=True) and np.array_equal(one[1], status) and np.array_equal(one[2], iters)))
