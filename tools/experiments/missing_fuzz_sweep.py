"""Differential fuzz of the metric missing-data path (mean imputation on the moments, re-imputed per replicate) against the oracle: random path models, ragged blocks,
modes, schemes, scaled or not, NaN cells scattered over a random subset of the columns (up to 15 % of a column); fit + bootstrap replicates on explicit index lists.
Also Scale.NUM / RAW models with incomplete rows (the NaN-aware Mode-A products, solver_nmx.h).  Seeds A .. B from the command line.  (A test-side tool: generators and checkers live in
tests/test_gpu_fuzz.py, whose parametrised tests run the first seeds.)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


import test_gpu_fuzz as f


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        for fn in (f._missing_case_check, f._nmx_case_check):
            try:
                hist[fn(seed)] += 1
            except Exception:
                tb = traceback.format_exc().splitlines()
                bad.append((fn.__name__, seed, tb[-1][:400]))
    print("outcomes", dict(hist))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
