"""Differential fuzz of the categorical (Scale.ORD / NOM / NUM mixes) path against the oracle: random path models, block sizes, category counts (2 .. 12), modes
and schemes; the fit (iterations, weights, loadings, path coefficients, scores) and bootstrap replicates on explicit index lists -- through the wave step where it covers
the model and the workgroup step elsewhere.  Seeds A .. B from the command line; prints the route histogram and the failures.  (A test-side tool: generator and checker live in
tests/test_gpu_fuzz.py.)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import test_gpu_fuzz as f

if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    big = (sys.argv[3] == "small" and "small") or (sys.argv[3] == "big") if len(sys.argv) > 3 else False      # big: make_cat_big_case (up to 80 MVs, 16 categories); small: make_cat_small_case (30 ... 90 rows)
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        try:
            hist[f._cat_case_check(seed, big)] += 1
        except Exception:
            tb = traceback.format_exc().splitlines()
            bad.append((seed, tb[-1][:400]))
    print("routes", dict(hist))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
