"""A longer run of the edge-case fuzz of tests/test_gpu_fuzz.py (fuzz_cases.make_degenerate_case: tiny samples, duplicated / constant columns, iteration caps, tolerance extremes)
and of the base fuzz (make_case).  Seeds A .. B."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import fuzz_cases as fc
import test_gpu_fuzz as f

if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        X, model, nonmetric, kind = fc.make_degenerate_case(seed)
        try:
            hist["kind%d/%s" % (kind, f._model_check(X, model, nonmetric, seed))] += 1
        except Exception:
            bad.append((seed, kind, traceback.format_exc().splitlines()[-1][:500]))
    print("outcomes", dict(sorted(hist.items())))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
