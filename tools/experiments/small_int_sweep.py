"""The base fuzz checker on small samples of integer-valued items (fuzz_cases.make_small_int_case).  Seeds A .. B."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import fuzz_cases as fc
import test_gpu_fuzz as f
a, b = int(sys.argv[1]), int(sys.argv[2])
hist, bad = collections.Counter(), []
for seed in range(a, b):
    X, model, nonmetric = fc.make_small_int_case(seed)
    try:
        hist[f._model_check(X, model, nonmetric, seed)] += 1
    except Exception:
        bad.append((seed, traceback.format_exc().splitlines()[-1][:400]))
print("outcomes", dict(hist)); print("failures", len(bad))
for x in bad[:30]: print(x)
