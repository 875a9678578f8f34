"""A longer run of tests/test_gpu_fuzz.py::_rare_indicator_check (an item that is constant in some replicates).  Seeds A .. B."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import test_gpu_fuzz as f
a, b = int(sys.argv[1]), int(sys.argv[2])
hist, bad = collections.Counter(), []
for seed in range(a, b):
    try:
        hist[f._rare_indicator_check(seed)] += 1
    except Exception:
        bad.append((seed, traceback.format_exc().splitlines()[-1][:400]))
print("outcomes", dict(hist)); print("failures", len(bad))
for x in bad[:30]: print(x)
