"""A longer run of the differential fuzz of tests/test_gpu_fuzz.py (narrow and wide models on the int8 route: every solver form against the LDS solver and
the oracle): seeds A .. B from the command line.  Prints the solver histogram and the failures."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import pytest
import test_gpu_fuzz as f
a, b = int(sys.argv[1]), int(sys.argv[2])
hist, bad = collections.Counter(), []
for seed in range(a, b):
    for name, fn in (("narrow", f._narrow_case_check), ("wide", f.test_random_wide_model_bootstrap)):
        try:
            code = fn(seed)
            hist[(name, code)] += 1
        except pytest.skip.Exception:
            hist[(name, "skip")] += 1
        except Exception:
            bad.append((name, seed, traceback.format_exc().splitlines()[-1][:300]))
print("solvers", dict(hist))
print("failures", len(bad))
for x in bad[:20]: print(x)
