"""A longer run of the hostile-data fuzz of tests/test_gpu_fuzz.py (scales 1e-6 .. 1e6, offsets of up to 1e7 standard deviations, heavy tails, wild cells, items read as
numbers): the int8 digit-plane route against the fp64 route and the oracle.  Seeds A .. B."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import test_gpu_fuzz as f

if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        try:
            hist[f._hostile_case_check(seed)] += 1
        except Exception:
            bad.append((seed, traceback.format_exc().splitlines()[-1][:500]))
    print("outcomes", dict(sorted(hist.items())))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
