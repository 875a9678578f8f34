"""A longer run of the many-LV / many-MV fuzz of tests/test_gpu_fuzz.py (fuzz_cases.make_huge_case: 9 ... 48 LVs, up to 230 MVs).  Seeds A .. B; prints the route histogram."""
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
            hist[f._huge_case_check(seed)] += 1
        except BaseException:
            bad.append((seed, traceback.format_exc().splitlines()[-1][:500]))
    print("routes", dict(sorted(hist.items())))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
