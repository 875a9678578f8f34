"""Differential fuzz of the two-stage higher-order-construct bootstrap (Scale.NUM; plspm_model_attach_second_stage) against the oracle's fit_two_stage: random stage-2 path
models in which one LV is a HOC of two or three first-stage constituents (stage 1 = the path with the HOC expanded in place, every constituent inheriting the HOC's edges:
estimator.py:60-74), ragged blocks, Mode A / B, the three schemes; replicates on explicit index lists, the first of them the data themselves.  Seeds A .. B.  (Generator and checker: tests/test_gpu_fuzz.py.)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


import test_gpu_fuzz as f


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        for fn in (f._hoc_case_check, f._hoc_ord_case_check):
            try:
                hist[fn.__name__[1:9] + fn(seed)] += 1
            except Exception:
                tb = traceback.format_exc().splitlines()
                bad.append((fn.__name__, seed, tb[-1][:400]))
    print("outcomes", dict(hist))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
