import os, sys, traceback
ROOT = os.getcwd()
for p in ("plspm-python_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
np.set_printoptions(linewidth=220, precision=6, suppress=False)
import fuzz_cases as fc, test_gpu_fuzz as f
gen = "edge"
args = sys.argv[1:]
if args and not args[0].isdigit():
    gen, args = args[0], args[1:]
for seed in map(int, args):
    if gen == "smallint":
        X, model, nonmetric = fc.make_small_int_case(seed); kind = -1
    else:
        X, model, nonmetric, kind = fc.make_degenerate_case(seed)
    const = [p for p in range(X.shape[1]) if X[:, p].std() == 0]
    print("=== seed", seed, X.shape, model.modes, model.scheme, "scaled", model.scaled, "constant col", const, "blocks", [list(map(int, b)) for b in model.blocks])
    try:
        print(f._model_check(X, model, nonmetric, seed))
    except Exception:
        tb = traceback.format_exc().splitlines()
        print("\n".join(l[:260] for l in tb[-14:]))
