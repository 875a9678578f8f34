"""Differential fuzz of the categorical (Scale.ORD / NOM / NUM mixes) path against the oracle: random path models, block sizes, category counts (2 .. 12), modes
and schemes; the fit (iterations, weights, loadings, path coefficients, scores) and bootstrap replicates on explicit index lists -- through the wave step where it covers
the model and the workgroup step elsewhere.  Seeds A .. B from the command line; prints the route histogram and the failures.  (A test-side tool: it drives the
checker of tests/test_gpu_categorical.py.)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import plspm_oracle as orc
import test_gpu_categorical as tc
from test_gpu_parity import _random_dag, _ragged
from helpers import assert_close


def make_case(seed):
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.integers(2, 7))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 6)) for _ in range(L)]
    n = int(rng.integers(60, 900))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    P = X.shape[1]
    kind = int(rng.integers(0, 3))            # 0 all ORD, 1 ORD / NOM mix, 2 with NUM columns
    scales, data = [], X.copy()
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    for p in range(P):
        s = "ORD" if kind == 0 else ("ORD", "NOM")[int(rng.integers(0, 2))] if kind == 1 else ("ORD", "NOM", "NUM")[int(rng.integers(0, 3))]
        scales.append(s)
        if s != "NUM":
            c = int(rng.integers(2, 13))
            data[:, p] = np.clip(np.round((c + 1) / 2.0 + float(rng.uniform(0.6, 1.4)) * c / 5.0 * Z[:, p]), 1, c)
    all_a = bool(rng.integers(0, 2))
    modes = "".join("A" if all_a or sizes[l] == 1 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    model = orc.Model(blocks, C, modes, scheme, True, tol=1e-6, scales=scales)
    return data, model


def check(seed):
    from plspm import _native
    data, model = make_case(seed)
    n = data.shape[0]
    tag = "seed %d L=%d P=%d n=%d %s %s %s" % (seed, model.L, data.shape[1], n, model.modes, model.scheme, "".join(s[0] for s in model.scales))
    try:
        r = orc.fit(data, model)
    except orc.NotConverged:
        r = None
    except Exception as e:                          # singular blocks etc.: the oracle cannot finish -- the device must not report success
        r = e
    nm, g = tc.gpu_fit_cat(data, model)
    route = "wave" if nm.get_option("last_nm_wave") == 1 else "group"
    if r is None:
        assert g["status"] == 1, tag + ": oracle did not converge, device status %d" % g["status"]
        return route + "/notconv"
    if isinstance(r, Exception):
        assert g["status"] != 0, tag + ": oracle raised %r, device status 0" % (r,)
        return route + "/oracle-raised"
    if not all(np.all(np.isfinite(r[k])) for k in ("weights", "loadings", "path_coef", "scores")):
        return route + "/oracle-nonfinite"
    tc.check_fit(g, r, tag)
    B = 6
    rs = np.random.RandomState(seed)
    idx = rs.randint(n, size=(B, n)).astype(np.int32)
    rows, status, iters = nm.bootstrap(B, idx=idx)
    Pm = len(model.scales)
    rows = tc._rows_in_data_order(rows, g["inv"], Pm, model.L, nm.n_eff)
    corr = orc.correction(n)
    compared = 0
    for b in range(B):
        try:
            mine, its = orc.bootstrap_replicate(data, model, idx[b], corr)
        except Exception:
            assert status[b] != 0, tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        if not np.all(np.isfinite(mine)):
            continue
        assert status[b] == 0 and its == iters[b], tag + " replicate %d: status %d iterations %d vs oracle %d" % (b, status[b], iters[b], its)
        assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        compared += 1
    # the device's own resampling: the same records whatever the route (wave step / workgroup step), where both exist
    if route == "wave":
        a = nm.bootstrap(40, seed=seed)
        nm.set_option("nm_wave", 0)
        w = nm.bootstrap(40, seed=seed)
        nm.set_option("nm_wave", 1)
        assert np.array_equal(a[1], w[1]) and np.array_equal(a[2], w[2]), tag + ": wave / workgroup step disagree on status or iterations"
        ok = a[1] == 0
        assert_close(a[0][ok], w[0][ok], 1e-9, 1e-11, what=tag + " wave vs workgroup step")
    return route + "/%d" % compared


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        try:
            hist[check(seed)] += 1
        except Exception:
            tb = traceback.format_exc().splitlines()
            bad.append((seed, tb[-1][:400]))
    print("routes", dict(hist))
    print("failures", len(bad))
    for x in bad[:40]:
        print(x)
