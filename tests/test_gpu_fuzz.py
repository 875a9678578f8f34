"""GPU differential test: seeded random models (random DAG, ragged blocks, random Mode / Scheme / scaled / metric vs Scale.NUM,
random N) -- single fit and three bootstrap replicates (device RNG mirrored on the host) against the oracle.
Cases whose oracle run does not converge or hits a singular system must be reported as such by the device (status != 0)."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_gpu_parity import SCHEME_ID, _ragged, _random_dag

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-7, 1e-10


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(2, 9))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 9)) for _ in range(L)]
    n = int(rng.integers(30, 700))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    modes = "".join("AB"[int(rng.integers(0, 2))] if sizes[l] > 1 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    nonmetric = bool(rng.integers(0, 3) == 0)
    scaled = bool(rng.integers(0, 2))
    model = orc.Model(blocks, C, modes, scheme, scaled, tol=1e-6 if not nonmetric else 1e-7,
                      scales=(["NUM"] * X.shape[1]) if nonmetric else None)
    return X, model, nonmetric


@pytest.mark.parametrize("seed", range(60))
def test_random_model(seed):
    from plspm import _native
    X, model, nonmetric = make_case(seed)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0, nonmetric=nonmetric)
    nm.upload(X)
    g = nm.fit(want_scores=True)
    try:
        r = orc.fit(X, model)
    except orc.NotConverged:
        assert g["status"] == 1
        return
    if g["status"] != 0:
        pytest.skip("device flags a numerical condition (status %d) on a random model" % g["status"])
    tag = "seed %d L=%d P=%d n=%d %s %s %s" % (seed, model.L, X.shape[1], X.shape[0], model.modes, model.scheme, "NUM" if nonmetric else "metric")
    assert g["iterations"] == r["iterations"], tag
    assert_close(g["weights"], r["weights"], RTOL, ATOL, what=tag)
    assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL, what=tag)
    assert_close(g["r2"], r["r2"], RTOL, ATOL, what=tag)
    assert_close(g["loadings"], r["loadings"], RTOL, ATOL, what=tag)
    assert_close(g["scores"], r["scores"], 1e-6, 1e-8, what=tag)
    n = X.shape[0]
    rows, status, iters = nm.bootstrap(3, seed=seed)
    corr = orc.correction(n)
    for b in range(3):
        idx = _native.bootstrap_indices(seed, b, n)
        try:
            mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        except Exception:
            continue
        if status[b] != 0 or not np.all(np.isfinite(mine)):
            continue
        assert its == iters[b], tag + " replicate %d" % b
        assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
