"""GPU parity of solver_quad_kernel (csrc/solver_quad.h: the wave solver's fixed lane roles on four waves per replicate) -- the bootstrap solver
of metric Mode-A models with 65 ... 128 MVs and at most 16 LVs -- through the C-ABI: against the oracle (reference arithmetic on the resampled
data) and the split rows / LDS variants on the same moment matrices.  Tolerances: oracle 1e-8 (north_star asks 1e-6); between solver variants
1e-10; iteration counts and status words equal."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_gpu_parity import native_model
from test_solver_hostemu_quad import _shaped

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-8, 1e-11


def _three_solvers(nm, B, seed, idx=None):
    out = {}
    for name, (rows_opt, quad_opt, code) in {"quad": (1, 1, 5), "split": (1, 0, 4), "lds": (0, 0, 1)}.items():
        nm.set_option("solver_rows", rows_opt)
        nm.set_option("solver_quad", quad_opt)
        out[name] = nm.bootstrap(B, seed=seed, idx=idx)
        # (a wide inner model's generic workspace may not fit the split rows solver's LDS share: the LDS solver then)
        assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in ((code, 1) if name == "split" else (code,)), name
    nm.set_option("solver_rows", 1); nm.set_option("solver_quad", 1)
    return out


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_quad_solver_equals_split_rows_and_lds_solvers_and_the_oracle(scheme, scaled):
    """10 MVs x 12 LVs (the size next to the headline in tools/size_bench.py)."""
    from plspm import _native
    C = orc.chain_C(12)
    X, blocks = orc.synth(3000, C, 10, seed=9)
    model = orc.Model(blocks, C, "A" * 12, scheme, scaled)
    nm = native_model(model)
    nm.upload(X)
    assert nm.get_option("solver_quad") == 1
    out = _three_solvers(nm, 600, 3)
    rows, status, iters = out["quad"]
    assert np.all(status == 0)
    for other in ("split", "lds"):
        assert np.array_equal(status, out[other][1]) and np.array_equal(iters, out[other][2]), other
        assert_close(rows, out[other][0], 1e-10, 1e-13, what=other)
    corr = orc.correction(3000)
    for r in (0, 347, 599):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(3, r, 3000), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def _dag(L, fan):
    C = np.zeros((L, L), dtype=np.int64)
    for i in range(1, L):
        for j in range(max(0, i - fan), i):
            C[i, j] = 1
    return C


@pytest.mark.parametrize("sizes,fan", [([64, 64], 1), ([3, 64], 1), ([8] * 16, 1), ([1] * 10 + [9, 9, 9, 9, 9, 20], 2), ([1, 17, 40, 9, 5], 4), ([60, 1, 10], 2),
                                       ([9, 8, 7, 12, 5, 6, 11, 10, 4], 6), ([8] * 16, 3), ([13, 13, 13, 13, 12, 13, 13, 13, 13], 8)])
@pytest.mark.parametrize("scheme", ["factorial", "path"])
def test_quad_solver_model_shapes(sizes, fan, scheme):
    """Ragged blocks, one-LV sides, 128 MVs, 16 LVs, one to eight predecessors (in-register LDL', Cholesky in the staging area)."""
    from plspm import _native
    L = len(sizes)
    C = _dag(L, fan)
    X, blocks = _shaped(C, sizes, seed=4, N=700)
    model = orc.Model(blocks, C, "A" * L, scheme, True)
    nm = native_model(model)
    nm.upload(X)
    out = _three_solvers(nm, 130, 11)
    rows, status, iters = out["quad"]
    assert np.array_equal(status, out["split"][1]) and np.array_equal(status, out["lds"][1])
    ok = status == 0
    assert ok.sum() >= 120
    assert np.array_equal(iters[ok], out["split"][2][ok]) and np.array_equal(iters[ok], out["lds"][2][ok])
    assert_close(rows[ok], out["split"][0][ok], 1e-10, 1e-13, what="split")
    assert_close(rows[ok], out["lds"][0][ok], 1e-10, 1e-13, what="lds")
    corr = orc.correction(700)
    r = int(np.flatnonzero(ok)[-1])
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(11, r, 700), corr)
    assert its == iters[r]
    assert_close(rows[r], mine, RTOL, ATOL)


def test_quad_solver_status_codes_sign_rule_and_fallbacks():
    C = orc.chain_C(4)
    X, blocks = _shaped(C, [20, 25, 15, 30], seed=9)
    tight = orc.Model(blocks, C, "AAAA", "centroid", True, max_iter=2, tol=1e-14)
    nm = native_model(tight)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(70, seed=2)
    assert nm.get_option("last_solver") == 5 and np.all(status == 1) and np.all(iters == 3)
    # most MVs of two blocks negated: the sign rule's votes (weights.py:62-64) are cast per wave and met in LDS
    Xn = X.copy()
    Xn[:, blocks[0][:15]] *= -1.0
    Xn[:, blocks[3][:20]] *= -1.0
    model = orc.Model(blocks, C, "AAAA", "path", True)
    nm = native_model(model)
    nm.upload(Xn)
    out = _three_solvers(nm, 70, 2)
    assert np.all(out["quad"][1] == 0)
    assert_close(out["quad"][0], out["split"][0], 1e-10, 1e-13)
    # a Mode-B block, 17 LVs, no usable block boundary: not the quad solver's
    for sizes, modes in (([30, 20, 30], "ABA"), ([5] * 17, "A" * 17), ([66, 4], "AA")):
        L = len(sizes)
        Xo, bo = _shaped(orc.chain_C(L), sizes, seed=2)
        nmo = native_model(orc.Model(bo, orc.chain_C(L), modes, "centroid", True))
        nmo.upload(Xo)
        nmo.bootstrap(70, seed=1)
        assert nmo.get_option("last_solver") in (1, 4), sizes


def test_quad_solver_full_size_batch_properties():
    """10k x 120 x 12, 5,000 replicates: every record converged, equal to the split rows solver's, independent of the batch it travels in."""
    C = orc.chain_C(12)
    X, blocks = orc.synth(10000, C, 10, seed=0)
    model = orc.Model(blocks, C, "A" * 12, "path", True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") == 5 and np.all(status == 0)
    nm.set_option("solver_quad", 0)
    rows_s, status_s, iters_s = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_solver") == 4 and np.array_equal(iters, iters_s)
    assert_close(rows, rows_s, 1e-10, 1e-13)
    nm.set_option("solver_quad", 1)
    part, _, _ = nm.bootstrap(700, seed=1, rep_offset=4300)
    assert np.array_equal(part, rows[4300:])
