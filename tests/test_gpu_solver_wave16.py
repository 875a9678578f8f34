"""GPU parity of solver_wave16_kernel (csrc/solver_wave16.h: one wave per replicate, four matrix entries per pair lane) -- the bootstrap solver of metric
Mode-A models with at most 64 MVs and 9 ... 16 LVs -- through the C-ABI: against the oracle (reference arithmetic on the resampled data) and the rows / LDS
variants on the same moment matrices.  Tolerances: oracle 1e-8 (north_star asks 1e-6); between solver variants 1e-10; iteration counts and status words equal."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_gpu_parity import native_model
from test_solver_hostemu_quad import _shaped
from test_solver_hostemu_wave16 import _dag

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-8, 1e-11


def _three_solvers(nm, B, seed, idx=None):
    out = {}
    for name, (rows_opt, wave_opt, codes) in {"wave16": (1, 1, (6,)), "rows": (1, 0, (2, 1)), "lds": (0, 0, (1,))}.items():
        nm.set_option("solver_rows", rows_opt)
        nm.set_option("solver_wave", wave_opt)
        out[name] = nm.bootstrap(B, seed=seed, idx=idx)
        # (a wide inner model's generic workspace may not fit the rows solver's LDS share: the LDS solver then)
        assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in codes, name
    nm.set_option("solver_rows", 1); nm.set_option("solver_wave", 1)
    return out


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_wave16_solver_equals_rows_and_lds_solvers_and_the_oracle(scheme, scaled):
    """5 MVs x 12 LVs."""
    from plspm import _native
    C = _dag(12, 2)
    X, blocks = orc.synth(3000, C, 5, seed=9)
    model = orc.Model(blocks, C, "A" * 12, scheme, scaled)
    nm = native_model(model)
    nm.upload(X)
    out = _three_solvers(nm, 700, 3)
    rows, status, iters = out["wave16"]
    assert np.all(status == 0)
    for other in ("rows", "lds"):
        assert np.array_equal(status, out[other][1]) and np.array_equal(iters, out[other][2]), other
        assert_close(rows, out[other][0], 1e-10, 1e-13, what=other)
    corr = orc.correction(3000)
    for r in (0, 347, 699):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(3, r, 3000), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("sizes,fan", [([4] * 16, 1), ([4] * 16, 5), ([1] * 8 + [7] * 8, 2), ([1, 17, 2, 9, 5, 3, 3, 3, 4, 1], 4), ([30, 1, 1, 1, 1, 1, 1, 1, 1], 2),
                                       ([5, 4, 3, 6, 2, 5, 4, 3, 6, 2, 5, 4, 3], 6), ([2] * 11, 10), ([3] * 9, 1)])
@pytest.mark.parametrize("scheme", ["factorial", "path"])
def test_wave16_solver_model_shapes(sizes, fan, scheme):
    """Ragged blocks, 64 MVs, 9 ... 16 LVs, one to ten predecessors (in-register LDL', Cholesky in the scratch area), partly filled row groups of the pair lanes."""
    from plspm import _native
    L = len(sizes)
    C = _dag(L, fan)
    X, blocks = _shaped(C, sizes, seed=4, N=700)
    model = orc.Model(blocks, C, "A" * L, scheme, True)
    nm = native_model(model)
    nm.upload(X)
    out = _three_solvers(nm, 130, 11)
    rows, status, iters = out["wave16"]
    assert np.array_equal(status, out["rows"][1]) and np.array_equal(status, out["lds"][1])
    ok = status == 0
    assert ok.sum() >= 120
    assert np.array_equal(iters[ok], out["rows"][2][ok]) and np.array_equal(iters[ok], out["lds"][2][ok])
    assert_close(rows[ok], out["rows"][0][ok], 1e-10, 1e-13, what="rows")
    assert_close(rows[ok], out["lds"][0][ok], 1e-10, 1e-13, what="lds")
    corr = orc.correction(700)
    r = int(np.flatnonzero(ok)[-1])
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(11, r, 700), corr)
    assert its == iters[r]
    assert_close(rows[r], mine, RTOL, ATOL)


def test_wave16_solver_status_codes_sign_rule_and_fallbacks():
    sizes = [5, 6, 4, 7, 5, 4, 6, 5, 4, 5]
    C = _dag(10, 2)
    X, blocks = _shaped(C, sizes, seed=9)
    tight = orc.Model(blocks, C, "A" * 10, "centroid", True, max_iter=2, tol=1e-14)
    nm = native_model(tight)
    nm.upload(X); nm.set_option("gram_path", 2)
    rows, status, iters = nm.bootstrap(70, seed=2)
    assert nm.get_option("last_solver") == 6 and np.all(status == 1) and np.all(iters == 3)
    Xn = X.copy()
    Xn[:, blocks[0][:4]] *= -1.0
    Xn[:, blocks[9][:4]] *= -1.0
    nm = native_model(orc.Model(blocks, C, "A" * 10, "path", True))
    nm.upload(Xn); nm.set_option("gram_path", 2)
    out = _three_solvers(nm, 70, 2)
    assert np.all(out["wave16"][1] == 0)
    assert_close(out["wave16"][0], out["rows"][0], 1e-10, 1e-13)
    # Mode-B blocks (round 5, last part): this solver's as well -- against the rows and LDS solvers
    nmo = native_model(orc.Model(blocks, C, "BABABABABB", "factorial", True))
    nmo.upload(X); nmo.set_option("gram_path", 2)
    outb = _three_solvers(nmo, 70, 1)
    assert np.all(outb["wave16"][1] == 0) and np.array_equal(outb["wave16"][2], outb["rows"][2])
    assert_close(outb["wave16"][0], outb["rows"][0], 1e-10, 1e-12)
    assert_close(outb["wave16"][0], outb["lds"][0], 1e-10, 1e-12)
    # Mode-B blocks whose inverses exceed the staging area: not this solver's
    Xw, bw = _shaped(_dag(9, 1), [25, 25, 2, 2, 2, 2, 2, 2, 2], seed=3)
    nmw = native_model(orc.Model(bw, _dag(9, 1), "BB" + "A" * 7, "centroid", True))
    nmw.upload(Xw); nmw.set_option("gram_path", 2)
    nmw.bootstrap(70, seed=1)
    assert nmw.get_option("last_solver") in (1, 2)


def test_wave16_solver_full_size_batch_properties():
    """10k x 60 x 12, 5,000 replicates: every record converged, equal to the rows solver's, independent of the batch it travels in."""
    C = _dag(12, 2)
    X, blocks = orc.synth(10000, C, 5, seed=0)
    model = orc.Model(blocks, C, "A" * 12, "path", True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") == 6 and np.all(status == 0)
    nm.set_option("solver_wave", 0)
    rows_s, status_s, iters_s = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_solver") == 2 and np.array_equal(iters, iters_s)
    assert_close(rows, rows_s, 1e-10, 1e-13)
    nm.set_option("solver_wave", 1)
    part, _, _ = nm.bootstrap(700, seed=1, rep_offset=4300)
    assert np.array_equal(part, rows[4300:])


@pytest.mark.parametrize("scheme", ["centroid", "path"])
@pytest.mark.parametrize("sizes,modes", [([5] * 12, "BBBBBBBBBBBB"), ([5] * 12, "ABABABABABAB"), ([1, 13, 2, 6, 3, 3, 17, 4, 5, 2], "BBABBABBAB"), ([4] * 16, "B" * 16)])
def test_wave16_solver_mode_b_blocks(sizes, modes, scheme):
    """Mode-B blocks at 9 ... 16 LVs: oracle 1e-8 (lstsq on the resampled data), rows / LDS solvers 1e-10, equal iteration counts."""
    from plspm import _native
    L = len(sizes)
    C = _dag(L, 2)
    X, blocks = _shaped(C, sizes, seed=14, N=900)
    model = orc.Model(blocks, C, modes, scheme, True)
    nm = native_model(model)
    nm.upload(X); nm.set_option("gram_path", 2)
    out = _three_solvers(nm, 130, 5)
    rows, status, iters = out["wave16"]
    ok = status == 0
    assert ok.sum() >= 120 and np.array_equal(status, out["rows"][1]) and np.array_equal(status, out["lds"][1])
    assert np.array_equal(iters[ok], out["rows"][2][ok]) and np.array_equal(iters[ok], out["lds"][2][ok])
    assert_close(rows[ok], out["rows"][0][ok], 1e-10, 1e-12, what="rows")
    assert_close(rows[ok], out["lds"][0][ok], 1e-10, 1e-12, what="lds")
    corr = orc.correction(900)
    r = int(np.flatnonzero(ok)[0])
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(5, r, 900), corr)
    assert its == iters[r]
    assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("scheme", ["centroid", "path"])
@pytest.mark.parametrize("sizes,fan", [([3] * 20, 2), ([2] * 32, 1), ([1] * 10 + [4] * 7, 5), ([5, 1, 2, 3, 1, 4, 2, 1, 3, 2, 1, 1, 6, 2, 3, 1, 2, 4, 1, 2, 3, 1, 1, 2], 3)])
def test_wave16_solver_for_17_to_32_lvs(sizes, fan, scheme):
    """solver_wave16_kernel<32> (sixteen matrix entries per pair lane, one wave per SIMD: 324 registers, three problems per CU): all-Mode-A models of 17 ... 32 LVs
    against the rows / LDS solvers (1e-10, equal iteration counts) and the oracle (1e-8)."""
    from plspm import _native
    L = len(sizes)
    C = _dag(L, fan)
    X, blocks = _shaped(C, sizes, seed=31, N=700)
    model = orc.Model(blocks, C, "A" * L, scheme, True)
    nm = native_model(model)
    nm.upload(X); nm.set_option("gram_path", 2)
    out = {}
    for name, (rows_opt, wave_opt, codes) in {"wave32": (1, 1, (8,)), "rows": (1, 0, (2, 1)), "lds": (0, 0, (1,))}.items():
        nm.set_option("solver_rows", rows_opt); nm.set_option("solver_wave", wave_opt)
        out[name] = nm.bootstrap(130, seed=11)
        assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in codes, (name, nm.get_option("last_solver"))
    rows, status, iters = out["wave32"]
    ok = status == 0
    assert ok.sum() >= 120 and np.array_equal(status, out["lds"][1])
    assert np.array_equal(iters[ok], out["lds"][2][ok]) and np.array_equal(iters[ok], out["rows"][2][ok])
    assert_close(rows[ok], out["lds"][0][ok], 1e-10, 1e-13, what="lds")
    assert_close(rows[ok], out["rows"][0][ok], 1e-10, 1e-13, what="rows")
    r = int(np.flatnonzero(ok)[-1])
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(11, r, 700), orc.correction(700))
    assert its == iters[r]
    assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("L,per", [(12, 5), (12, 10), (20, 3)])
def test_plspm_api_bootstrap_on_models_of_the_new_solver_classes(L, per):
    """Plspm(bootstrap=True) through the host API on chain models of 12 x 5 (wave solver for 9 ... 16 LVs), 12 x 10 (quad solver) and 20 x 3 MVs (17 ... 32 LVs):
    the fit against the oracle, every replicate used, the bootstrap means beside the original weights."""
    import sys, os
    import pandas as pd
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    C = orc.chain_C(L)
    X, blocks = orc.synth(2000, C, per, seed=3)
    cols = ["x%d" % i for i in range(X.shape[1])]
    names = ["LV%02d" % l for l in range(L)]
    s = c.Structure()
    for l in range(1, L):                                      # (orc.chain_C: edges l - 1 -> l and l - 3 -> l)
        s.add_path([names[l - 1]], [names[l]])
        if l >= 3:
            s.add_path([names[l - 3]], [names[l]])
    cfg = c.Config(s.path(), scaled=True)
    for l in range(L):
        cfg.add_lv(names[l], Mode.A, *[c.MV(cols[i]) for i in blocks[l]])
    m = Plspm(pd.DataFrame(X, columns=cols), cfg, Scheme.PATH, 100, 1e-6, bootstrap=True, bootstrap_iterations=600, seed=1)
    ref = orc.fit(X, orc.Model(blocks, C, "A" * L, "path", True, tol=1e-6))
    om = m.outer_model()
    assert_close(om.loc[cols, "weight"].values, ref["weights"], 1e-8, 1e-11)
    bw = m.bootstrap().weights()
    assert m.bootstrap().used() == 600 and np.all(np.isfinite(bw[["mean", "std.error"]].values))
    assert float(np.abs(bw["original"] - bw["mean"]).max()) < 0.01
