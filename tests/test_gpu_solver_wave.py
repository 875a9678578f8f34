"""GPU parity of solver_wave_kernel (csrc/solver_wave.h: one wave per replicate with fixed lane roles, coalesced triangle load + LDS
transpose) -- the bootstrap solver of metric Mode-A models with at most 64 MVs and 8 LVs -- through the C-ABI: against the oracle
(reference arithmetic on the resampled data), the reference's own bootstrap rows (goldens g3 / g4) and the rows / LDS variants on the
same moment matrices.  Tolerances: oracle / reference 1e-8 (north_star asks 1e-6); between solver variants 1e-11; iteration counts equal."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, load, satisfaction_oracle_inputs
from test_gpu_parity import native_model

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-8, 1e-11


WAVE = (3, 7)      # last_solver of the wave class: 3 = solver_wave_kernel (rounds 3 / 4; option solver_wave 3), 7 = solver_wave16_kernel<8> (round 5: the default)


def _three_solvers(nm, B, seed, idx=None):
    """The default route (wave class), the rows and the LDS solver on the same batch -- and, where the default is the round-5 form, the round-3 wave kernel
    (set_option("solver_wave", 3)): equal iteration counts and status words, records to 1e-11."""
    out = {}
    for name, (rows_opt, wave_opt, codes) in {"wave": (1, 1, WAVE), "rows": (1, 0, (2,)), "lds": (0, 0, (1,))}.items():
        nm.set_option("solver_rows", rows_opt)
        nm.set_option("solver_wave", wave_opt)
        out[name] = nm.bootstrap(B, seed=seed, idx=idx)
        assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in codes, name
        if name == "wave" and nm.get_option("last_solver") == 7:
            nm.set_option("solver_wave", 3)
            rows3, status3, iters3 = nm.bootstrap(B, seed=seed, idx=idx)
            assert nm.get_option("last_solver") == 3 and np.array_equal(status3, out[name][1])
            ok = status3 == 0
            assert np.array_equal(iters3[ok], out[name][2][ok])
            assert_close(rows3[ok], out[name][0][ok], 1e-11, 1e-13, what="round-3 wave kernel")
    nm.set_option("solver_rows", 1); nm.set_option("solver_wave", 1)
    return out


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_wave_solver_equals_rows_and_lds_solvers_and_the_oracle(scheme, scaled):
    from plspm import _native
    X, blocks = orc.synth(3000, orc.satisfaction_C(), 10, seed=9)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", scheme, scaled)
    nm = native_model(model)
    nm.upload(X)
    assert nm.get_option("solver_wave") == 1
    out = _three_solvers(nm, 700, 3)
    rows, status, iters = out["wave"]
    assert np.all(status == 0)
    for other in ("rows", "lds"):
        assert np.array_equal(status, out[other][1]) and np.array_equal(iters, out[other][2]), other
        assert_close(rows, out[other][0], 1e-11, 1e-13, what=other)
    corr = orc.correction(3000)
    for r in (0, 347, 699):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(3, r, 3000), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def test_wave_solver_headline_workload_and_reference_rows():
    """BASELINE configs[2]: 10k x 60 x 6, Mode A, PATH, 5,000 replicates -- the default route is the wave solver; rows of the reference
    itself on explicit indices (golden g3, made by importing the real reference)."""
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in WAVE and np.all(status == 0)
    nm.set_option("solver_wave", 0)
    rows_r, status_r, iters_r = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_solver") == 2 and np.array_equal(iters, iters_r) and np.array_equal(status, status_r)
    assert_close(rows, rows_r, 1e-11, 1e-13)
    nm.set_option("solver_wave", 1)
    gold = load("g3_synth10k_path")
    idx = np.stack([np.random.RandomState(int(s)).randint(10000, size=10000) for s in gold["boot_seeds"]]).astype(np.int32)
    r2, s2, i2 = nm.bootstrap(len(idx), idx=idx)
    assert nm.get_option("last_solver") in WAVE and np.all(s2 == 0) and np.array_equal(i2, gold["boot_iters"])
    assert_close(r2, gold["boot_rows"], RTOL, ATOL)
    # ragged batches: a replicate's record does not depend on the batch it travels in
    a = nm.bootstrap(257, seed=1)[0]
    assert np.array_equal(a, rows[:257])


def test_wave_solver_reference_rows_satisfaction():
    """Identical resample indices -> the reference's replicate rows (bootstrap.py:56-64; golden g4 made by importing the reference)."""
    g = load("g4_satisfaction_boot")
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False)
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    nm.set_option("gram_path", 2)
    rows, status, iters = nm.bootstrap(8, idx=g["idx"])
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in WAVE and np.all(status == 0)
    assert np.array_equal(iters, g["A_centroid_0/iters"])
    P, L, ne = 27, 6, nm.n_eff
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    mine = np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)
    assert_close(mine, g["A_centroid_0/rows"], RTOL, ATOL)


@pytest.mark.parametrize("sizes", [[1] * 8, [8] * 8, [3, 1], [1, 17, 2, 9, 5], [16, 16, 16, 16], [15, 1, 17, 31], [63, 1], [32, 32], [1, 62, 1], [7, 9, 5, 11, 3, 13, 1, 15]])
@pytest.mark.parametrize("scheme", ["factorial", "path"])
def test_wave_solver_model_shapes(sizes, scheme):
    """Block boundaries on / next to / across the sixteen-column seams of the segmented product, one MV per LV, 64 MVs, 2 .. 8 LVs; PATH
    with up to 7 predecessors (more than four go through the Cholesky in LDS scratch).  Against the LDS solver and the oracle."""
    from plspm import _native
    from test_gpu_parity import _ragged
    L = len(sizes)
    C = orc.chain_C(L) if scheme == "factorial" else np.tril(np.ones((L, L), dtype=np.int64), -1)
    X, blocks = _ragged(900, C, sizes, seed=4)
    model = orc.Model(blocks, C, "A" * L, scheme, True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(130, seed=11)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_solver") in WAVE
    nm.set_option("solver_rows", 0)
    rows0, status0, iters0 = nm.bootstrap(130, seed=11)
    assert nm.get_option("last_solver") == 1
    assert np.array_equal(status, status0) and np.array_equal(iters, iters0)
    ok = status == 0
    assert ok.sum() >= 100
    assert_close(rows[ok], rows0[ok], 1e-10, 1e-12)
    corr = orc.correction(900)
    r = int(np.flatnonzero(ok)[0])
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(11, r, 900), corr)
    assert its == iters[r]
    assert_close(rows[r], mine, RTOL, ATOL)


def test_wave_solver_status_codes_and_fallbacks():
    """Not-converged counter (weights.py:181-186), a constant MV, collinear predecessor scores (minimum-norm coefficients, golden g14);
    9 LVs and Mode-B blocks whose inverses exceed the staging area are outside the class and take the rows solver."""
    from test_oracle_golden import g14_case
    X, blocks, _ = satisfaction_oracle_inputs()
    tight = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True, max_iter=2, tol=1e-12)
    nm = native_model(tight)
    nm.upload(X, tight.mv_order.astype(np.int32)); nm.set_option("gram_path", 2)
    rows, status, iters = nm.bootstrap(64, seed=2)
    assert nm.get_option("last_solver") in WAVE and np.all(status == 1) and np.all(iters == 3)
    Xc = X.copy(); Xc[:, blocks[2][1]] = 3.0
    nm = native_model(orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True))
    nm.upload(Xc, tight.mv_order.astype(np.int32)); nm.set_option("gram_path", 2)
    flat = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True)
    rows, status, iters = nm.bootstrap(32, seed=2)              # a constant MV: weight 0, loading 0, the replicate counts -- as in the reference (solver_core.h treated_sd)
    assert np.all(status == 0) and nm.get_option("last_solver") in WAVE
    from plspm import _native
    mine, its = orc.bootstrap_replicate(Xc, flat, _native.bootstrap_indices(2, 5, Xc.shape[0]), orc.correction(Xc.shape[0]))
    P = Xc.shape[1]
    inv = np.empty(P, dtype=np.int64); inv[flat.mv_order] = np.arange(P)
    got = np.concatenate((rows[5, :P][inv], rows[5, P:P + 6 + 2 * nm.n_eff], rows[5, P + 6 + 2 * nm.n_eff:][inv]))
    assert its == iters[5] and mine[P + 6 + 2 * nm.n_eff + blocks[2][1]] == 0.0
    assert_close(got, mine, 1e-8, 1e-11)
    g = load("g14_rank_deficient")
    Xb, blocks_b, Cb = g14_case(g, "b")
    model = orc.Model(blocks_b, Cb, "AAAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(Xb, model.mv_order.astype(np.int32)); nm.set_option("gram_path", 2)
    out = _three_solvers(nm, 40, 5)
    assert np.array_equal(out["wave"][1], out["lds"][1]) and np.array_equal(out["wave"][2], out["lds"][2])
    ok = out["wave"][1] == 0
    assert ok.sum() >= 30
    assert_close(out["wave"][0][ok], out["lds"][0][ok], 1e-9, 1e-11)
    Xw, bw = orc.synth(300, orc.chain_C(2), 30, seed=2)                  # two Mode-B blocks of 30 MVs: 1,800 doubles of inverses > 1,056
    nm = native_model(orc.Model(bw, orc.chain_C(2), "BB", "path", True))
    nm.upload(Xw); nm.set_option("gram_path", 2)
    nm.bootstrap(16, seed=1)
    assert nm.get_option("last_solver") == 2
    C9 = orc.chain_C(9)
    X9, b9 = orc.synth(300, C9, 3, seed=1)
    nm = native_model(orc.Model(b9, C9, "A" * 9, "path", True))
    nm.upload(X9); nm.set_option("gram_path", 2)
    nm.bootstrap(16, seed=1)
    assert nm.get_option("last_solver") == 6                   # nine LVs: the wave solver for 9 ... 16 LVs (round 5, solver_wave16.h; tests/test_gpu_solver_wave16.py)
    C17 = orc.chain_C(17)
    X17, b17 = orc.synth(300, C17, 3, seed=1)
    nm = native_model(orc.Model(b17, C17, "A" * 17, "path", True))
    nm.upload(X17); nm.set_option("gram_path", 2)
    nm.bootstrap(16, seed=1)
    assert nm.get_option("last_solver") == 8                   # 17 ... 32 LVs, all Mode A: solver_wave16_kernel<32>
    nm = native_model(orc.Model(b17, C17, "A" * 16 + "B", "path", True))
    nm.upload(X17); nm.set_option("gram_path", 2)
    nm.bootstrap(16, seed=1)
    assert nm.get_option("last_solver") == 2                   # ... with a Mode-B block: the rows solver


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("modes", ["BBBBBB", "ABABAB"])
def test_wave_solver_mode_b_blocks(modes, scheme):
    """Round 4: Mode-B blocks on the wave solver (inverse of every S_bb by a Gauss-Jordan sweep on the block's MV lanes, once per problem; the
    reference solves lstsq(X_b, z) in every iteration, mode.py:50-52).  Same iteration counts as the rows and LDS solvers (Cholesky factor +
    triangular solves), records to 1e-10; replicates against the oracle on the same indices; the reference's rows of golden g14 (a duplicated
    and a linearly dependent MV inside Mode-B blocks: gelsd's minimum-norm weights) on explicit indices."""
    from plspm import _native
    from test_oracle_golden import g14_case
    X, blocks = orc.synth(3000, orc.satisfaction_C(), 10, seed=9)
    for scaled in (False, True):
        model = orc.Model(blocks, orc.satisfaction_C(), modes, scheme, scaled)
        nm = native_model(model)
        nm.upload(X)
        out = _three_solvers(nm, 300, 3)
        rows, status, iters = out["wave"]
        assert np.all(status == 0)
        for other in ("rows", "lds"):
            assert np.array_equal(out[other][1], status) and np.array_equal(out[other][2], iters), other
            assert_close(out[other][0], rows, 1e-10, 1e-12)
        corr = orc.correction(3000)
        for r in (0, 299):
            mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(3, r, 3000), corr)
            assert its == iters[r]
            assert_close(rows[r], mine, RTOL, ATOL)
    g = load("g14_rank_deficient")
    Xa, blocks_a, Ca = g14_case(g, "a")
    key = "a_%s_%s_1" % ("B" if modes == "BBBBBB" else "M", scheme)
    model = orc.Model(blocks_a, Ca, modes if modes == "BBBBBB" else "BABABA", scheme, True)
    nm = native_model(model)
    nm.upload(Xa, model.mv_order.astype(np.int32)); nm.set_option("gram_path", 2)
    if key + "/boot_rows" in g.files:
        rows, status, iters = nm.bootstrap(len(g["idx"]), idx=g["idx"].astype(np.int32))
        assert nm.get_option("last_solver") in WAVE and np.all(status == 0) and np.array_equal(iters, g[key + "/boot_iters"])
        inv = np.empty(len(model.mv_order), dtype=np.int64); inv[model.mv_order] = np.arange(len(model.mv_order))
        P, L = len(inv), model.L
        ne = (rows.shape[1] - 2 * P - L) // 2
        mine = np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)
        assert_close(mine, g[key + "/boot_rows"], RTOL, 1e-8)
    out = _three_solvers(nm, 64, 5)
    assert np.array_equal(out["wave"][1], out["lds"][1]) and np.array_equal(out["wave"][2], out["lds"][2])
    assert_close(out["wave"][0], out["lds"][0], 1e-8, 1e-10)


@pytest.mark.parametrize("sizes,modes", [([8, 8, 8, 8], "BBBB"), ([12, 3, 9], "BAB"), ([13, 16, 2, 7], "BBAB"), ([16, 16, 16, 16], "BBBB"),
                                         ([17, 8, 5], "BBA"), ([32, 4, 4], "BAB"), ([20, 20, 20], "ABB")])
def test_wave_solver_mode_b_block_widths(sizes, modes):
    """Mode-B blocks at the widths where the inverse phase changes form (rows of 8 / 12 / 16 registers per lane; wider than 16 MVs the sweep over
    LDS-resident matrices): wave solver against the rows and LDS solvers on the same moment matrices and against the oracle."""
    from plspm import _native
    from test_gpu_parity import _ragged
    L = len(sizes)
    C = orc.chain_C(L)
    X, blocks = _ragged(900, C, sizes, seed=13)
    for scheme, scaled in (("centroid", True), ("path", False)):
        model = orc.Model(blocks, C, modes, scheme, scaled)
        nm = native_model(model)
        nm.upload(X)
        out = _three_solvers(nm, 200, 4)
        rows, status, iters = out["wave"]
        assert np.all(status == 0), (sizes, modes, scheme)
        for other in ("rows", "lds"):
            assert np.array_equal(out[other][1], status) and np.array_equal(out[other][2], iters), other
            assert_close(out[other][0], rows, 1e-10, 1e-12, what="%s %s %s" % (sizes, modes, other))
        corr = orc.correction(900)
        for r in (0, 199):
            mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(4, r, 900), corr)
            assert its == iters[r]
            assert_close(rows[r], mine, RTOL, ATOL)
