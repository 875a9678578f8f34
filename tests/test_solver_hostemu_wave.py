"""CPU check of the wave solver source (csrc/solver_wave.h: one 64-lane wave per problem, three lane roles) through the
std::thread emulation build in tests/hostemu/ (one emulated thread per lane, the device's butterfly order of the stop-rule sum):
against the data-level oracle, the reference's bootstrap rows and the rows / LDS variants of the same solver.
Tolerance vs the oracle: 1e-9 relative (fp64 both sides; the formulations differ); vs the rows variant 1e-11."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, effect_pairs, packed_scatter, satisfaction_oracle_inputs
from test_solver_hostemu import EMU, HERE, RTOL, SCHEME_ID, _ptr, dense_from_packed, run_emu


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    return ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))


def run_wave(lib, X, model, counts=None, shift=None):
    """solve_problem_wave<8> on the dense upper-triangular moment matrix of the device-ordered columns; returns the record pieces in
    DATA column order, or None when the model is outside the wave solver's class."""
    order = model.mv_order
    Xdev = np.ascontiguousarray(X[:, order])
    P, L = Xdev.shape[1], model.L
    Mp, shift, PA = packed_scatter(Xdev, counts, shift)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs] + [0], dtype=np.int32)
    et = np.array([p[1] for p in pairs] + [0], dtype=np.int32)
    ne = len(pairs)
    row = np.full(2 * P + L + 2 * ne + 2, np.nan)
    iters, status = ctypes.c_int(0), ctypes.c_int(-1)
    Md = np.ascontiguousarray(dense_from_packed(Mp, PA, P))
    shift = np.ascontiguousarray(shift, dtype=np.float64)
    rc = lib.hostemu_solve_wave(P, L, PA, SCHEME_ID[model.scheme], int(model.scaled), model.max_iter, ctypes.c_double(model.tol),
                                _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte), _ptr(mode, ctypes.c_int), _ptr(shift), ne,
                                _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(Md), _ptr(row), ctypes.byref(iters), ctypes.byref(status))
    if rc:
        return None
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    assert row[-2] == status.value and row[-1] == iters.value
    return dict(weights=row[:P][inv], r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne],
                loadings=row[P + L + 2 * ne:2 * P + L + 2 * ne][inv], iterations=iters.value, status=status.value, row=row, pairs=pairs)


def check(e, r, tag=""):
    assert e["status"] == 0, tag
    assert e["iterations"] == r["iterations"], tag
    assert e["pairs"] == r["effect_pairs"], tag
    for key in ("weights", "loadings"):
        assert_close(e[key], r[key], RTOL, what=tag + " " + key)
    for key in ("r2", "total", "direct"):
        assert_close(e[key], r[key], RTOL, 1e-13, what=tag + " " + key)


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_wave_satisfaction_vs_oracle_and_rows_variant(emu, scheme, scaled):
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", scheme, scaled)
    e = run_wave(emu, X, model)
    check(e, orc.fit(X, model), "wave %s/%d" % (scheme, scaled))
    base = run_emu(emu, X, model, rows=True)
    assert e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-11, 1e-13)


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_wave_synth_60_columns_and_weighted(emu, scheme):
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", scheme, True)
    check(run_wave(emu, X, model), orc.fit(X, model))
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 2000, 2000)
    counts = np.bincount(idx, minlength=2000)
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_wave(emu, X, model, counts=counts, shift=shift)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(2000))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


@pytest.mark.parametrize("tag", ["A_centroid_0"])
def test_wave_reference_bootstrap_rows(emu, tag):
    """Weighted scatter (multiplicities) == the reference run on data.iloc[idx] (bootstrap.py:56-64), golden g4 from the real reference."""
    from helpers import load
    g = load("g4_satisfaction_boot")
    X, blocks, _ = satisfaction_oracle_inputs()
    m, scheme, scaled = tag.split("_")
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", scheme, bool(int(scaled)))
    shift = X[:, model.mv_order].mean(axis=0)
    for idx, ref_row, it in zip(g["idx"], g[tag + "/rows"], g[tag + "/iters"]):
        e = run_wave(emu, X, model, counts=np.bincount(idx, minlength=X.shape[0]), shift=shift)
        assert e["status"] == 0 and e["iterations"] == int(it)
        assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), ref_row, RTOL, 1e-12, what=tag)


def test_wave_model_shapes(emu):
    """Ragged blocks (1 .. 17 MVs), L = 2 .. 8, 64 MVs exactly, a chain and a dense DAG with 4 predecessors: every lane role at its limits."""
    rs = np.random.RandomState(4)
    cases = []
    cases.append((orc.chain_C(2), [3, 1]))
    cases.append((orc.chain_C(8), [8] * 8))                                   # 64 MVs, 8 LVs: every pair lane live
    cases.append((orc.chain_C(5), [1, 17, 2, 9, 5]))
    C = np.zeros((6, 6), dtype=np.int64)
    for i in range(1, 6):
        for j in range(max(0, i - 4), i):
            C[i, j] = 1                                                       # up to 4 predecessors
    cases.append((C, [4, 3, 5, 2, 6, 7]))
    cases.append((orc.satisfaction_C(), [5, 5, 5, 4, 4, 4]))
    for C, sizes in cases:
        L = C.shape[0]
        N = 400
        eta = np.zeros((N, L))
        for j in range(L):
            eta[:, j] = 0.5 * eta[:, C[j] == 1].sum(axis=1) + rs.standard_normal(N)
        cols, blocks, c0 = [], [], 0
        for j, k in enumerate(sizes):
            lam = np.linspace(0.5, 0.9, k)
            cols.append(eta[:, [j]] * lam + 0.6 * rs.standard_normal((N, k)))
            blocks.append(np.arange(c0, c0 + k)); c0 += k
        X = np.column_stack(cols) + rs.standard_normal(c0)
        for scheme in ("centroid", "factorial", "path"):
            model = orc.Model(blocks, C, "A" * L, scheme, True)
            e = run_wave(emu, X, model)
            assert e is not None
            check(e, orc.fit(X, model), "L=%d %s %s" % (L, sizes, scheme))


def test_wave_status_codes_and_rank_deficient_predecessors(emu):
    from helpers import load
    from test_oracle_golden import g14_case
    X, blocks, _ = satisfaction_oracle_inputs()
    tight = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True, max_iter=2, tol=1e-12)
    e = run_wave(emu, X, tight)
    assert e["status"] == 1 and e["iterations"] == 3           # counter runs to max_iter+1 before giving up (weights.py:181-186)
    Xc = X.copy(); Xc[:, blocks[2][1]] = 3.0                    # a constant MV
    cm = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True)                                     # a constant MV: the reference centres it to zeros -- weight 0, loading 0, the estimate counts (solver_core.h treated_sd)
    e, r = run_wave(emu, Xc, cm), orc.fit(Xc, cm)
    assert e["status"] == 0 and e["iterations"] == r["iterations"] and e["loadings"][blocks[2][1]] == 0.0
    assert_close(e["weights"], r["weights"], 1e-9, 1e-12); assert_close(e["loadings"], r["loadings"], 1e-9, 1e-12)
    # exactly collinear predecessor scores (a cloned LV): the minimum-norm coefficients of the reference's pinv (golden g14)
    g = load("g14_rank_deficient")
    Xb, blocks_b, Cb = g14_case(g, "b")
    for scheme in ("path", "centroid"):
        model = orc.Model(blocks_b, Cb, "AAAAAAA", scheme, True)
        e = run_wave(emu, Xb, model)
        key = "b_A_%s" % scheme
        assert e is not None and e["status"] == 0 and e["iterations"] == int(g[key + "/iters"])
        assert_close(e["weights"], g[key + "/weights"], RTOL, what=key)
        r = orc.fit(Xb, model)
        assert_close(e["direct"], r["direct"], RTOL, 1e-12)
        assert_close(e["r2"], r["r2"], RTOL, 1e-12)


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("modes", ["BBBBBB", "ABABAB", "BAAAAB"])
def test_wave_mode_b_blocks_vs_oracle_and_rows_variant(emu, scheme, modes):
    """Round 4: Mode-B blocks in the wave formulation -- the inverse of every S_bb by an out-of-place Gauss-Jordan sweep on the block's MV
    lanes, once per problem; the outer step is a k-term product per lane (mode.py:50-52).  Against the oracle (lstsq on the data) and the
    rows variant (Cholesky factor + two triangular solves per iteration): same iteration counts."""
    X, blocks, _ = satisfaction_oracle_inputs()
    for scaled in (False, True):
        model = orc.Model(blocks, orc.satisfaction_C(), modes, scheme, scaled)
        e = run_wave(emu, X, model)
        assert e is not None
        check(e, orc.fit(X, model), "wave %s %s/%d" % (modes, scheme, scaled))
        base = run_emu(emu, X, model, rows=True)
        assert e["iterations"] == base["iterations"]
        assert_close(e["row"], base["row"], 1e-10, 1e-12)
    Xs, bs = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(bs, orc.satisfaction_C(), modes, scheme, True)
    rng = np.random.default_rng(6)
    idx = rng.integers(0, 2000, 2000)
    e = run_wave(emu, Xs, model, counts=np.bincount(idx, minlength=2000), shift=Xs[:, model.mv_order].mean(axis=0))
    mine, its = orc.bootstrap_replicate(Xs, model, idx, orc.correction(2000))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def test_wave_mode_b_rank_deficient_blocks_take_the_minimum_norm_route(emu):
    """A duplicated MV and an MV that is a linear combination of others inside Mode-B blocks (golden g14 from the real reference: gelsd's
    minimum-norm weights), blocks of 1 MV, ragged sizes."""
    from helpers import load
    from test_oracle_golden import g14_case
    g = load("g14_rank_deficient")
    Xa, blocks_a, Ca = g14_case(g, "a")
    from helpers import case_modes
    for mtag in ("B", "M"):
        for scheme in ("centroid", "factorial", "path"):
            for scaled in (0, 1):
                key = "a_%s_%s_%d" % (mtag, scheme, scaled)
                model = orc.Model(blocks_a, Ca, case_modes(mtag, mixed="BABABA"), scheme, bool(scaled))
                e = run_wave(emu, Xa, model)
                assert e is not None and e["status"] == 0 and e["iterations"] == int(g[key + "/iters"]), key
                assert_close(e["weights"], g[key + "/weights"], RTOL, what=key)
    rs = np.random.RandomState(9)
    C = orc.chain_C(4)
    sizes = [1, 13, 2, 6]
    N = 500
    eta = np.zeros((N, 4))
    for j in range(4):
        eta[:, j] = 0.5 * eta[:, C[j] == 1].sum(axis=1) + rs.standard_normal(N)
    cols, blocks, c0 = [], [], 0
    for j, k in enumerate(sizes):
        cols.append(eta[:, [j]] * np.linspace(0.5, 0.9, k) + 0.6 * rs.standard_normal((N, k)))
        blocks.append(np.arange(c0, c0 + k)); c0 += k
    X = np.column_stack(cols)
    for modes in ("BBBB", "BABA"):
        for scheme in ("centroid", "path"):
            model = orc.Model(blocks, C, modes, scheme, True)
            e = run_wave(emu, X, model)
            assert e is not None
            check(e, orc.fit(X, model), "%s %s" % (modes, scheme))


def test_wave_declines_models_outside_its_class(emu):
    X, blocks, _ = satisfaction_oracle_inputs()
    Xw, bw = orc.synth(300, orc.chain_C(2), 30, seed=2)
    assert run_wave(emu, Xw, orc.Model(bw, orc.chain_C(2), "BB", "path", True)) is None                    # two Mode-B blocks of 30: their inverses exceed the staging area
    C = orc.chain_C(9)
    Xs, bs = orc.synth(300, C, 3, seed=1)
    assert run_wave(emu, Xs, orc.Model(bs, C, "A" * 9, "path", True)) is None                              # 9 LVs


def test_wave_sign_rule_case(emu):
    from helpers import load
    g = load("g5_sign_rule")
    model = orc.Model([np.arange(0, 2), np.arange(2, 7), np.arange(7, 11)], g["C"], "AAA", "path", True)
    e = run_wave(emu, g["X"], model)
    r = orc.fit(g["X"], model)
    check(e, r)
    assert r["sign"][0] == -1                                   # the flipped LV: loadings / effects carry the sign, weights do not


def test_wave_thread_sanitizer_clean():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_tsan.so"])
    code = ("import sys; sys.path[:0]=[%r,%r]; import ctypes, numpy as np; import plspm_oracle as orc; import test_solver_hostemu_wave as t;"
            "from helpers import satisfaction_oracle_inputs;"
            "lib=ctypes.CDLL(%r); X,b,_=satisfaction_oracle_inputs();"
            "[t.run_wave(lib, X, orc.Model(b, orc.satisfaction_C(), m, s, True)) for s in ('centroid','path') for m in ('AAAAAA', 'BABABB')];"
            "print('tsan-run-done')") % (HERE, os.path.join(os.path.dirname(HERE), "oracle"), os.path.join(EMU, "libplspm_hostemu_tsan.so"))
    tsan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "tsan-run-done" in r.stdout, r.stderr[-2000:]
    assert "data race" not in r.stderr, r.stderr[-4000:]


def test_reciprocal_forms_reach_the_last_bits_from_a_24_bit_seed(emu):
    """wave_rsqrt / wave_rcp: the device refines v_rsq_f64 / v_rcp_f64 (~2^-26) with two Newton steps; the emulation runs the same two steps
    on the exact value rounded to 24 bits -- both must land within two units in the last place, whatever the exponent."""
    emu.hostemu_wave_rsqrt.restype = ctypes.c_double; emu.hostemu_wave_rsqrt.argtypes = [ctypes.c_double]
    emu.hostemu_wave_rcp.restype = ctypes.c_double; emu.hostemu_wave_rcp.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(0)
    xs = np.concatenate((rng.uniform(0.5, 2.0, 400), 10.0 ** rng.uniform(-300, 300, 400), [1.0, 4.0, 2.0 ** -1000, 2.0 ** 1001, 3.0]))
    worst_s = worst_r = 0.0
    for x in xs:
        worst_s = max(worst_s, abs(emu.hostemu_wave_rsqrt(float(x)) * np.sqrt(np.longdouble(x)) - 1))
        worst_r = max(worst_r, abs(emu.hostemu_wave_rcp(float(x)) * np.longdouble(x) - 1), abs(emu.hostemu_wave_rcp(float(-x)) * np.longdouble(-x) - 1))
    assert worst_s < 4.5e-16 and worst_r < 4.5e-16, (worst_s, worst_r)
    assert not np.isfinite(emu.hostemu_wave_rcp(0.0)) and not np.isfinite(emu.hostemu_wave_rsqrt(0.0))      # (inf seed, NaN after the refinement -- as on the device; callers test the pivot first)


@pytest.mark.parametrize("sizes,modes", [([8, 8, 8, 8], "BBBB"), ([12, 3, 9], "BAB"), ([13, 16, 2, 7], "BBAB"), ([16, 16, 16, 16], "BBBB"),
                                         ([17, 8, 5], "BBA"), ([32, 4, 4], "BAB"), ([20, 20, 20], "ABB")])
def test_wave_mode_b_block_widths(emu, sizes, modes):
    """Mode-B blocks at the widths where the inverse phase changes form: rows of 8 / 12 / 16 registers per lane, and -- wider than 16 MVs -- the
    sweep over LDS-resident matrices; mixed with Mode-A blocks and with narrower Mode-B blocks that wait out the widest one's steps."""
    rs = np.random.RandomState(11)
    L = len(sizes)
    C = orc.chain_C(L)
    N = 600
    eta = np.zeros((N, L))
    for j in range(L):
        eta[:, j] = 0.5 * eta[:, C[j] == 1].sum(axis=1) + rs.standard_normal(N)
    cols, blocks, c0 = [], [], 0
    for j, k in enumerate(sizes):
        cols.append(eta[:, [j]] * np.linspace(0.5, 0.9, k) + 0.6 * rs.standard_normal((N, k)))
        blocks.append(np.arange(c0, c0 + k)); c0 += k
    X = np.column_stack(cols) + rs.standard_normal(c0)
    for scheme, scaled in (("centroid", True), ("path", False)):
        model = orc.Model(blocks, C, modes, scheme, scaled)
        e = run_wave(emu, X, model)
        assert e is not None
        check(e, orc.fit(X, model), "%s %s %s" % (sizes, modes, scheme))
        base = run_emu(emu, X, model, rows=True)
        assert e["iterations"] == base["iterations"]
        assert_close(e["row"], base["row"], 1e-10, 1e-12)
