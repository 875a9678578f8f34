"""CPU check of the device solver source (csrc/solver_core.h) through the std::thread emulation build in
tests/hostemu/: second-moment formulation vs the data-level oracle, all Mode x Scheme x scaled cases,
plus weighted (bootstrap-count) problems, thread-count invariance and a ThreadSanitizer run.
Tolerance: 1e-9 relative (fp64 both sides; the formulations differ, SURVEY.md A.7 measured <= 6e-13)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import (assert_close, case_modes, effect_pairs, packed_index_np, packed_scatter, satisfaction_oracle_inputs)

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL = 1e-9


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    return ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def dense_from_packed(Mp, PA, P):
    """[(P+1) x cov_ld(P)] upper-triangular moment matrix (the layout the int8 Gram writes for the rows solver) out of the tile-packed one."""
    PS = (P + 1) | 1
    T = PA // 16
    idx = np.arange(P + 1)
    Md = np.zeros((P + 1, PS))
    Md[:, :P + 1] = np.triu(Mp[packed_index_np(T, idx[:, None], idx[None, :])])      # upper triangle only, as the Gram writes it
    Md[np.tril_indices(P + 1, -1)] = np.nan                                            # nothing may read below the diagonal
    return Md


def run_emu(lib, X, model, counts=None, shift=None, nthreads=4, packed=None, rows=False, split=False):
    """`packed`: precomputed (Mp, shift, PA) of the DEVICE-ordered columns instead of the scatter of X (X then only feeds the score check).
    `rows`: the one-wave-per-problem variant (solve_problem_rows<64>) on the dense moment matrix, 64 emulated lanes; with `split` its
    two-threads-per-MV form for 65 ... 128 MVs (solve_problem_rows<64, true>, 256 emulated threads)."""
    order = model.mv_order
    Xdev = np.ascontiguousarray(X[:, order])
    P, L = Xdev.shape[1], model.L
    Mp, shift, PA = packed if packed is not None else packed_scatter(Xdev, counts, shift)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs], dtype=np.int32)
    et = np.array([p[1] for p in pairs], dtype=np.int32)
    ne = len(pairs)
    row = np.zeros(2 * P + L + 2 * ne + 2); cl = np.zeros((P, L)); pc = np.zeros((L, L)); lc = np.zeros((L, L))
    ind = np.zeros(max(ne, 1)); sw = np.zeros(P); sc = np.zeros(L); cov = np.zeros((P, P)); mean = np.zeros(P)
    sign = np.zeros(L, dtype=np.int8); iters = ctypes.c_int(0); status = ctypes.c_int(-1)
    shift = np.ascontiguousarray(shift, dtype=np.float64)
    if rows:
        Md = np.ascontiguousarray(dense_from_packed(Mp, PA, P))
        rc = lib.hostemu_solve_rows(P, L, PA, SCHEME_ID[model.scheme], int(model.scaled), model.max_iter, ctypes.c_double(model.tol),
                                    _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte), _ptr(mode, ctypes.c_int), _ptr(shift), ne,
                                    _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(Md), 256 if split else 64, _ptr(row), _ptr(cl), _ptr(pc),
                                    _ptr(lc), _ptr(ind), _ptr(sw), _ptr(sc), _ptr(cov), _ptr(mean), _ptr(sign, ctypes.c_int8),
                                    ctypes.byref(iters), ctypes.byref(status))
        assert rc == 0
    else:
        lib.hostemu_solve(P, L, PA, SCHEME_ID[model.scheme], int(model.scaled), model.max_iter, ctypes.c_double(model.tol),
                          _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte), _ptr(mode, ctypes.c_int), _ptr(shift), ne,
                          _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(Mp), nthreads, _ptr(row), _ptr(cl), _ptr(pc),
                          _ptr(lc), _ptr(ind), _ptr(sw), _ptr(sc), _ptr(cov), _ptr(mean), _ptr(sign, ctypes.c_int8),
                          ctypes.byref(iters), ctypes.byref(status))
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)          # data column -> device column
    w = row[:P][inv]; ld = row[P + L + 2 * ne:2 * P + L + 2 * ne][inv]
    scores = ((Xdev - shift) * sw) @ np.equal.outer(np.repeat(np.arange(L), np.diff(boff)), np.arange(L)).astype(float) + sc
    return dict(weights=w, r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne], loadings=ld,
                crossloadings=cl[inv], path_coef=pc, lv_cov=lc, indirect=ind[:ne], sign=sign, iterations=iters.value,
                status=status.value, pairs=pairs, scores=scores, cov=cov, mean=mean, row=row, inv=inv)


def check(e, r, tag=""):
    assert e["status"] == 0, tag
    assert e["iterations"] == r["iterations"], tag
    assert_close(e["weights"], r["weights"], RTOL, what=tag + " weights")
    assert_close(e["loadings"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(e["crossloadings"], r["crossloadings"], RTOL, 1e-13, what=tag + " crossloadings")
    assert_close(e["path_coef"], r["path_coef"], RTOL, 1e-13, what=tag + " path")
    assert_close(e["r2"], r["r2"], RTOL, 1e-13, what=tag + " r2")
    assert e["pairs"] == r["effect_pairs"], tag
    assert_close(e["total"], r["total"], RTOL, 1e-13)
    assert_close(e["direct"], r["direct"], RTOL, 1e-13)
    assert_close(e["indirect"], r["indirect"], RTOL, 1e-13)
    assert_close(e["scores"], r["scores"], 1e-8, 1e-10, what=tag + " scores")
    assert np.array_equal(e["sign"], r["sign"].astype(np.int8)), tag


def test_packed_index_matches_numpy(emu):
    emu.hostemu_packed_index.restype = ctypes.c_long
    for T in (2, 4, 5, 13, 14):                       # even: pairs of tiles per 32-column group; odd: a partner-less last tile
        PA = 16 * T
        seen = set()
        for p in range(PA):
            for q in range(PA):
                a = emu.hostemu_packed_index(T, p, q)
                assert a == int(packed_index_np(T, np.array(p), np.array(q)))
                assert a == emu.hostemu_packed_index(T, q, p)
                seen.add(a)
        assert len(seen) <= T * (T + 1) // 2 * 256 and max(seen) < T * (T + 1) // 2 * 256


@pytest.mark.parametrize("modes", ["A", "B", "M"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_satisfaction_all_cases(emu, modes, scheme, scaled):
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, scaled)
    check(run_emu(emu, X, model), orc.fit(X, model), "%s/%s/%d" % (modes, scheme, scaled))


@pytest.mark.parametrize("modes", ["A", "B", "M"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_rows_variant_satisfaction_all_cases(emu, modes, scheme, scaled):
    """solve_problem_rows<64> (one wave per problem, S rows in registers, dense moment matrix) against the oracle and the LDS variant."""
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, scaled)
    e = run_emu(emu, X, model, rows=True)
    check(e, orc.fit(X, model), "rows %s/%s/%d" % (modes, scheme, scaled))
    base = run_emu(emu, X, model)
    assert e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-11, 1e-13)
    assert_close(e["cov"], base["cov"], 1e-13, 1e-15)


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial"), ("M", "centroid")])
def test_rows_variant_synth_60_columns_and_weighted(emu, modes, scheme):
    from helpers import load
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, True)
    check(run_emu(emu, X, model, rows=True), orc.fit(X, model))
    rng = np.random.default_rng(5)
    idx = rng.integers(0, 2000, 2000)
    counts = np.bincount(idx, minlength=2000)
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_emu(emu, X, model, counts=counts, shift=shift, rows=True)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(2000))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def test_rows_variant_rank_deficient_and_status(emu):
    from helpers import load
    X, blocks, _ = satisfaction_oracle_inputs()
    Xd = X.copy()
    Xd[:, blocks[1][1]] = Xd[:, blocks[1][0]]                       # duplicated MV in a Mode-B block: minimum-norm weights
    model = orc.Model(blocks, orc.satisfaction_C(), "ABABAB", "path", True)
    e, base = run_emu(emu, Xd, model, rows=True), run_emu(emu, Xd, model)
    assert e["status"] == base["status"] == 0 and e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-9, 1e-11)
    tight = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True, max_iter=2, tol=1e-12)
    e = run_emu(emu, X, tight, rows=True)
    assert e["status"] == 1 and e["iterations"] == 3


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial"), ("M", "centroid")])
def test_synth_2000(emu, modes, scheme):
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, True)
    check(run_emu(emu, X, model), orc.fit(X, model))


@pytest.mark.parametrize("modes,scheme", [("B", "factorial"), ("A", "path"), ("B", "path")])
def test_chain20(emu, modes, scheme):
    C = orc.chain_C(20)
    X, blocks = orc.synth(3000, C, 10, seed=3)
    model = orc.Model(blocks, C, modes * 20, scheme, True)
    check(run_emu(emu, X, model, nthreads=8), orc.fit(X, model))


@pytest.mark.parametrize("scheme", ["centroid", "path"])
def test_sign_rule_case(emu, scheme):
    from helpers import load
    g = load("g5_sign_rule")
    model = orc.Model([np.arange(0, 2), np.arange(2, 7), np.arange(7, 11)], g["C"], "AAA", scheme, True)
    e = run_emu(emu, g["X"], model)
    check(e, orc.fit(g["X"], model))
    assert e["sign"][0] == -1


@pytest.mark.parametrize("tag", ["A_centroid_0", "B_path_1", "M_factorial_1"])
def test_bootstrap_counts_vs_oracle_and_reference_rows(emu, tag):
    """Weighted scatter (multiplicities) == the reference run on data.iloc[idx] (bootstrap.py:56-64)."""
    from helpers import load
    g = load("g4_satisfaction_boot")
    X, blocks, _ = satisfaction_oracle_inputs()
    m, scheme, scaled = tag.split("_")
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(m), scheme, bool(int(scaled)))
    shift = X[:, model.mv_order].mean(axis=0)
    for idx, ref_row, it in zip(g["idx"], g[tag + "/rows"], g[tag + "/iters"]):
        counts = np.bincount(idx, minlength=X.shape[0])
        e = run_emu(emu, X, model, counts=counts, shift=shift)
        assert e["status"] == 0 and e["iterations"] == int(it)
        mine = np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"]))
        assert_close(mine, ref_row, RTOL, 1e-12, what=tag)


def test_thread_count_invariance(emu):
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "ABABAB", "path", True)
    base = run_emu(emu, X, model, nthreads=1)["row"]
    for nt in (2, 3, 7, 16):
        other = run_emu(emu, X, model, nthreads=nt)["row"]
        assert_close(other, base, 1e-12, 1e-14)                # group reductions sum per-thread partials: order depends on the group size
        assert np.array_equal(other, run_emu(emu, X, model, nthreads=nt)["row"])    # but a fixed group size is bit-reproducible


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial")])
def test_odd_tile_count_layout(emu, modes, scheme):
    """70 MVs + the ones column = 71 columns: 5 tiles of 16 (row pitch 80) instead of 6 -- the geometry metric handles use for
    5 <= T <= 15 (configs[4]: 201 columns -> 13 tiles).  The solver must read the partner-less last tile correctly."""
    C = orc.chain_C(7)
    X, blocks = orc.synth(600, C, 10, seed=23)
    model = orc.Model(blocks, C, modes * 7, scheme, True)
    Xdev = np.ascontiguousarray(X[:, model.mv_order])
    packed = packed_scatter(Xdev, odd_tiles=True)
    assert packed[2] == 80
    check(run_emu(emu, X, model, packed=packed), orc.fit(X, model), "odd tiles " + modes)
    idx = np.random.RandomState(3).randint(600, size=600)
    shift = Xdev.mean(axis=0)
    w = run_emu(emu, X, model, packed=packed_scatter(Xdev, np.bincount(idx, minlength=600), shift, odd_tiles=True))
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(600))
    assert w["status"] == 0 and w["iterations"] == its
    assert_close(np.concatenate((w["weights"], w["r2"], w["total"], w["direct"], w["loadings"])), mine, RTOL, 1e-12)


def test_not_converged_status(emu):
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False, max_iter=2, tol=1e-30)
    e = run_emu(emu, X, model)
    assert e["status"] == 1 and e["iterations"] == 3      # counter runs to max_iter+1 before giving up (weights.py:181-186)
    with pytest.raises(orc.NotConverged):
        orc.fit(X, model)


@pytest.mark.parametrize("which,modes,scheme,scaled", [("a", "BBBBBB", "centroid", True), ("a", "BBBBBB", "path", False), ("a", "BABABA", "factorial", True),
                                                       ("b", "AAAAAAA", "path", True), ("b", "BBBBBBB", "path", True), ("b", "AAAAAAA", "centroid", True)])
def test_rank_deficient_least_squares_minimum_norm(emu, which, modes, scheme, scaled):
    """Collinear Mode-B blocks (duplicated / linearly dependent MV) and exactly collinear predecessor scores (a cloned LV): the device
    solver falls back from Cholesky to the eigen-truncated pseudo-inverse (solver_core.h jacobi_pinv) and lands on the minimum-norm
    solution the reference's gelsd / pinv return (golden g14, made from the real reference) -- fit and weighted (bootstrap) problems."""
    from helpers import load
    from test_oracle_golden import g14_case
    g = load("g14_rank_deficient")
    X, blocks, C = g14_case(g, which)
    model = orc.Model(blocks, C, modes, scheme, scaled)
    e = run_emu(emu, X, model)
    check(e, orc.fit(X, model), which + modes + scheme)
    key = ("a_%s_%s_%d" % ("B" if modes == "BBBBBB" else "M", scheme, int(scaled))) if which == "a" else "b_%s_%s" % (modes[0], scheme)
    assert e["iterations"] == int(g[key + "/iters"])
    assert_close(e["weights"], g[key + "/weights"], RTOL, what=key)                     # straight against the reference's numbers
    assert_close(e["path_coef"], g[key + "/path_coef"], RTOL, 1e-12, what=key)
    if which == "a":
        assert abs(e["weights"][0] - e["weights"][5]) < 1e-12                           # the duplicated MV shares its weight with its twin
    else:
        assert abs(e["path_coef"][5, 1] - e["path_coef"][5, 2]) < 1e-10                 # the clone shares its coefficient
    if key + "/boot_rows" in g.files:
        idx = g["idx"][0]
        w = run_emu(emu, X, model, counts=np.bincount(idx, minlength=250), shift=X[:, model.mv_order].mean(axis=0))
        assert w["status"] == 0 and w["iterations"] == int(g[key + "/boot_iters"][0])
        P, L, ne = X.shape[1], model.L, len(w["pairs"])
        mine = np.concatenate((w["weights"], w["r2"], w["total"], w["direct"], w["loadings"]))
        assert_close(mine, g[key + "/boot_rows"][0], RTOL, 1e-11, what=key + " bootstrap row")


@pytest.mark.parametrize("modes", ["AAAAAA", "ABBABA"])
@pytest.mark.parametrize("rows", [False, True])
def test_zero_variance_column_follows_the_reference(emu, modes, rows):
    """A constant MV is centred to exact zeros by the reference: Mode-A weight 0 (Mode B: the minimum-norm answer gives it 0 as well), pandas' corrwith NaN for its
    cross-loadings, which `(crossloadings * odm).sum(axis=1)` skips -- loading 0, and the estimate counts (round 6: found by tests/golden/sweep_oracle_vs_reference.py `edge`;
    until then the device flagged PLSPM_NONFINITE, i.e. dropped every bootstrap replicate in which a rare binary item came out constant).  LDS and rows solver."""
    X, blocks, _ = satisfaction_oracle_inputs()
    X = X.copy(); X[:, blocks[2][1]] = 3.0
    model = orc.Model(blocks, orc.satisfaction_C(), modes, "centroid", True)
    ref = orc.fit(X, model)
    assert ref["loadings"][blocks[2][1]] == 0.0 and abs(ref["weights"][blocks[2][1]]) < 1e-15
    got = run_emu(emu, X, model, rows=rows)
    assert got["status"] == 0 and got["iterations"] == ref["iterations"]
    assert got["loadings"][blocks[2][1]] == 0.0
    assert_close(got["weights"], ref["weights"], RTOL, 1e-12, what="weights")
    assert_close(got["loadings"], ref["loadings"], RTOL, 1e-12, what="loadings")
    assert_close(got["r2"], ref["r2"], RTOL, 1e-12)


def test_thread_sanitizer_clean():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_tsan.so"])
    code = ("import sys; sys.path[:0]=[%r,%r]; import ctypes, numpy as np; import plspm_oracle as orc; import test_solver_hostemu as t;"
            "from helpers import satisfaction_oracle_inputs;"
            "lib=ctypes.CDLL(%r); X,b,_=satisfaction_oracle_inputs();"
            "[t.run_emu(lib, X, orc.Model(b, orc.satisfaction_C(), m, s, True), nthreads=6) for m in ('AAAAAA','BBBBBB') for s in ('centroid','path')];"
            "[t.run_emu(lib, X, orc.Model(b, orc.satisfaction_C(), m, 'path', True), rows=True) for m in ('AAAAAA','BBBBBB')];"      # one thread per MV
            "Xw,mw=t._wide_model(10, 12, 'M', 'path', 21); t.run_emu(lib, Xw, mw, rows=True, split=True);"                              # two threads per MV
            "print('tsan-run-done')") % (HERE, os.path.join(os.path.dirname(HERE), "oracle"), os.path.join(EMU, "libplspm_hostemu_tsan.so"))
    tsan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "tsan-run-done" in r.stdout, r.stderr[-2000:]
    assert "data race" not in r.stderr, r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------- operator seam (csrc/solver_ops.h)
def _scheme_np(name, C, y):
    """reference scheme.py:27-28 / 36-37 / 45-54 in NumPy."""
    if name == "centroid":
        return np.sign(np.corrcoef(y, rowvar=False) * (C + C.T))
    if name == "factorial":
        return np.cov(y, rowvar=False) * (C + C.T)
    E = C.astype(np.float64)
    for i in range(C.shape[0]):
        follow = C[i, :] == 1
        if follow.any():
            E[follow, i] = np.linalg.pinv(y[:, follow]) @ y[:, i]
        predec = C[:, i] == 1
        if predec.any():
            E[predec, i] = np.corrcoef(np.column_stack((y[:, predec], y[:, i])), rowvar=False)[:, -1][:-1]
    return E


@pytest.mark.parametrize("name", ["centroid", "factorial", "path"])
def test_operator_seam_scheme_on_moments(emu, name):
    rs = np.random.RandomState(12)
    C = orc.satisfaction_C()
    y = rs.standard_normal((400, 6)) @ (np.eye(6) + 0.4 * np.tril(rs.standard_normal((6, 6)), -1)).T * np.array([1, 2, 0.5, 3, 1, 0.2]) + np.array([0.5, -1, 2, 0, 3, -0.2])
    Mp, shift, PA = packed_scatter(y)
    E = np.zeros((6, 6))
    Cb = np.ascontiguousarray(C.astype(np.uint8))
    emu.hostemu_op_inner_weights(6, PA, SCHEME_ID[name], _ptr(Cb, ctypes.c_ubyte), _ptr(np.ascontiguousarray(shift)), _ptr(Mp), 4, _ptr(E))
    assert_close(E, _scheme_np(name, C, y), 1e-10, 1e-13, what=name)


@pytest.mark.parametrize("mode", [0, 1])
def test_operator_seam_outer_weights_on_moments(emu, mode):
    rs = np.random.RandomState(13)
    X = rs.standard_normal((300, 7)) + np.array([1, 0, -2, 0.5, 0, 3, 1])
    X[:, 6] = X[:, 0] - 2 * X[:, 3]                                  # rank deficient: Mode B must return the minimum-norm weights
    z = rs.standard_normal(300) + 0.3 * X[:, 1]
    Mp, shift, PA = packed_scatter(np.column_stack((X, z)))
    w = np.zeros(7)
    emu.hostemu_op_outer_weights.restype = ctypes.c_int
    assert emu.hostemu_op_outer_weights(mode, 7, PA, _ptr(np.ascontiguousarray(shift)), _ptr(Mp), 3, _ptr(w)) == 0
    want = X.T @ z / 300 if mode == 0 else np.linalg.lstsq(X, z, rcond=None)[0]
    assert_close(w, want, 1e-9, 1e-12)


def _wide_model(P_per, L, modes, scheme, seed):
    """L LVs in a chain with a few extra paths, P_per MVs each."""
    C = np.zeros((L, L))
    for i in range(1, L):
        C[i, i - 1] = 1
        if i >= 3: C[i, i - 3] = 1
    X, blocks = orc.synth(1500, C, P_per, seed=seed)
    return X, orc.Model(blocks, C, case_modes(modes) if L == 6 else [("B" if (modes == "B" or (modes == "M" and l % 3 == 0)) else "A") for l in range(L)], scheme, True)


@pytest.mark.parametrize("modes,scheme,P_per,L", [("A", "path", 10, 12), ("M", "factorial", 10, 12), ("B", "centroid", 9, 9), ("A", "centroid", 16, 8), ("M", "path", 13, 5)])
def test_split_rows_variant_for_65_to_128_columns(emu, modes, scheme, P_per, L):
    """solve_problem_rows<64, true>: two threads per MV on either side of a block boundary (round 4, models of 65 ... 128 MVs) against the oracle
    and the LDS variant -- plain and with bootstrap counts."""
    X, model = _wide_model(P_per, L, modes, scheme, seed=21)
    assert 64 < X.shape[1] <= 128
    e = run_emu(emu, X, model, rows=True, split=True)
    check(e, orc.fit(X, model), "split rows %s/%s" % (modes, scheme))
    base = run_emu(emu, X, model)
    assert e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-10, 1e-13)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, X.shape[0], X.shape[0])
    counts = np.bincount(idx, minlength=X.shape[0])
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_emu(emu, X, model, counts=counts, shift=shift, rows=True, split=True)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(X.shape[0]))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def test_split_rows_variant_needs_a_block_boundary(emu):
    """One block wider than 64 MVs on either side of every boundary: the emulation entry (like the host's rows_split_block test) declines."""
    C = np.array([[0, 0], [1, 0]], dtype=float)
    X, blocks = orc.synth(300, C, 35, seed=3)                     # 70 MVs in two blocks of 35: boundary at 35 -> 35 | 35 fits
    model = orc.Model(blocks, C, ["A", "A"], "centroid", True)
    check(run_emu(emu, X, model, rows=True, split=True), orc.fit(X, model))
    blocks2 = [list(blocks[0]) + list(blocks[1][:31]), list(blocks[1][31:])]          # 66 | 4: no boundary leaves both sides <= 64
    model2 = orc.Model(blocks2, C, ["A", "A"], "centroid", True)
    with pytest.raises(AssertionError):
        run_emu(emu, X, model2, rows=True, split=True)
