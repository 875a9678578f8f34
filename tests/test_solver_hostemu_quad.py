"""CPU check of the quad solver source (csrc/solver_quad.h: the wave solver's lane roles on four waves per problem -- metric Mode-A models of
65 ... 128 MVs and at most 16 LVs) through the std::thread emulation build in tests/hostemu/ (256 emulated threads, butterfly sums per emulated
wave, ballots per wave): against the data-level oracle, bootstrap replicates of the oracle and the split rows / LDS variants of the same solver.
Tolerance vs the oracle: 1e-9 relative (fp64 both sides; the formulations differ); vs the split rows variant 1e-10."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, effect_pairs, packed_scatter
from test_solver_hostemu import EMU, HERE, RTOL, SCHEME_ID, _ptr, _wide_model, dense_from_packed, run_emu
from test_solver_hostemu_wave import check


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    return ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))


def run_quad(lib, X, model, counts=None, shift=None, entry="hostemu_solve_quad"):
    """solve_problem_quad<16> on the dense upper-triangular moment matrix of the device-ordered columns; returns the record pieces in DATA
    column order, or None when the model is outside the quad solver's class."""
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
    rc = getattr(lib, entry)(P, L, PA, SCHEME_ID[model.scheme], int(model.scaled), model.max_iter, ctypes.c_double(model.tol),
                                _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte), _ptr(mode, ctypes.c_int), _ptr(shift), ne,
                                _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(Md), _ptr(row), ctypes.byref(iters), ctypes.byref(status))
    if rc:
        return None
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    assert row[-2] == status.value and row[-1] == iters.value
    assert status.value != 0 or not np.isnan(row).any()         # every entry of the record written
    return dict(weights=row[:P][inv], r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne],
                loadings=row[P + L + 2 * ne:2 * P + L + 2 * ne][inv], iterations=iters.value, status=status.value, row=row, pairs=pairs)


@pytest.mark.parametrize("scheme,scaled,P_per,L", [("path", True, 10, 12), ("factorial", True, 10, 12), ("centroid", False, 10, 12), ("centroid", True, 16, 8),
                                                   ("path", False, 8, 16), ("factorial", False, 13, 5), ("path", True, 7, 10)])
def test_quad_vs_oracle_split_rows_variant_and_bootstrap_replicate(emu, scheme, scaled, P_per, L):
    X, model = _wide_model(P_per, L, "A", scheme, seed=21)
    model = orc.Model(model.blocks, model.C, "A" * L, scheme, scaled)
    assert 64 < X.shape[1] <= 128
    e = run_quad(emu, X, model)
    assert e is not None
    check(e, orc.fit(X, model), "quad %s/%d %dx%d" % (scheme, scaled, P_per, L))
    base = run_emu(emu, X, model, rows=True, split=True)
    assert e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-10, 1e-13)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, X.shape[0], X.shape[0])
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_quad(emu, X, model, counts=np.bincount(idx, minlength=X.shape[0]), shift=shift)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(X.shape[0]))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def _shaped(C, sizes, seed, N=500):
    rs = np.random.RandomState(seed)
    L = C.shape[0]
    eta = np.zeros((N, L))
    for j in range(L):
        eta[:, j] = 0.4 * eta[:, C[j] == 1].sum(axis=1) + rs.standard_normal(N)
    cols, blocks, c0 = [], [], 0
    for j, k in enumerate(sizes):
        lam = np.linspace(0.5, 0.9, k)
        cols.append(eta[:, [j]] * lam + 0.6 * rs.standard_normal((N, k)))
        blocks.append(np.arange(c0, c0 + k)); c0 += k
    return np.column_stack(cols) + rs.standard_normal(c0), blocks


def test_quad_model_shapes(emu):
    """Ragged blocks (1 ... 64 MVs), L = 2 ... 16, 128 MVs exactly, a one-LV side, five and more predecessors (Cholesky in the staging area):
    every lane role at its limits."""
    cases = []
    cases.append((orc.chain_C(2), [64, 64]))                                  # 128 MVs: every MV thread live, one LV per side
    cases.append((orc.chain_C(2), [3, 64]))
    cases.append((orc.chain_C(16), [8] * 16))                                 # 128 MVs, 16 LVs: every pair thread live
    cases.append((orc.chain_C(16), [1] * 10 + [9, 9, 9, 9, 9, 20]))
    cases.append((orc.chain_C(5), [1, 17, 40, 9, 5]))
    cases.append((orc.chain_C(3), [60, 1, 10]))
    C = np.zeros((9, 9), dtype=np.int64)
    for i in range(1, 9):
        for j in range(max(0, i - 6), i):
            C[i, j] = 1                                                       # up to 6 predecessors
    cases.append((C, [9, 8, 7, 12, 5, 6, 11, 10, 4]))
    for C, sizes in cases:
        L = C.shape[0]
        X, blocks = _shaped(C, sizes, seed=4)
        for scheme in ("centroid", "factorial", "path"):
            model = orc.Model(blocks, C, "A" * L, scheme, True)
            e = run_quad(emu, X, model)
            assert e is not None, sizes
            check(e, orc.fit(X, model), "L=%d %s %s" % (L, sizes, scheme))


def test_quad_declines_models_outside_its_class(emu):
    X, blocks = _shaped(orc.chain_C(2), [66, 4], seed=2)                      # no block boundary leaves both sides <= 64
    assert run_quad(emu, X, orc.Model(blocks, orc.chain_C(2), "AA", "centroid", True)) is None
    X, blocks = _shaped(orc.chain_C(3), [30, 20, 30], seed=2)                 # Mode-B block
    assert run_quad(emu, X, orc.Model(blocks, orc.chain_C(3), "ABA", "centroid", True)) is None
    X, blocks = _shaped(orc.chain_C(3), [20, 20, 20], seed=2)                 # 60 MVs: the wave solver's class
    assert run_quad(emu, X, orc.Model(blocks, orc.chain_C(3), "AAA", "centroid", True)) is None
    X, blocks = _shaped(orc.chain_C(17), [5] * 17, seed=2)                    # 17 LVs
    assert run_quad(emu, X, orc.Model(blocks, orc.chain_C(17), "A" * 17, "centroid", True)) is None


def test_quad_status_codes_sign_rule_and_rank_deficient_predecessors(emu):
    C = orc.chain_C(4)
    X, blocks = _shaped(C, [20, 25, 15, 30], seed=9)
    tight = orc.Model(blocks, C, "AAAA", "centroid", True, max_iter=2, tol=1e-14)
    e = run_quad(emu, X, tight)
    assert e["status"] == 1 and e["iterations"] == 3           # counter runs to max_iter+1 before giving up (weights.py:181-186)
    Xc = X.copy(); Xc[:, blocks[2][1]] = 3.0                    # a constant MV
    cm = orc.Model(blocks, C, "AAAA", "centroid", True)                                     # a constant MV: the reference centres it to zeros -- weight 0, loading 0, the estimate counts (solver_core.h treated_sd)
    e, r = run_quad(emu, Xc, cm), orc.fit(Xc, cm)
    assert e["status"] == 0 and e["iterations"] == r["iterations"] and e["loadings"][blocks[2][1]] == 0.0
    assert_close(e["weights"], r["weights"], 1e-9, 1e-12); assert_close(e["loadings"], r["loadings"], 1e-9, 1e-12)
    # sign rule: most MVs of the first side's blocks negated -> the votes of waves 0 / 1 flip those LVs (weights.py:62-64)
    Xn = X.copy()
    Xn[:, blocks[0][:15]] *= -1.0
    Xn[:, blocks[3][:20]] *= -1.0
    for scheme in ("centroid", "path"):
        model = orc.Model(blocks, C, "AAAA", scheme, True)
        check(run_quad(emu, Xn, model), orc.fit(Xn, model), "sign " + scheme)
    # exactly collinear predecessor scores (a cloned LV block): the minimum-norm coefficients of the reference's pinv
    C5 = np.zeros((5, 5), dtype=np.int64)
    C5[2, 0] = C5[2, 1] = C5[3, 2] = C5[4, 0] = C5[4, 1] = C5[4, 3] = 1
    Xb, blocks_b = _shaped(C5, [15, 15, 14, 16, 12], seed=11)
    Xb[:, blocks_b[1]] = Xb[:, blocks_b[0]]                     # LV 1 == LV 0
    for scheme in ("path", "centroid"):
        model = orc.Model(blocks_b, C5, "AAAAA", scheme, True)
        e = run_quad(emu, Xb, model)
        r = orc.fit(Xb, model)
        assert e["status"] == 0 and e["iterations"] == r["iterations"]
        assert_close(e["weights"], r["weights"], RTOL)
        assert_close(e["direct"], r["direct"], RTOL, 1e-12)
        assert_close(e["r2"], r["r2"], RTOL, 1e-12)


def test_quad_thread_sanitizer_clean():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_tsan.so"])
    code = ("import sys; sys.path[:0]=[%r,%r]; import ctypes, numpy as np; import plspm_oracle as orc; import test_solver_hostemu as t; import test_solver_hostemu_quad as q;"
            "lib=ctypes.CDLL(%r);"
            "Xw,mw=t._wide_model(10, 12, 'A', 'path', 21); assert q.run_quad(lib, Xw, mw) is not None;"
            "Xw,mw=t._wide_model(8, 16, 'A', 'centroid', 5); assert q.run_quad(lib, Xw, orc.Model(mw.blocks, mw.C, 'A' * 16, 'centroid', True)) is not None;"
            "print('tsan-run-done')") % (HERE, os.path.join(os.path.dirname(HERE), "oracle"), os.path.join(EMU, "libplspm_hostemu_tsan.so"))
    tsan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", OPENBLAS_NUM_THREADS="1")      # (NumPy's BLAS pool is not under test)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "tsan-run-done" in r.stdout, r.stderr[-2000:]
    assert "data race" not in r.stderr, r.stderr[-4000:]
