"""CPU check of the NON-METRIC (Scale.NUM / RAW) device solver source (csrc/solver_core.h: nm_prepare / nm_step / nm_finish)
through the std::thread emulation build.  The streaming convergence pass the GPU runs as a kernel (sum over observations of
(|y_old| - |y_new|)^2, reference weights.py:120) is played here by NumPy from the two score maps the step leaves in the
state, split into several partial sums like the kernel's workgroups.  Checked against the data-level oracle (pinned on the
reference) for every Mode x Scheme, a bootstrap-weighted problem, and the iteration counts."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, case_modes, effect_pairs, load, packed_scatter
from test_oracle_golden import RUSSA_BLOCKS, RUSSA_C, russa_inputs

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL = 1e-9


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    lib = ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))
    lib.hostemu_cov_doubles.restype = ctypes.c_long
    lib.hostemu_nm_state_doubles.restype = ctypes.c_long
    return lib


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def run_nm_emu(lib, X, model, counts=None, shift=None, nthreads=4, nparts=5):
    order = model.mv_order
    Xdev = np.ascontiguousarray(X[:, order])
    n, P = Xdev.shape
    L = model.L
    Mp, shift, PA = packed_scatter(Xdev, counts, shift)
    shift = np.ascontiguousarray(shift, dtype=np.float64)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    n_chol = int(sum(2 * (boff[l + 1] - boff[l]) ** 2 for l in range(L) if mode[l] == 1))
    S = np.zeros(lib.hostemu_cov_doubles(P))
    state = np.zeros(lib.hostemu_nm_state_doubles(P, L, n_chol))
    args = (P, L, PA, SCHEME_ID[model.scheme], model.max_iter, ctypes.c_double(model.tol), _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte),
            _ptr(mode, ctypes.c_int), _ptr(shift))
    lib.hostemu_nm_prepare(*args, _ptr(Mp), nthreads, _ptr(S), _ptr(state))
    o = 8
    sl = {}
    for name, size in (("a_old", P), ("a_new", P), ("c_old", P), ("c_new", P), ("k_old", L), ("k_new", L), ("sd", P), ("mu", P)):
        sl[name] = slice(o, o + size); o += size
    Xs = Xdev - shift
    cw = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    lv_of = np.repeat(np.arange(L), np.diff(boff))
    onehot = (lv_of[:, None] == np.arange(L)[None, :]).astype(float)
    partial = np.zeros(nparts)

    def stop_rule_terms():
        y_old = (Xs * state[sl["c_old"]]) @ onehot + state[sl["k_old"]]
        y_new = (Xs * state[sl["c_new"]]) @ onehot + state[sl["k_new"]]
        return ((np.abs(y_old) - np.abs(y_new)) ** 2).sum(axis=1) * cw                 # what the nm_conv kernel accumulates
    early = False
    for _ in range(model.max_iter + 5):
        active = lib.hostemu_nm_step(*args, nthreads, _ptr(S), _ptr(state), _ptr(partial), nparts)
        if not active:
            # the step either decided on the exact value of the previous pass or stopped on its own upper bound (nm_step: the
            # quadratic form on the correlation matrix): whichever it was, state[4] must not be below the exact value of the step
            # the score maps now describe
            exact = float(stop_rule_terms().sum())
            assert exact <= state[4] * (1.0 + 1e-9) + 1e-18, (exact, state[4])
            early = abs(state[4] - partial.sum()) > 1e-15 * max(1.0, abs(state[4])) and state[1] == 0.0 and state[4] < model.tol
            if early:
                assert state[4] <= 50.0 * max(exact, 1e-300) + 1e-12                    # ... and the bound is tight where it is used
            break
        d = stop_rule_terms()
        partial = np.array([chunk.sum() for chunk in np.array_split(d, nparts)])
    run_nm_emu.early_stops = getattr(run_nm_emu, "early_stops", 0) + int(early)
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs], dtype=np.int32); et = np.array([p[1] for p in pairs], dtype=np.int32)
    ne = len(pairs)
    row = np.zeros(2 * P + L + 2 * ne + 2); cl = np.zeros((P, L)); pc = np.zeros((L, L)); lc = np.zeros((L, L))
    ind = np.zeros(max(ne, 1)); sw = np.zeros(P); sc = np.zeros(L); cov = np.zeros((P, P)); mean = np.zeros(P)
    iters = ctypes.c_int(0); status = ctypes.c_int(-1)
    lib.hostemu_nm_finish(*args, ne, _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), nthreads, _ptr(S), _ptr(state), _ptr(row), _ptr(cl), _ptr(pc),
                          _ptr(lc), _ptr(ind), _ptr(sw), _ptr(sc), _ptr(cov), _ptr(mean), ctypes.byref(iters), ctypes.byref(status))
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    scores = (Xs * sw) @ onehot + sc
    return dict(weights=row[:P][inv], r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne],
                loadings=row[P + L + 2 * ne:2 * P + L + 2 * ne][inv], crossloadings=cl[inv], path_coef=pc, indirect=ind[:ne],
                iterations=iters.value, status=status.value, pairs=pairs, scores=scores, cov=cov)


def check_nm(e, r, tag=""):
    assert e["status"] == 0, tag
    assert e["iterations"] == r["iterations"], "%s: iterations %d vs %d" % (tag, e["iterations"], r["iterations"])
    assert_close(e["weights"], r["weights"], RTOL, what=tag + " weights")
    assert_close(e["loadings"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(e["crossloadings"], r["crossloadings"], RTOL, 1e-13, what=tag + " crossloadings")
    assert_close(e["path_coef"], r["path_coef"], RTOL, 1e-13)
    assert_close(e["r2"], r["r2"], RTOL, 1e-13)
    assert e["pairs"] == r["effect_pairs"]
    assert_close(e["total"], r["total"], RTOL, 1e-13)
    assert_close(e["scores"], r["scores"], 1e-8, 1e-10, what=tag + " scores")


@pytest.mark.parametrize("modes", ["AAA", "BBB", "ABA"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_russa_nonmetric_all_cases(emu, modes, scheme):
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=["NUM"] * 9)
    check_nm(run_nm_emu(emu, X, model), orc.fit(X, model), modes + "/" + scheme)


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial"), ("M", "centroid")])
def test_synth2000_nonmetric(emu, modes, scheme):
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, True, tol=1e-7, scales=["NUM"] * 60)
    check_nm(run_nm_emu(emu, X, model, nparts=7), orc.fit(X, model))


@pytest.mark.parametrize("tag", ["AAA_centroid_NUM", "ABA_path_NUM"])
def test_nonmetric_bootstrap_counts_vs_reference_rows(emu, tag):
    g = load("g8_nonmetric_russa")
    X = russa_inputs()
    modes, scheme, _ = tag.split("_")
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=["NUM"] * 9)
    shift = X[:, model.mv_order].mean(axis=0)
    for idx, ref_row, it in zip(g["idx"], g[tag + "/boot_rows"], g[tag + "/boot_iters"]):
        counts = np.bincount(idx, minlength=47)
        e = run_nm_emu(emu, X, model, counts=counts, shift=shift)
        assert e["status"] == 0 and e["iterations"] == int(it)
        mine = np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"]))
        assert_close(mine, ref_row, RTOL, 1e-12, what=tag)


def test_nonmetric_not_converged(emu):
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, "AAA", "centroid", True, max_iter=2, tol=1e-30, scales=["NUM"] * 9)
    e = run_nm_emu(emu, X, model)
    assert e["status"] == 1 and e["iterations"] == 3


def test_the_stop_rule_bound_is_exercised(emu):
    """nm_step's upper bound on the stop-rule value (no pass over the observations in the converged iteration) must actually end
    problems in this suite -- with the reference's iteration counts, which every test above asserts."""
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 4, seed=21)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True, tol=1e-6, scales=["NUM"] * 24)
    before = getattr(run_nm_emu, "early_stops", 0)
    e = run_nm_emu(emu, X, model)
    check_nm(e, orc.fit(X, model), "bound")
    assert getattr(run_nm_emu, "early_stops", 0) == before + 1
