"""CPU check of the one-launch non-metric (Scale.NUM / RAW) wave solver source (csrc/solver_wave16.h, template flag NM; round 6) through the std::thread
emulation build in tests/hostemu/: 64 emulated lanes run prepare + every step + finish of a problem; a step stops on the quadratic upper bound of the
reference's score criterion (weights.py:120) and continues speculatively otherwise, leaving the score map of every step it continued behind.  Checked here:
  * the record against the data-level oracle (pinned on the reference) for every Mode x Scheme, iteration counts included;
  * the stored maps: played back on the observations in NumPy they give the oracle's own sequence of criterion values -- every continued step at or above the
    tolerance (what the GPU's verification pass establishes), the bound of the last step below it;
  * a bootstrap-weighted problem (row multiplicities) against the oracle's replicate;
  * the replay seam (force_T): stopping behind exactly j steps reproduces the oracle with max_iter-free early stop, i.e. the weights after j steps."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, case_modes, effect_pairs, packed_scatter
from test_solver_hostemu import dense_from_packed
from test_oracle_golden import RUSSA_BLOCKS, RUSSA_C, russa_inputs

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


def run_nmwave(lib, X, model, counts=None, shift=None, force_T=0):
    order = model.mv_order
    Xdev = np.ascontiguousarray(X[:, order])
    n, P = Xdev.shape
    L = model.L
    Mp, shift, PA = packed_scatter(Xdev, counts, shift)
    shift = np.ascontiguousarray(shift, dtype=np.float64)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs] + [0], dtype=np.int32)
    et = np.array([p[1] for p in pairs] + [0], dtype=np.int32)
    ne = len(pairs)
    row = np.full(2 * P + L + 2 * ne + 2, np.nan)
    iters, status, steps = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int(-1)
    Md = np.ascontiguousarray(dense_from_packed(Mp, PA, P))
    maps = np.full((model.max_iter + 2, P + L + 1), np.nan)       # per step: c_p | k_l | the quadratic bound of the step
    rc = lib.hostemu_solve_nmwave16(P, L, PA, SCHEME_ID[model.scheme], 1, model.max_iter, ctypes.c_double(model.tol), _ptr(boff, ctypes.c_int), _ptr(C, ctypes.c_ubyte),
                                    _ptr(mode, ctypes.c_int), _ptr(shift), ne, _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(Md), _ptr(row), ctypes.byref(iters),
                                    ctypes.byref(status), _ptr(maps), int(force_T), ctypes.byref(steps))
    if rc:
        return None
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    assert row[-2] == status.value and row[-1] == iters.value and steps.value == iters.value
    # the criterion of every step the problem continued behind, from the stored maps, on the observations (what nm_conv_dense_kernel accumulates)
    Xs = Xdev - shift
    cw = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    lv_of = np.repeat(np.arange(L), np.diff(boff))
    onehot = (lv_of[:, None] == np.arange(L)[None, :]).astype(float)
    T = iters.value
    assert not np.isnan(maps[:T]).any() and np.isnan(maps[T:]).all()          # maps of steps 0 .. T - 1, nothing else
    ys = [(Xs * maps[j, :P]) @ onehot + maps[j, P:P + L] for j in range(T)]
    conv = [float((((np.abs(ys[j - 1]) - np.abs(ys[j])) ** 2).sum(axis=1) * cw).sum()) for j in range(1, T)]
    for j in range(1, T):                                                        # the value stored beside map j bounds the criterion of step j from above
        assert conv[j - 1] <= maps[j, P + L] * (1 + 1e-9), (j, conv[j - 1], maps[j, P + L])
    return dict(weights=row[:P][inv], r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne],
                loadings=row[P + L + 2 * ne:2 * P + L + 2 * ne][inv], iterations=T, status=status.value, row=row, conv=conv, scores=ys)


def _check(e, o, tag):
    assert e["status"] == 0, tag
    assert e["iterations"] == o["iterations"], tag
    assert_close(e["weights"], o["weights"], RTOL, 1e-12, what=tag)
    assert_close(e["r2"], o["r2"], RTOL, 1e-12, what=tag)
    assert_close(e["loadings"], o["loadings"], RTOL, 1e-12, what=tag)
    assert all(c >= 1e-7 for c in e["conv"]), (tag, e["conv"])               # every step it continued behind: the reference continues there too


@pytest.mark.parametrize("modes", ["A", "B", "M"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_nmwave_russa_vs_oracle(emu, modes, scheme):
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, case_modes(modes, 3, "ABA"), scheme, True, tol=1e-7, scales=["NUM"] * X.shape[1])
    e = run_nmwave(emu, X, model)
    assert e is not None
    _check(e, orc.fit(X, model), "russa %s %s" % (modes, scheme))


@pytest.mark.parametrize("scheme,modes,L,per", [("path", "AAAAAA", 6, 10), ("centroid", "ABABAB", 6, 5), ("factorial", "AAAAAAAAAAAA", 12, 5), ("path", "ABBAABBAAB", 10, 6),
                                                ("centroid", "A" * 20, 20, 3), ("path", "A" * 24, 24, 2)])
def test_nmwave_synthetic_vs_oracle_maps_and_bootstrap_replicate(emu, scheme, modes, L, per):
    C = orc.satisfaction_C() if L == 6 else orc.chain_C(L)
    X, blocks = orc.synth(700, C, per, seed=31)
    model = orc.Model(blocks, C, modes, scheme, True, tol=1e-7, scales=["NUM"] * X.shape[1])
    e = run_nmwave(emu, X, model)
    assert e is not None
    o = orc.fit(X, model)
    _check(e, o, "synthetic %s %s" % (scheme, modes))
    # the scores of the last stored map are the oracle's scores one step before the end; from step 1 on they are population-standardised
    for y in e["scores"][1:]:
        assert_close(y.std(axis=0), np.ones(L), 1e-9)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, X.shape[0], X.shape[0])
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_nmwave(emu, X, model, counts=np.bincount(idx, minlength=X.shape[0]), shift=shift)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(X.shape[0]))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def test_nmwave_replay_stops_where_it_is_told(emu):
    """force_T = j (the replay of a replicate whose stop the verification moved): the record of exactly j steps == the oracle run with a tolerance
    that lets it stop at step j."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(500, C, 4, seed=2)
    base = orc.Model(blocks, C, "AAAAAA", "path", True, tol=1e-9, scales=["NUM"] * X.shape[1])
    full = run_nmwave(emu, X, base)
    assert full["iterations"] >= 3
    for j in range(1, full["iterations"]):
        e = run_nmwave(emu, X, base, force_T=j)
        assert e["iterations"] == j and e["status"] == 0
        loose = orc.Model(blocks, C, "AAAAAA", "path", True, tol=full["conv"][j - 1] * 1.0001 if j - 1 < len(full["conv"]) else 1e-9, scales=["NUM"] * X.shape[1])
        o = orc.fit(X, loose)
        assert o["iterations"] == j
        assert_close(e["weights"], o["weights"], RTOL, 1e-12)


def test_nmwave_max_iter_and_not_converged(emu):
    C = orc.satisfaction_C()
    X, blocks = orc.synth(400, C, 3, seed=9)
    model = orc.Model(blocks, C, "AAAAAA", "centroid", True, tol=1e-30, max_iter=4, scales=["NUM"] * X.shape[1])
    e = run_nmwave(emu, X, model)
    assert e["status"] == 1 and e["iterations"] == 5                          # weights.py:183-186: stops at iteration > max_iter and raises
    with pytest.raises(orc.NotConverged):
        orc.fit(X, model)
