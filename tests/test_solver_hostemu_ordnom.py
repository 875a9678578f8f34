"""CPU check of the categorical (Scale.ORD / NOM) non-metric device solver source (csrc/solver_nmg.h) through the std::thread
emulation build: indicator-column ("aug") formulation on raw second moments vs the data-level oracle pinned on the reference.
The streaming convergence pass is played by NumPy from the two score maps over the aug columns, as in the NUM/RAW emulation test."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, effect_pairs, load, packed_scatter
from test_oracle_golden import (LIKERT_BLOCKS, LIKERT_C, LIKERT_CASES, RUSSA_C, RUSSA_CAT_BLOCKS, RUSSA_CAT_SCALES, russa_cat_inputs)

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
KIND = {"NUM": 0, "RAW": 0, "ORD": 1, "NOM": 2}
RTOL = 1e-9


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    lib = ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))
    lib.hostemu_cov_doubles.restype = ctypes.c_long
    lib.hostemu_nmg_state_doubles.restype = ctypes.c_long
    return lib


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def build_aug(X, model):
    """Device columns: MVs grouped by LV (path order); NUM/RAW -> the raw column, ORD/NOM -> indicator columns of the rank codes."""
    cols, mv_off, mv_kind, lmv_off, boff, mv_data_col = [], [0], [], [0], [0], []
    for b in model.blocks:
        for p in b:
            kind = KIND[model.scales[p]]
            if kind == 0:
                cols.append(X[:, p][:, None])
            else:
                cols.append(orc.dummy_matrix(orc.rank_column(X[:, p])))
            mv_off.append(mv_off[-1] + cols[-1].shape[1]); mv_kind.append(kind); mv_data_col.append(p)
        lmv_off.append(len(mv_kind)); boff.append(mv_off[-1])
    return (np.ascontiguousarray(np.column_stack(cols)), np.array(mv_off, dtype=np.int32), np.array(mv_kind, dtype=np.int32),
            np.array(lmv_off, dtype=np.int32), np.array(boff, dtype=np.int32), np.array(mv_data_col))


def run_cat_emu(lib, X, model, counts=None, nthreads=4, nparts=3):
    Xaug, mv_off, mv_kind, lmv_off, boff, mv_data_col = build_aug(X, model)
    n, Q = Xaug.shape
    Pm, L = len(mv_kind), model.L
    Mp, _, PA = packed_scatter(Xaug, counts, np.zeros(Q))
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    S = np.zeros(lib.hostemu_cov_doubles(Q))
    state = np.zeros(lib.hostemu_nmg_state_doubles(Q, Pm, L, _ptr(mv_off, ctypes.c_int), _ptr(mv_kind, ctypes.c_int), _ptr(lmv_off, ctypes.c_int)))
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs], dtype=np.int32); et = np.array([p[1] for p in pairs], dtype=np.int32)
    ne = len(pairs)
    row = np.zeros(2 * Pm + L + 2 * ne + 2); cl = np.zeros((Pm, L)); pc = np.zeros((L, L)); sw = np.zeros(Q); sc = np.zeros(L)
    cov = np.zeros((Pm, Pm)); iters = ctypes.c_int(0); status = ctypes.c_int(-1)
    partial = np.zeros(nparts)

    def call(op):
        return lib.hostemu_nmg(op, Q, Pm, L, PA, SCHEME_ID[model.scheme], model.max_iter, ctypes.c_double(model.tol), _ptr(boff, ctypes.c_int),
                               _ptr(C, ctypes.c_ubyte), _ptr(mode, ctypes.c_int), _ptr(mv_off, ctypes.c_int), _ptr(mv_kind, ctypes.c_int),
                               _ptr(lmv_off, ctypes.c_int), _ptr(Mp), nthreads, _ptr(S), _ptr(state), _ptr(partial), nparts, ne,
                               _ptr(ef, ctypes.c_int), _ptr(et, ctypes.c_int), _ptr(row), _ptr(cl), _ptr(pc), _ptr(sw), _ptr(sc), _ptr(cov),
                               ctypes.byref(iters), ctypes.byref(status))
    call(0)
    o = 8 + 2 * Q
    c_old, c_new = slice(o, o + Q), slice(o + Q, o + 2 * Q)
    k_old, k_new = slice(o + 2 * Q, o + 2 * Q + L), slice(o + 2 * Q + L, o + 2 * Q + 2 * L)
    cw = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    lv_of = np.repeat(np.arange(L), np.diff(boff))
    onehot = (lv_of[:, None] == np.arange(L)[None, :]).astype(float)
    for _ in range(model.max_iter + 5):
        if not call(1):
            break
        y_old = (Xaug * state[c_old]) @ onehot + state[k_old]
        y_new = (Xaug * state[c_new]) @ onehot + state[k_new]
        d = ((np.abs(y_old) - np.abs(y_new)) ** 2).sum(axis=1) * cw
        partial[:] = [chunk.sum() for chunk in np.array_split(d, nparts)]
    call(2)
    inv = np.empty(Pm, dtype=np.int64); inv[mv_data_col] = np.arange(Pm)
    scores = (Xaug * sw) @ onehot + sc
    return dict(weights=row[:Pm][inv], r2=row[Pm:Pm + L], total=row[Pm + L:Pm + L + ne], loadings=row[Pm + L + 2 * ne:2 * Pm + L + 2 * ne][inv],
                crossloadings=cl[inv], path_coef=pc, iterations=iters.value, status=status.value, scores=scores, cov=cov)


def check(e, r, tag=""):
    assert e["status"] == 0, tag
    assert e["iterations"] == r["iterations"], "%s: iterations %d vs %d" % (tag, e["iterations"], r["iterations"])
    assert_close(e["weights"], r["weights"], RTOL, what=tag + " weights")
    assert_close(e["loadings"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(e["crossloadings"], r["crossloadings"], RTOL, 1e-12, what=tag)
    assert_close(e["path_coef"], r["path_coef"], RTOL, 1e-12)
    assert_close(e["r2"], r["r2"], RTOL, 1e-12)
    assert_close(e["scores"], r["scores"], 1e-8, 1e-10, what=tag + " scores")


@pytest.mark.parametrize("modes", ["AAA", "BBB"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_russa_categorical(emu, modes, scheme):
    X = russa_cat_inputs()
    model = orc.Model(RUSSA_CAT_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=RUSSA_CAT_SCALES)
    check(run_cat_emu(emu, X, model), orc.fit(X, model), modes + "/" + scheme)


@pytest.mark.parametrize("tag", ["ordA", "ordB", "mixM"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_likert(emu, tag, scheme):
    g = load("g11_ordnom")
    modes, scales = LIKERT_CASES[tag]
    model = orc.Model(LIKERT_BLOCKS, LIKERT_C, modes, scheme, True, tol=1e-7, scales=scales)
    check(run_cat_emu(emu, g["likert"], model), orc.fit(g["likert"], model), tag + "/" + scheme)


def test_all_numeric_model_agrees_with_the_num_solver_path(emu):
    """KIND_NUM everywhere: the general solver must reproduce the NUM/RAW results (russa, golden g8)."""
    from test_oracle_golden import RUSSA_BLOCKS, russa_inputs
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, "ABA", "path", True, tol=1e-7, scales=["NUM"] * 9)
    check(run_cat_emu(emu, X, model), orc.fit(X, model))


@pytest.mark.parametrize("modes,scheme", [("AAA", "centroid"), ("BBB", "path")])
def test_categorical_bootstrap_counts_with_absent_categories(emu, modes, scheme):
    """A resample misses categories: their indicator columns are all-zero, rank codes are taken over the present ones
    (util.rank on the resampled column) -- weighted moments must reproduce the oracle run on data[idx]."""
    X = russa_cat_inputs()
    model = orc.Model(RUSSA_CAT_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=RUSSA_CAT_SCALES)
    rs = np.random.RandomState(31)
    corr = orc.correction(47)
    done = 0
    for _ in range(12):
        idx = rs.randint(47, size=47)
        try:
            mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        except Exception:
            continue
        if not np.all(np.isfinite(mine)):
            continue
        e = run_cat_emu(emu, X, model, counts=np.bincount(idx, minlength=47))
        assert e["status"] == 0 and e["iterations"] == its
        got = np.concatenate((e["weights"], e["r2"], e["total"]))
        ne = len(e["total"])
        assert_close(got, np.concatenate((mine[:9], mine[9:12], mine[12:12 + ne])), 1e-8, 1e-11)
        assert_close(e["loadings"], mine[-9:], 1e-8, 1e-11)
        done += 1
    assert done >= 6
