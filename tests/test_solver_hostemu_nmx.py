"""CPU check of the non-metric missing-data solver source (csrc/solver_nmx.h) through the std::thread emulation build: complete
rows on the moments, incomplete rows explicitly, the streaming convergence pass played by NumPy on the uploaded matrix (whose
incomplete rows are zero rows) -- against the NaN-aware oracle pinned on the reference (golden g13, russa.missing CSV)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, effect_pairs, load, packed_scatter
from test_oracle_golden import RUSSA_C, RUSSA_M_BLOCKS, russa_missing_matrix

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
I32 = ctypes.c_int
RTOL = 1e-9


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    lib = ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))
    lib.hostemu_cov_doubles.restype = ctypes.c_long
    lib.hostemu_nm_state_doubles.restype = ctypes.c_long
    lib.hostemu_nmx_state_doubles.restype = ctypes.c_long
    return lib


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def split_incomplete(Xdev):
    """What the host does before the upload: NaN cells -> column mean, incomplete rows listed with their masks."""
    miss = np.isnan(Xdev)
    rows = np.flatnonzero(miss.any(axis=1))
    filled = np.where(miss, np.nanmean(Xdev, axis=0), Xdev)
    return filled, rows, (~miss[rows]).astype(np.float64)


def run_nmx_emu(lib, Xnan, model, counts=None, nthreads=4, nparts=3):
    order = model.mv_order
    Xdev = np.ascontiguousarray(Xnan[:, order])
    n, P = Xdev.shape
    L = model.L
    filled, rows, Mk = split_incomplete(Xdev)
    K = len(rows)
    shift = filled.mean(axis=0)
    stored = filled - shift
    Xk = np.ascontiguousarray(stored[rows] * Mk)
    Xup = stored.copy(); Xup[rows] = 0.0                   # the uploaded matrix: incomplete rows zeroed, ones column too
    ones = np.ones(n); ones[rows] = 0.0
    cw = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    PA = ((P + 1 + 31) // 32) * 32
    Xa = np.zeros((n, PA)); Xa[:, :P] = Xup; Xa[:, P] = ones
    M = (Xa * cw[:, None]).T @ Xa
    from helpers import packed_index_np
    T = PA // 16
    Mp = np.zeros(T * (T + 1) // 2 * 256)
    pp, qq = np.meshgrid(np.arange(PA), np.arange(PA), indexing="ij")
    Mp[packed_index_np(T, pp.ravel(), qq.ravel())] = M.ravel()
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    C = np.ascontiguousarray(model.C.astype(np.uint8))
    mode = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    n_chol = int(sum(2 * (boff[l + 1] - boff[l]) ** 2 for l in range(L) if mode[l] == 1))
    S = np.zeros(lib.hostemu_cov_doubles(P))
    head = lib.hostemu_nm_state_doubles(P, L, n_chol)
    state = np.zeros(lib.hostemu_nmx_state_doubles(P, L, n_chol, K))
    state[head:head + K] = cw[rows]
    pairs = effect_pairs(model.C)
    ef = np.array([p[0] for p in pairs], dtype=np.int32); et = np.array([p[1] for p in pairs], dtype=np.int32)
    ne = len(pairs)
    row = np.zeros(2 * P + L + 2 * ne + 2); cl = np.zeros((P, L)); pc = np.zeros((L, L)); sw = np.zeros(P); sc = np.zeros(L); cov = np.zeros((P, P))
    iters = I32(0); status = I32(-1)
    partial = np.zeros(nparts)
    Xk_c, Mk_c = np.ascontiguousarray(Xk), np.ascontiguousarray(Mk)
    if K == 0:
        Xk_c, Mk_c = np.zeros(1), np.zeros(1)

    def call(op):
        return lib.hostemu_nmx(op, int(all(k == "RAW" for k in model.scales)), P, L, PA, SCHEME_ID[model.scheme], model.max_iter, ctypes.c_double(model.tol), _ptr(boff, I32), _ptr(C, ctypes.c_ubyte),
                               _ptr(mode, I32), K, _ptr(Xk_c), _ptr(Mk_c), _ptr(Mp), nthreads, _ptr(S), _ptr(state), _ptr(partial), nparts, ne,
                               _ptr(ef, I32), _ptr(et, I32), _ptr(row), _ptr(cl), _ptr(pc), _ptr(sw), _ptr(sc), _ptr(cov), ctypes.byref(iters), ctypes.byref(status))
    call(0)
    o = 8 + 2 * P
    lv_of = np.repeat(np.arange(L), np.diff(boff))
    onehot = (lv_of[:, None] == np.arange(L)[None, :]).astype(float)
    for _ in range(model.max_iter + 5):
        if not call(1):
            break
        yo = (Xup * state[o:o + P]) @ onehot + state[o + 2 * P:o + 2 * P + L]            # the kernel adds k_l to EVERY streamed row
        yn = (Xup * state[o + P:o + 2 * P]) @ onehot + state[o + 2 * P + L:o + 2 * P + 2 * L]
        d = ((np.abs(yo) - np.abs(yn)) ** 2).sum(axis=1) * cw
        partial[:] = [chunk.sum() for chunk in np.array_split(d, nparts)]
    call(2)
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    scores = (Xup * sw) @ onehot + sc
    # incomplete rows: explicit scores Yn from the state
    x0 = head + K + 2 * P + K * P + K * L
    scores[rows] = state[x0:x0 + K * L].reshape(K, L)
    return dict(weights=row[:P][inv], r2=row[P:P + L], total=row[P + L:P + L + ne], direct=row[P + L + ne:P + L + 2 * ne],
                loadings=row[P + L + 2 * ne:2 * P + L + 2 * ne][inv], crossloadings=cl[inv], path_coef=pc, iterations=iters.value, status=status.value,
                scores=scores, pairs=pairs)


def check(e, r, tag=""):
    assert e["status"] == 0, tag
    assert e["iterations"] == r["iterations"], "%s: iterations %d vs %d" % (tag, e["iterations"], r["iterations"])
    assert_close(e["weights"], r["weights"], RTOL, what=tag + " weights")
    assert_close(e["loadings"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(e["crossloadings"], r["crossloadings"], RTOL, 1e-12, what=tag + " crossloadings")
    assert_close(e["path_coef"], r["path_coef"], RTOL, 1e-12)
    assert_close(e["r2"], r["r2"], RTOL, 1e-12)
    assert e["pairs"] == r["effect_pairs"]
    assert_close(e["total"], r["total"], RTOL, 1e-12)
    assert_close(e["scores"], r["scores"], 1e-8, 1e-10, what=tag + " scores")


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_russa_missing(emu, scheme):
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "AAA", scheme, True, tol=1e-7, scales=["NUM"] * 9)
    check(run_nmx_emu(emu, X, model), orc.fit(X, model), scheme)


@pytest.mark.parametrize("tag", ["A_path", "M_centroid", "A_factorial"])
def test_synthetic_missing(emu, tag):
    g = load("g13_nonmetric_missing")
    modes, scheme = tag.split("_")
    blocks = [np.arange(4 * j, 4 * j + 4) for j in range(6)]
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA" if modes == "A" else "AAAABB", scheme, True, tol=1e-7, scales=["NUM"] * 24)
    check(run_nmx_emu(emu, g["synth"], model, nthreads=7), orc.fit(g["synth"], model), tag)


def test_russa_missing_raw_scale(emu):
    """Scale.RAW only: the treated values are used as they are, so columns with holes are scaled differently from Scale.NUM."""
    g = load("g13_nonmetric_missing")
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "AAA", "centroid", True, tol=1e-7, scales=["RAW"] * 9)
    e = run_nmx_emu(emu, X, model)
    check(e, orc.fit(X, model), "raw")
    assert_close(e["weights"], g["russa_raw_centroid/weights"], 1e-8)
    assert np.abs(e["weights"] - g["russa_centroid/weights"]).max() > 1e-5


def test_bootstrap_weights_vs_oracle_replicates(emu):
    g = load("g13_nonmetric_missing")
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "AAA", "centroid", True, tol=1e-7, scales=["NUM"] * 9)
    corr = orc.correction(47)
    for k, idx in enumerate(g["idx47"]):
        counts = np.bincount(idx, minlength=47).astype(np.float64)
        e = run_nmx_emu(emu, X, model, counts)
        want, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert e["status"] == 0 and e["iterations"] == its
        mine = np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"]))
        assert_close(mine, want, RTOL, 1e-11, what="replicate %d" % k)
        assert_close(mine, g["russa_centroid/boot_rows"][k], 1e-8, 1e-10)


def test_complete_data_agrees_with_the_plain_num_solver(emu):
    from test_oracle_golden import RUSSA_BLOCKS, russa_inputs
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, "ABA", "path", True, tol=1e-7, scales=["NUM"] * 9)
    check(run_nmx_emu(emu, X, model), orc.fit(X, model))


def test_mode_b_block_with_a_hole_is_flagged(emu):
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "BAA", "centroid", True, tol=1e-7, scales=["NUM"] * 9)
    assert run_nmx_emu(emu, X, model)["status"] != 0
