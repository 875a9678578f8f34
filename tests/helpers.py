"""Shared helpers for the parity tests (test infrastructure)."""
import hashlib
import os

import numpy as np
import pandas as pd

import plspm_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAT_PREFIX = dict(IMAG="imag", EXPE="expe", QUAL="qual", VAL="val", SAT="sat", LOY="loy")
SAT_ADD_ORDER = ["IMAG", "EXPE", "VAL", "QUAL", "SAT", "LOY"]   # as tests/golden/make_golden.py (and the reference test)
MODES = {"A": "AAAAAA", "B": "BBBBBB"}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def satisfaction_frame():
    return pd.read_csv(os.path.join(GOLDEN, "ref_data", "satisfaction.csv"), index_col=0)


def satisfaction_oracle_inputs():
    """X in data-column (add_lv) order + blocks (path order) for the satisfaction fixtures."""
    sat = satisfaction_frame()
    cols = []
    for lv in SAT_ADD_ORDER:
        cols += [c for c in sat.columns if c.startswith(SAT_PREFIX[lv])]
    X = sat[cols].values.astype(np.float64)
    blocks = [np.array([i for i, c in enumerate(cols) if c.startswith(SAT_PREFIX[lv])]) for lv in orc.SAT_LVS]
    return X, blocks, cols


def case_modes(name, L=6, mixed="ABABAB"):
    if name == "M":
        return mixed
    return name * L


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def assert_close(a, b, rtol, atol=0.0, what=""):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol,
                               err_msg=what)


# ----------------------------------------------------------------------------------------------
# NumPy restatement of the device data layouts (independent of the C++ in csrc/solver_core.h)
def packed_index_np(T, p, q):
    """csrc/solver_core.h packed_index in NumPy.  Odd T: the last tile holds the 16 consecutive columns 16(T-1) .. 16T-1."""
    p = np.asarray(p); q = np.asarray(q)
    last = 16 * (T - 1)

    def tile_pos(c):
        lone = (T % 2 == 1) & (c >= last)
        return np.where(lone, T - 1, 2 * (c >> 5) + (c & 1)), np.where(lone, c - last, (c & 31) >> 1)
    tp, ip = tile_pos(p)
    tq, iq = tile_pos(q)
    swap = ~((tp < tq) | ((tp == tq) & (ip <= iq)))
    t = np.where(swap, tq, tp); u = np.where(swap, tp, tq)
    r = np.where(swap, iq, ip); c = np.where(swap, ip, iq)
    tile = t * T - t * (t - 1) // 2 + (u - t)
    return (tile * 4 + (r >> 2)) * 64 + (r & 3) * 16 + c


def padded_width(P, odd_tiles=False):
    """Device row pitch: whole 32-column groups; with odd_tiles the 16-column granularity metric handles use for 5 <= T <= 15."""
    t16 = (P + 1 + 15) // 16
    if odd_tiles and t16 % 2 == 1 and 5 <= t16 <= 15:
        return 16 * t16
    return ((P + 1 + 31) // 32) * 32


def packed_scatter(Xdev, counts=None, shift=None, odd_tiles=False):
    """Augmented raw scatter sum_i c_i [x'_i,1][x'_i,1]^T of the shifted data in the tile-packed layout."""
    n, P = Xdev.shape
    PA = padded_width(P, odd_tiles)
    T = PA // 16
    if shift is None:
        shift = Xdev.mean(axis=0)
    Xa = np.zeros((n, PA))
    Xa[:, :P] = Xdev - shift
    Xa[:, P] = 1.0
    c = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    M = (Xa * c[:, None]).T @ Xa
    out = np.zeros(T * (T + 1) // 2 * 256)
    pp, qq = np.meshgrid(np.arange(PA), np.arange(PA), indexing="ij")
    out[packed_index_np(T, pp.ravel(), qq.ravel())] = M.ravel()
    return out, shift, PA


def effect_pairs(C):
    """(from, to) pairs with a directed path from -> to, from-major (the rows of the reference's effects frame)."""
    L = C.shape[0]
    reach = (np.asarray(C) != 0)
    for _ in range(L):
        reach = reach | ((reach.astype(int) @ reach.astype(int)) > 0)
    return [(f, t) for f in range(L) for t in range(L) if f != t and reach[t, f]]


# ----------------------------------------------------------------------------------------------
# When may the device report a numerical condition the oracle did not?  (tests/test_gpu_fuzz.py)
DEGENERATE_EIG_RTOL = 1e-10      # two orders above csrc/solver_core.h PLSPM_EIG_RTOL (1e-12) / PLSPM_PIVOT_RTOL (1e-13): the device's rank decision


def oracle_conditioning(X, model):
    """Smallest relative eigenvalue the oracle's own arithmetic meets on these rows: over the correlation matrices of the Mode-B blocks (mode.py:50-52,
    lstsq on the block) and of every LV's predecessor scores (scheme.py:48-50, inner_model.py:69: OLS on the predecessors), and the smallest column
    standard deviation relative to the largest.  Returns (value, what)."""
    worst, what = 1.0, "none"
    sd = X.std(axis=0)
    if not np.all(np.isfinite(sd)) or sd.max() == 0:
        return 0.0, "non-finite or constant data"
    if sd.min() / sd.max() < worst:
        worst, what = float(sd.min() / sd.max()), "column std ratio"
    if sd.min() == 0:
        return 0.0, "zero-variance MV"
    for l, blk in enumerate(model.blocks):
        if model.modes[l] == "B" and len(blk) > 1:
            ev = np.linalg.eigvalsh(np.corrcoef(X[:, blk], rowvar=False))
            if ev[0] / ev[-1] < worst:
                worst, what = float(ev[0] / ev[-1]), "Mode-B block %d" % l
    try:
        scores = orc.fit(X, model)["scores"]
    except Exception:                                      # noqa: BLE001 -- the oracle itself cannot estimate these rows
        return 0.0, "oracle fails on these rows"
    if not np.all(np.isfinite(scores)):
        return 0.0, "oracle scores not finite"
    for i in range(model.L):
        pred = np.flatnonzero(model.C[i])
        if len(pred) > 1:
            ev = np.linalg.eigvalsh(np.corrcoef(scores[:, pred], rowvar=False))
            if ev[0] / ev[-1] < worst:
                worst, what = float(ev[0] / ev[-1]), "predecessors of LV %d" % i
    return worst, what


def assert_device_status_justified(status, X, model, tag=""):
    """A device status the oracle did not share on the same rows: PLSPM_NOT_CONVERGED (1) is never acceptable (iteration counts are held equal);
    PLSPM_SINGULAR / PLSPM_NONFINITE (2, 3) only when the oracle's own systems on these rows are degenerate to working precision."""
    assert status != 1, "%s: the device did not converge where the oracle did" % tag
    cond, what = oracle_conditioning(X, model)
    assert cond < DEGENERATE_EIG_RTOL, "%s: device status %d on rows whose worst conditioning is %.3g (%s): not a degenerate case" % (tag, status, cond, what)
