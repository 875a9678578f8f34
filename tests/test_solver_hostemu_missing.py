"""CPU check of the metric missing-data stage (csrc/solver_core.h impute_collapse) through the std::thread emulation build:
the weighted Gram of [mean-filled data | missing indicators | 1] collapsed on the moments must equal the moments of the
RESAMPLED data imputed with its own column means (reference util.impute via Config.treat per replicate, bootstrap.py:57), and
the solver run on the collapsed moments must reproduce the oracle's replicate."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, load, packed_index_np, packed_scatter, padded_width
from test_solver_hostemu import run_emu

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    return ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def aug_matrix(Xnan):
    """[filled | indicators], ind_of -- what plspm/weights.py uploads for a metric model with missing values."""
    means = np.nanmean(Xnan, axis=0)
    filled = np.where(np.isnan(Xnan), means, Xnan)
    incomplete = np.flatnonzero(np.isnan(Xnan).any(axis=0))
    ind_of = np.full(Xnan.shape[1], -1, dtype=np.int32)
    ind_of[incomplete] = Xnan.shape[1] + np.arange(len(incomplete))
    return np.column_stack([filled, np.isnan(Xnan[:, incomplete]).astype(np.float64)]), ind_of


def collapse(lib, Xaug, ind_of, counts, shift_data, nthreads=4):
    P, Qa = len(ind_of), Xaug.shape[1]
    shift = np.concatenate([shift_data, np.zeros(Qa - P)])
    Min, _, PAa = packed_scatter(Xaug, counts, shift)
    PAs = padded_width(P)
    out = np.zeros((PAs // 16) * (PAs // 16 + 1) // 2 * 256)
    lib.hostemu_impute_collapse(P, Qa, PAa // 16, PAs // 16, _ptr(ind_of, ctypes.c_int), _ptr(Min), _ptr(out), nthreads)
    return out, PAs


def missing_case(n=300, seed=5, holes=60):
    C = orc.satisfaction_C()
    X, blocks = orc.synth(n, C, 4, seed=seed)
    rs = np.random.RandomState(seed)
    Xn = X.copy()
    for _ in range(holes):
        Xn[rs.randint(n), rs.randint(X.shape[1] - 6)] = np.nan       # the last columns stay complete
    return Xn, blocks, C


@pytest.mark.parametrize("nthreads", [1, 5])
def test_collapsed_moments_equal_moments_of_reimputed_resample(emu, nthreads):
    Xn, blocks, C = missing_case()
    n, P = Xn.shape
    Xaug, ind_of = aug_matrix(Xn)
    shift = Xaug[:, :P].mean(axis=0)
    rs = np.random.RandomState(1)
    for rep in range(4):
        idx = np.arange(n) if rep == 0 else rs.randint(n, size=n)
        counts = np.bincount(idx, minlength=n).astype(np.float64)
        out, PAs = collapse(emu, Xaug, ind_of, counts, shift, nthreads)
        Ximp = orc.impute(Xn[idx])                       # util.impute on the resampled rows
        want, _, _ = packed_scatter(Ximp, None, shift)
        pp, qq = np.meshgrid(np.arange(P + 1), np.arange(P + 1), indexing="ij")
        slots = packed_index_np(PAs // 16, pp.ravel(), qq.ravel())           # (diagonal tiles also hold a mirrored, unused half)
        assert_close(out[slots], want[slots], 1e-11, 1e-9, what="replicate %d" % rep)


def test_column_without_present_cells_poisons_the_replicate(emu):
    Xn, blocks, C = missing_case(n=40, holes=0)
    Xn[:38, 3] = np.nan
    Xaug, ind_of = aug_matrix(Xn)
    counts = np.zeros(40); counts[:38] = 1; counts[0] = 3           # the two present cells of column 3 are not drawn
    out, PAs = collapse(emu, Xaug, ind_of, counts, Xaug[:, :Xn.shape[1]].mean(axis=0))
    T = PAs // 16
    assert np.isnan(out[packed_index_np(T, np.array([3]), np.array([3]))][0])
    assert np.isfinite(out[packed_index_np(T, np.array([2]), np.array([4]))][0])


@pytest.mark.parametrize("modes,scheme,scaled", [("AAAAAA", "centroid", True), ("ABABAB", "path", False)])
def test_solver_on_collapsed_moments_reproduces_oracle_replicates(emu, modes, scheme, scaled):
    Xn, blocks, C = missing_case()
    n, P = Xn.shape
    model = orc.Model(blocks, C, modes, scheme, scaled)
    order = model.mv_order
    Xaug, ind_of = aug_matrix(Xn[:, order])              # device column order
    shift = Xaug[:, :P].mean(axis=0)
    corr = orc.correction(n)
    rs = np.random.RandomState(3)
    for rep in range(3):
        idx = np.arange(n) if rep == 0 else rs.randint(n, size=n)
        counts = np.bincount(idx, minlength=n).astype(np.float64)
        Mp, PAs = collapse(emu, Xaug, ind_of, counts, shift)
        e = run_emu(emu, orc.impute(Xn[idx]), model, packed=(Mp, shift, PAs))
        row, its = orc.bootstrap_replicate(Xn, model, idx, corr)
        assert e["status"] == 0 and e["iterations"] == its
        mine = np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"]))
        assert_close(mine, row, 1e-9, 1e-11, what="replicate %d" % rep)
