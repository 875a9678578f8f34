"""CPU check of the two-stage higher-order-construct machinery (csrc/solver_hoc.h) through the std::thread emulation build:
stage 1 (NUM solver on the replicate's moments) -> stage-2 moment matrix by congruence -> stage 2, with the stage-2 convergence
pass evaluated on the ORIGINAL columns through the composed score maps -- against oracle.fit_two_stage, which is pinned on the
reference (golden g12: mobi, fit + bootstrap replicates)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close, effect_pairs, load, packed_scatter, padded_width
from test_oracle_golden import MOBI_C1, MOBI_STAGE2, mobi_hoc_inputs, mobi_hoc_model

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
I32 = ctypes.c_int


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    lib = ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))
    lib.hostemu_cov_doubles.restype = ctypes.c_long
    lib.hostemu_nm_state_doubles.restype = ctypes.c_long
    return lib


def _ptr(a, t=ctypes.c_double):
    return a.ctypes.data_as(ctypes.POINTER(t))


def state_slices(P, L):
    o, sl = 8, {}
    for name, size in (("a_old", P), ("a_new", P), ("c_old", P), ("c_new", P), ("k_old", L), ("k_new", L), ("sd", P), ("mu", P)):
        sl[name] = slice(o, o + size); o += size
    return sl


class NmProblem:
    """One non-metric problem on packed moments; the caller plays the streaming convergence pass."""

    def __init__(self, lib, Mp, P, boff, C, modes, scheme, max_iter, tol, nthreads=4):
        self.lib, self.P, self.L, self.boff = lib, P, len(boff) - 1, np.ascontiguousarray(boff, dtype=np.int32)
        self.C = np.ascontiguousarray(np.asarray(C).astype(np.uint8))
        self.mode = np.array([0 if m == "A" else 1 for m in modes], dtype=np.int32)
        self.shift = np.zeros(P)
        n_chol = int(sum(2 * (self.boff[l + 1] - self.boff[l]) ** 2 for l in range(self.L) if self.mode[l] == 1))
        self.S = np.zeros(lib.hostemu_cov_doubles(P))
        self.state = np.zeros(lib.hostemu_nm_state_doubles(P, self.L, n_chol))
        self.args = (P, self.L, padded_width(P), SCHEME_ID[scheme], max_iter, ctypes.c_double(tol), _ptr(self.boff, I32), _ptr(self.C, ctypes.c_ubyte),
                     _ptr(self.mode, I32), _ptr(self.shift))
        self.nthreads, self.max_iter, self.sl = nthreads, max_iter, state_slices(P, self.L)
        lib.hostemu_nm_prepare(*self.args, _ptr(Mp), nthreads, _ptr(self.S), _ptr(self.state))

    def run(self, conv, nparts=3):
        partial = np.zeros(nparts)
        for _ in range(self.max_iter + 5):
            if not self.lib.hostemu_nm_step(*self.args, self.nthreads, _ptr(self.S), _ptr(self.state), _ptr(partial), nparts):
                break
            partial = np.array([chunk.sum() for chunk in np.array_split(conv(self), nparts)])

    def finish(self, Cpath):
        pairs = effect_pairs(Cpath)
        ef = np.array([p[0] for p in pairs], dtype=np.int32); et = np.array([p[1] for p in pairs], dtype=np.int32)
        ne, P, L = len(pairs), self.P, self.L
        row = np.zeros(2 * P + L + 2 * ne + 2); cl = np.zeros((P, L)); pc = np.zeros((L, L)); lc = np.zeros((L, L))
        ind = np.zeros(max(ne, 1)); sw = np.zeros(P); sc = np.zeros(L); cov = np.zeros((P, P)); mean = np.zeros(P)
        iters = I32(0); status = I32(-1)
        self.lib.hostemu_nm_finish(*self.args, ne, _ptr(ef, I32), _ptr(et, I32), self.nthreads, _ptr(self.S), _ptr(self.state), _ptr(row), _ptr(cl),
                                   _ptr(pc), _ptr(lc), _ptr(ind), _ptr(sw), _ptr(sc), _ptr(cov), _ptr(mean), ctypes.byref(iters), ctypes.byref(status))
        return row[:2 * P + L + 2 * ne], iters.value, status.value


def two_stage_emu(lib, X, model1, stage2, C2, modes2, counts=None, nthreads=4):
    n, P1 = X.shape                                       # columns already in stage-1 device order (blocks contiguous)
    L1 = model1.L
    boff1 = np.concatenate(([0], np.cumsum([len(b) for b in model1.blocks]))).astype(np.int32)
    shift = X.mean(axis=0)
    Xs = X - shift
    cw = np.ones(n) if counts is None else np.asarray(counts, dtype=np.float64)
    M1, _, PA1 = packed_scatter(X, counts, shift)
    onehot1 = (np.repeat(np.arange(L1), np.diff(boff1))[:, None] == np.arange(L1)[None, :]).astype(float)

    def conv1(pb):
        s = pb.state
        yo = (Xs * s[pb.sl["c_old"]]) @ onehot1 + s[pb.sl["k_old"]]
        yn = (Xs * s[pb.sl["c_new"]]) @ onehot1 + s[pb.sl["k_new"]]
        return ((np.abs(yo) - np.abs(yn)) ** 2).sum(axis=1) * cw
    st1 = NmProblem(lib, M1, P1, boff1, model1.C, model1.modes, model1.scheme, model1.max_iter, model1.tol, nthreads)
    st1.run(conv1)
    iters1, ok = int(st1.state[2]), st1.state[1] == 0.0
    c1 = np.ascontiguousarray(st1.state[st1.sl["c_new"]]); k1 = np.ascontiguousarray(st1.state[st1.sl["k_new"]])
    # stage-2 descriptors (what plspm_model_attach_second_stage derives)
    L2 = len(stage2)
    lv_first, boff2, col2_lv1, col2_p1 = [0], [0], [], []
    for kind, ref in stage2:
        if kind == "lv":
            lv_first.append(lv_first[-1] + 1)
            col2_p1.extend(range(boff1[ref], boff1[ref + 1])); col2_lv1.extend([-1] * (boff1[ref + 1] - boff1[ref]))
        else:
            lv_first.append(lv_first[-1] + len(ref))
            col2_lv1.extend(ref); col2_p1.extend([-1] * len(ref))
        boff2.append(len(col2_p1))
    P2 = len(col2_p1)
    hcol = [a for a in range(P2) if col2_lv1[a] >= 0]
    hidx = [-1] * P2
    for h, a in enumerate(hcol):
        hidx[a] = h
    arr = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    lv_first, boff2, col2_lv1, col2_p1, hcol_a, hidx = arr(lv_first), arr(boff2), arr(col2_lv1), arr(col2_p1), arr(hcol or [0]), arr(hidx)
    PA2 = padded_width(P2)
    M2 = np.zeros((PA2 // 16) * (PA2 // 16 + 1) // 2 * 256)
    lib.hostemu_hoc_moments(P1, L1, P2, L2, PA1 // 16, PA2 // 16, _ptr(boff1, I32), _ptr(boff2, I32), _ptr(lv_first, I32), _ptr(col2_lv1, I32),
                            _ptr(col2_p1, I32), len(hcol), _ptr(hcol_a, I32), _ptr(hidx, I32), _ptr(M1), _ptr(c1), _ptr(k1), int(ok), _ptr(M2), nthreads)
    lv_cols = np.array([boff1[j] for j in lv_first])       # stage-1 column range of every stage-2 LV
    onehot2 = np.zeros((P1, L2))
    for l in range(L2):
        onehot2[lv_cols[l]:lv_cols[l + 1], l] = 1.0
    pseudo = np.zeros(8 + 4 * P1 + 2 * L2)

    def conv2(pb):
        lib.hostemu_hoc_compose(P1, L1, P2, L2, _ptr(boff1, I32), _ptr(boff2, I32), _ptr(lv_first, I32), _ptr(col2_lv1, I32), _ptr(c1), _ptr(k1),
                                _ptr(pb.state), _ptr(pseudo), nthreads)
        o = 8 + 2 * P1
        yo = (Xs * pseudo[o:o + P1]) @ onehot2 + pseudo[o + 2 * P1:o + 2 * P1 + L2]
        yn = (Xs * pseudo[o + P1:o + 2 * P1]) @ onehot2 + pseudo[o + 2 * P1 + L2:o + 2 * P1 + 2 * L2]
        return ((np.abs(yo) - np.abs(yn)) ** 2).sum(axis=1) * cw
    st2 = NmProblem(lib, M2, P2, boff2, C2, modes2, model1.scheme, model1.max_iter, model1.tol, nthreads)
    st2.run(conv2)
    row, iters2, status = st2.finish(np.asarray(C2))
    return row, iters1, iters2, status, M2, PA2


@pytest.mark.parametrize("tag", ["path_B", "centroid_A"])
def test_mobi_two_stage_fit_and_weighted_replicates(emu, tag):
    g = load("g12_hoc_two_stage")
    X, blocks, _ = mobi_hoc_inputs()
    model1 = mobi_hoc_model(tag, blocks)
    C2, modes2 = g[tag + "/path2"], model1.modes[0] + "AAAA"
    corr = orc.correction(250)
    for k, idx in enumerate([np.arange(250)] + list(g["idx"][:2])):
        counts = np.bincount(idx, minlength=250).astype(np.float64)
        row, it1, it2, status, _, _ = two_stage_emu(emu, X, model1, MOBI_STAGE2, C2, modes2, counts)
        r = orc.fit_two_stage(X[idx], model1, MOBI_STAGE2, C2, modes2, corr)
        assert status == 0 and it1 == r["iterations1"] and it2 == r["iterations"], (k, it1, it2, r["iterations1"], r["iterations"])
        want = np.concatenate((r["weights"], r["r2"], r["total"], r["direct"], r["loadings"]))
        assert_close(row, want, 1e-9, 1e-11, what="%s replicate %d" % (tag, k))
        assert_close(row, g[tag + "/rows"][k], 1e-8, 1e-10, what="%s vs reference golden %d" % (tag, k))


def test_stage2_moments_equal_moments_of_the_extended_data(emu):
    """M2 by congruence == scatter of [plain columns | stage-1 scores] (what the reference's stage 2 receives)."""
    X, blocks, _ = mobi_hoc_inputs()
    model1 = mobi_hoc_model("path_B", blocks)
    g = load("g12_hoc_two_stage")
    row, it1, it2, status, M2, PA2 = two_stage_emu(emu, X, model1, MOBI_STAGE2, g["path_B/path2"], "BAAAA")
    r1 = orc.fit(X, model1)
    shift = X.mean(axis=0)
    X2 = np.column_stack([X[:, :10] - shift[:10], r1["scores"][:, 2], r1["scores"][:, 3], X[:, 17:] - shift[17:]])
    want, _, _ = packed_scatter(X2, None, np.zeros(X2.shape[1]))
    from helpers import packed_index_np
    pp, qq = np.meshgrid(np.arange(X2.shape[1] + 1), np.arange(X2.shape[1] + 1), indexing="ij")
    slots = packed_index_np(PA2 // 16, pp.ravel(), qq.ravel())
    assert_close(M2[slots], want[slots], 1e-10, 1e-9)


def test_failed_first_stage_poisons_the_second(emu):
    X, blocks, _ = mobi_hoc_inputs()
    model1 = mobi_hoc_model("path_B", blocks)
    model1.max_iter = 2                                   # stage 1 cannot converge in 2 iterations
    g = load("g12_hoc_two_stage")
    row, it1, it2, status, M2, _ = two_stage_emu(emu, X, model1, MOBI_STAGE2, g["path_B/path2"], "BAAAA")
    assert np.isnan(M2).any() and status != 0
