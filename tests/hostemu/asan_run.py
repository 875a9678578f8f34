"""TEST INFRASTRUCTURE: drives every device solver source (compiled for the CPU with -fsanitize=address) through the emulation
runners of the CPU tests; run by tests/test_solver_hostemu_tsan.py::test_address_sanitizer_clean with libasan preloaded."""
import sys, os, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "oracle")]
import numpy as np
import plspm_oracle as orc
LIB = os.path.join(HERE, "hostemu", "libplspm_hostemu_asan.so")
def lib():
    l = ctypes.CDLL(LIB)
    for n in ("hostemu_cov_doubles", "hostemu_nm_state_doubles", "hostemu_nmg_state_doubles", "hostemu_nmx_state_doubles"):
        getattr(l, n).restype = ctypes.c_long
    return l
l = lib()
import test_solver_hostemu as t0
from helpers import satisfaction_oracle_inputs, load
X, b, _ = satisfaction_oracle_inputs()
for m in ("AAAAAA", "BBBBBB", "ABABAB"):
    for s in ("centroid", "factorial", "path"):
        t0.run_emu(l, X, orc.Model(b, orc.satisfaction_C(), m, s, True), nthreads=5)
for m in ("AAAAAA", "BBBBBB"):                                  # rows solver (one thread per MV), its split form (two per MV, 65-128 MVs) and the wave solver
    t0.run_emu(l, X, orc.Model(b, orc.satisfaction_C(), m, "path", True), rows=True)
Xw, mw = t0._wide_model(10, 12, "M", "path", 21)
t0.run_emu(l, Xw, mw, rows=True, split=True)
import test_solver_hostemu_wave as tw
Xs, bs = orc.synth(500, orc.satisfaction_C(), 10, seed=4)
for m in ("AAAAAA", "BBBBBB"):
    tw.run_wave(l, Xs, orc.Model(bs, orc.satisfaction_C(), m, "path", True))
print("metric ok")
import test_solver_hostemu_nonmetric as t1
from test_oracle_golden import RUSSA_BLOCKS, RUSSA_C, russa_inputs
for m in ("AAA", "ABA", "BBB"):
    t1.run_nm_emu(l, russa_inputs(), orc.Model(RUSSA_BLOCKS, RUSSA_C, m, "path", True, tol=1e-7, scales=["NUM"] * 9), nthreads=5)
print("nm ok")
import test_solver_hostemu_ordnom as t2
for m in ("AAA", "BBB"):
    t2.run_cat_emu(l, t2.russa_cat_inputs(), orc.Model(t2.RUSSA_CAT_BLOCKS, t2.RUSSA_C, m, "centroid", True, tol=1e-7, scales=t2.RUSSA_CAT_SCALES), nthreads=5)
g = load("g11_ordnom")
for tag in ("ordA", "ordB", "mixM"):
    modes, scales = t2.LIKERT_CASES[tag]
    t2.run_cat_emu(l, g["likert"], orc.Model(t2.LIKERT_BLOCKS, t2.LIKERT_C, modes, "path", True, tol=1e-7, scales=scales), nthreads=5)
print("nmg ok")
import test_solver_hostemu_nmx as t3
for s in ("centroid", "path"):
    t3.run_nmx_emu(l, t3.russa_missing_matrix(), orc.Model(t3.RUSSA_M_BLOCKS, t3.RUSSA_C, "AAA", s, True, tol=1e-7, scales=["NUM"] * 9), nthreads=5)
g13 = load("g13_nonmetric_missing")
blocks = [np.arange(4 * j, 4 * j + 4) for j in range(6)]
t3.run_nmx_emu(l, g13["synth"], orc.Model(blocks, orc.satisfaction_C(), "AAAABB", "centroid", True, tol=1e-7, scales=["NUM"] * 24), nthreads=5)
print("nmx ok")
import test_solver_hostemu_hoc as t4
Xm, blocksm, _ = t4.mobi_hoc_inputs(); g12 = load("g12_hoc_two_stage")
t4.two_stage_emu(l, Xm, t4.mobi_hoc_model("path_B", blocksm), t4.MOBI_STAGE2, g12["path_B/path2"], "BAAAA", nthreads=5)
print("hoc ok")
import test_solver_hostemu_missing as t5
Xn, bl, C = t5.missing_case(); Xaug, ind = t5.aug_matrix(Xn)
t5.collapse(l, Xaug, ind, None, Xaug[:, :Xn.shape[1]].mean(axis=0), nthreads=5)
print("impute ok; asan-run-done")
