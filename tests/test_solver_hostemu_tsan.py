"""ThreadSanitizer pass over the non-metric device solver sources (solver_core.h nm_*, solver_nmg.h, solver_nmx.h, solver_hoc.h)
in the std::thread emulation build: the executor's barriers stand in for __syncthreads, so a data race here is a missing
barrier on the GPU.  (The metric solver has its own TSan test in test_solver_hostemu.py.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "hostemu")

SNIPPETS = {
    "nm": ("import test_solver_hostemu_nonmetric as t; from test_oracle_golden import RUSSA_BLOCKS, RUSSA_C, russa_inputs;"
           "lib=t.ctypes.CDLL(LIB); lib.hostemu_cov_doubles.restype=t.ctypes.c_long; lib.hostemu_nm_state_doubles.restype=t.ctypes.c_long;"
           "[t.run_nm_emu(lib, russa_inputs(), orc.Model(RUSSA_BLOCKS, RUSSA_C, m, s, True, tol=1e-7, scales=['NUM']*9), nthreads=6) for m in ('AAA','ABA') for s in ('centroid','path')]"),
    "nmg": ("import test_solver_hostemu_ordnom as t;"
            "lib=t.ctypes.CDLL(LIB); lib.hostemu_cov_doubles.restype=t.ctypes.c_long; lib.hostemu_nmg_state_doubles.restype=t.ctypes.c_long;"
            "[t.run_cat_emu(lib, t.russa_cat_inputs(), orc.Model(t.RUSSA_CAT_BLOCKS, t.RUSSA_C, m, 'centroid', True, tol=1e-7, scales=t.RUSSA_CAT_SCALES), nthreads=6) for m in ('AAA','BBB')]"),
    "nmx": ("import test_solver_hostemu_nmx as t;"
            "lib=t.ctypes.CDLL(LIB); lib.hostemu_cov_doubles.restype=t.ctypes.c_long; lib.hostemu_nm_state_doubles.restype=t.ctypes.c_long;"
            "lib.hostemu_nmx_state_doubles.restype=t.ctypes.c_long;"
            "[t.run_nmx_emu(lib, t.russa_missing_matrix(), orc.Model(t.RUSSA_M_BLOCKS, t.RUSSA_C, 'AAA', s, True, tol=1e-7, scales=['NUM']*9), nthreads=6) for s in ('centroid','path')]"),
    "hoc": ("import test_solver_hostemu_hoc as t; from helpers import load;"
            "lib=t.ctypes.CDLL(LIB); lib.hostemu_cov_doubles.restype=t.ctypes.c_long; lib.hostemu_nm_state_doubles.restype=t.ctypes.c_long;"
            "X,blocks,_=t.mobi_hoc_inputs(); g=load('g12_hoc_two_stage');"
            "t.two_stage_emu(lib, X, t.mobi_hoc_model('path_B', blocks), t.MOBI_STAGE2, g['path_B/path2'], 'BAAAA', nthreads=6)"),
    "impute": ("import test_solver_hostemu_missing as t; import numpy as np;"
               "lib=t.ctypes.CDLL(LIB); Xn,blocks,C=t.missing_case(); Xaug,ind=t.aug_matrix(Xn);"
               "t.collapse(lib, Xaug, ind, None, Xaug[:, :Xn.shape[1]].mean(axis=0), nthreads=6)"),
}


@pytest.mark.parametrize("which", sorted(SNIPPETS))
def test_thread_sanitizer_clean(which):
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_tsan.so"])
    lib = os.path.join(EMU, "libplspm_hostemu_tsan.so")
    code = ("import sys; sys.path[:0]=[%r,%r]; import plspm_oracle as orc; LIB=%r; %s; print('tsan-run-done')"
            % (HERE, os.path.join(os.path.dirname(HERE), "oracle"), lib, SNIPPETS[which]))
    tsan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "tsan-run-done" in r.stdout, r.stderr[-3000:]
    assert "data race" not in r.stderr, r.stderr[-6000:]


def test_address_sanitizer_clean():
    """The same solver sources under AddressSanitizer (GPU ASan is not available on the pool: the CPU build is where out-of-bounds
    accesses of the workspace / state carving would show)."""
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_asan.so"])
    asan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libasan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(EMU, "asan_run.py")], env=env, capture_output=True, text=True, timeout=900)
    assert "asan-run-done" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
