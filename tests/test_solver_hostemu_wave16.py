"""CPU check of the wave solver source for 9 ... 16 LVs (csrc/solver_wave16.h: one 64-lane wave per problem, four matrix entries per pair lane, V in LDS)
through the std::thread emulation build in tests/hostemu/: against the data-level oracle, bootstrap replicates of the oracle and the rows variant of the
same solver.  Tolerance vs the oracle: 1e-9 relative (fp64 both sides; the formulations differ); vs the rows variant 1e-10."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_solver_hostemu import EMU, HERE, RTOL, run_emu
from test_solver_hostemu_quad import _shaped, run_quad
from test_solver_hostemu_wave import check


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu.so"])
    return ctypes.CDLL(os.path.join(EMU, "libplspm_hostemu.so"))


def run_w16(lib, X, model, counts=None, shift=None):
    return run_quad(lib, X, model, counts, shift, entry="hostemu_solve_wave16")


def _dag(L, fan):
    C = np.zeros((L, L), dtype=np.int64)
    for i in range(1, L):
        for j in range(max(0, i - fan), i):
            C[i, j] = 1
    return C


@pytest.mark.parametrize("scheme,scaled,sizes,fan", [("path", True, [5] * 12, 2), ("factorial", True, [5] * 12, 2), ("centroid", False, [5] * 12, 1), ("path", False, [4] * 16, 3),
                                                     ("centroid", True, [4] * 16, 1), ("factorial", False, [3] * 9, 4), ("path", True, [6] * 10, 2)])
def test_wave16_vs_oracle_rows_variant_and_bootstrap_replicate(emu, scheme, scaled, sizes, fan):
    L = len(sizes)
    C = _dag(L, fan)
    X, blocks = _shaped(C, sizes, seed=21, N=600)
    model = orc.Model(blocks, C, "A" * L, scheme, scaled)
    e = run_w16(emu, X, model)
    assert e is not None
    check(e, orc.fit(X, model), "wave16 %s/%d %s" % (scheme, scaled, sizes))
    base = run_emu(emu, X, model, rows=True)
    assert e["iterations"] == base["iterations"]
    assert_close(e["row"], base["row"], 1e-10, 1e-13)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, X.shape[0], X.shape[0])
    shift = X[:, model.mv_order].mean(axis=0)
    e = run_w16(emu, X, model, counts=np.bincount(idx, minlength=X.shape[0]), shift=shift)
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(X.shape[0]))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)


def test_wave16_model_shapes(emu):
    """Ragged blocks (1 ... 30 MVs), L = 9 ... 16, 64 MVs exactly, five and more predecessors (Cholesky in the scratch area), every row group of the pair
    lanes partly filled (L = 9, 10, 13)."""
    cases = [([4] * 16, 1), ([1] * 8 + [7] * 8, 2), ([1, 17, 2, 9, 5, 3, 3, 3, 4, 1], 4), ([30, 1, 1, 1, 1, 1, 1, 1, 1], 2), ([5, 4, 3, 6, 2, 5, 4, 3, 6, 2, 5, 4, 3], 6),
             ([2] * 11, 10)]
    for sizes, fan in cases:
        L = len(sizes)
        C = _dag(L, fan)
        X, blocks = _shaped(C, sizes, seed=4)
        for scheme in ("centroid", "factorial", "path"):
            model = orc.Model(blocks, C, "A" * L, scheme, True)
            e = run_w16(emu, X, model)
            assert e is not None, sizes
            check(e, orc.fit(X, model), "L=%d %s %s" % (L, sizes, scheme))


def test_wave16_declines_models_outside_its_class(emu):
    for sizes, modes in (([5] * 8, "A" * 8), ([3] * 17, "A" * 17), ([30, 30, 1, 1, 1, 1, 1, 1, 1], "BB" + "A" * 7), ([7] * 10, "A" * 10)):      # 8 LVs: the wave solver's; 17 LVs; Mode-B inverses beyond the staging area; 70 MVs
        L = len(sizes)
        X, blocks = _shaped(orc.chain_C(L), sizes, seed=2)
        assert run_w16(emu, X, orc.Model(blocks, orc.chain_C(L), modes, "centroid", True)) is None, sizes


def test_wave16_status_codes_sign_rule_and_rank_deficient_predecessors(emu):
    sizes = [5, 6, 4, 7, 5, 4, 6, 5, 4, 5]
    C = _dag(10, 2)
    X, blocks = _shaped(C, sizes, seed=9)
    tight = orc.Model(blocks, C, "A" * 10, "centroid", True, max_iter=2, tol=1e-14)
    e = run_w16(emu, X, tight)
    assert e["status"] == 1 and e["iterations"] == 3           # counter runs to max_iter+1 before giving up (weights.py:181-186)
    Xc = X.copy(); Xc[:, blocks[2][1]] = 3.0                    # a constant MV
    cm = orc.Model(blocks, C, "A" * 10, "centroid", True)                                     # a constant MV: the reference centres it to zeros -- weight 0, loading 0, the estimate counts (solver_core.h treated_sd)
    e, r = run_w16(emu, Xc, cm), orc.fit(Xc, cm)
    assert e["status"] == 0 and e["iterations"] == r["iterations"] and e["loadings"][blocks[2][1]] == 0.0
    assert_close(e["weights"], r["weights"], 1e-9, 1e-12); assert_close(e["loadings"], r["loadings"], 1e-9, 1e-12)
    Xn = X.copy()
    Xn[:, blocks[0][:4]] *= -1.0                                # most MVs of LVs 0 and 9 negated: the sign rule flips them (weights.py:62-64)
    Xn[:, blocks[9][:4]] *= -1.0
    for scheme in ("centroid", "path"):
        model = orc.Model(blocks, C, "A" * 10, scheme, True)
        check(run_w16(emu, Xn, model), orc.fit(Xn, model), "sign " + scheme)
    Xb = X.copy()
    Xb[:, blocks[1][:5]] = Xb[:, blocks[0]]                     # LV 1's first five MVs == LV 0's ...
    Xb[:, blocks[1][5]] = Xb[:, blocks[0][0]]                   # ... and the sixth a copy too: collinear predecessor scores for LV 2
    for scheme in ("path", "centroid"):
        model = orc.Model(blocks, C, "A" * 10, scheme, True)
        e = run_w16(emu, Xb, model)
        r = orc.fit(Xb, model)
        assert e["status"] == r.get("status", 0) or e["status"] == 0
        if e["status"] == 0:
            assert e["iterations"] == r["iterations"]
            assert_close(e["weights"], r["weights"], RTOL)


def test_wave16_thread_sanitizer_clean():
    subprocess.check_call(["make", "-s", "-C", EMU, "libplspm_hostemu_tsan.so"])
    code = ("import sys; sys.path[:0]=[%r,%r]; import ctypes, numpy as np; import plspm_oracle as orc; import test_solver_hostemu_wave16 as q;"
            "lib=ctypes.CDLL(%r);"
            "X,b=q._shaped(q._dag(12, 2), [5] * 12, seed=3); assert q.run_w16(lib, X, orc.Model(b, q._dag(12, 2), 'A' * 12, 'path', True)) is not None;"
            "X,b=q._shaped(q._dag(16, 6), [4] * 16, seed=5); assert q.run_w16(lib, X, orc.Model(b, q._dag(16, 6), 'A' * 16, 'centroid', True)) is not None;"
            "print('tsan-run-done')") % (HERE, os.path.join(os.path.dirname(HERE), "oracle"), os.path.join(EMU, "libplspm_hostemu_tsan.so"))
    tsan = subprocess.run(["bash", "-c", "ls /usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so | head -1"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", OPENBLAS_NUM_THREADS="1")      # (NumPy's BLAS pool is not under test)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "tsan-run-done" in r.stdout, r.stderr[-2000:]
    assert "data race" not in r.stderr, r.stderr[-4000:]


def test_wave16_source_at_eight_lvs_equals_the_wave_solver(emu):
    """solve_problem_wave16<8> (set_option("solver_wave", 2): V in LDS, the product stream's second copy w V for the Q sums, a folded into E) on the wave solver's
    own class: the oracle at 1e-9, the wave solver's record at 1e-11, equal iteration counts; the reference's bootstrap rows of golden g4."""
    from helpers import load, satisfaction_oracle_inputs
    from test_solver_hostemu_wave import run_wave
    X, blocks, _ = satisfaction_oracle_inputs()
    for scheme in ("centroid", "factorial", "path"):
        for scaled in (False, True):
            model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", scheme, scaled)
            e = run_quad(emu, X, model, entry="hostemu_solve_wave16_l8")
            check(e, orc.fit(X, model), "wave16<8> %s/%d" % (scheme, scaled))
            base = run_wave(emu, X, model)
            assert e["iterations"] == base["iterations"]
            assert_close(e["row"], base["row"], 1e-11, 1e-13)
    for sizes, fan in (([8] * 8, 1), ([3, 1], 1), ([1, 17, 2, 9, 5], 2), ([4, 3, 5, 2, 6, 7], 4), ([63, 1], 1), ([7, 9, 5, 11, 3, 13, 1, 15], 7)):
        L = len(sizes)
        C = _dag(L, fan)
        Xs, bs = _shaped(C, sizes, seed=4)
        for scheme in ("centroid", "path"):
            model = orc.Model(bs, C, "A" * L, scheme, True)
            e = run_quad(emu, Xs, model, entry="hostemu_solve_wave16_l8")
            assert e is not None
            check(e, orc.fit(Xs, model), "wave16<8> L=%d %s %s" % (L, sizes, scheme))


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("modes", ["BBBBBB", "ABABAB", "BAAAAB"])
def test_wave16_source_with_mode_b_blocks(emu, scheme, modes):
    """Mode-B blocks (round 5, last part: the wave solver's block inverses on this workspace).  LMAX = 8 on the satisfaction model against the oracle, the wave
    solver (1e-11) and a bootstrap replicate; LMAX = 16 on ten / twelve LVs against the oracle and the rows variant."""
    from helpers import satisfaction_oracle_inputs
    from test_solver_hostemu_wave import run_wave
    X, blocks, _ = satisfaction_oracle_inputs()
    for scaled in (False, True):
        model = orc.Model(blocks, orc.satisfaction_C(), modes, scheme, scaled)
        e = run_quad(emu, X, model, entry="hostemu_solve_wave16_l8")
        assert e is not None
        check(e, orc.fit(X, model), "wave16<8> %s %s/%d" % (modes, scheme, scaled))
        base = run_wave(emu, X, model)
        assert e["iterations"] == base["iterations"]
        assert_close(e["row"], base["row"], 1e-11, 1e-13)
    Xs, bs = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(bs, orc.satisfaction_C(), modes, scheme, True)
    rng = np.random.default_rng(6)
    idx = rng.integers(0, 2000, 2000)
    e = run_quad(emu, Xs, model, counts=np.bincount(idx, minlength=2000), shift=Xs[:, model.mv_order].mean(axis=0), entry="hostemu_solve_wave16_l8")
    mine, its = orc.bootstrap_replicate(Xs, model, idx, orc.correction(2000))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)
    for sizes in ([5] * 12, [1, 13, 2, 6, 3, 3, 17, 4, 5, 2]):
        L = len(sizes)
        C = _dag(L, 2)
        X16, b16 = _shaped(C, sizes, seed=12, N=600)
        m16 = (modes * 3)[:L]
        model = orc.Model(b16, C, m16, scheme, True)
        e = run_w16(emu, X16, model)
        assert e is not None
        check(e, orc.fit(X16, model), "wave16<16> %s %s %s" % (m16, sizes, scheme))
        base = run_emu(emu, X16, model, rows=True)
        assert e["iterations"] == base["iterations"]
        assert_close(e["row"], base["row"], 1e-10, 1e-12)


def test_wave16_source_mode_b_rank_deficient_blocks_take_the_minimum_norm_route(emu):
    """Golden g14 from the real reference (a duplicated MV and a linearly dependent MV inside Mode-B blocks: gelsd's minimum-norm weights) on the LMAX = 8 form."""
    from helpers import case_modes, load
    from test_oracle_golden import g14_case
    g = load("g14_rank_deficient")
    Xa, blocks_a, Ca = g14_case(g, "a")
    for mtag in ("B", "M"):
        for scheme in ("centroid", "path"):
            key = "a_%s_%s_1" % (mtag, scheme)
            model = orc.Model(blocks_a, Ca, case_modes(mtag, mixed="BABABA"), scheme, True)
            e = run_quad(emu, Xa, model, entry="hostemu_solve_wave16_l8")
            assert e is not None and e["status"] == 0 and e["iterations"] == int(g[key + "/iters"]), key
            assert_close(e["weights"], g[key + "/weights"], RTOL, what=key)


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_wave16_source_at_32_lvs(emu, scheme):
    """solve_problem_wave16<32> (sixteen matrix entries per pair lane): all-Mode-A models of 17 ... 32 LVs against the oracle, the LDS variant and a bootstrap replicate."""
    for sizes, fan in (([3] * 20, 2), ([2] * 32, 1), ([1] * 10 + [4] * 7, 5), ([5, 1, 2, 3, 1, 4, 2, 1, 3, 2, 1, 1, 6, 2, 3, 1, 2, 4, 1, 2, 3, 1, 1, 2], 3)):
        L = len(sizes)
        C = _dag(L, fan)
        X, blocks = _shaped(C, sizes, seed=31, N=600)
        model = orc.Model(blocks, C, "A" * L, scheme, True)
        e = run_quad(emu, X, model, entry="hostemu_solve_wave16_l32")
        assert e is not None, sizes
        check(e, orc.fit(X, model), "wave16<32> L=%d %s" % (L, scheme))
        base = run_emu(emu, X, model)
        assert e["iterations"] == base["iterations"]
        assert_close(e["row"], base["row"], 1e-10, 1e-13)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, X.shape[0], X.shape[0])
    e = run_quad(emu, X, model, counts=np.bincount(idx, minlength=X.shape[0]), shift=X[:, model.mv_order].mean(axis=0), entry="hostemu_solve_wave16_l32")
    mine, its = orc.bootstrap_replicate(X, model, idx, orc.correction(X.shape[0]))
    assert e["status"] == 0 and e["iterations"] == its
    assert_close(np.concatenate((e["weights"], e["r2"], e["total"], e["direct"], e["loadings"])), mine, RTOL, 1e-12)
    Xs, bs = _shaped(orc.chain_C(16), [4] * 16, seed=2)
    assert run_quad(emu, Xs, orc.Model(bs, orc.chain_C(16), "A" * 16, scheme, True), entry="hostemu_solve_wave16_l32") is None      # 16 LVs: the <16> form's
