"""GPU tests of the one-launch Scale.NUM / RAW bootstrap (round 6: csrc/solver_wave16.h NM -> kernels_solver.h solver_nmwave_kernel, verification in
csrc/plspm_nonmetric.hip run_nonmetric_wave): prepare + every step + finish of a replicate in one launch that stops on the quadratic upper bound of the
reference's score criterion (weights.py:120) and continues speculatively otherwise; the steps it continued behind are verified on the observations (a lower
bound from the first rows, the exact fixed-order sum for what that leaves open), and a replicate whose exact criterion was below the tolerance although its
bound was not is replayed with the reference's stop.  Held against: the oracle on the mirrored Philox draws (rtol 1e-8, equal iteration counts), the
per-iteration launches of rounds 1-5 (set_option "nm_wave16" 0: equal status / iteration counts, rows to 1e-9), and -- with the test seams that make the
lower bound fail ("nm_verify_rows") and the solver overshoot ("nm_bound_shift": the bound times 2^k is still an upper bound) -- the exact pass and the replay."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import assert_close
from test_gpu_parity import SCHEME_ID

pytestmark = pytest.mark.gpu


def _handle(X, blocks, C, modes, scheme, tol=1e-6, max_iter=100):
    from plspm import _native
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    md = np.array([0 if m == "A" else 1 for m in modes], dtype=np.int32)
    nm = _native.NativeModel(boff, C.astype(np.uint8), md, SCHEME_ID[scheme], True, max_iter, tol, 0, nonmetric=True)
    nm.upload(X)
    nm.set_option("gram_path", 2)
    return nm


def _oracle_rows(nm, X, model, seed, reps, rows, status, iters):
    from plspm import _native
    n = X.shape[0]
    corr = orc.correction(n)
    for b in reps:
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(seed, b, n), corr)
        assert status[b] == 0 and its == iters[b], (b, its, iters[b])
        assert_close(rows[b], mine, 1e-8, 1e-11, what="replicate %d" % b)


CASES = [("path", "AAAAAA", 6, 10, 4000), ("centroid", "AAAAAA", 6, 10, 1500), ("factorial", "ABABAB", 6, 6, 1500), ("path", "AAAAAAAAAAAA", 12, 5, 1500),
         ("centroid", "ABBAABBAAB", 10, 6, 900), ("factorial", "AA", 2, 7, 700), ("path", "A" * 20, 20, 3, 1200)]


@pytest.mark.parametrize("scheme,modes,L,per,n", CASES)
def test_wave_route_vs_oracle_and_per_iteration_launches(scheme, modes, L, per, n):
    C = orc.satisfaction_C() if L == 6 else orc.chain_C(L)
    X, blocks = orc.synth(n, C, per, seed=41)
    model = orc.Model(blocks, C, modes, scheme, True, tol=1e-6, scales=["NUM"] * X.shape[1])
    nm = _handle(X, blocks, C, modes, scheme)
    B = 600
    rows, status, iters = nm.bootstrap(B, seed=17)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_nm_wave16") == 1 and nm.get_option("last_solver") == (9 if L <= 8 else (10 if L <= 16 else 11))
    assert np.all(status == 0)
    _oracle_rows(nm, X, model, 17, (0, 1, B // 2, B - 1), rows, status, iters)
    nm.set_option("nm_wave16", 0)
    rows_l, status_l, iters_l = nm.bootstrap(B, seed=17)
    assert nm.get_option("last_nm_wave16") == 0
    assert np.array_equal(status, status_l) and np.array_equal(iters, iters_l)
    assert_close(rows, rows_l, 1e-9, 1e-12)


def test_headline_shape_5000_replicates_and_sub_batch_independence():
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    C = orc.satisfaction_C()
    model = orc.Model(blocks, C, "AAAAAA", "path", True, tol=1e-6, scales=["NUM"] * 60)
    nm = _handle(X, blocks, C, "AAAAAA", "path")
    rows, status, iters = nm.bootstrap(5000, seed=1)
    assert nm.get_option("last_nm_wave16") == 1 and np.all(status == 0)
    assert nm.get_option("last_nm_flagged") == 0 and nm.get_option("last_nm_replayed") == 0        # the lower bound of an eighth of the rows confirms every step
    _oracle_rows(nm, X, model, 1, (0, 4999), rows, status, iters)
    part, st_p, it_p = nm.bootstrap(700, seed=1, rep_offset=1000)                                  # a replicate's record does not depend on the batch it travels in
    assert np.array_equal(part, rows[1000:1700]) and np.array_equal(it_p, iters[1000:1700])


def test_exact_pass_when_the_lower_bound_is_inconclusive():
    """One percent of the rows (at least 128) cannot lift step 2 of these replicates over the tolerance: the (replicate, step) pairs go through the exact pass
    over all rows, which confirms them -- same records, nothing replayed."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(10000, C, 5, seed=3)
    nm = _handle(X, blocks, C, "AAAAAA", "path", tol=1e-5)
    rows, status, iters = nm.bootstrap(500, seed=5)
    assert nm.get_option("last_nm_flagged") == 0
    nm.set_option("nm_verify_rows", 1)
    rows_b, status_b, iters_b = nm.bootstrap(500, seed=5)
    assert nm.get_option("last_nm_flagged") > 0 and nm.get_option("last_nm_replayed") == 0
    assert np.array_equal(rows, rows_b) and np.array_equal(iters, iters_b) and np.array_equal(status, status_b)


@pytest.mark.parametrize("shift", [20, 60])
def test_replay_when_the_bound_overshoots(shift):
    """nm_bound_shift k: the solver stops on bound x 2^k < tol -- a valid but useless bound, so every replicate runs on behind the reference's stop; the
    verification finds the first step whose exact criterion is below the tolerance and the replay stops there: records, status and iteration counts of the
    unshifted run (where the bound decides) and of the oracle."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(3000, C, 6, seed=8)
    model = orc.Model(blocks, C, "AABAAB", "centroid", True, tol=1e-6, scales=["NUM"] * X.shape[1])
    nm = _handle(X, blocks, C, "AABAAB", "centroid")
    B = 300
    rows, status, iters = nm.bootstrap(B, seed=9)
    assert nm.get_option("last_nm_replayed") == 0
    nm.set_option("nm_bound_shift", shift)
    rows_s, status_s, iters_s = nm.bootstrap(B, seed=9)
    assert nm.get_option("last_nm_replayed") == B and nm.get_option("last_nm_flagged") >= B
    assert np.array_equal(status, status_s) and np.array_equal(iters, iters_s)
    assert_close(rows_s, rows, 1e-12, 1e-14)
    _oracle_rows(nm, X, model, 9, (0, B - 1), rows_s, status_s, iters_s)


def test_not_converged_and_many_steps():
    """A tolerance nothing reaches within max_iter: every replicate is reported PLSPM_NOT_CONVERGED at max_iter + 1 steps (weights.py:183-186), as on the
    per-iteration launches; the verification walks all its steps in rounds."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(800, C, 3, seed=2)
    nm = _handle(X, blocks, C, "AAAAAA", "factorial", tol=1e-300, max_iter=9)
    rows, status, iters = nm.bootstrap(200, seed=3)
    assert nm.get_option("last_nm_wave16") == 1
    nm.set_option("nm_wave16", 0)
    rows_l, status_l, iters_l = nm.bootstrap(200, seed=3)
    assert np.array_equal(status, status_l) and np.array_equal(iters, iters_l)
    assert np.all(status[iters == 10] == 1)
    ok = status == 0
    assert_close(rows[ok], rows_l[ok], 1e-9, 1e-12)


def test_plspm_api_bootstrap_on_num_data_takes_the_wave_route():
    import pandas as pd
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    X, blocks = orc.synth(3000, orc.satisfaction_C(), 5, seed=12)
    cols = ["%s%d" % (lv.lower(), k) for lv in orc.SAT_LVS for k in range(5)]
    frame = pd.DataFrame(X, columns=cols)
    structure = c.Structure()
    for frm, to in orc.SAT_EDGES:
        structure.add_path([frm], [to])
    cfg = c.Config(structure.path(), default_scale=Scale.NUM)
    for lv in orc.SAT_LVS:
        cfg.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
    m = Plspm(frame, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=2000, processes=1, seed=4)
    w = m.bootstrap().weights()
    assert m.bootstrap().used() == 2000 and np.all(np.isfinite(w.values))
    assert_close(w["original"].values, m.outer_model().loc[w.index, "weight"].values, 1e-12)
    assert np.all(np.abs(w["mean"].values - w["original"].values) < 5 * w["std.error"].values + 1e-9)
