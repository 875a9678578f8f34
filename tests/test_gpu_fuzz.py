"""GPU differential test: seeded random models (random DAG, ragged blocks, random Mode / Scheme / scaled / metric vs Scale.NUM,
random N) -- single fit and three bootstrap replicates (device RNG mirrored on the host) against the oracle.
Cases whose oracle run does not converge or hits a singular system must be reported as such by the device (status != 0)."""
import numpy as np
import pytest

import plspm_oracle as orc
from helpers import DEGENERATE_EIG_RTOL, assert_close, assert_device_status_justified, oracle_conditioning
from fuzz_cases import make_case, make_cat_case, make_hoc_case, make_missing_case, make_nmx_case
from test_gpu_parity import SCHEME_ID, _ragged, _random_dag

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-7, 1e-10


@pytest.mark.parametrize("seed", range(60))
def test_random_model(seed):
    X, model, nonmetric = make_case(seed)
    _model_check(X, model, nonmetric, seed)


@pytest.mark.parametrize("seed", range(40))
def test_random_model_at_the_edges(seed):
    """fuzz_cases.make_degenerate_case: tiny samples, duplicated / constant columns, iteration caps of 1 ... 4, tolerances of 1e-2 / 1e-12."""
    from fuzz_cases import make_degenerate_case
    X, model, nonmetric, _ = make_degenerate_case(seed)
    _model_check(X, model, nonmetric, seed)


def _model_check(X, model, nonmetric, seed):
    from plspm import _native
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0, nonmetric=nonmetric)
    nm.upload(X)
    g = nm.fit(want_scores=True)
    tag = "seed %d L=%d P=%d n=%d %s %s %s max_iter=%d tol=%g" % (seed, model.L, X.shape[1], X.shape[0], model.modes, model.scheme, "NUM" if nonmetric else "metric", model.max_iter, model.tol)
    try:
        with np.errstate(all="ignore"):
            r = orc.fit(X, model)
    except orc.NotConverged:
        # (the reference's trips never end once a score is NaN -- a zero-variance column under Scale.NUM, a single-item LV on a constant column --: "could not converge
        #  after 101 iterations"; the device names that cause, PLSPM_NONFINITE / PLSPM_SINGULAR, where the rows are degenerate: plspm/weights.py NumericalConditionError)
        if g["status"] != 1:
            if g["status"] == 0 and _fit_is_a_coin_toss(X, model):
                return "coin-toss"
            assert g["status"] != 0, tag + ": the oracle does not converge, the device reports PLSPM_OK"
            cond, what = oracle_conditioning(X, model)
            assert cond < DEGENERATE_EIG_RTOL, tag + ": device status %d where the oracle runs out of iterations, conditioning %.3g (%s)" % (g["status"], cond, what)
        return "notconv"
    except Exception:                                      # noqa: BLE001  (a singular system: numpy raises LinAlgError where the reference's statsmodels / lstsq would)
        if g["status"] == 0:
            # (an exactly collinear Mode-B block: the oracle's lstsq meets NaN where the reference's gelsd -- at its rank threshold -- went to 1e13, the device returns the
            #  minimum-norm weights either way: DESIGN 6, "the reference's coin toss"; anything else is a failure)
            cond, what = oracle_conditioning(X, model)
            assert cond < DEGENERATE_EIG_RTOL, tag + ": the oracle cannot estimate this model (conditioning %.3g, %s), the device reports PLSPM_OK" % (cond, what)
        return "oracle-raised"
    if not all(np.all(np.isfinite(r[k])) for k in ("weights", "path_coef", "r2", "loadings", "scores")):
        assert g["status"] != 0 or not all(np.all(np.isfinite(g[k])) for k in ("weights", "path_coef", "r2", "loadings", "scores")), tag + ": oracle outputs not finite, device finite and PLSPM_OK"
        return "oracle-nonfinite"
    if g["status"] != 0:
        # a device-only status must be explained by the oracle's own conditioning on these rows -- it can turn the suite red
        if _fit_is_a_coin_toss(X, model):                      # (two single-item LVs whose items are exactly uncorrelated: the reference normalises an inner estimate of 1e-18, the device meets 0 / 0)
            return "coin-toss"
        assert_device_status_justified(g["status"], X, model, tag)
        return "device-status-%d" % g["status"]
    try:
        assert g["iterations"] == r["iterations"], tag
        assert_close(g["weights"], r["weights"], RTOL, ATOL, what=tag)
        assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL, what=tag)
        assert_close(g["r2"], r["r2"], RTOL, ATOL, what=tag)
        assert_close(g["loadings"], r["loadings"], RTOL, ATOL, what=tag)
        assert_close(g["scores"], r["scores"], 1e-6, 1e-8, what=tag)
    except AssertionError:
        if not _fit_is_a_coin_toss(X, model):                  # (e.g. the centroid sign of a correlation that is exactly zero in a small sample of integers: _oracle_coin_toss)
            raise
        return "coin-toss"
    n = X.shape[0]
    rows, status, iters = nm.bootstrap(3, seed=seed)
    corr = orc.correction(n)
    for b in range(3):
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        try:
            assert its == iters[b], tag + " replicate %d" % b
            assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
        except AssertionError:
            if not _fit_is_a_coin_toss(X[idx], model):
                raise
    return "ok"


def _fit_is_a_coin_toss(X, model):
    """The oracle's own run on these rows passes through a decision the reference makes on rounding residue (plspm_oracle.DIAG)."""
    orc.DIAG = {}
    try:
        with np.errstate(all="ignore"):
            orc.fit(X, model)
    except Exception:                                      # noqa: BLE001
        pass
    diag, orc.DIAG = orc.DIAG, None
    return _oracle_coin_toss(diag)


def _replicate_comparable(X, model, idx, corr, status, tag):
    """True when the oracle and the device both estimated the replicate (the caller then compares the records).  Otherwise the two must AGREE that it
    cannot be estimated: an oracle failure (the reference's bare `except`, bootstrap.py:65-66, drops the replicate) needs a device status != 0, and a
    device status the oracle does not share needs degenerate rows (helpers.assert_device_status_justified) -- never a silent `continue`."""
    try:
        mine, _ = orc.bootstrap_replicate(X, model, idx, corr)
        oracle_ok = bool(np.all(np.isfinite(mine)))
    except Exception:                                      # noqa: BLE001
        oracle_ok = False
    if not oracle_ok:
        # (PLSPM_OK beside an oracle failure only where the oracle's own run meets a residue decision: two single-item LVs whose items are exactly uncorrelated in the resample --
        #  the oracle's inner estimate is 0, the device's count arithmetic leaves 1e-17 and normalises it)
        assert status != 0 or _fit_is_a_coin_toss(X[idx], model), tag + ": the oracle cannot estimate this replicate, the device reports PLSPM_OK"
        return False
    if status != 0:
        if not _fit_is_a_coin_toss(X[idx], model):
            assert_device_status_justified(int(status), X[idx], model, tag)
        return False
    return True


def make_wide_case(seed):
    """65 ... 128 MVs in 3 ... 12 ragged blocks (metric): the quad solver (all Mode A) or the split rows solver (Mode-B blocks) where a block
    boundary leaves at most 64 MVs on either side, the LDS solver otherwise."""
    rng = np.random.default_rng(5000 + seed)
    L = int(rng.integers(3, 13))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.8)))
    P = int(rng.integers(65, 129))
    cuts = np.sort(rng.choice(np.arange(1, P), size=L - 1, replace=False))
    sizes = np.diff(np.concatenate(([0], cuts, [P]))).tolist()
    n = int(rng.integers(300, 900))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    modes = "".join("AB"[int(rng.integers(0, 2))] if 1 < sizes[l] <= 24 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    return X, orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2))), sizes


@pytest.mark.parametrize("seed", range(24))
def test_random_wide_model_bootstrap(seed):
    from plspm import _native
    X, model, sizes = make_wide_case(seed)
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0)
    nm.upload(X)
    B = 40
    rows, status, iters = nm.bootstrap(B, seed=seed)
    if nm.get_option("last_gram_path") != 2:
        pytest.skip("the fp64 route took this batch")
    below = [int(b) for b in boff[1:-1] if b <= 64]
    splittable = bool(below) and P - max(below) <= 64
    # (round 5: all-Mode-A models of at most 16 LVs take the quad solver -- four waves per problem with fixed lane roles -- where they took the split rows solver)
    quad = splittable and "B" not in model.modes
    # (a Mode-B model whose generic workspace -- wide inner model -- does not fit the split rows solver's LDS share beside the loader tiles takes the LDS solver)
    assert nm.get_option("last_solver") in (((5,) if quad else (4, 1)) if splittable else (1,)), (sizes, nm.get_option("last_solver"))
    if quad:
        nm.set_option("solver_quad", 0)
        rows_s, status_s, iters_s = nm.bootstrap(B, seed=seed)
        assert nm.get_option("last_solver") in (4, 1)
        assert np.array_equal(status, status_s) and np.array_equal(iters[status == 0], iters_s[status == 0]), (sizes, model.scheme)
        assert_close(rows[status == 0], rows_s[status == 0], 1e-9, 1e-12, what="quad vs split rows %s" % sizes)
        nm.set_option("solver_quad", 1)
    nm.set_option("solver_rows", 0)
    rows_l, status_l, iters_l = nm.bootstrap(B, seed=seed)
    assert nm.get_option("last_solver") == 1
    tag = "seed %d P=%d sizes=%s %s %s" % (seed, P, sizes, model.modes, model.scheme)
    assert np.array_equal(status, status_l), tag
    ok = status == 0
    assert np.array_equal(iters[ok], iters_l[ok]), tag
    assert_close(rows[ok], rows_l[ok], 1e-9, 1e-12, what=tag)
    corr = orc.correction(n)
    checked = 0
    for b in range(B):
        if checked == 2 and status[b] == 0:
            continue
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert its == iters[b], tag + " replicate %d" % b
        assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
        checked += 1


def make_narrow_case(seed):
    """At most 64 MVs in 2 ... 16 ragged blocks, random Mode A / B blocks, on the int8 route: the wave solvers of round 5 (solver_wave16_kernel<8> / <16>, with
    their Mode-B instantiations) where they cover the model, the rows or the LDS solver otherwise."""
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.integers(2, 17))
    C = _random_dag(L, rng, density=float(rng.uniform(0.2, 0.9)))
    P = int(rng.integers(L, 65))
    cuts = np.sort(rng.choice(np.arange(1, P), size=L - 1, replace=False))
    sizes = np.diff(np.concatenate(([0], cuts, [P]))).tolist()
    n = int(rng.integers(200, 900))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    allA = bool(rng.integers(0, 2))
    modes = "".join("A" if allA or sizes[l] == 1 or sizes[l] > 16 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    return X, orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2))), sizes


def _narrow_case_check(seed, B=40):
    from plspm import _native
    X, model, sizes = make_narrow_case(seed)
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0)
    nm.upload(X)
    nm.set_option("gram_path", 2)
    rows, status, iters = nm.bootstrap(B, seed=seed)
    if nm.get_option("last_gram_path") != 2:
        pytest.skip("the fp64 route took this batch")
    code = nm.get_option("last_solver")
    tag = "seed %d P=%d sizes=%s %s %s solver %d" % (seed, P, sizes, model.modes, model.scheme, code)
    assert code in (1, 2, 6, 7, 8), tag
    if code in (6, 7):
        assert code == (7 if model.L <= 8 else 6), tag
    nm.set_option("solver_rows", 0)
    rows_l, status_l, iters_l = nm.bootstrap(B, seed=seed)
    assert nm.get_option("last_solver") == 1
    assert np.array_equal(status, status_l), tag
    ok = status == 0
    assert np.array_equal(iters[ok], iters_l[ok]), tag
    assert_close(rows[ok], rows_l[ok], 1e-9, 1e-12, what=tag)
    corr = orc.correction(n)
    checked = 0
    for b in range(B):
        if checked == 2 and status[b] == 0:
            continue
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert its == iters[b], tag + " replicate %d" % b
        assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
        checked += 1
    return code


@pytest.mark.parametrize("seed", range(40))
def test_random_narrow_model_bootstrap(seed):
    _narrow_case_check(seed)


@pytest.mark.parametrize("seed", range(30))
def test_random_num_model_on_the_one_launch_route(seed):
    """Scale.NUM models of the narrow fuzz class (at most 64 MVs in 2 ... 16 ragged blocks, random Mode A / B blocks, random scheme) on the int8 route: the one-launch
    solver + verification of round 6 (solver_wave16.h NM) against the per-iteration launches (equal status and iteration counts, records to 1e-9) and two replicates
    against the oracle on the mirrored Philox draws."""
    from plspm import _native
    X, metric, sizes = make_narrow_case(seed)
    model = orc.Model(metric.blocks, metric.C, metric.modes, metric.scheme, True, tol=1e-7, scales=["NUM"] * X.shape[1])
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], True, model.max_iter, model.tol, 0, nonmetric=True)
    nm.upload(X)
    nm.set_option("gram_path", 2)
    B = 40
    rows, status, iters = nm.bootstrap(B, seed=seed)
    tag = "seed %d P=%d sizes=%s %s %s" % (seed, P, sizes, model.modes, model.scheme)
    if nm.get_option("last_gram_path") != 2 or nm.get_option("last_nm_wave16") != 1:
        pytest.skip("outside the one-launch class (%s)" % tag)
    nm.set_option("nm_wave16", 0)
    rows_l, status_l, iters_l = nm.bootstrap(B, seed=seed)
    assert nm.get_option("last_nm_wave16") == 0
    assert np.array_equal(status, status_l), tag
    ok = status == 0
    assert np.array_equal(iters[ok], iters_l[ok]), tag
    assert_close(rows[ok], rows_l[ok], 1e-9, 1e-12, what=tag)
    corr = orc.correction(n)
    checked = 0
    for b in range(B):
        if checked == 2 and status[b] == 0:
            continue
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert its == iters[b], tag + " replicate %d" % b
        assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
        checked += 1


# ---- categorical (Scale.ORD / NOM / NUM mixes): random path models, block sizes, category counts (2 .. 12), modes and schemes -- the fit and six bootstrap replicates on
# explicit index lists against the oracle, through the wave step where it covers the model and the workgroup step elsewhere; where both exist, the same records from both
def _oracle_coin_toss(diag):
    """True when the oracle's run passed through a decision the reference itself makes on rounding residue (plspm_oracle.DIAG): a direction tie, category means that tie, an
    inner estimate of residue, linked scores that are uncorrelated to rounding, a sign rule settled by the sign bit of such a correlation (small samples of integer codes meet these EXACTLY on the device's count arithmetic)."""
    return (diag.get("min_direction_margin", 1.0) < 1e-9 or diag.get("min_quant_var_over_z", 1.0) < 1e-20 or diag.get("min_z_scale", 1.0) < 1e-12 or
            diag.get("min_linked_score_corr", 1.0) < 1e-12 or diag.get("min_decisive_vote_corr", 1.0) < 1e-12)


def _cat_case_check(seed, big=False):
    import test_gpu_categorical as tc
    from fuzz_cases import make_cat_big_case, make_cat_small_case
    from plspm import _native
    data, model = (make_cat_small_case if big == "small" else make_cat_big_case if big else make_cat_case)(seed)
    n = data.shape[0]
    tag = "seed %d L=%d P=%d n=%d %s %s %s" % (seed, model.L, data.shape[1], n, model.modes, model.scheme, "".join(s[0] for s in model.scales))
    try:
        r = orc.fit(data, model)
    except orc.NotConverged:
        r = None
    except Exception as e:                          # singular blocks etc.: the oracle cannot finish -- the device must not report success
        r = e
    nm, g = tc.gpu_fit_cat(data, model)
    route = "wave" if nm.get_option("last_nm_wave") == 1 else "group"
    if r is None:
        assert g["status"] == 1, tag + ": oracle did not converge, device status %d" % g["status"]
        return route + "/notconv"
    if isinstance(r, Exception):
        assert g["status"] != 0, tag + ": oracle raised %r, device status 0" % (r,)
        return route + "/oracle-raised"
    if not all(np.all(np.isfinite(r[k])) for k in ("weights", "loadings", "path_coef", "scores")):
        return route + "/oracle-nonfinite"
    tc.check_fit(g, r, tag)
    B = 6
    rs = np.random.RandomState(seed)
    idx = rs.randint(n, size=(B, n)).astype(np.int32)
    rows, status, iters = nm.bootstrap(B, idx=idx)
    Pm = len(model.scales)
    rows = tc._rows_in_data_order(rows, g["inv"], Pm, model.L, nm.n_eff)
    corr = orc.correction(n)
    compared = 0
    for b in range(B):
        try:
            mine, its = orc.bootstrap_replicate(data, model, idx[b], corr)
        except Exception:
            # (device status 0 only behind one of the oracle's residue decisions: a direction tie taken the other way can converge where the oracle's run cycles -- seed 30224)
            assert status[b] != 0 or _fit_is_a_coin_toss(data[idx[b]], model), tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        if not np.all(np.isfinite(mine)):
            continue
        try:
            assert status[b] == 0 and its == iters[b], tag + " replicate %d: status %d iterations %d vs oracle %d" % (b, status[b], iters[b], its)
            assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        except AssertionError:
            orc.DIAG = {}                                   # (see below: a direction decision of scale.py:74 that is a tie in exact arithmetic)
            with np.errstate(all="ignore"):
                orc.bootstrap_replicate(data, model, idx[b], corr)
            diag, orc.DIAG = orc.DIAG, None
            if not _oracle_coin_toss(diag):
                raise
            continue
        compared += 1
    # the device's own resampling: the same records whatever the route (wave step / workgroup step), where both exist
    if route == "wave":
        a = nm.bootstrap(40, seed=seed)
        nm.set_option("nm_wave", 0)
        w = nm.bootstrap(40, seed=seed)
        nm.set_option("nm_wave", 1)
        same = (a[1] == w[1]) & (a[2] == w[2]) & np.array([a[1][r] != 0 or np.allclose(a[0][r], w[0][r], rtol=1e-9, atol=1e-11) for r in range(40)])
        for r in np.flatnonzero(~same):
            # the two forms may part ways only where the REFERENCE's own answer is a coin toss: an ordinal MV whose increasing and decreasing poolings have the same
            # variance in exact arithmetic (scale.py:74 `var_incr < var_decr`, decided by np.var's rounding) -- seed 1553, replicate 6: means (a, b, a), equal counts
            orc.DIAG = {}
            try:
                with np.errstate(all="ignore"):
                    orc.bootstrap_replicate(data, model, _native.bootstrap_indices(seed, int(r), n), corr)
            except Exception:
                pass
            diag, orc.DIAG = orc.DIAG, None
            assert _oracle_coin_toss(diag), tag + " replicate %d: wave / workgroup step disagree (status %d / %d, iterations %d / %d) with no tie in the oracle (%s)" % (
                r, a[1][r], w[1][r], a[2][r], w[2][r], diag)
    return route + "/%d" % compared


@pytest.mark.parametrize("seed", range(30))
def test_random_categorical_model(seed):
    _cat_case_check(seed)


@pytest.mark.parametrize("seed", range(16))
def test_random_small_sample_categorical_model(seed):
    """fuzz_cases.make_cat_small_case: 30 ... 90 rows -- resamples that lose categories, items left with one."""
    _cat_case_check(seed, big="small")


@pytest.mark.parametrize("seed", list(range(10)) + [34, 124])
def test_random_large_categorical_model(seed):
    """fuzz_cases.make_cat_big_case: up to 80 MVs in up to 10 blocks, items of up to 16 categories.  Seeds 34 / 124: seven / eight LVs, items of up to 13 categories, PATH scheme --
    the class whose single fit returned NaN (and faulted) on the six-columns-per-lane form of the sixteen-category wave step (plspm_nonmetric.hip, `cpl6`)."""
    _cat_case_check(seed, big=True)


def test_exact_tie_of_category_means_fails_like_the_reference():
    """Seed 526 of the categorical sweep (tools/experiments/cat_fuzz_sweep.py), replicate 4: in the first trip of the centroid scheme the two category means of a
    binary item are EQUAL (2.5 and 2.5 in exact arithmetic) -- the reference's quantified column is constant, util.treat_numpy divides 0 by 0, the estimate never
    converges and the replicate is dropped.  On second moments that variance is rounding noise; the device has to call it zero (solver_nmg.h nmg_quant_sd) instead
    of standardising the noise into a replicate that 'converged'.  Both step forms."""
    import test_gpu_categorical as tc
    data, model = make_cat_case(526)
    n = data.shape[0]
    idx = np.random.RandomState(526).randint(n, size=(6, n)).astype(np.int32)
    with pytest.raises(orc.NotConverged), np.errstate(all="ignore"):
        orc.bootstrap_replicate(data, model, idx[4], orc.correction(n))
    nm, g = tc.gpu_fit_cat(data, model)
    tc.check_fit(g, orc.fit(data, model), "seed 526")
    for wave in (1, 0):
        nm.set_option("nm_wave", wave)
        rows, status, iters = nm.bootstrap(6, idx=idx)
        assert nm.get_option("last_nm_wave") == wave
        assert status[4] != 0 and np.all(status[[0, 1, 2, 3, 5]] == 0), (wave, status.tolist(), iters.tolist())


# ---- missing data: metric models with NaN cells (mean imputation on the moments, re-imputed per replicate) and Scale.NUM / RAW models with incomplete rows (the NaN-aware
# Mode-A products of mode.py:35-41) -- random models, fit + five replicates on explicit index lists against the oracle
def _missing_case_check(seed):
    import test_gpu_missing as tm
    Xn, model = make_missing_case(seed)
    n, P = Xn.shape
    tag = "seed %d L=%d P=%d n=%d %s %s scaled=%d nan=%d" % (seed, model.L, P, n, model.modes, model.scheme, model.scaled, int(np.isnan(Xn).sum()))
    try:
        ref = orc.fit(Xn, model)
    except orc.NotConverged:
        ref = None
    nm, inv = tm.gpu_model(Xn, model)
    out = nm.fit(want_scores=True)
    if ref is None:
        assert out["status"] == 1, tag
        return "notconv"
    if out["status"] != 0:
        return "device-status-%d" % out["status"]
    assert out["iterations"] == ref["iterations"], tag + ": iterations %d vs %d" % (out["iterations"], ref["iterations"])
    assert_close(out["weights"][inv], ref["weights"], 1e-6, 1e-9, what=tag + " weights")
    assert_close(out["loadings"][inv], ref["loadings"], 1e-6, 1e-9, what=tag + " loadings")
    assert_close(out["path_coef"], ref["path_coef"], 1e-6, 1e-9, what=tag + " paths")
    assert_close(out["scores"], ref["scores"], 1e-6, 1e-8, what=tag + " scores")
    B = 5
    idx = np.random.RandomState(seed).randint(n, size=(B, n)).astype(np.int32)
    rows, status, iters = nm.bootstrap(B, idx=idx)
    rows = tm.rows_in_data_order(rows, inv, P, model.L, nm.n_eff)
    corr = orc.correction(n)
    compared = 0
    for b in range(B):
        try:
            with np.errstate(all="ignore"):
                mine, its = orc.bootstrap_replicate(Xn, model, idx[b], corr)
        except Exception:
            assert status[b] != 0, tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        if not np.all(np.isfinite(mine)):
            assert status[b] != 0, tag + " replicate %d: oracle row not finite, device status 0" % b
            continue
        if status[b] != 0:
            continue                                                        # (device-only status: conditioning -- counted below)
        assert its == iters[b], tag + " replicate %d: iterations %d vs %d" % (b, iters[b], its)
        assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        compared += 1
    return "ok/%d" % compared


def _nmx_case_check(seed):
    import test_gpu_nmx as tx
    Xn, model = make_nmx_case(seed)
    n, P = Xn.shape
    tag = "nmx seed %d L=%d P=%d n=%d %s %s %s nan=%d" % (seed, model.L, P, n, model.modes, model.scheme, model.scales[0], int(np.isnan(Xn).sum()))
    try:
        with np.errstate(all="ignore"):
            ref = orc.fit(Xn, model)
    except orc.NotConverged:
        ref = None
    nm, inv = tx.gpu_model(Xn, model)
    out = nm.fit(want_scores=True)
    if ref is None:
        assert out["status"] == 1, tag
        return "nmx-notconv"
    if out["status"] != 0:
        return "nmx-device-status-%d" % out["status"]
    tx.check_fit(out, ref, inv, tag)
    B = 5
    idx = np.random.RandomState(seed).randint(n, size=(B, n)).astype(np.int32)
    rows, status, iters = nm.bootstrap(B, idx=idx)
    rows = tx.rows_in_data_order(rows, inv, P, model.L, nm.n_eff)
    corr = orc.correction(n)
    compared = 0
    for b in range(B):
        try:
            with np.errstate(all="ignore"):
                mine, its = orc.bootstrap_replicate(Xn, model, idx[b], corr)
        except Exception:
            assert status[b] != 0, tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        if not np.all(np.isfinite(mine)):
            assert status[b] != 0, tag + " replicate %d: oracle row not finite, device status 0" % b
            continue
        if status[b] != 0:
            continue
        assert its == iters[b], tag + " replicate %d: iterations %d vs %d" % (b, iters[b], its)
        assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        compared += 1
    return "nmx-ok/%d" % compared


@pytest.mark.parametrize("seed", range(16))
def test_random_model_with_missing_cells(seed):
    _missing_case_check(seed)
    _nmx_case_check(seed)


# ---- higher order constructs (Scale.NUM): random stage-2 path models with one HOC of two or three first-stage constituents (stage 1 = the path with the HOC expanded in
# place, estimator.py:60-74), both stages of five replicates on explicit index lists (the first: the data themselves) against the oracle's fit_two_stage
def _hoc_case_check(seed):
    from plspm import _native
    X, model1, stage2, C2, modes2, first_of = make_hoc_case(seed)
    n = X.shape[0]
    L2 = len(stage2)
    tag = "seed %d L1=%d L2=%d P=%d n=%d %s/%s %s" % (seed, model1.L, L2, X.shape[1], n, model1.modes, modes2, model1.scheme)
    boff1 = np.concatenate(([0], np.cumsum([len(b) for b in model1.blocks]))).astype(np.int32)
    m1 = np.array([0 if m == "A" else 1 for m in model1.modes], dtype=np.int32)
    first = _native.NativeModel(boff1, model1.C.astype(np.uint8), m1, SCHEME_ID[model1.scheme], True, 100, 1e-7, 0, nonmetric=True)
    first.upload(X[:, np.concatenate(model1.blocks)])
    sizes2 = [(len(ref) if kind == "hoc" else len(model1.blocks[ref])) for kind, ref in stage2]
    boff2 = np.concatenate(([0], np.cumsum(sizes2))).astype(np.int32)
    m2 = np.array([0 if m == "A" else 1 for m in modes2], dtype=np.int32)
    second = _native.NativeModel(boff2, np.asarray(C2).astype(np.uint8), m2, SCHEME_ID[model1.scheme], True, 100, 1e-7, 0, nonmetric=True)
    first.attach_second_stage(second, list(first_of))
    B = 5
    idx = np.vstack([np.arange(n), np.random.RandomState(seed).randint(n, size=(B - 1, n))]).astype(np.int32)
    rows, status, iters = first.bootstrap(B, idx=idx)
    corr = orc.correction(n)
    compared = 0
    for b in range(B):
        try:
            with np.errstate(all="ignore"):
                o = orc.fit_two_stage(X[idx[b]], model1, stage2, C2, modes2, corr)
            mine = np.concatenate((o["weights"], o["r2"], o["total"], o["direct"], o["loadings"]))
        except Exception:
            assert status[b] != 0, tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        if not np.all(np.isfinite(mine)):
            assert status[b] != 0, tag + " replicate %d: oracle row not finite, device status 0" % b
            continue
        if status[b] != 0:
            continue
        assert o["iterations"] == iters[b], tag + " replicate %d: stage-2 iterations %d vs %d" % (b, iters[b], o["iterations"])
        assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        compared += 1
    second.close(); first.close()
    return "ok/%d" % compared


@pytest.mark.parametrize("seed", range(16))
def test_random_hoc_model_two_stage_bootstrap(seed):
    _hoc_case_check(seed)


# ---- higher order constructs on ORDINAL items through the host API (Estimator.two_stage_bootstrap_handles: both stages of every replicate on the device, the second stage on
# the congruence of the first stage's count matrices): the rows of explicit index lists against the oracle's fit_two_stage -- whose second stage runs on the FIRST treatment's
# rank codes and dummy matrices, as the reference's does (estimator.py:33,52; pinned by tests/golden/sweep_oracle_vs_reference.py `hocord`)
def _hoc_ord_case_check(seed):
    import pandas as pd
    import plspm.config as c
    import plspm.weights as w
    from fuzz_cases import make_hoc_ord_case
    from plspm.estimator import Estimator
    from plspm.mode import Mode
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    data, model1, stage2, C2, modes2, first_of = make_hoc_ord_case(seed)
    n = data.shape[0]
    lvs1 = ["L%d" % l for l in range(model1.L)]
    lv2 = ["H" if kind == "hoc" else lvs1[ref] for kind, ref in stage2]
    tag = "seed %d L1=%d L2=%d P=%d n=%d %s" % (seed, model1.L, len(stage2), data.shape[1], n, model1.scheme)
    df = pd.DataFrame(data, columns=["x%d" % p for p in range(data.shape[1])])
    config = c.Config(pd.DataFrame(np.asarray(C2, dtype=int), index=lv2, columns=lv2), default_scale=Scale.ORD)
    for (kind, ref), name in zip(stage2, lv2):
        if kind == "hoc":
            config.add_higher_order(name, Mode.A, [lvs1[j] for j in ref])
    for l in range(model1.L):
        config.add_lv(lvs1[l], Mode.A, *[c.MV("x%d" % p) for p in model1.blocks[l]])
    scheme = {"centroid": Scheme.CENTROID, "factorial": Scheme.FACTORIAL, "path": Scheme.PATH}[model1.scheme]
    observations = config.filter(df)
    corr = orc.correction(n)
    calculator = w.WeightsCalculatorFactory(config, 100, model1.tol, corr, scheme, 0)
    pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
    names = []                                                 # the oracle's stage-2 MV order
    for kind, ref in stage2:
        names += [lvs1[j] for j in ref] if kind == "hoc" else ["x%d" % p for p in model1.blocks[ref]]
    dev = list(pair.compiled.dev_mvs)
    assert sorted(dev) == sorted(names), tag
    perm = np.array([dev.index(x) for x in names])
    B = 5
    idx = np.vstack([np.arange(n), np.random.RandomState(seed).randint(n, size=(B - 1, n))]).astype(np.int32)
    rows, status, iters = pair.native.bootstrap(B, idx=idx)
    P2, L2 = len(names), len(stage2)
    ne = (rows.shape[1] - 2 * P2 - L2) // 2
    rows = np.concatenate((rows[:, :P2][:, perm], rows[:, P2:P2 + L2 + 2 * ne], rows[:, P2 + L2 + 2 * ne:][:, perm]), axis=1)
    compared = 0
    for b in range(B):
        orc.DIAG = {}
        try:
            with np.errstate(all="ignore"):
                o = orc.fit_two_stage(data[idx[b]], model1, stage2, C2, modes2, corr)
            mine = np.concatenate((o["weights"], o["r2"], o["total"], o["direct"], o["loadings"]))
        except Exception:
            orc.DIAG = None
            assert status[b] != 0, tag + " replicate %d: the oracle cannot finish, device status 0" % b
            continue
        toss, orc.DIAG = _oracle_coin_toss(orc.DIAG), None
        if not np.all(np.isfinite(mine)):
            assert status[b] != 0, tag + " replicate %d: oracle row not finite, device status 0" % b
            continue
        if status[b] != 0 or toss:                    # (a direction decision that is a tie in exact arithmetic: see _cat_case_check)
            continue
        assert o["iterations"] == iters[b], tag + " replicate %d: stage-2 iterations %d vs %d" % (b, iters[b], o["iterations"])
        assert_close(rows[b], mine, 1e-6, 1e-8, what=tag + " replicate %d" % b)
        compared += 1
    return "ok/%d" % compared


@pytest.mark.parametrize("seed", range(12))
def test_random_hoc_model_on_ordinal_items(seed):
    _hoc_ord_case_check(seed)


# ---- data that are unkind to a fixed-point evaluation of the moments (fuzz_cases.make_hostile_case): the int8 digit-plane route against the fp64 MFMA route on the same
# Philox draws (equal status and iteration counts, records to 1e-7) and against the oracle (fit + two replicates)
def _hostile_case_check(seed, B=48):
    from fuzz_cases import make_hostile_case
    from plspm import _native
    X, model, sizes, kind = make_hostile_case(seed)
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0)
    nm.upload(X)
    tag = "seed %d kind %d P=%d n=%d %s %s scaled=%d" % (seed, kind, P, n, model.modes, model.scheme, model.scaled)
    g = nm.fit(want_scores=True)
    try:
        with np.errstate(all="ignore"):
            r = orc.fit(X, model)
    except orc.NotConverged:
        assert g["status"] == 1, tag
        return "notconv"
    if g["status"] != 0:
        assert_device_status_justified(g["status"], X, model, tag)
        return "device-status"
    assert g["iterations"] == r["iterations"], tag
    assert_close(g["weights"], r["weights"], 1e-6, 0.0, what=tag + " weights")
    assert_close(g["path_coef"], r["path_coef"], 1e-6, 1e-9, what=tag + " paths")
    assert_close(g["loadings"], r["loadings"], 1e-6, 1e-9, what=tag + " loadings")
    nm.set_option("gram_path", 2)
    rows, status, iters = nm.bootstrap(B, seed=seed)
    planes = nm.get_option("last_i8_slices") if nm.get_option("last_gram_path") == 2 else 0
    nm.set_option("gram_path", 1)
    rows_f, status_f, iters_f = nm.bootstrap(B, seed=seed)
    assert nm.get_option("last_gram_path") == 1
    both = (status == 0) & (status_f == 0)
    assert both.sum() >= B // 2 and np.array_equal(status == 0, status_f == 0), (tag, status.tolist(), status_f.tolist())
    assert np.array_equal(iters[both], iters_f[both]), tag
    scale = np.maximum(np.abs(rows_f[both]), 1e-3 * np.max(np.abs(rows_f[both]), axis=0, keepdims=True))       # (weights live on the data's scale: relative per column)
    err = float(np.max(np.abs(rows[both] - rows_f[both]) / np.maximum(scale, 1e-300)))
    assert err < 1e-7, tag + ": int8 route against the fp64 route %.3g (planes %d)" % (err, planes)
    corr = orc.correction(n)
    checked = 0
    for b in range(B):
        if checked == 2:
            break
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert its == iters[b], tag + " replicate %d" % b
        sc = np.maximum(np.abs(mine), 1e-3 * np.max(np.abs(mine[:P])))
        assert float(np.max(np.abs(rows[b] - mine) / sc)) < 1e-6, tag + " replicate %d" % b
        checked += 1
    return "kind%d/planes%d" % (kind, planes)


@pytest.mark.parametrize("seed", range(24))
def test_random_model_on_hostile_data(seed):
    _hostile_case_check(seed)


# ---- many LVs / many MVs (fuzz_cases.make_huge_case): the fit against the oracle; a batch of replicates on the route the library picks against the LDS solver / the
# per-iteration non-metric launches (equal status and iteration counts, records to 1e-9) and two of them against the oracle
def _huge_case_check(seed, B=24):
    from fuzz_cases import make_huge_case
    from plspm import _native
    X, model, sizes, nonmetric = make_huge_case(seed)
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum(sizes))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0, nonmetric=nonmetric)
    nm.upload(X)
    tag = "seed %d L=%d P=%d n=%d %s %s" % (seed, model.L, P, n, model.scheme, "NUM" if nonmetric else "metric")
    try:
        g = nm.fit(want_scores=True)
    except _native.NativeBackendError as e:                # the documented size limit of the LDS-resident solvers: PLSPM_E_LIMIT by name, nothing else
        assert "(102)" in str(e) and "LDS" in str(e), tag + ": " + str(e)
        return "E_LIMIT"
    try:
        with np.errstate(all="ignore"):
            r = orc.fit(X, model)
    except orc.NotConverged:
        assert g["status"] == 1, tag
        return "notconv"
    if g["status"] != 0:
        assert_device_status_justified(g["status"], X, model, tag)
        return "device-status"
    assert g["iterations"] == r["iterations"], tag
    assert_close(g["weights"], r["weights"], 1e-6, 1e-9, what=tag + " weights")
    assert_close(g["path_coef"], r["path_coef"], 1e-6, 1e-9, what=tag + " paths")
    assert_close(g["loadings"], r["loadings"], 1e-6, 1e-9, what=tag + " loadings")
    assert_close(g["scores"], r["scores"], 1e-6, 1e-8, what=tag + " scores")
    rows, status, iters = nm.bootstrap(B, seed=seed)
    route = "gram%d/solver%d/wave16-%d" % (nm.get_option("last_gram_path"), nm.get_option("last_solver"), nm.get_option("last_nm_wave16") if nonmetric else 0)
    if nonmetric:
        nm.set_option("nm_wave16", 0)
    else:
        nm.set_option("solver_rows", 0); nm.set_option("solver_wave", 0); nm.set_option("solver_quad", 0)
    rows_l, status_l, iters_l = nm.bootstrap(B, seed=seed)
    assert np.array_equal(status, status_l), tag + " " + route
    ok = status == 0
    assert ok.sum() >= B // 2, tag
    assert np.array_equal(iters[ok], iters_l[ok]), tag + " " + route
    assert_close(rows[ok], rows_l[ok], 1e-9, 1e-12, what=tag + " " + route)
    corr = orc.correction(n)
    checked = 0
    for b in range(B):
        if checked == 2:
            break
        idx = _native.bootstrap_indices(seed, b, n)
        if not _replicate_comparable(X, model, idx, corr, status[b], tag + " replicate %d" % b):
            continue
        mine, its = orc.bootstrap_replicate(X, model, idx, corr)
        assert its == iters[b], tag + " replicate %d" % b
        assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d" % b)
        checked += 1
    return route


@pytest.mark.parametrize("seed", range(12))
def test_random_model_with_many_lvs(seed):
    _huge_case_check(seed)


# ---- a rare 0/1 indicator that comes out CONSTANT in some replicates (fuzz_cases.make_rare_indicator_case; the oracle is pinned on the reference's own rows for the same
# cases by tests/golden/sweep_bootstrap_rows_vs_reference.py): weight 0, loading 0, the replicate counts -- on both Gram routes
def _rare_indicator_check(seed):
    from fuzz_cases import make_rare_indicator_case
    from plspm import _native
    case = make_rare_indicator_case(seed)
    if case is None:
        return "skipped"
    X, model, idx = case
    n, P = X.shape
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0)
    nm.upload(X)
    tag = "seed %d L=%d P=%d n=%d %s %s scaled=%d" % (seed, model.L, P, n, model.modes, model.scheme, model.scaled)
    corr = orc.correction(n)
    out = []
    for path in (2, 1):
        nm.set_option("gram_path", path)
        rows, status, iters = nm.bootstrap(len(idx), idx=idx)
        flat_ok = 0
        for b in range(len(idx)):
            try:
                with np.errstate(all="ignore"):
                    mine, its = orc.bootstrap_replicate(X, model, idx[b], corr)
            except Exception:                              # noqa: BLE001  (e.g. the indicator is the only item of its LV: the reference fails too)
                assert status[b] != 0, tag + " replicate %d (gram_path %d): the oracle cannot finish, device status 0" % (b, path)
                continue
            if status[b] != 0:
                assert_device_status_justified(int(status[b]), np.delete(X[idx[b]], np.flatnonzero(X[idx[b]].std(axis=0) == 0), axis=1) if False else X[idx[b]], model, tag)
                continue
            assert its == iters[b], tag + " replicate %d (gram_path %d): iterations %d vs %d" % (b, path, iters[b], its)
            assert_close(rows[b], mine, 1e-6, 1e-9, what=tag + " replicate %d (gram_path %d)" % (b, path))
            flat_ok += b >= 3
        out.append(flat_ok)
    return "flat replicates compared %d/%d" % tuple(out)


@pytest.mark.parametrize("seed", range(24))
def test_rare_indicator_constant_in_some_replicates(seed):
    _rare_indicator_check(seed)


@pytest.mark.parametrize("seed", range(40))
def test_random_model_on_a_small_sample_of_integers(seed):
    """fuzz_cases.make_small_int_case: 20 ... 80 rows of five- / seven-point items read as numbers -- covariances that are exactly zero, blocks that are exactly collinear."""
    from fuzz_cases import make_small_int_case
    X, model, nonmetric = make_small_int_case(seed)
    _model_check(X, model, nonmetric, seed)
