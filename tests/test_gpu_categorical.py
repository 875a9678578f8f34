"""GPU parity of the categorical (Scale.ORD / NOM, optimal scaling) non-metric path -- SURVEY.md 8(f) rank 4 -- through the
C-ABI (plspm_model_set_categorical) and through the host API, against the oracle (pinned on the reference, golden g11) and the
reference's own expected CSVs (tests/test_regression_nonmetric.py:94-120).  Tolerance: 1e-6 relative (north_star), scores 1e-7."""
import os

import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from helpers import GOLDEN, assert_close, load
from test_oracle_golden import (LIKERT_BLOCKS, LIKERT_C, LIKERT_CASES, RUSSA_C, RUSSA_CAT_BLOCKS, RUSSA_CAT_COLS, RUSSA_CAT_SCALES,
                                russa_cat_inputs)
from test_solver_hostemu_ordnom import build_aug

pytestmark = pytest.mark.gpu
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL, ATOL = 1e-6, 1e-9


def gpu_fit_cat(X, model):
    from plspm import _native
    Xaug, mv_off, mv_kind, lmv_off, boff, mv_data_col = build_aug(X, model)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], True, model.max_iter, model.tol, 0, nonmetric=True,
                             categorical=(mv_off, mv_kind))
    nm.upload(Xaug)
    out = nm.fit(want_scores=True, want_cov=True)
    Pm = len(mv_kind)
    inv = np.empty(Pm, dtype=np.int64); inv[mv_data_col] = np.arange(Pm)
    out["inv"] = inv
    out["weights_d"] = out["weights"][inv]; out["loadings_d"] = out["loadings"][inv]; out["crossloadings_d"] = out["crossloadings"][inv]
    out["pairs"] = list(zip(nm.eff_from.tolist(), nm.eff_to.tolist()))
    return nm, out


def check_fit(g, r, tag=""):
    assert g["status"] == 0, tag
    assert g["iterations"] == r["iterations"], "%s: iterations %d vs oracle %d" % (tag, g["iterations"], r["iterations"])
    assert_close(g["weights_d"], r["weights"], RTOL, what=tag + " weights")
    assert_close(g["loadings_d"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(g["crossloadings_d"], r["crossloadings"], RTOL, ATOL)
    assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL)
    assert_close(g["r2"], r["r2"], RTOL, ATOL)
    assert g["pairs"] == r["effect_pairs"]
    assert_close(g["total"], r["total"], RTOL, ATOL)
    assert_close(g["direct"], r["direct"], RTOL, ATOL)
    assert_close(g["scores"], r["scores"], 1e-7, 1e-9, what=tag + " scores")


@pytest.mark.parametrize("modes", ["AAA", "BBB"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_russa_categorical_vs_oracle_and_reference_golden(modes, scheme):
    X = russa_cat_inputs()
    model = orc.Model(RUSSA_CAT_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=RUSSA_CAT_SCALES)
    _, g = gpu_fit_cat(X, model)
    check_fit(g, orc.fit(X, model), modes + "/" + scheme)
    gold = load("g11_ordnom")
    key = "russa_%s_%s" % (modes, scheme)
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
    assert_close(g["loadings_d"], gold[key + "/loadings"], RTOL)
    assert_close(g["r2"], gold[key + "/r2"], RTOL, ATOL)
    assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9)


@pytest.mark.parametrize("tag", ["ordA", "ordB", "mixM"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_likert_vs_oracle_and_reference_golden(tag, scheme):
    gold = load("g11_ordnom")
    modes, scales = LIKERT_CASES[tag]
    model = orc.Model(LIKERT_BLOCKS, LIKERT_C, modes, scheme, True, tol=1e-7, scales=scales)
    _, g = gpu_fit_cat(gold["likert"], model)
    check_fit(g, orc.fit(gold["likert"], model), tag + "/" + scheme)
    key = "likert_%s_%s" % (tag, scheme)
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
    assert_close(g["path_coef"], gold[key + "/path_coef"], RTOL, ATOL)


def _rows_in_data_order(rows, inv, Pm, L, ne):
    return np.concatenate((rows[:, :Pm][:, inv], rows[:, Pm:Pm + L + 2 * ne], rows[:, Pm + L + 2 * ne:][:, inv]), axis=1)


@pytest.mark.parametrize("modes,scheme", [("AAA", "centroid"), ("BBB", "path")])
def test_bootstrap_explicit_indices_with_absent_categories(modes, scheme):
    """N = 47 resamples regularly lose a category of `demo` / `gnpr`: the device re-ranks the present categories like util.rank
    on the resampled column.  Replicates the oracle cannot finish (singular / not converged) must be flagged, not reported."""
    X = russa_cat_inputs()
    model = orc.Model(RUSSA_CAT_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=RUSSA_CAT_SCALES)
    nm, g = gpu_fit_cat(X, model)
    rs = np.random.RandomState(31)
    idx = rs.randint(47, size=(24, 47)).astype(np.int32)
    rows, status, iters = nm.bootstrap(24, idx=idx)
    rows = _rows_in_data_order(rows, g["inv"], 9, 3, nm.n_eff)
    corr = orc.correction(47)
    compared = 0
    for b in range(24):
        try:
            mine, its = orc.bootstrap_replicate(X, model, idx[b], corr)
        except Exception:
            assert status[b] != 0
            continue
        if not np.all(np.isfinite(mine)):
            continue
        assert status[b] == 0 and its == iters[b], "replicate %d: %d/%d vs %d" % (b, status[b], iters[b], its)
        assert_close(rows[b], mine, RTOL, 1e-8, what="replicate %d" % b)
        compared += 1
    assert compared >= 12


def test_bootstrap_device_resampling_likert_spot_checks():
    from plspm import _native
    gold = load("g11_ordnom")
    modes, scales = LIKERT_CASES["mixM"]
    model = orc.Model(LIKERT_BLOCKS, LIKERT_C, modes, "path", True, tol=1e-7, scales=scales)
    X = gold["likert"]
    nm, g = gpu_fit_cat(X, model)
    rows, status, iters = nm.bootstrap(200, seed=9)
    assert np.all(status == 0)
    rows = _rows_in_data_order(rows, g["inv"], 16, 4, nm.n_eff)
    corr = orc.correction(X.shape[0])
    for r in (0, 77, 199):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(9, r, X.shape[0]), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)
    # sharding invariance: replicate ids, not call boundaries, key the streams
    again, st2, _ = nm.bootstrap(50, seed=9, rep_offset=150)
    assert np.array_equal(_rows_in_data_order(again, g["inv"], 16, 4, nm.n_eff), rows[150:])


def _russa_config(mode):
    import plspm.config as c
    from plspm.scale import Scale
    s = c.Structure(); s.add_path(["AGRI", "IND"], ["POLINS"])
    config = c.Config(s.path(), default_scale=Scale.NUM)
    config.add_lv("AGRI", mode, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    config.add_lv("IND", mode, c.MV("gnpr", Scale.ORD), c.MV("labo", Scale.ORD))
    config.add_lv("POLINS", mode, c.MV("ecks"), c.MV("death"), c.MV("demo", Scale.NOM), c.MV("inst"))
    return config


@pytest.mark.parametrize("mode_name,fname", [("A", "russa.categorical.inner_summary.csv"), ("B", "russa.categorical.mode_b.inner_summary.csv")])
def test_api_reproduces_reference_russa_categorical_tests(mode_name, fname):
    """Mirrors reference tests/test_regression_nonmetric.py:94-120 (expected values: R plspm)."""
    import plspm.util as util
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    ref = os.path.join(GOLDEN, "ref_data")
    russa = pd.read_csv(os.path.join(ref, "russa.csv"), index_col=0)
    calc = Plspm(russa, _russa_config(Mode.A if mode_name == "A" else Mode.B), Scheme.CENTROID, 100, 0.0000001)
    expected = pd.read_csv(os.path.join(ref, fname), index_col=0)
    np.testing.assert_allclose(util.sort_cols(expected.drop(["type"], axis=1)).sort_index(),
                               util.sort_cols(calc.inner_summary().drop(["type", "r_squared_adj"], axis=1)).sort_index().astype(float))
    pd.testing.assert_series_equal(expected.loc[:, "type"].sort_index(), calc.inner_summary().loc[:, "type"].sort_index())
    # the frames carry the logical MVs (not the indicator columns)
    assert sorted(calc.outer_model().index) == sorted(RUSSA_CAT_COLS)
    assert calc.crossloadings().shape == (9, 3) and calc.scores().shape == (47, 3)
    gold = load("g11_ordnom")
    key = "russa_%s_centroid" % (mode_name * 3)
    om = calc.outer_model().loc[RUSSA_CAT_COLS]
    assert_close(om["weight"].values, gold[key + "/weights"], RTOL)
    assert_close(om["loading"].values, gold[key + "/loadings"], RTOL)


def test_api_bootstrap_categorical():
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    russa = pd.read_csv(os.path.join(GOLDEN, "ref_data", "russa.csv"), index_col=0)
    calc = Plspm(russa, _russa_config(Mode.A), Scheme.CENTROID, 100, 0.0000001, bootstrap=True, bootstrap_iterations=200, seed=3)
    boot = calc.bootstrap()
    w = boot.weights()
    assert sorted(w.index) == sorted(RUSSA_CAT_COLS)
    assert np.all(np.isfinite(w[["original", "mean", "std.error"]].values))
    om = calc.outer_model()
    assert_close(w.loc[om.index, "original"].values, om["weight"].values, 1e-12)
    assert boot.paths().shape[0] == 2 and boot.r_squared().shape[0] == 1


def test_wide_indicator_model_uses_the_block_staged_stop_rule_pass():
    """60 five-point ORD items = 300 indicator columns: the coefficient tile of 64 replicates (300 KB) no longer fits LDS as a whole,
    the dense pass stages it per LV block.  Must agree with the gathering pass and with the oracle."""
    import os
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(1500, C, 10, seed=31)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
    model = orc.Model(blocks, C, "AAAAAA", "centroid", True, tol=1e-6, scales=["ORD"] * 60)
    nm, g = gpu_fit_cat(likert, model)
    check_fit(g, orc.fit(likert, model), "wide")
    dense = nm.bootstrap(70, seed=4)
    nm.set_option("conv_pass", 1)                           # the gathering stop-rule pass
    gathered = nm.bootstrap(70, seed=4)
    nm.set_option("conv_pass", 0)
    assert np.all(dense[1] == 0)
    assert np.array_equal(dense[1], gathered[1]) and np.array_equal(dense[2], gathered[2])
    assert_close(dense[0], gathered[0], 1e-11, 1e-13)
    rows = _rows_in_data_order(dense[0], g["inv"], 60, 6, nm.n_eff)
    mine, its = orc.bootstrap_replicate(likert, model, _native.bootstrap_indices(4, 69, 1500), orc.correction(1500))
    assert its == dense[2][69]
    assert_close(rows[69], mine, RTOL, ATOL)


@pytest.mark.parametrize("scale", ["ORD", "NOM"])
def test_stop_rule_pass_on_category_codes_gives_the_dense_pass_bits(scale):
    """All-indicator models on the blocked dense pass: nm_conv_codes_kernel adds the coefficient of the ONE column a row has set per MV
    (16 category codes per row tile and MV) instead of multiplying five 0/1 columns through -- the same additions in the same order, so
    records and iteration counts are bit-identical to the dense pass ("nm_codes" 0), and both follow the oracle.  1,500 rows (the last
    tile holds pad rows), 300 indicator columns, uneven category counts (3 .. 5 per item)."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(1500, C, 10, seed=37)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
    likert[:, ::3] = np.clip(likert[:, ::3], 2, 4)          # every third item has three categories
    model = orc.Model(blocks, C, "AAAAAA", "centroid", True, tol=1e-6, scales=[scale] * 60)
    nm, g = gpu_fit_cat(likert, model)
    assert nm.get_option("nm_codes") == 1
    on = nm.bootstrap(200, seed=6)
    assert nm.get_option("last_nm_codes") == 1 and nm.get_option("last_gram_path") == 2
    nm.set_option("nm_codes", 0)
    off = nm.bootstrap(200, seed=6)
    assert nm.get_option("last_nm_codes") == 0
    nm.set_option("nm_codes", 1)
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[2], off[2])
    assert np.array_equal(on[0], off[0])
    rows = _rows_in_data_order(on[0], g["inv"], 60, 6, nm.n_eff)
    ok = np.flatnonzero(on[1] == 0)
    r = int(ok[-1])
    mine, its = orc.bootstrap_replicate(likert, model, _native.bootstrap_indices(6, r, 1500), orc.correction(1500))
    assert its == on[2][r]
    assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("scale,scheme", [("ORD", "path"), ("NOM", "centroid"), ("ORD", "factorial")])
def test_stop_rule_pass_as_int8_matrix_product(scale, scheme):
    """kernels_nmp.h (round 5): the score-based stop rule (weights.py:120) of all-indicator models as an exact int8 MFMA product -- indicator bytes x
    seven base-256 digit planes of the score maps, the digits put together again per (row, replicate).  Held against the pass on category codes
    ("nm_mfma" 0) on the VALUE of the criterion every replicate was decided on (test seam plspm_nonmetric_criteria: agreement far below anything a
    decision could see), on the iteration counts and -- the pass only decides when to stop -- on the bits of the records; and against the oracle.
    1,500 rows (the last tile holds pad rows, 94 tiles cut into row chunks), uneven category counts, replicates that lose categories."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(1500, C, 10, seed=53)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
    likert[:, ::3] = np.clip(likert[:, ::3], 2, 4)
    model = orc.Model(blocks, C, "AAAAAA", scheme, True, tol=1e-6, scales=[scale] * 60)
    nm, g = gpu_fit_cat(likert, model)
    assert nm.get_option("nm_mfma") == 1
    sub = nm.bootstrap(330, seed=8)                       # round 6 default: the step's own upper bound stops, the pass over the first row chunks says "go on"
    bound = nm.nonmetric_criteria(330)
    nm.set_option("nm_subset", 0)                           # ... and every pass over all rows, as in round 5: what the two pass kernels are compared on
    on = nm.bootstrap(330, seed=8)
    assert nm.get_option("last_nm_mfma") == 1 and nm.get_option("last_nm_codes") == 1 and nm.get_option("last_gram_path") == 2
    crit_on = nm.nonmetric_criteria(330)
    assert np.array_equal(sub[0], on[0]) and np.array_equal(sub[1], on[1]) and np.array_equal(sub[2], on[2])
    okb = on[1] == 0
    assert np.all(bound[okb] >= crit_on[okb] * (1 - 1e-9)) and np.all(bound[okb] < 1e-6)      # the value a replicate stopped on: the upper bound of the exact criterion
    nm.set_option("nm_mfma", 0)
    off = nm.bootstrap(330, seed=8)
    assert nm.get_option("last_nm_mfma") == 0 and nm.get_option("last_nm_codes") == 1
    crit_off = nm.nonmetric_criteria(330)
    nm.set_option("nm_mfma", 1); nm.set_option("nm_subset", 4)
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[2], off[2])
    assert np.array_equal(on[0], off[0])
    assert np.all(np.isfinite(crit_off)) and np.all(crit_off > 0) and np.all(crit_off[on[1] == 0] < 1e-6)
    assert_close(crit_on, crit_off, 1e-9, 1e-20)
    rows = _rows_in_data_order(on[0], g["inv"], 60, 6, nm.n_eff)
    ok = np.flatnonzero(on[1] == 0)
    for r in (int(ok[0]), int(ok[-1])):
        mine, its = orc.bootstrap_replicate(likert, model, _native.bootstrap_indices(8, r, 1500), orc.correction(1500))
        assert its == on[2][r]
        assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("shape", ["fourteen_five_point_items", "eight_ten_point_items"])
def test_matrix_product_pass_on_blocks_of_up_to_128_columns(shape):
    """LV blocks of 65 .. 128 indicator columns take TWO k-steps of the instruction (nmp::conv_mfma_kernel<4, 2>: the second MFMA of a plane chains through the
    accumulator): 14 five-point items per LV (70 columns) and 8 ten-point items (80 columns; the CMAX = 16 wave step).  Against the pass on category codes:
    criterion values to 1e-9, identical iteration counts and records; a replicate against the oracle."""
    from plspm import _native
    C = orc.chain_C(2)
    if shape == "fourteen_five_point_items":
        X, blocks = orc.synth(2000, C, 14, seed=71)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(3 + 1.1 * Z), 1, 5)
        model = orc.Model(blocks, C, "AA", "centroid", True, tol=1e-6, scales=["ORD"] * 28)
    else:
        X, blocks = orc.synth(3000, C, 8, seed=73)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(5.5 + 2.0 * Z), 1, 10)
        model = orc.Model(blocks, C, "AA", "path", True, tol=1e-6, scales=["ORD", "NOM"] * 8)
    nm, g = gpu_fit_cat(data, model)
    sub = nm.bootstrap(200, seed=12)                      # (round 6 default: upper bound + lower bound from the first row chunks)
    nm.set_option("nm_subset", 0)
    on = nm.bootstrap(200, seed=12)
    assert nm.get_option("last_nm_mfma") == 1 and nm.get_option("last_nm_wave") == 1 and nm.get_option("last_gram_path") == 2
    crit_on = nm.nonmetric_criteria(200)
    assert np.array_equal(sub[0], on[0]) and np.array_equal(sub[1], on[1]) and np.array_equal(sub[2], on[2])
    nm.set_option("nm_mfma", 0)
    off = nm.bootstrap(200, seed=12)
    assert nm.get_option("last_nm_mfma") == 0 and nm.get_option("last_nm_codes") == 1
    crit_off = nm.nonmetric_criteria(200)
    nm.set_option("nm_mfma", 1); nm.set_option("nm_subset", 4)
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[2], off[2]) and np.array_equal(on[0], off[0])
    assert_close(crit_on, crit_off, 1e-9, 1e-20)
    ok = np.flatnonzero(on[1] == 0)
    assert ok.size >= 150
    r = int(ok[-1])
    mine, its = orc.bootstrap_replicate(data, model, _native.bootstrap_indices(12, r, data.shape[0]), orc.correction(data.shape[0]))
    rows = _rows_in_data_order(on[0], g["inv"], len(model.scales), 2, nm.n_eff)
    assert its == on[2][r]
    assert_close(rows[r], mine, RTOL, ATOL)


@pytest.mark.parametrize("shape", ["likert60", "chain3"])
def test_count_matrices_written_by_the_int8_product(shape):
    """Round 5: on all-indicator data the int8 product's sums ARE the co-occurrence counts the wave step streams -- the one-plane launch writes them as uint16
    (upper triangle; nmg_kernel<4> mirrors it through LDS) instead of fp64 slots that a scatter pass re-reads ("nm_direct16" 0: that path).  Same integers,
    so the records are the same bits.  301 / 49 count columns: five tiles of 64 with a ragged last one / a single ragged tile."""
    if shape == "likert60":
        C = orc.satisfaction_C()
        X, blocks = orc.synth(1500, C, 10, seed=59)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
        model = orc.Model(blocks, C, "AAAAAA", "path", True, tol=1e-6, scales=["ORD"] * 60)
    else:
        C = orc.chain_C(3)
        X, blocks = orc.synth(900, C, 4, seed=61)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        likert = np.clip(np.round(2.5 + 0.9 * Z), 1, 4)
        model = orc.Model(blocks, C, "AAA", "factorial", True, tol=1e-6, scales=["NOM"] * 12)
    from plspm import _native
    nm, g = gpu_fit_cat(likert, model)
    assert nm.get_option("nm_direct16") == 1
    on = nm.bootstrap(300, seed=10)
    assert nm.get_option("last_nm_direct16") == 1 and nm.get_option("last_nm_wave") == 1 and nm.get_option("last_gram_path") == 2
    nm.set_option("nm_direct16", 0)
    off = nm.bootstrap(300, seed=10)
    assert nm.get_option("last_nm_direct16") == 0 and nm.get_option("last_nm_wave") == 1
    nm.set_option("nm_direct16", 1)
    assert np.all(on[1] == 0)
    assert np.array_equal(on[1], off[1]) and np.array_equal(on[2], off[2]) and np.array_equal(on[0], off[0])
    # explicit index lists take the same route (a multiplicity above 127 would fall back to the fp64 Gram: not here)
    idx = np.stack([_native.bootstrap_indices(10, r, likert.shape[0]) for r in range(40)])
    ex = nm.bootstrap(40, idx=idx)
    assert np.array_equal(ex[0], on[0][:40]) and np.array_equal(ex[2], on[2][:40])


@pytest.mark.parametrize("case", ["likert60_path", "likert60_nom_centroid", "chain8_factorial", "tiny_blocks", "eight_categories", "ten_point_items", "sixteen_categories"])
def test_wave_step_agrees_with_the_workgroup_step(case):
    """kernels_nmw.h (round 5): the categorical iteration as ONE WAVE per problem -- count matrix streamed 16 bytes per lane and row, the pooling
    of the ordinal quantification in registers, the block quadratic form as a second matrix-vector product on the block diagonal -- against the
    workgroup step it restates (nmg_kernel<1>, "nm_wave" 0): same iteration counts, records equal to 1e-10 (the sums that cross lanes are wave
    reductions: a different order of the same terms), for fits and bootstraps; every shape class of its instantiations: 2 .. 8 LVs, 2 .. 8
    categories per item (and -- the second set of instantiations -- 9 .. 16), several LV blocks inside one lane's eight columns, replicates that lose categories, ORD and NOM; and both against the oracle."""
    from plspm import _native
    rng = np.random.default_rng(5)
    if case in ("likert60_path", "likert60_nom_centroid"):
        C = orc.satisfaction_C()
        X, blocks = orc.synth(1500, C, 10, seed=31)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(3 + 1.1 * Z), 1, 5)
        data[:, ::3] = np.clip(data[:, ::3], 2, 4)
        scale, scheme = ("ORD", "path") if case == "likert60_path" else ("NOM", "centroid")
        model = orc.Model(blocks, C, "A" * 6, scheme, True, tol=1e-6, scales=[scale] * 60)
    elif case == "chain8_factorial":
        C = orc.chain_C(8)
        X, blocks = orc.synth(900, C, 4, seed=3)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(2.5 + 1.0 * Z), 1, 4)
        model = orc.Model(blocks, C, "A" * 8, "factorial", True, tol=1e-6, scales=["ORD", "NOM"] * 16)
    elif case == "tiny_blocks":
        C = orc.chain_C(5)
        X, blocks = orc.synth(400, C, 1, seed=8)               # one two-category item per LV: five blocks of two columns inside ONE lane's eight columns
        X2, _ = orc.synth(400, C, 1, seed=9)
        data = (np.concatenate((X, X2), axis=1) > 0).astype(float) + 1.0
        blocks = [np.array([l, 5 + l]) for l in range(5)]
        model = orc.Model(blocks, C, "A" * 5, "path", True, tol=1e-6, scales=["ORD"] * 10)
    elif case == "ten_point_items":
        # the reference's own example data (mobi / ECSI) are ten-point items: the CMAX = 16 instantiation (one wave per SIMD); an MV's ten columns lie across
        # two or three lanes of the column role
        C = orc.chain_C(3)
        X, blocks = orc.synth(3000, C, 5, seed=14)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(5.5 + 2.0 * Z), 1, 10)
        model = orc.Model(blocks, C, "AAA", "path", True, tol=1e-6, scales=["ORD"] * 15)
    elif case == "sixteen_categories":
        C = orc.chain_C(2)
        X, blocks = orc.synth(4000, C, 4, seed=16)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(8.5 + 3.2 * Z), 1, 16)
        data[:, 1] = np.clip(data[:, 1], 4, 12)                   # (one item with nine)
        model = orc.Model(blocks, C, "AA", "factorial", True, tol=1e-6, scales=["ORD", "NOM"] * 4)
    else:
        C = orc.chain_C(2)
        X, blocks = orc.synth(2500, C, 6, seed=12)
        Z = (X - X.mean(axis=0)) / X.std(axis=0)
        data = np.clip(np.round(4.5 + 1.6 * Z), 1, 8)             # eight categories per item
        model = orc.Model(blocks, C, "AA", "centroid", True, tol=1e-6, scales=["ORD"] * 12)
    nm, g = gpu_fit_cat(data, model)
    assert nm.get_option("nm_wave") == 1 and nm.get_option("last_nm_wave") == 1
    check_fit(g, orc.fit(data, model), case)
    B = 300
    wave = nm.bootstrap(B, seed=6)
    assert nm.get_option("last_nm_wave") == 1
    if case == "ten_point_items":                               # round 6: the ten-category instantiation (two waves per SIMD) against the sixteen-category one
        assert nm.get_option("nm_c10") == 1
        nm.set_option("nm_c10", 0)
        wide = nm.bootstrap(B, seed=6)
        nm.set_option("nm_c10", 1)
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(wave, wide))
    nm.set_option("nm_wave", 0)
    fit0 = nm.fit(want_scores=True)
    group = nm.bootstrap(B, seed=6)
    assert nm.get_option("last_nm_wave") == 0
    nm.set_option("nm_wave", 1)
    assert fit0["iterations"] == g["iterations"]
    assert_close(g["weights"], fit0["weights"], 1e-10, 1e-13)
    assert_close(g["scores"], fit0["scores"], 1e-10, 1e-12)
    assert np.array_equal(wave[1], group[1]) and np.array_equal(wave[2], group[2]), (np.flatnonzero(wave[2] != group[2])[:8], wave[1].sum(), group[1].sum())
    ok = wave[1] == 0
    assert ok.sum() >= 10
    assert_close(wave[0][ok], group[0][ok], 1e-10, 1e-12)
    r = int(np.flatnonzero(ok)[-1])
    mine, its = orc.bootstrap_replicate(data, model, _native.bootstrap_indices(6, r, data.shape[0]), orc.correction(data.shape[0]))
    Pm = len(model.scales)
    rows = _rows_in_data_order(wave[0], g["inv"], Pm, C.shape[0], nm.n_eff)
    assert its == wave[2][r]
    assert_close(rows[r], mine, RTOL, ATOL)


def test_full_size_categorical_bootstrap_properties():
    """The categorical counterpart of the headline workload at FULL size -- 10,000 rows x 60 five-point items (300 indicator columns) x 6 LVs, 2,000 replicates --
    through size-independent properties: (i) every route of round 5 gives the same records (stop rule as int8 matrix product / on category codes; count matrices
    from the int8 product / through the scatter pass; wave step / workgroup step to 1e-10 with equal iteration counts); (ii) sharding invariance: the replicates of
    two calls with `rep_offset` are the bits of one call; (iii) a replicate against the oracle on the resampled DATA; (iv) the criterion every replicate stopped
    on is below the tolerance and agrees between the two passes."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(10000, C, 10, seed=0)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
    model = orc.Model(blocks, C, "AAAAAA", "path", True, tol=1e-6, scales=["ORD"] * 60)
    nm, g = gpu_fit_cat(likert, model)
    B = 2000
    sub = nm.bootstrap(B, seed=1)                           # round 6 default: the step's upper bound stops, a lower bound from the first row chunks says "go on"
    assert nm.get_option("last_nm_exact") == 0              # ... and decides every step of this workload without the pass over all rows
    nm.set_option("nm_subset", 0)                           # the comparisons below: every pass over all rows (round 5's protocol)
    base = nm.bootstrap(B, seed=1)
    assert np.array_equal(sub[0], base[0]) and np.array_equal(sub[1], base[1]) and np.array_equal(sub[2], base[2])
    assert nm.get_option("last_nm_wave") == 1 and nm.get_option("last_nm_mfma") == 1 and nm.get_option("last_nm_direct16") == 1 and nm.get_option("last_gram_path") == 2
    assert np.all(base[1] == 0) and base[2].min() >= 3
    crit = nm.nonmetric_criteria(B)
    assert np.all(crit < 1e-6) and np.all(crit > 0)
    a = nm.bootstrap(1200, seed=1)
    b = nm.bootstrap(800, seed=1, rep_offset=1200)
    assert np.array_equal(np.concatenate((a[0], b[0])), base[0]) and np.array_equal(np.concatenate((a[2], b[2])), base[2])
    for key in ("nm_mfma", "nm_direct16"):
        nm.set_option(key, 0)
        other = nm.bootstrap(B, seed=1)
        assert nm.get_option("last_" + key) == 0
        if key == "nm_mfma": assert_close(nm.nonmetric_criteria(B), crit, 1e-9, 1e-20)
        nm.set_option(key, 1)
        assert np.array_equal(other[0], base[0]) and np.array_equal(other[1], base[1]) and np.array_equal(other[2], base[2]), key
    nm.set_option("nm_wave", 0)
    group = nm.bootstrap(B, seed=1)
    nm.set_option("nm_wave", 1)
    assert nm.get_option("last_nm_wave") == 0 and np.array_equal(group[2], base[2])
    assert_close(group[0], base[0], 1e-10, 1e-12)
    rows = _rows_in_data_order(base[0], g["inv"], 60, 6, nm.n_eff)
    r = B - 1
    mine, its = orc.bootstrap_replicate(likert, model, _native.bootstrap_indices(1, r, 10000), orc.correction(10000))
    assert its == base[2][r]
    assert_close(rows[r], mine, RTOL, ATOL)


def test_categorical_bootstrap_beyond_one_histogram_window_takes_the_int8_route():
    """70,000 rows: non-metric bootstraps with on-device draws now stay on the digit-plane Gram (counts from the 131,072-row byte
    histogram) and take their stop-rule passes' row multiplicities from its int8 counts -- on category codes for all-indicator data.
    Same bits as the multiply-add pass; the fp64 route (row lists from the global histogram, gathering pass) agrees to 1e-9 with
    identical iteration counts."""
    C = orc.chain_C(3)
    X, blocks = orc.synth(70000, C, 4, seed=41)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(2.5 + 0.9 * Z), 1, 4)
    model = orc.Model(blocks, C, "AAA", "factorial", True, tol=1e-6, scales=["ORD"] * 12)
    nm, g = gpu_fit_cat(likert, model)
    on = nm.bootstrap(48, seed=2)
    assert nm.get_option("last_gram_path") == 2 and nm.get_option("last_nm_codes") == 1 and np.all(on[1] == 0)
    nm.set_option("nm_codes", 0)
    off = nm.bootstrap(48, seed=2)
    assert nm.get_option("last_nm_codes") == 0
    assert np.array_equal(on[0], off[0]) and np.array_equal(on[2], off[2])
    nm.set_option("gram_path", 1)
    f64 = nm.bootstrap(48, seed=2)
    assert nm.get_option("last_gram_path") == 1
    assert np.array_equal(on[2], f64[2]) and np.all(f64[1] == 0)
    assert_close(on[0], f64[0], 1e-9, 1e-12)


def _likert_model(n, per, scheme, scale, seed, tol=1e-6):
    C = orc.satisfaction_C()
    X, blocks = orc.synth(n, C, per, seed=seed)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
    return likert, orc.Model(blocks, C, "AAAAAA", scheme, True, tol=tol, scales=[scale] * (6 * per))


@pytest.mark.parametrize("scheme,scale", [("path", "ORD"), ("centroid", "NOM")])
def test_one_launch_categorical_bootstrap_against_the_launch_by_launch_forms(scheme, scale):
    """Round 6: a batch of an all-indicator, all-Mode-A model as ONE solver launch (kernels_nmw.h ONE: every step stops on its own upper bound of the reference's
    score criterion, weights.py:120, or goes on speculatively, leaving its score map behind) + the verification on the observations (lower bound from the row
    chunks a step asks for, exact pass, replay).  Bit-identical records, status and iteration counts to the launch-by-launch forms -- with the bound + row subsets
    ("nm_cat_one" 0) and with every pass over all rows ("nm_subset" 0, round 5) --, a replicate against the oracle; then the seams: a safety factor of 1 (the lower
    bound fails for some steps: exact pass, nothing replayed) and a bound scaled by 2^30 (every replicate overshoots: every one replayed)."""
    from plspm import _native
    likert, model = _likert_model(3000, 6, scheme, scale, seed=61)
    nm, g = gpu_fit_cat(likert, model)
    B = 400
    one = nm.bootstrap(B, seed=4)
    assert nm.get_option("last_nm_one") == 1 and nm.get_option("last_nm_wave") == 1 and nm.get_option("last_nm_replayed") == 0
    nm.set_option("nm_cat_one", 0)
    step = nm.bootstrap(B, seed=4)
    assert nm.get_option("last_nm_one") == 0
    nm.set_option("nm_subset", 0)
    full = nm.bootstrap(B, seed=4)
    nm.set_option("nm_subset", 4); nm.set_option("nm_cat_one", 1)
    for other in (step, full):
        assert np.array_equal(one[1], other[1]) and np.array_equal(one[2], other[2]) and np.array_equal(one[0], other[0])
    assert np.all(one[1] == 0) and one[2].min() >= 3
    rows = _rows_in_data_order(one[0], g["inv"], likert.shape[1], 6, nm.n_eff)
    for r in (0, B - 1):
        mine, its = orc.bootstrap_replicate(likert, model, _native.bootstrap_indices(4, r, likert.shape[0]), orc.correction(likert.shape[0]))
        assert its == one[2][r]
        assert_close(rows[r], mine, RTOL, ATOL)
    nm.set_option("nm_subset", 1)                            # the lower bound with no safety margin: some steps are left to the exact pass, which confirms them
    tight = nm.bootstrap(B, seed=4)
    assert nm.get_option("last_nm_flagged") > 0 and nm.get_option("last_nm_replayed") == 0
    assert np.array_equal(tight[0], one[0]) and np.array_equal(tight[2], one[2])
    nm.set_option("nm_subset", 4)
    nm.set_option("nm_bound_shift", 30)                      # a useless (but valid) bound: every replicate runs on behind the reference's stop and is moved back
    over = nm.bootstrap(B, seed=4)
    assert nm.get_option("last_nm_replayed") == B
    assert np.array_equal(over[1], one[1]) and np.array_equal(over[2], one[2])
    assert_close(over[0], one[0], 1e-12, 1e-14)
    nm.set_option("nm_bound_shift", 0)


def test_one_launch_categorical_not_converged_and_sharding():
    """max_iter reached inside the one launch: PLSPM_NOT_CONVERGED at max_iter + 1 steps (weights.py:183-186), as the launch-by-launch form reports it; and a
    replicate's record does not depend on the batch it travels in."""
    likert, model = _likert_model(2000, 5, "factorial", "ORD", seed=67)
    nm, g = gpu_fit_cat(likert, model)
    base = nm.bootstrap(300, seed=2)
    assert nm.get_option("last_nm_one") == 1
    part = nm.bootstrap(120, seed=2, rep_offset=100)
    assert np.array_equal(part[0], base[0][100:220]) and np.array_equal(part[2], base[2][100:220])
    stubborn = orc.Model(model.blocks, model.C, "AAAAAA", "factorial", True, tol=1e-300, max_iter=6, scales=["ORD"] * 30)
    nm2, _ = gpu_fit_cat(likert, stubborn)
    one = nm2.bootstrap(200, seed=3)
    assert nm2.get_option("last_nm_one") == 1
    nm2.set_option("nm_cat_one", 0)
    step = nm2.bootstrap(200, seed=3)
    assert np.array_equal(one[1], step[1]) and np.array_equal(one[2], step[2])
    assert np.all(one[1][one[2] == 7] == 1) and np.any(one[2] == 7)
