"""GPU: the host API's frames on seeded random models against golden g17 (tests/golden/make_golden_g17.py: the REAL reference's outer_model, inner_model with standard errors /
t / p, inner_summary, path_coefficients, crossloadings, effects, unidimensionality and goodness_of_fit on 12 metric / Scale.NUM models of fuzz_cases.make_case, 6 categorical
ones of make_cat_case, 3 + 3 with NaN cells).  The statistics behind them are host arithmetic on device outputs (plspm/inner_model.py from the device's L x L covariance instead of statsmodels,
plspm/unidimensionality.py by eigh of the device covariance blocks instead of sklearn's PCA ...): the reference's own data sets pin them on four models, this on eighteen more."""
import hashlib

import numpy as np
import pandas as pd
import pytest

import fuzz_cases as fc
from helpers import assert_close, load

pytestmark = pytest.mark.gpu
G = load("g17_api_frames")
TAGS = [str(t) for t in G["tags"]]


def _plspm(tag, check_sha=True, **kwargs):
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    kind = [k for k in ("metric", "missing", "nmx", "cat", "hocnum", "hocord", "flat", "raw") if tag.startswith(k)][0]
    seed = int(tag[len(kind):])
    hoc = None
    if kind.startswith("hoc"):
        X, model, stage2, C2, modes2, _ = (fc.make_hoc_case if kind == "hocnum" else fc.make_hoc_ord_case)(seed)
        hoc = (stage2, C2, modes2)
    else:
        X, model = {"metric": lambda s: fc.make_case(s)[:2], "cat": fc.make_cat_case, "missing": fc.make_missing_case, "nmx": fc.make_nmx_case,
                    "flat": lambda s: fc.make_degenerate_case(s)[:2], "raw": fc.make_raw_case}[kind](seed)
    if check_sha:
        assert hashlib.sha256(np.ascontiguousarray(X, dtype=np.float64).tobytes()).hexdigest() == str(G[tag + "/x_sha"]), "the generator no longer reproduces the matrix g17 was made from"
    scale = {"NUM": Scale.NUM, "RAW": Scale.RAW, "ORD": Scale.ORD, "NOM": Scale.NOM}
    lvs = ["L%d" % l for l in range(model.L)]
    df = pd.DataFrame(X, columns=["x%d" % p for p in range(X.shape[1])])
    if hoc is None:
        cfg = c.Config(pd.DataFrame(np.asarray(model.C, dtype=int), index=lvs, columns=lvs), scaled=model.scaled, default_scale=(Scale.NUM if model.scales is not None else None))
    else:
        lv2 = ["H" if k == "hoc" else lvs[ref] for k, ref in hoc[0]]
        cfg = c.Config(pd.DataFrame(np.asarray(hoc[1], dtype=int), index=lv2, columns=lv2), scaled=True, default_scale=scale[model.scales[0]])
        for (k, ref), name, mode in zip(hoc[0], lv2, hoc[2]):
            if k == "hoc":
                cfg.add_higher_order(name, Mode.A if mode == "A" else Mode.B, [lvs[j] for j in ref])
    for l in range(model.L):
        cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B,
                   *[c.MV("x%d" % p, scale[model.scales[p]] if (model.scales is not None and hoc is None) else None) for p in model.blocks[l]])
    scheme = {"centroid": Scheme.CENTROID, "factorial": Scheme.FACTORIAL, "path": Scheme.PATH}[model.scheme]
    return Plspm(df, cfg, scheme, 100, model.tol, **kwargs)


@pytest.mark.parametrize("tag", TAGS)
def test_api_frames_vs_reference_on_random_models(tag):
    m = _plspm(tag)
    for name, frame in (("outer_model", m.outer_model()), ("inner_model", m.inner_model()), ("inner_summary", m.inner_summary()), ("path_coefficients", m.path_coefficients()),
                        ("crossloadings", m.crossloadings())) + ((("unidimensionality", m.unidimensionality()),) if not (tag.startswith("hoc") or tag.startswith("flat")) else ()):
        num = frame.select_dtypes(include=[np.number])
        index, columns = [str(i) for i in G[tag + "/" + name + "/index"]], [str(x) for x in G[tag + "/" + name + "/columns"]]
        assert sorted(str(i) for i in frame.index) == sorted(index), (name, list(frame.index)[:6], index[:6])
        assert [str(x) for x in num.columns] == columns, (name, list(num.columns), columns)
        mine = num.loc[index].values.astype(float)
        want = G[tag + "/" + name + "/values"]
        # (p-values of strongly significant paths are 1e-30 and below: they pass on the absolute tolerance)
        assert np.array_equal(np.isnan(mine), np.isnan(want)), (tag, name)        # (blocks with a missing cell report NaN diagnostics: unidimensionality.py:39)
        assert_close(np.nan_to_num(mine), np.nan_to_num(want), 1e-6, 1e-9, what="%s %s" % (tag, name))
    eff = m.effects()
    assert [str(x) for x in eff["from"]] == [str(x) for x in G[tag + "/effects/from"]] and [str(x) for x in eff["to"]] == [str(x) for x in G[tag + "/effects/to"]]
    assert_close(eff[["direct", "indirect", "total"]].values.astype(float), G[tag + "/effects/values"], 1e-6, 1e-9, what=tag + " effects")
    assert_close(float(m.goodness_of_fit()), float(G[tag + "/gof"]), 1e-6, what=tag + " goodness_of_fit")


# ---- Plspm(..., bootstrap=True) through the host API on random models of every kind (not in g17: any seed): the call returns, nearly every replicate is used, the frames carry the
# fit's labels and its values in `original`, the replicate means sit near them, and the same seed gives the same table twice
API_BOOT = [("metric", s) for s in range(8)] + [("cat", s) for s in range(6)] + [("missing", s) for s in range(4)] + [("nmx", s) for s in range(4)] + \
           [("hocnum", s) for s in range(4)] + [("hocord", s) for s in range(4)]


def _api_bootstrap_check(kind, seed, B=120):
    m = _plspm("%s%d" % (kind, seed), check_sha=False, bootstrap=True, bootstrap_iterations=B, seed=seed + 1)
    boot = m.bootstrap()
    assert boot.used() >= int(0.8 * B), (kind, seed, boot.used())
    om = m.outer_model()
    w = boot.weights()
    assert sorted(w.index) == sorted(om.index)
    np.testing.assert_allclose(w.loc[om.index, "original"].values, om["weight"].values, rtol=1e-12, atol=1e-14)
    ld = boot.loading()
    np.testing.assert_allclose(ld.loc[om.index, "original"].values, om["loading"].values, rtol=1e-9, atol=1e-12)
    paths = boot.paths()
    pc = m.path_coefficients()
    for label, row in paths.iterrows():
        src, dst = [x.strip() for x in str(label).split("->")]
        assert abs(row["original"] - pc.loc[dst, src]) <= 1e-9 * max(1.0, abs(row["original"])), (kind, seed, label)
    for frame in (w, ld, paths, boot.r_squared(), boot.total_effects()):
        vals = frame[["original", "mean", "std.error", "perc.025", "perc.975"]].values.astype(float)
        assert np.all(np.isfinite(vals)), (kind, seed)
        assert np.all(frame["perc.025"].values <= frame["perc.975"].values + 1e-12)
    assert np.all(np.abs(w["mean"] - w["original"]) <= 8 * w["std.error"] + 1e-3 * np.abs(w["original"]).max()), (kind, seed)
    again = _plspm("%s%d" % (kind, seed), check_sha=False, bootstrap=True, bootstrap_iterations=B, seed=seed + 1).bootstrap()
    assert np.array_equal(again.weights().values, w.values, equal_nan=True)
    return boot.used()


@pytest.mark.parametrize("kind,seed", API_BOOT)
def test_api_bootstrap_on_random_models(kind, seed):
    _api_bootstrap_check(kind, seed)


def test_api_is_indifferent_to_column_order_dtypes_index_and_unused_columns():
    """Config.filter picks the configured MVs in add_lv order (config.py:262-270): shuffled columns, integer / float32 dtypes, a string index and columns the model never
    names must not change a single frame (bootstrap included: same seed, same replicates)."""
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    X, model = fc.make_case(3)[:2]
    Xi = np.round(X * 4.0)                                    # integer-valued: int64 / float32 / float64 columns then hold the same numbers
    lvs = ["L%d" % l for l in range(model.L)]
    names = ["x%d" % p for p in range(X.shape[1])]

    def run(df):
        cfg = c.Config(pd.DataFrame(np.asarray(model.C, dtype=int), index=lvs, columns=lvs), scaled=model.scaled)
        for l in range(model.L):
            cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B, *[c.MV(names[p]) for p in model.blocks[l]])
        return Plspm(df, cfg, Scheme.PATH, 100, 1e-6, bootstrap=True, bootstrap_iterations=100, seed=9)
    plain = run(pd.DataFrame(Xi, columns=names))
    rs = np.random.RandomState(2)
    odd = pd.DataFrame(Xi, columns=names)
    odd["unused_a"] = rs.standard_normal(len(odd)); odd["unused_b"] = 7
    odd = odd[list(rs.permutation(odd.columns))]
    for k, col in enumerate(names):
        odd[col] = odd[col].astype(["int64", "float32", "float64"][k % 3])
    odd.index = ["row%05d" % i for i in range(len(odd))]
    other = run(odd)
    for name in ("outer_model", "inner_summary", "path_coefficients", "crossloadings", "effects"):
        a, b = getattr(plain, name)(), getattr(other, name)()
        assert list(a.index) == list(b.index) and list(a.columns) == list(b.columns), name
        an, bn = a.select_dtypes(include=[np.number]).values, b.select_dtypes(include=[np.number]).values
        assert np.array_equal(an, bn, equal_nan=True), name
    assert np.array_equal(plain.scores().values, other.scores().values) and list(other.scores().index) == list(odd.index)
    assert np.array_equal(plain.bootstrap().weights().values, other.bootstrap().weights().values)


def test_categorical_model_beyond_65535_rows():
    """All-indicator data sets of more than 65,535 rows leave the uint16 count matrices (a count may exceed 65,535): fit and two replicates of a 70,000-row ordinal model
    against the oracle."""
    import plspm_oracle as orc
    import test_gpu_categorical as tc
    from plspm import _native
    C = orc.chain_C(3)
    X, blocks = orc.synth(70000, C, 3, seed=23)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    data = np.clip(np.round(3.0 + 1.1 * Z), 1, 5)
    model = orc.Model(blocks, C, "AAA", "path", True, tol=1e-6, scales=["ORD"] * 9)
    nm, g = tc.gpu_fit_cat(data, model)
    tc.check_fit(g, orc.fit(data, model), "70,000 rows")
    rows, status, iters = nm.bootstrap(8, seed=4)
    assert np.all(status == 0)
    rows = tc._rows_in_data_order(rows, g["inv"], 9, 3, nm.n_eff)
    for r in (0, 7):
        mine, its = orc.bootstrap_replicate(data, model, _native.bootstrap_indices(4, r, 70000), orc.correction(70000))
        assert its == iters[r]
        assert_close(rows[r], mine, 1e-6, 1e-9, what="replicate %d" % r)
