#!/opt/conda/bin/python3.9
"""Golden-vector generator (build container only; the reference never travels to the GPU box).

Run:   PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

Imports the REAL reference from /root/reference (through oracle/refshim.py, plumbing shims only),
drives it through its public API / the same internal calls its bootstrap worker makes
(bootstrap.py:56-64), and stores inputs + outputs as small .npz fixtures in tests/golden/.
Fixtures are data only.  Iteration counts are observed by wrapping _MetricWeights.iterate with a
call counter (the reference does not expose them); the solver runs twice per estimate()
(estimator.py:39,52) so the count is halved.
"""
import hashlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refshim  # noqa: E402

refshim.load_reference()
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import plspm.config as c  # noqa: E402
import plspm.weights as refw  # noqa: E402
import plspm.inner_model as refim  # noqa: E402
import plspm.bootstrap as refboot  # noqa: E402
from plspm.estimator import Estimator  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.scheme import Scheme  # noqa: E402

import plspm_oracle as orc  # noqa: E402  (only for the synthetic generator + structures)

_calls = {"n": 0}
_orig_iterate = refw._MetricWeights.iterate


def _counting_iterate(self, scheme):
    _calls["n"] += 1
    return _orig_iterate(self, scheme)


refw._MetricWeights.iterate = _counting_iterate
_orig_nm_iterate = refw._NonmetricWeights.iterate


_per_solver = []          # iterate() calls of every _NonmetricWeights instance, in creation order (two per HOC estimate)


def _counting_nm_iterate(self, scheme):
    _calls["n"] += 1
    if not _per_solver or _per_solver[-1][0] is not self:
        _per_solver.append([self, 0])
    _per_solver[-1][1] += 1
    return _orig_nm_iterate(self, scheme)


refw._NonmetricWeights.iterate = _counting_nm_iterate
from plspm.scale import Scale  # noqa: E402

SCHEMES = {"centroid": Scheme.CENTROID, "factorial": Scheme.FACTORIAL, "path": Scheme.PATH}


def path_frame(C, lvs):
    return pd.DataFrame(np.asarray(C, dtype=int), index=lvs, columns=lvs)


def build_config(C, lvs, blocks_names, modes, scaled, add_order=None, default_scale=None, mv_scales=None):
    cfg = c.Config(path_frame(C, lvs), scaled=scaled, default_scale=default_scale)
    for lv in (add_order or lvs):
        i = lvs.index(lv)
        cfg.add_lv(lv, Mode.A if modes[i] == "A" else Mode.B, *[c.MV(n, (mv_scales or {}).get(n)) for n in blocks_names[i]])
    return cfg


def run_fit(df, cfg, scheme, lvs, want_scores=True, tol=1e-6):
    _calls["n"] = 0
    m = Plspm(df, cfg, SCHEMES[scheme], 100, tol)
    iters = _calls["n"] // 2
    data_cols = [mv for lv in cfg._Config__mvs for mv in cfg._Config__mvs[lv]]  # add_lv order == filtered column order
    om = m.outer_model()
    eff = m.effects()
    out = dict(
        mv_names=np.array(data_cols),
        weights=om.loc[data_cols, "weight"].values.astype(float),
        loadings=om.loc[data_cols, "loading"].values.astype(float),
        crossloadings=m.crossloadings().loc[data_cols, lvs].values.astype(float),
        path_coef=m.path_coefficients().loc[lvs, lvs].values.astype(float),
        r2=m.inner_summary().loc[lvs, "r_squared"].values.astype(float),
        eff_from=np.array([lvs.index(x) for x in eff["from"]]),
        eff_to=np.array([lvs.index(x) for x in eff["to"]]),
        eff_direct=eff["direct"].values.astype(float),
        eff_indirect=eff["indirect"].values.astype(float),
        eff_total=eff["total"].values.astype(float),
        iters=np.array(iters),
    )
    if want_scores:
        out["scores"] = m.scores().loc[:, lvs].values.astype(float)
    return m, out


def boot_rows(df, cfg, scheme, lvs, idx_list, effects_index, tol=1e-6):
    """Per-replicate rows exactly as BootstrapProcess.run builds them (bootstrap.py:56-64)."""
    filtered = cfg.filter(df)
    n = filtered.shape[0]
    corr = np.sqrt(n / (n - 1))
    calc = refw.WeightsCalculatorFactory(cfg, 100, tol, corr, SCHEMES[scheme])
    est = Estimator(cfg)
    cols = list(filtered.columns)
    rows, iters = [], []
    for idx in idx_list:
        _calls["n"] = 0
        fd, sc, w = est.estimate(calc, filtered.iloc[idx, :])
        iters.append(_calls["n"] // 2)
        im = refim.InnerModel(cfg.path(), sc)
        r2 = im.r_squared().loc[lvs].values.astype(float)
        eff = im.effects()
        tot = eff.loc[effects_index, "total"].values.astype(float)
        dire = eff.loc[effects_index, "direct"].values.astype(float)
        ld = (sc.apply(lambda s: fd.corrwith(s)) * cfg.odm(cfg.path())).sum(axis=1).loc[cols].values.astype(float)
        wv = w.loc[cols, "weight"].values.astype(float)
        rows.append(np.concatenate((wv, r2, tot, dire, ld)))
    return np.array(rows), np.array(iters)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def save(name, **arrays):
    p = os.path.join(HERE, name + ".npz")
    np.savez_compressed(p, **arrays)
    print("wrote %-40s %7.1f KB" % (name + ".npz", os.path.getsize(p) / 1024))


def main():
    t0 = time.time()
    sat = pd.read_csv(os.path.join(HERE, "ref_data", "satisfaction.csv"), index_col=0)
    lvs = orc.SAT_LVS
    Csat = orc.satisfaction_C()
    prefixes = dict(IMAG="imag", EXPE="expe", QUAL="qual", VAL="val", SAT="sat", LOY="loy")
    sat_blocks = [[col for col in sat.columns if col.startswith(prefixes[lv])] for lv in lvs]

    # ---- G1: satisfaction x {A,B,mixed} x {C,F,P} x scaled {F,T}; add_lv order as the reference test (VAL before QUAL)
    add_order = ["IMAG", "EXPE", "VAL", "QUAL", "SAT", "LOY"]
    g1 = {}
    for modes_name, modes in (("A", "AAAAAA"), ("B", "BBBBBB"), ("M", "ABABAB")):
        for scheme in SCHEMES:
            for scaled in (False, True):
                cfg = build_config(Csat, lvs, sat_blocks, modes, scaled, add_order)
                _, out = run_fit(sat, cfg, scheme, lvs)
                key = "%s_%s_%d" % (modes_name, scheme, int(scaled))
                for k, v in out.items():
                    g1[key + "/" + k] = v
    save("g1_satisfaction", **g1)

    # ---- G2: synthetic N=2000 x 60 x 6 (matrix regenerated from seed; SHA-256 recorded)
    X2, blocks2 = orc.synth(2000, Csat, 10, seed=7)
    names2 = ["x%d" % i for i in range(X2.shape[1])]
    df2 = pd.DataFrame(X2, columns=names2)
    bn2 = [[names2[i] for i in b] for b in blocks2]
    g2 = {"sha256": np.array(sha(X2)), "seed": np.array(7), "n": np.array(2000)}
    for modes_name, modes in (("A", "AAAAAA"), ("B", "BBBBBB"), ("M", "BABABA")):
        for scheme in SCHEMES:
            for scaled in (False, True):
                cfg = build_config(Csat, lvs, bn2, modes, scaled)
                _, out = run_fit(df2, cfg, scheme, lvs, want_scores=False)
                key = "%s_%s_%d" % (modes_name, scheme, int(scaled))
                for k, v in out.items():
                    if k != "mv_names":
                        g2[key + "/" + k] = v
    save("g2_synth2000", **g2)

    # ---- G3: synthetic 10k x 60 x 6, Mode A, PATH, scaled (BASELINE.json config 2) + score checksums
    X3, blocks3 = orc.synth(10000, Csat, 10, seed=0)
    names3 = ["x%d" % i for i in range(X3.shape[1])]
    df3 = pd.DataFrame(X3, columns=names3)
    bn3 = [[names3[i] for i in b] for b in blocks3]
    cfg3 = build_config(Csat, lvs, bn3, "AAAAAA", True)
    m3, out3 = run_fit(df3, cfg3, "path", lvs)
    sc3 = out3.pop("scores")
    out3.pop("mv_names")
    g3 = dict(out3)
    g3.update(sha256=np.array(sha(X3)), seed=np.array(0), n=np.array(10000),
              scores_head=sc3[:64], scores_colsum=sc3.sum(axis=0), scores_gram=sc3.T @ sc3,
              scores_rows=np.arange(0, 10000, 97), scores_sample=sc3[::97])
    # G4b: bootstrap rows for 4 seeded index vectors on the 10k data
    eff_index = list(m3.effects().index)
    seeds = [11, 12, 13, 14]
    idx_list = [np.random.RandomState(s).randint(10000, size=10000) for s in seeds]
    rows, its = boot_rows(df3, cfg3, "path", lvs, idx_list, eff_index)
    g3.update(boot_seeds=np.array(seeds), boot_rows=rows, boot_iters=its)
    save("g3_synth10k_path", **g3)

    # ---- G4a: satisfaction bootstrap-by-index (8 explicit index vectors), Mode A centroid unscaled + Mode B path scaled
    rs = np.random.RandomState(2024)
    idx8 = rs.randint(250, size=(8, 250)).astype(np.int32)
    g4 = {"idx": idx8}
    for tag, modes, scheme, scaled in (("A_centroid_0", "AAAAAA", "centroid", False), ("B_path_1", "BBBBBB", "path", True),
                                       ("M_factorial_1", "ABABAB", "factorial", True)):
        cfg = build_config(Csat, lvs, sat_blocks, modes, scaled, add_order)
        m, _ = run_fit(sat, cfg, scheme, lvs)
        rows, its = boot_rows(sat, cfg, scheme, lvs, list(idx8), list(m.effects().index))
        g4[tag + "/rows"] = rows
        g4[tag + "/iters"] = its
    save("g4_satisfaction_boot", **g4)

    # ---- G5: sign-rule stress: LV "a" (2 MVs) negatively related to LV "b" (5 MVs) and "c" (4 MVs)
    rs = np.random.RandomState(5)
    n5 = 300
    ea = rs.standard_normal(n5)
    eb = -0.8 * ea + 0.6 * rs.standard_normal(n5)
    ec = -0.5 * ea + 0.4 * eb + 0.7 * rs.standard_normal(n5)
    X5 = np.column_stack([ea[:, None] * np.array([0.8, 0.7]) + 0.5 * rs.standard_normal((n5, 2)),
                          eb[:, None] * np.array([0.9, 0.8, 0.7, 0.6, 0.8]) + 0.5 * rs.standard_normal((n5, 5)),
                          ec[:, None] * np.array([0.9, 0.8, 0.7, 0.6]) + 0.5 * rs.standard_normal((n5, 4))])
    lv5 = ["a", "b", "c"]
    C5 = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    names5 = ["m%d" % i for i in range(11)]
    bn5 = [names5[0:2], names5[2:7], names5[7:11]]
    df5 = pd.DataFrame(X5, columns=names5)
    g5 = {"X": X5, "C": C5}
    for scheme in SCHEMES:
        for modes in ("AAA", "BBB"):
            cfg = build_config(C5, lv5, bn5, modes, True)
            _, out = run_fit(df5, cfg, scheme, lv5)
            out.pop("mv_names")
            for k, v in out.items():
                g5["%s_%s/%s" % (modes, scheme, k)] = v
    save("g5_sign_rule", **g5)

    # ---- G6: summary statistics (reference _create_summary on a fixed sample matrix)
    rs = np.random.RandomState(6)
    samp = rs.standard_normal((37, 5)) * np.array([1, 2, 0.5, 3, 1]) + np.array([0, 1, -1, 2, 0.3])
    orig = pd.Series(np.array([0.1, 1.2, -0.9, 2.5, 0.2]), index=list("abcde"))
    summ = refboot._create_summary(pd.DataFrame(samp, columns=list("abcde")), orig)
    save("g6_summary", samples=samp, original=orig.values, summary=summ.values.astype(float),
         columns=np.array(list(summ.columns)))

    # ---- G7: reduced BASELINE config 5: N=4000 x 200 x 20, Mode B, FACTORIAL (+ Mode A path), chain structure
    L7 = 20
    C7 = orc.chain_C(L7)
    X7, blocks7 = orc.synth(4000, C7, 10, seed=3)
    lv7 = ["L%02d" % i for i in range(L7)]
    names7 = ["x%d" % i for i in range(X7.shape[1])]
    df7 = pd.DataFrame(X7, columns=names7)
    bn7 = [[names7[i] for i in b] for b in blocks7]
    g7 = {"sha256": np.array(sha(X7)), "seed": np.array(3), "n": np.array(4000)}
    for tag, modes, scheme in (("B_factorial_1", "B" * L7, "factorial"), ("A_path_1", "A" * L7, "path"),
                               ("B_centroid_1", "B" * L7, "centroid")):
        cfg = build_config(C7, lv7, bn7, modes, True)
        _, out = run_fit(df7, cfg, scheme, lv7, want_scores=False)
        out.pop("mv_names")
        for k, v in out.items():
            g7[tag + "/" + k] = v
    save("g7_chain20", **g7)
    # ---- G8: non-metric NUM / RAW: russa (reference tests/test_regression_nonmetric.py), mobi (seminr), synthetic
    russa = pd.read_csv(os.path.join(HERE, "ref_data", "russa.csv"), index_col=0)
    lv8 = ["AGRI", "IND", "POLINS"]
    C8 = np.array([[0, 0, 0], [0, 0, 0], [1, 1, 0]])
    bn8 = [["gini", "farm", "rent"], ["gnpr", "labo"], ["ecks", "death", "demo", "inst"]]
    g8 = {}
    for modes in ("AAA", "BBB", "ABA"):
        for scheme in SCHEMES:
            for kind, dflt, mvs in (("NUM", Scale.NUM, None), ("RAW", Scale.RAW, None), ("MIX", Scale.RAW, {"gini": Scale.NUM, "ecks": Scale.NUM})):
                cfg = build_config(C8, lv8, bn8, modes, True, add_order=["POLINS", "AGRI", "IND"], default_scale=dflt, mv_scales=mvs)
                m, out = run_fit(russa, cfg, scheme, lv8, tol=1e-7)
                key = "%s_%s_%s" % (modes, scheme, kind)
                for k, v in out.items():
                    g8[key + "/" + k] = v
    rs = np.random.RandomState(88)
    idx47 = rs.randint(47, size=(6, 47)).astype(np.int32)
    g8["idx"] = idx47
    for tag, modes, scheme in (("AAA_centroid_NUM", "AAA", "centroid"), ("ABA_path_NUM", "ABA", "path")):
        cfg = build_config(C8, lv8, bn8, modes, True, add_order=["POLINS", "AGRI", "IND"], default_scale=Scale.NUM)
        m, _ = run_fit(russa, cfg, scheme, lv8, tol=1e-7)
        rows, its = boot_rows(russa, cfg, scheme, lv8, list(idx47), list(m.effects().index), tol=1e-7)
        g8[tag + "/boot_rows"] = rows
        g8[tag + "/boot_iters"] = its
    save("g8_nonmetric_russa", **g8)

    g9 = {"sha256": np.array(sha(X2)), "seed": np.array(7), "n": np.array(2000)}
    for modes_name, modes in (("A", "AAAAAA"), ("B", "BBBBBB"), ("M", "BABABA")):
        for scheme in SCHEMES:
            cfg = build_config(Csat, lvs, bn2, modes, True, default_scale=Scale.NUM)
            _, out = run_fit(df2, cfg, scheme, lvs, want_scores=False, tol=1e-7)
            key = "%s_%s" % (modes_name, scheme)
            for k, v in out.items():
                if k != "mv_names":
                    g9[key + "/" + k] = v
    save("g9_nonmetric_synth2000", **g9)
    # ---- G10: metric data with missing values (mean imputation, config.py:300 + util.py:61-68; rows with an all-NaN block dropped, config.py:273-285)
    satm = sat.copy()
    rs = np.random.RandomState(10)
    for _ in range(40):
        satm.iloc[rs.randint(250), 1 + rs.randint(27)] = np.nan       # column 0 is gender
    satm.loc[satm.index[7], [col for col in sat.columns if col.startswith("loy")]] = np.nan      # a whole LOY block missing -> row dropped
    g10 = {"data": satm[[col for lv in add_order for col in sat.columns if col.startswith(prefixes[lv])]].values.astype(float)}
    for modes_name, modes in (("A", "AAAAAA"), ("M", "ABABAB")):
        for scheme in ("centroid", "path"):
            for scaled in (False, True):
                cfg = build_config(Csat, lvs, sat_blocks, modes, scaled, add_order)
                _, out = run_fit(satm, cfg, scheme, lvs)
                key = "%s_%s_%d" % (modes_name, scheme, int(scaled))
                for k, v in out.items():
                    g10[key + "/" + k] = v
    # bootstrap replicates on explicit indices: Config.treat re-imputes every resampled data set (bootstrap.py:57, config.py:300)
    rs10 = np.random.RandomState(1010)
    g10["idx"] = rs10.randint(249, size=(5, 249))
    for tag, modes, scheme, scaled in (("A_centroid_1", "AAAAAA", "centroid", True), ("M_path_0", "ABABAB", "path", False)):
        cfg = build_config(Csat, lvs, sat_blocks, modes, scaled, add_order)
        m, _ = run_fit(satm, cfg, scheme, lvs)
        rows, its = boot_rows(satm, cfg, scheme, lvs, list(g10["idx"]), list(m.effects().index))
        g10[tag + "/boot_rows"] = rows
        g10[tag + "/boot_iters"] = its
    save("g10_metric_missing", **g10)
    # ---- G11: ORD / NOM optimal scaling (scale.py:42-89): russa categorical (reference tests/test_regression_nonmetric.py:94-120) + Likert-style synthetic
    g11 = {}
    cat = {"gnpr": Scale.ORD, "labo": Scale.ORD, "demo": Scale.NOM}
    for modes in ("AAA", "BBB"):
        for scheme in SCHEMES:
            cfg = build_config(C8, lv8, bn8, modes, True, add_order=["IND", "POLINS", "AGRI"], default_scale=Scale.NUM, mv_scales=cat)
            _, out = run_fit(russa, cfg, scheme, lv8, tol=1e-7)
            for k, v in out.items():
                g11["russa_%s_%s/%s" % (modes, scheme, k)] = v
    rs = np.random.RandomState(111)
    Cl = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [1, 1, 0, 0], [0, 1, 1, 0]])
    lvl = ["L0", "L1", "L2", "L3"]
    Xl, blocksl = orc.synth(400, Cl, 4, seed=21)
    likert = np.clip(np.round(3 + 1.2 * Xl / Xl.std(axis=0)), 1, 5)                  # 5-point items
    namesl = ["q%d" % i for i in range(16)]
    dfl = pd.DataFrame(likert, columns=namesl)
    bnl = [[namesl[i] for i in b] for b in blocksl]
    g11["likert"] = likert
    for tag, modes, mvs in (("ordA", "AAAA", {n: Scale.ORD for n in namesl}), ("ordB", "BBBB", {n: Scale.ORD for n in namesl}),
                            ("mixM", "ABAB", {n: [Scale.ORD, Scale.NOM, Scale.NUM, Scale.RAW][i % 4] for i, n in enumerate(namesl)})):
        for scheme in SCHEMES:
            cfg = build_config(Cl, lvl, bnl, modes, True, default_scale=Scale.NUM, mv_scales=mvs)
            _, out = run_fit(dfl, cfg, scheme, lvl, tol=1e-7)
            for k, v in out.items():
                if k != "mv_names":
                    g11["likert_%s_%s/%s" % (tag, scheme, k)] = v
    save("g11_ordnom", **g11)
    # ---- G12: higher order construct, two-stage approach (estimator.py:43-52; reference tests/test_regression_seminr.py:49-74), fit + bootstrap rows
    mobi = pd.read_csv(os.path.join(HERE, "ref_data", "mobi.csv"), index_col=0)
    g12 = {}
    rs12 = np.random.RandomState(1212)
    g12["idx"] = rs12.randint(250, size=(4, 250))
    for tag, scheme, qmode in (("path_B", "path", Mode.B), ("centroid_A", "centroid", Mode.A)):
        def hoc_config():
            st = c.Structure()
            st.add_path(["Expectation", "Quality"], ["Satisfaction"])
            st.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
            cfg = c.Config(st.path(), default_scale=Scale.NUM)
            cfg.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
            cfg.add_lv_with_columns_named("Expectation", Mode.A, mobi, "CUEX")
            cfg.add_lv_with_columns_named("Quality", qmode, mobi, "PERQ")
            cfg.add_lv_with_columns_named("Loyalty", Mode.A, mobi, "CUSL")
            cfg.add_lv_with_columns_named("Image", Mode.A, mobi, "IMAG")
            cfg.add_lv_with_columns_named("Complaints", Mode.A, mobi, "CUSCO")
            cfg.add_lv_with_columns_named("Value", Mode.A, mobi, "PERV")
            return cfg
        cfg = hoc_config()
        filtered = cfg.filter(mobi)
        corr = np.sqrt(250 / 249)
        calc = refw.WeightsCalculatorFactory(cfg, 100, 1e-8, corr, SCHEMES[scheme])
        rows, its1, its2 = [], [], []
        for idx in [np.arange(250)] + list(g12["idx"]):
            est = Estimator(cfg)
            del _per_solver[:]
            fd, sc, w = est.estimate(calc, filtered.iloc[idx, :])
            cfg2 = est.config()
            its1.append(_per_solver[0][1]); its2.append(_per_solver[1][1])
            lvs2 = list(cfg2.path())
            mvs2 = [mv for lv in lvs2 for mv in cfg2.mvs(lv)]
            im = refim.InnerModel(cfg2.path(), sc)
            eff = im.effects()
            ld = (sc.apply(lambda s: fd.corrwith(s)) * cfg2.odm(cfg2.path())).sum(axis=1).loc[mvs2].values.astype(float)
            rows.append(np.concatenate((w.loc[mvs2, "weight"].values.astype(float), im.r_squared().loc[lvs2].values.astype(float),
                                        eff.loc[:, "total"].values.astype(float), eff.loc[:, "direct"].values.astype(float), ld)))
        g12[tag + "/lvs2"] = np.array(lvs2); g12[tag + "/mvs2"] = np.array(mvs2)
        g12[tag + "/path2"] = cfg2.path().values.astype(int)
        g12[tag + "/eff_from"] = np.array([lvs2.index(x) for x in eff["from"]]); g12[tag + "/eff_to"] = np.array([lvs2.index(x) for x in eff["to"]])
        g12[tag + "/rows"] = np.array(rows); g12[tag + "/iters1"] = np.array(its1); g12[tag + "/iters2"] = np.array(its2)
    save("g12_hoc_two_stage", **g12)
    # ---- G13: non-metric data with missing values (weights.py:88-98, mode.py:35-39, scale.py:27-30; reference
    #      tests/test_regression_nonmetric.py:122-138): russa with the test's three NaNs + a synthetic case, fits and bootstrap rows
    g13 = {}
    russam = russa.copy()
    russam.iloc[0, 0] = np.nan; russam.iloc[3, 3] = np.nan; russam.iloc[5, 5] = np.nan
    rs13 = np.random.RandomState(1313)
    g13["idx47"] = rs13.randint(47, size=(6, 47))
    for scheme in SCHEMES:
        cfg = build_config(C8, lv8, bn8, "AAA", True, add_order=["AGRI", "IND", "POLINS"], default_scale=Scale.NUM)
        m, out = run_fit(russam, cfg, scheme, lv8, tol=1e-7)
        for k, v in out.items():
            g13["russa_%s/%s" % (scheme, k)] = v
        rows, its = boot_rows(russam, cfg, scheme, lv8, list(g13["idx47"]), list(m.effects().index), tol=1e-7)
        g13["russa_%s/boot_rows" % scheme] = rows
        g13["russa_%s/boot_iters" % scheme] = its
    # Scale.RAW only: the MVs keep the treated values (scale.py:38-39), which differ from Scale.NUM where cells are missing
    cfg = build_config(C8, lv8, bn8, "AAA", True, add_order=["AGRI", "IND", "POLINS"], default_scale=Scale.RAW)
    m, out = run_fit(russam, cfg, "centroid", lv8, tol=1e-7)
    for k, v in out.items():
        g13["russa_raw_centroid/%s" % k] = v
    rows, its = boot_rows(russam, cfg, "centroid", lv8, list(g13["idx47"][:3]), list(m.effects().index), tol=1e-7)
    g13["russa_raw_centroid/boot_rows"] = rows
    g13["russa_raw_centroid/boot_iters"] = its
    Cs = orc.satisfaction_C()
    Xs, blocks_s = orc.synth(300, Cs, 4, seed=13)
    Xm = Xs.copy()
    for _ in range(45):
        Xm[rs13.randint(300), rs13.randint(16)] = np.nan               # LVs 0-3 get holes, LVs 4-5 stay complete (Mode B allowed there)
    names13 = ["v%d" % i for i in range(24)]
    df13 = pd.DataFrame(Xm, columns=names13)
    bn13 = [[names13[i] for i in b] for b in blocks_s]
    g13["synth"] = Xm
    g13["idx300"] = rs13.randint(300, size=(3, 300))
    for tag, modes, scheme in (("A_path", "AAAAAA", "path"), ("M_centroid", "AAAABB", "centroid"), ("A_factorial", "AAAAAA", "factorial")):
        cfg = build_config(Cs, orc.SAT_LVS, bn13, modes, True, default_scale=Scale.NUM)
        m, out = run_fit(df13, cfg, scheme, orc.SAT_LVS, tol=1e-7)
        for k, v in out.items():
            if k != "mv_names":
                g13["synth_%s/%s" % (tag, k)] = v
        rows, its = boot_rows(df13, cfg, scheme, orc.SAT_LVS, list(g13["idx300"]), list(m.effects().index), tol=1e-7)
        g13["synth_%s/boot_rows" % tag] = rows
        g13["synth_%s/boot_iters" % tag] = its
    save("g13_nonmetric_missing", **g13)
    print("done in %.1f s" % (time.time() - t0))


def g14_rank_deficient():
    """G14: rank-deficient least squares.  (a) Mode-B blocks with a duplicated MV and an MV that is a linear combination of three
    others -- scipy.linalg.lstsq (gelsd) returns the minimum-norm weights (mode.py:51); (b) two LVs with identical blocks and
    identical edges, i.e. exactly collinear predecessor scores -- statsmodels' pinv returns equal coefficients in the PATH scheme
    (scheme.py:50) and in the inner model (inner_model.py:69).  Fits + bootstrap rows on explicit indices."""
    sat = pd.read_csv(os.path.join(HERE, "ref_data", "satisfaction.csv"), index_col=0)
    lvs = orc.SAT_LVS
    Csat = orc.satisfaction_C()
    prefixes = dict(IMAG="imag", EXPE="expe", QUAL="qual", VAL="val", SAT="sat", LOY="loy")
    blocks = [[col for col in sat.columns if col.startswith(prefixes[lv])] for lv in lvs]
    g = {}
    df = sat.copy()
    df["imag1dup"] = df["imag1"]
    df["satlin"] = 0.5 * df["sat1"] - 2.0 * df["sat2"] + df["sat3"]
    ba = [list(b) for b in blocks]
    ba[0].append("imag1dup"); ba[4].append("satlin")
    cols_a = [n for b in ba for n in b]
    g["a/X"] = df[cols_a].values.astype(float)
    g["a/block_sizes"] = np.array([len(b) for b in ba])
    rs = np.random.RandomState(1414)
    g["idx"] = rs.randint(250, size=(4, 250)).astype(np.int32)
    for modes_name, modes in (("B", "BBBBBB"), ("M", "BABABA")):
        for scheme in SCHEMES:
            for scaled in (False, True):
                cfg = build_config(Csat, lvs, ba, modes, scaled)
                m, out = run_fit(df, cfg, scheme, lvs)
                key = "a_%s_%s_%d" % (modes_name, scheme, int(scaled))
                assert list(out.pop("mv_names")) == cols_a
                for k, v in out.items():
                    g[key + "/" + k] = v
                if scheme == "path" or (modes_name == "B" and scheme == "centroid" and scaled):
                    rows, its = boot_rows(df, cfg, scheme, lvs, list(g["idx"]), list(m.effects().index))
                    g[key + "/boot_rows"] = rows
                    g[key + "/boot_iters"] = its
    # (b) EXPE2 = a clone of EXPE (same columns under new names, same edges)
    lv7 = ["IMAG", "EXPE", "EXPE2", "QUAL", "VAL", "SAT", "LOY"]
    edges = [("IMAG", "EXPE"), ("IMAG", "EXPE2"), ("IMAG", "SAT"), ("IMAG", "LOY"), ("EXPE", "QUAL"), ("EXPE", "VAL"), ("EXPE", "SAT"),
             ("EXPE2", "QUAL"), ("EXPE2", "VAL"), ("EXPE2", "SAT"), ("QUAL", "VAL"), ("QUAL", "SAT"), ("VAL", "SAT"), ("SAT", "LOY")]
    C7 = np.zeros((7, 7), dtype=int)
    for frm, to in edges:
        C7[lv7.index(to), lv7.index(frm)] = 1
    dfb = sat.copy()
    bb = []
    for lv in lv7:
        if lv == "EXPE2":
            names = []
            for col in blocks[1]:
                dfb[col + "x"] = dfb[col]
                names.append(col + "x")
            bb.append(names)
        else:
            bb.append(list(blocks[lvs.index(lv)]))
    cols_b = [n for b in bb for n in b]
    g["b/X"] = dfb[cols_b].values.astype(float)
    g["b/block_sizes"] = np.array([len(b) for b in bb])
    g["b/C"] = C7
    for modes_name, modes in (("A", "AAAAAAA"), ("B", "BBBBBBB")):
        for scheme in SCHEMES:
            cfg = build_config(C7, lv7, bb, modes, True)
            m, out = run_fit(dfb, cfg, scheme, lv7)
            key = "b_%s_%s" % (modes_name, scheme)
            assert list(out.pop("mv_names")) == cols_b
            for k, v in out.items():
                g[key + "/" + k] = v
            if scheme == "path":
                rows, its = boot_rows(dfb, cfg, scheme, lv7, list(g["idx"][:2]), list(m.effects().index))
                g[key + "/boot_rows"] = rows
                g[key + "/boot_iters"] = its
    save("g14_rank_deficient", **g)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_rank_deficient()
    else:
        main()
        g14_rank_deficient()
