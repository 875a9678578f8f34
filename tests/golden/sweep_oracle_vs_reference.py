#!/opt/conda/bin/python3.9
"""Oracle against the REAL reference on seeded random models (build container only, like make_golden.py: the reference never travels).

Run:   PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/sweep_oracle_vs_reference.py KIND A B        KIND = cat | catbig | missing | nmx | hoc | hocord | metric | edge | hostile | huge

The fixtures g1-g16 pin the oracle on the reference's own data sets and a handful of synthetic ones; the GPU fuzz (tests/test_gpu_fuzz.py) then holds the device against the ORACLE on
thousands of random models.  This sweep closes the triangle for the same generators (tests/fuzz_cases.py): the reference's public API (Plspm(...), fit only) on the model of seed s,
the oracle on the same matrix -- iteration counts equal, weights / loadings / path coefficients / R2 / scores to 1e-9.  A reference run that raises must be an oracle run that raises.
Prints the outcome histogram and every disagreement; writes nothing."""
import collections
import os
import sys
import traceback
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402  (loads the reference through oracle/refshim.py; its main() does not run on import)
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import plspm.config as c  # noqa: E402  (the reference's)
from plspm.mode import Mode  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.scale import Scale  # noqa: E402
import plspm_oracle as orc  # noqa: E402
import fuzz_cases as fc  # noqa: E402

warnings.filterwarnings("ignore")
SCALE = {"NUM": Scale.NUM, "RAW": Scale.RAW, "ORD": Scale.ORD, "NOM": Scale.NOM}
RTOL, ATOL = 1e-9, 1e-11


def close(a, b, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if not np.allclose(a, b, rtol=RTOL, atol=ATOL, equal_nan=True):
        raise AssertionError("%s: max abs diff %.3e" % (what, float(np.nanmax(np.abs(a - b)))))


def reference_fit(X, model, tol, hoc=None):
    """The reference through its public API.  hoc = (stage2, C2, modes2): LV names L0.. of stage 1, the HOC named H."""
    L = model.L
    lvs = ["L%d" % l for l in range(L)]
    names = [["x%d" % p for p in b] for b in model.blocks]
    df = pd.DataFrame(X, columns=["x%d" % p for p in range(X.shape[1])])
    scales = model.scales
    if hoc is None:
        cfg = c.Config(mg.path_frame(model.C, lvs), scaled=model.scaled, default_scale=(Scale.NUM if scales is not None else None))
        for l in range(L):
            cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B, *[c.MV(n, SCALE[scales[p]] if scales is not None else None) for n, p in zip(names[l], model.blocks[l])])
        order = lvs
    else:
        stage2, C2, modes2 = hoc
        lv2 = ["H" if kind == "hoc" else lvs[ref] for kind, ref in stage2]
        cfg = c.Config(mg.path_frame(C2, lv2), scaled=True, default_scale=SCALE[scales[0]])
        for (kind, ref), name, mode in zip(stage2, lv2, modes2):
            if kind == "hoc":
                cfg.add_higher_order(name, Mode.A if mode == "A" else Mode.B, [lvs[j] for j in ref])
        for l in range(L):
            cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B, *[c.MV(n) for n in names[l]])
        order = lv2
    mg._calls["n"] = 0
    del mg._per_solver[:]
    m = Plspm(df, cfg, mg.SCHEMES[model.scheme], 100, tol)
    iters = mg._per_solver[-1][1] if mg._per_solver else mg._calls["n"] // 2       # (non-metric: the last solver instance; metric: the solver runs twice per estimate)
    om = m.outer_model()
    return dict(iterations=iters, outer=om, path_coef=m.path_coefficients().loc[order, order].values.astype(float),
                r2=m.inner_summary().loc[order, "r_squared"].values.astype(float), scores=m.scores().loc[:, order].values.astype(float))


def check(kind, seed):
    hoc = None
    erratic = False
    if kind == "cat":
        X, model = fc.make_cat_case(seed)
    elif kind == "missing":
        X, model = fc.make_missing_case(seed)
    elif kind == "nmx":
        X, model = fc.make_nmx_case(seed)
    elif kind == "metric":
        X, model, nonmetric = fc.make_case(seed)
    elif kind == "smallint":
        X, model = fc.make_small_int_case(seed)[:2]
    elif kind == "catbig":
        X, model = fc.make_cat_big_case(seed)
    elif kind == "edge":
        X, model, _, edge_kind = fc.make_degenerate_case(seed)
        if edge_kind == 3:                                     # Plspm() clamps `iterations` below 100 to 100 (plspm.py:54-55): the cap of 1 ... 4 exists at the C-ABI only
            model = orc.Model(model.blocks, model.C, model.modes, model.scheme, model.scaled, max_iter=100, tol=model.tol, scales=model.scales)
        # an exact copy / multiple of a column inside a Mode-B block: scipy.linalg.lstsq (mode.py:51) cuts singular values at machine epsilon, and the copy's sits AT that
        # threshold -- the reference either returns the minimum-norm weights (what the oracle and the device always do: cut-off eps * max(M, N) / 1e-12) or keeps a singular
        # value of 1e-16 and iterates on weights of 1e13 until "could not converge".  Such disagreements are the reference's coin toss, reported apart.
        erratic = edge_kind == 1 and "B" in model.modes
    elif kind == "hostile":
        X, model = fc.make_hostile_case(seed)[:2]
    elif kind == "huge":
        X, model = fc.make_huge_case(seed)[:2]
    else:
        X, model, stage2, C2, modes2, _ = (fc.make_hoc_ord_case if kind == "hocord" else fc.make_hoc_case)(seed)
        hoc = (stage2, C2, modes2)
    tol = model.tol
    ref_err = orc_err = None
    try:
        ref = reference_fit(X, model, tol, hoc)
    except Exception as e:                                 # noqa: BLE001
        ref_err = e
    try:
        with np.errstate(all="ignore"):
            Xf = orc.filter_missing(X, model) if np.isnan(X).any() else X          # Config.filter (config.py:273-285): what Plspm() does before the estimate
            mine = orc.fit_two_stage(Xf, model, hoc[0], hoc[1], hoc[2]) if hoc else orc.fit(Xf, model)
    except Exception as e:                                 # noqa: BLE001
        orc_err = e
    if ref_err is not None or orc_err is not None:
        if (ref_err is None) != (orc_err is None):
            if erratic and orc_err is None:
                return "reference-erratic (gelsd at its rank threshold)"
            raise AssertionError("reference %r / oracle %r" % (ref_err, orc_err))
        return "both-raise"
    if not np.all(np.isfinite(ref["scores"])):
        return "reference-nonfinite" if not np.all(np.isfinite(mine["scores"])) else "reference-nonfinite-ORACLE-FINITE"
    if erratic and ref["iterations"] != mine["iterations"]:
        return "reference-erratic (gelsd at its rank threshold)"
    assert ref["iterations"] == mine["iterations"], "iterations %d vs oracle %d" % (ref["iterations"], mine["iterations"])
    if hoc is None:
        names = ["x%d" % p for p in range(X.shape[1])]
    else:                                                  # stage-2 MV order: stage-2 LV by LV, a HOC's MVs named after its constituents
        names = []
        for kind2, ref2 in hoc[0]:
            names += ["L%d" % j for j in ref2] if kind2 == "hoc" else ["x%d" % p for p in model.blocks[ref2]]
    try:
        close(ref["outer"].loc[names, "weight"].values, mine["weights"], "weights")
    except AssertionError:
        if erratic:
            return "reference-erratic (gelsd at its rank threshold)"
        raise
    close(ref["outer"].loc[names, "loading"].values, mine["loadings"], "loadings")
    close(ref["path_coef"], mine["path_coef"], "path coefficients")
    close(ref["r2"], mine["r2"], "r2")
    close(ref["scores"], mine["scores"], "scores")                 # (the reference's scores carry the filtered rows only, like the oracle's)
    return "ok"


if __name__ == "__main__":
    kind, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        try:
            hist[check(kind, seed)] += 1
        except Exception:                                  # noqa: BLE001
            bad.append((seed, traceback.format_exc().splitlines()[-1][:300]))
    print(kind, "outcomes", dict(hist))
    print("disagreements", len(bad))
    for x in bad[:40]:
        print(x)
