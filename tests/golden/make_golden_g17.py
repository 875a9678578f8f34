#!/opt/conda/bin/python3.9
"""Golden g17 (build container only): the reference's PUBLIC frames on seeded random models -- outer_model, inner_model (estimates, standard errors, t, p), inner_summary,
path_coefficients, crossloadings, effects, unidimensionality, goodness_of_fit -- for the host-side statistics this backend computes from device outputs (plspm/inner_model.py,
inner_summary.py, outer_model.py, unidimensionality.py).  The fixtures of the reference's own data sets pin those on four models; this one adds 12 metric / Scale.NUM models of
tests/fuzz_cases.make_case, 6 categorical ones of make_cat_case and 3 + 3 with NaN cells (make_missing_case, make_nmx_case).  Data + expected outputs only (arrays and label strings).

Run:   PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_g17.py"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import plspm.config as c  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.scale import Scale  # noqa: E402
import fuzz_cases as fc  # noqa: E402

warnings.filterwarnings("ignore")
SCALE = {"NUM": Scale.NUM, "RAW": Scale.RAW, "ORD": Scale.ORD, "NOM": Scale.NOM}


def frames(X, model, hoc=None, unidim=True):
    """hoc = (stage2, C2, modes2) of fuzz_cases.make_hoc_case: the HOC is named H, its MVs after the constituents.  (The reference's unidimensionality() raises KeyError on a
    HOC model -- it looks the HOC's score columns up in the filtered data, unidimensionality.py:39 --: not part of those models' fixtures.)"""
    L = model.L
    lvs = ["L%d" % l for l in range(L)]
    df = pd.DataFrame(X, columns=["x%d" % p for p in range(X.shape[1])])
    scales = model.scales
    if hoc is None:
        cfg = c.Config(mg.path_frame(model.C, lvs), scaled=model.scaled, default_scale=(Scale.NUM if scales is not None else None))
    else:
        stage2, C2, modes2 = hoc
        lv2 = ["H" if kind == "hoc" else lvs[ref] for kind, ref in stage2]
        cfg = c.Config(mg.path_frame(C2, lv2), scaled=True, default_scale=SCALE[scales[0]])
        for (kind, ref), name, mode in zip(stage2, lv2, modes2):
            if kind == "hoc":
                cfg.add_higher_order(name, Mode.A if mode == "A" else Mode.B, [lvs[j] for j in ref])
    for l in range(L):
        cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B, *[c.MV("x%d" % p, SCALE[scales[p]] if (scales is not None and hoc is None) else None) for p in model.blocks[l]])
    m = Plspm(df, cfg, mg.SCHEMES[model.scheme], 100, model.tol)
    out = {}
    for name, fr in (("outer_model", m.outer_model()), ("inner_model", m.inner_model()), ("inner_summary", m.inner_summary()), ("path_coefficients", m.path_coefficients()),
                     ("crossloadings", m.crossloadings())) + ((("unidimensionality", m.unidimensionality()),) if (hoc is None and unidim) else ()):
        num = fr.select_dtypes(include=[np.number])
        out[name + "/values"] = num.values.astype(float)
        out[name + "/index"] = np.array([str(i) for i in fr.index])
        out[name + "/columns"] = np.array([str(x) for x in num.columns])
    eff = m.effects()
    out["effects/from"] = np.array([str(x) for x in eff["from"]]); out["effects/to"] = np.array([str(x) for x in eff["to"]])
    out["effects/values"] = eff[["direct", "indirect", "total"]].values.astype(float)
    out["gof"] = np.array(float(m.goodness_of_fit()))
    return out


def main():
    store = {}
    cases = []
    for seed in range(40):
        X, model, nonmetric = fc.make_case(seed)
        if any(len(b) < 2 for b in model.blocks):
            continue                                           # (goodness_of_fit / unidimensionality want blocks of two and more MVs)
        cases.append(("metric", seed, X, model))
        if sum(1 for k in cases if k[0] == "metric") == 12:
            break
    for seed in range(60):
        X, model = fc.make_cat_case(seed)
        if any(len(b) < 2 for b in model.blocks):
            continue
        cases.append(("cat", seed, X, model))
        if sum(1 for k in cases if k[0] == "cat") == 6:
            break
    for kind, gen, want in (("missing", fc.make_missing_case, 3), ("nmx", fc.make_nmx_case, 3)):      # NaN cells: metric (mean imputation) / Scale.NUM (incomplete rows)
        for seed in range(80):
            X, model = gen(seed)
            if any(len(b) < 2 for b in model.blocks):
                continue
            cases.append((kind, seed, X, model))
            if sum(1 for k in cases if k[0] == kind) == want:
                break
    for kind, gen, want in (("hocnum", fc.make_hoc_case, 3), ("hocord", fc.make_hoc_ord_case, 3)):      # two-stage estimates (the HOC block has two or three score columns)
        for seed in range(80):
            X, model, stage2, C2, modes2, _ = gen(seed)
            if any(len(b) < 2 for b in model.blocks):
                continue
            cases.append((kind, seed, X, (model, (stage2, C2, modes2))))
            if sum(1 for k in cases if k[0] == kind) == want:
                break
    # metric models with a CONSTANT column (fuzz_cases.make_degenerate_case kind 2): the reference centres it to zeros -- weight 0, loading 0 (pandas' sum skips the NaN of its
    # correlations), NaN cross-loadings; its unidimensionality() raises ValueError there (PCA on NaN) and is left out
    nflat = 0
    for seed in range(400):
        X, model, nonmetric, k2 = fc.make_degenerate_case(seed)
        if k2 != 2 or nonmetric or any(len(b) < 2 for b in model.blocks):
            continue
        cases.append(("flat", seed, X, model))
        nflat += 1
        if nflat == 4:
            break
    nraw = 0
    for seed in range(60):                                     # Scale.RAW / RAW + NUM mixes: Config.treat's scale rules (config.py:309-313)
        X, model = fc.make_raw_case(seed)
        if any(len(b) < 2 for b in model.blocks):
            continue
        cases.append(("raw", seed, X, model))
        nraw += 1
        if nraw == 4:
            break
    tags = []
    for kind, seed, X, model in cases:
        tag = "%s%d" % (kind, seed)
        hoc = None
        if isinstance(model, tuple):
            model, hoc = model
        try:
            out = frames(X, model, hoc, unidim=(kind != "flat"))
        except Exception as e:                                 # noqa: BLE001
            print(tag, "reference raised", repr(e)[:120])
            continue
        tags.append(tag)
        store[tag + "/x_sha"] = np.array(mg.sha(X))            # the matrix the generator must reproduce on the GPU box (NumPy's Generator streams are version-stable)
        for k, v in out.items():
            store[tag + "/" + k] = v
        print(tag, X.shape, model.modes, model.scheme, "gof %.6f" % float(out["gof"]), "HOC" if hoc else "")
    store["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(HERE, "g17_api_frames.npz"), **store)
    print("wrote g17_api_frames.npz:", len(tags), "models,", os.path.getsize(os.path.join(HERE, "g17_api_frames.npz")), "bytes")


if __name__ == "__main__":
    main()
