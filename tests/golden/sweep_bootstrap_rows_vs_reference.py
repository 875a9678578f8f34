#!/opt/conda/bin/python3.9
"""Bootstrap ROWS of the oracle against the REAL reference on seeded random metric models in which an item comes out CONSTANT in some replicates (build container only).

Run:   PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/sweep_bootstrap_rows_vs_reference.py A B [flat | cat | missing | nmx]

Model of seed s: fuzz_cases.make_case (metric ones), one column replaced by a rare 0/1 indicator (2 ... 6 ones among the rows); explicit index lists -- the data themselves, two
ordinary resamples and up to three resamples that miss every one (the column is constant there).  The reference's row is built as BootstrapProcess.run builds it
(bootstrap.py:56-64, through make_golden.boot_rows); the oracle's by bootstrap_replicate.  A constant column gives weight 0 and loading 0 in both and the replicate counts."""
import collections
import os
import sys
import traceback
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
import plspm_oracle as orc  # noqa: E402
import fuzz_cases as fc  # noqa: E402

warnings.filterwarnings("ignore")


def check(seed, kind="flat"):
    if kind == "flat":
        case = fc.make_rare_indicator_case(seed)
        if case is None:
            return "skipped (Scale.NUM)"
        X, model, idx_list = case
    else:
        # other data kinds on plain resamples: categorical models cut down to 40 ... 90 rows (resamples lose categories: util.rank re-ranks the present ones),
        # metric data with NaN cells (re-imputed per replicate), Scale.NUM data with incomplete rows
        rng = np.random.default_rng(19000 + seed)
        if kind == "cat":
            X, model = fc.make_cat_case(seed)
            X = X[rng.choice(X.shape[0], size=min(X.shape[0], int(rng.integers(40, 91))), replace=False)]
        elif kind == "missing":
            X, model = fc.make_missing_case(seed)
            X = orc.filter_missing(X, model)
        else:
            X, model = fc.make_nmx_case(seed)
        idx_list = [np.arange(X.shape[0])] + [rng.integers(0, X.shape[0], size=X.shape[0]) for _ in range(4)]
    n, P = X.shape
    lvs = ["L%d" % l for l in range(model.L)]
    names = ["x%d" % p for p in range(P)]
    df = pd.DataFrame(X, columns=names)
    from plspm.scale import Scale
    SC = {"NUM": Scale.NUM, "RAW": Scale.RAW, "ORD": Scale.ORD, "NOM": Scale.NOM}
    cfg = c.Config(mg.path_frame(model.C, lvs), scaled=model.scaled, default_scale=(Scale.NUM if model.scales is not None else None))
    for l in range(model.L):
        cfg.add_lv(lvs[l], Mode.A if model.modes[l] == "A" else Mode.B, *[c.MV(names[p], SC[model.scales[p]] if model.scales is not None else None) for p in model.blocks[l]])
    try:
        m = Plspm(df, cfg, mg.SCHEMES[model.scheme], 100, model.tol)
    except Exception:                                      # noqa: BLE001  (the full sample itself cannot be estimated: nothing to compare)
        return "full sample fails in the reference"
    eff_index = list(m.effects().index)
    corr = orc.correction(n)
    out = collections.Counter()
    for k, idx in enumerate(idx_list):
        ref_err = mine_err = None
        try:
            rows, its = mg.boot_rows(df, cfg, model.scheme, lvs, [idx], eff_index, tol=model.tol)
        except Exception as e:                             # noqa: BLE001
            ref_err = e
        try:
            with np.errstate(all="ignore"):
                mine, mits = orc.bootstrap_replicate(X, model, idx, corr)
        except Exception as e:                             # noqa: BLE001
            mine_err = e
        if ref_err is not None or mine_err is not None:
            assert (ref_err is None) == (mine_err is None), "replicate %d: reference %r / oracle %r" % (k, ref_err, mine_err)
            out["both-raise"] += 1
            continue
        # the reference's row is in filtered-column order = add_lv order = block order; the oracle's weights / loadings are in data-column order
        order = np.concatenate(model.blocks)
        ne = len(eff_index)
        mine_r = np.concatenate((mine[:P][order], mine[P:P + model.L + 2 * ne], mine[P + model.L + 2 * ne:][order]))
        assert its[0] == mits, "replicate %d: iterations %d vs oracle %d" % (k, its[0], mits)
        if not np.allclose(rows[0], mine_r, rtol=1e-9, atol=1e-11, equal_nan=True):
            raise AssertionError("replicate %d (constant: %s): max abs diff %.3e" % (k, k >= 3, float(np.nanmax(np.abs(rows[0] - mine_r)))))
        out["flat-ok" if (kind == "flat" and k >= 3) else "ok"] += 1
    return "+".join("%s:%d" % kv for kv in sorted(out.items()))


if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    kind = sys.argv[3] if len(sys.argv) > 3 else "flat"       # flat | cat | missing | nmx
    hist, bad = collections.Counter(), []
    for seed in range(a, b):
        try:
            hist[check(seed, kind)] += 1
        except Exception:                                  # noqa: BLE001
            bad.append((seed, traceback.format_exc().splitlines()[-1][:300]))
    print("outcomes", dict(hist))
    print("disagreements", len(bad))
    for x in bad[:40]:
        print(x)
