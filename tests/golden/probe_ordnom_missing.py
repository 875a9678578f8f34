#!/opt/conda/bin/python3.9
"""BUILD-CONTAINER ONLY (needs /root/reference + statsmodels; run under /opt/conda/bin/python3.9): what the real reference does with missing cells in
Scale.ORD / Scale.NOM columns.  Writes tests/golden/g16_ordnom_missing_probe.json -- one record per probed case:
{"case", "modes", "scheme", "outcome": "raises" | "estimates", "exception", "message", "raised_at", "weights", "iterations",
 "weights_rows_permuted", "max_abs_weight_change_under_row_permutation"}.

Why a probe and not a golden: the reference has no defined behaviour here.  util.rank / util.dummy (util.py:80-96) count NaN as one more category whose
indicator column is all zeros; scale.py:60-61,83-84 quantify through util.groupby_mean, whose dict gets one key PER NaN cell (NaN != NaN) and whose
sorted() leaves a NaN key wherever the first appearances of the categories put it.  The outcome is one of: statsmodels MissingDataError (Mode B; PATH
scheme), "Could not converge", a shape error (two or more NaNs in a column), all-zero weights (ORD), or an estimate whose category means are assigned to the
wrong categories -- which changes when the ROWS of the data set are permuted, something no PLS-PM estimate may do.  This script records all of it."""
import json
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
sys.path.insert(0, HERE)
sys.argv = [sys.argv[0], "none"]
import make_golden as mg  # noqa: E402  (loads the reference through oracle/refshim.py)
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from plspm.scale import Scale  # noqa: E402

LVS = ["AGRI", "IND", "POLINS"]
C = np.array([[0, 0, 0], [0, 0, 0], [1, 1, 0]])
BLOCKS = [["gini", "farm", "rent"], ["gnpr", "labo"], ["ecks", "death", "demo", "inst"]]
CAT = {"gnpr": Scale.ORD, "labo": Scale.ORD, "demo": Scale.NOM}


def attempt(frame, modes, scheme):
    try:
        cfg = mg.build_config(C, LVS, BLOCKS, modes, True, add_order=["IND", "POLINS", "AGRI"], default_scale=Scale.NUM, mv_scales=CAT)
        _, out = mg.run_fit(frame, cfg, scheme, LVS, tol=1e-7)
        return {"outcome": "estimates", "weights": [float(x) for x in out["weights"]], "mv_names": [str(x) for x in out["mv_names"]], "iterations": int(out["iters"])}
    except Exception as e:                                   # noqa: BLE001 -- the point of the probe
        tb = traceback.extract_tb(e.__traceback__)
        return {"outcome": "raises", "exception": type(e).__module__ + "." + type(e).__name__, "message": str(e)[:200],
                "raised_at": ["%s:%d" % (os.path.basename(f.filename), f.lineno) for f in tb if "/reference/" in f.filename][-3:]}


def main():
    russa = pd.read_csv(os.path.join(HERE, "ref_data", "russa.csv"), index_col=0)
    perm = np.random.RandomState(16).permutation(len(russa))
    cases = {"one NaN in an ORD column": [(0, "gnpr")], "one NaN in a NOM column": [(2, "demo")], "NaNs in two ORD columns and the NOM column": [(0, "gnpr"), (4, "labo"), (2, "demo")],
             "two NaNs in the NOM column": [(2, "demo"), (9, "demo")], "one NaN in an ORD and one in a NUM column": [(0, "gnpr"), (3, "gini")]}
    records = []
    for name, holes in cases.items():
        frame = russa.copy()
        for i, col in holes:
            frame.loc[frame.index[i], col] = np.nan
        for modes in ("AAA", "BBB"):
            for scheme in ("centroid", "factorial", "path"):
                rec = {"case": name, "holes": [[int(i), col] for i, col in holes], "modes": modes, "scheme": scheme}
                rec.update(attempt(frame, modes, scheme))
                if rec["outcome"] == "estimates":
                    again = attempt(frame.iloc[perm], modes, scheme)      # the same rows in another order
                    rec["rows_permuted"] = again
                    if again["outcome"] == "estimates":
                        rec["max_abs_weight_change_under_row_permutation"] = float(np.max(np.abs(np.array(again["weights"]) - np.array(rec["weights"]))))
                records.append(rec)
                print(name, modes, scheme, rec["outcome"], rec.get("exception", ""), rec.get("max_abs_weight_change_under_row_permutation", ""), flush=True)
    # control: the same permutation on COMPLETE data leaves the estimate alone (to rounding)
    ctl = attempt(russa, "AAA", "centroid"); ctl2 = attempt(russa.iloc[perm], "AAA", "centroid")
    control = float(np.max(np.abs(np.array(ctl["weights"]) - np.array(ctl2["weights"]))))
    with open(os.path.join(HERE, "g16_ordnom_missing_probe.json"), "w") as fh:
        json.dump({"generator": "tests/golden/probe_ordnom_missing.py", "reference": "plspm-python at /root/reference through oracle/refshim.py",
                   "control_complete_data_max_abs_weight_change_under_row_permutation": control, "records": records}, fh, indent=1)


if __name__ == "__main__":
    main()
