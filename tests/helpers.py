"""Shared helpers for the parity tests (test infrastructure)."""
import hashlib
import os

import numpy as np
import pandas as pd

import plspm_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAT_PREFIX = dict(IMAG="imag", EXPE="expe", QUAL="qual", VAL="val", SAT="sat", LOY="loy")
SAT_ADD_ORDER = ["IMAG", "EXPE", "VAL", "QUAL", "SAT", "LOY"]   # as tests/golden/make_golden.py (and the reference test)
MODES = {"A": "AAAAAA", "B": "BBBBBB"}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def satisfaction_frame():
    return pd.read_csv(os.path.join(GOLDEN, "ref_data", "satisfaction.csv"), index_col=0)


def satisfaction_oracle_inputs():
    """X in data-column (add_lv) order + blocks (path order) for the satisfaction fixtures."""
    sat = satisfaction_frame()
    cols = []
    for lv in SAT_ADD_ORDER:
        cols += [c for c in sat.columns if c.startswith(SAT_PREFIX[lv])]
    X = sat[cols].values.astype(np.float64)
    blocks = [np.array([i for i, c in enumerate(cols) if c.startswith(SAT_PREFIX[lv])]) for lv in orc.SAT_LVS]
    return X, blocks, cols


def case_modes(name, L=6, mixed="ABABAB"):
    if name == "M":
        return mixed
    return name * L


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def assert_close(a, b, rtol, atol=0.0, what=""):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol,
                               err_msg=what)
