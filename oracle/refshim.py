"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference, usable only in the build container.

Imports GoogleCloudPlatform/plspm-python from /root/reference under /opt/conda/bin/python3.9
(the only interpreter here with statsmodels) after applying four plumbing shims that touch no
least-squares / correlation / normalisation arithmetic:

  1. numpy.MachAr stub            (statsmodels 0.12.2 tools/numdiff.py needs it under numpy>=1.24)
  2. pandas.Int64Index & friends  (statsmodels 0.12.2 tsa/base/tsa_model.py)
  3. pandas.DataFrame.append      (removed in pandas 2; used by reference bootstrap.py:58-63,101-105,
                                   inner_model.py:52,74)
  4. statsmodels add_constant for DataFrame/Series input (old statsmodels calls pd.concat(x, 1))

Nothing from /root/reference is copied; it is imported in place.  This module never travels to the
GPU box in a usable form (the reference itself is absent there) and is used only by
tests/golden/make_golden.py and oracle/time_reference.py.
"""
import os
import sys
import warnings

REFERENCE_ROOT = os.environ.get("PLSPM_REFERENCE_ROOT", "/root/reference")


def load_reference():
    warnings.filterwarnings("ignore")
    import numpy as np
    import pandas as pd

    if not hasattr(np, "MachAr"):
        class MachAr:  # noqa: D401 - minimal stand-in exposing what statsmodels reads
            def __init__(self, *a, **k):
                fi = np.finfo(float)
                self.eps, self.tiny, self.huge = fi.eps, fi.tiny, fi.max
                self.epsneg, self.xmin, self.xmax = fi.epsneg, fi.tiny, fi.max
        np.MachAr = MachAr
    for name in ("Int64Index", "Float64Index", "UInt64Index"):
        if not hasattr(pd, name):
            setattr(pd, name, pd.Index)

    if not hasattr(pd.DataFrame, "append"):
        def _append(self, other, ignore_index=False, **_kw):
            if isinstance(other, pd.Series):
                other = other.to_frame().T
            elif isinstance(other, dict):
                other = pd.DataFrame([other])
            return pd.concat([self, other], ignore_index=ignore_index, sort=False)
        pd.DataFrame.append = _append

    import statsmodels.api as sm
    import statsmodels.tools.tools as smtools
    _orig_add_constant = smtools.add_constant

    def _add_constant(data, prepend=True, has_constant="skip"):
        if isinstance(data, pd.Series):
            data = data.to_frame()
        if isinstance(data, pd.DataFrame):
            out = data.copy()
            out.insert(0, "const", 1.0)
            return out
        return _orig_add_constant(data, prepend=prepend, has_constant=has_constant)
    sm.add_constant = _add_constant

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import plspm  # noqa: F401  (the reference package)
    assert os.path.realpath(plspm.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), plspm.__file__
    import plspm.config, plspm.plspm, plspm.mode, plspm.scheme, plspm.estimator  # noqa: E401,F401
    import plspm.weights, plspm.inner_model, plspm.bootstrap  # noqa: E401,F401
    return plspm
