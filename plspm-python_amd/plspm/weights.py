"""The weight-solver seam (reference plspm/weights.py:157-187), backed by libplspm_hip.so.

``WeightsCalculatorFactory(config, iterations, tolerance, correction, scheme)`` keeps the reference's
constructor, ``clone()``, ``config()`` and ``calculate(data, path) -> (final_data, scores, weights)``.
``calculate`` receives *treated* data exactly like the reference's (estimator.py:33,39) and therefore runs
the device solver with centring only.  ``Estimator`` uses ``run()`` instead, which uploads the raw filtered
data once and lets the device moments stage apply the treatment (config.py:299-305) itself -- that handle
then also serves the bootstrap.
"""
import numpy as np
import pandas as pd

from plspm import _native
from plspm._compile import augment, compile_model, with_missing_indicators
from plspm.scale import Scale
from plspm.scheme import Scheme


class ConvergenceError(Exception):
    pass


class NumericalConditionError(Exception):
    """The device solver flagged the full-sample fit: a zero-variance / non-finite input (PLSPM_NONFINITE) or a regression that
    the reference cannot solve either (PLSPM_SINGULAR).  Raised instead of returning frames built from meaningless numbers."""


class SolverResult:
    """Everything one device fit produced, still in device column order, plus the labels to unpack it."""

    def __init__(self, compiled, native, raw, index, builder=None):
        self.compiled = compiled
        self.native = native
        self.raw = raw
        self.index = index
        self.builder = builder          # builder(device_id) -> another uploaded handle of the same model + data (multi-GPU bootstrap)

    # frames with the reference's labels / ordering contracts
    def scores(self) -> pd.DataFrame:
        return pd.DataFrame(self.raw["scores"], index=self.index, columns=self.compiled.lvs)

    def weights(self) -> pd.DataFrame:
        return pd.DataFrame({"weight": self.raw["weights"]}, index=self.compiled.dev_mvs)

    def by_data_column(self, key):
        """A per-MV device vector / matrix re-ordered to the (used) data columns' order."""
        inv = self.compiled.inv_index
        return self.raw[key][inv[inv >= 0]]


class WeightsCalculatorFactory:
    """Calculates weights and scores from the data using the model, on the GPU."""

    def __init__(self, config, iterations: int, tolerance: float, correction: float, scheme: Scheme, device_id: int = 0, precision: str = "auto"):
        if precision not in ("auto", "strict"):
            raise ValueError("precision must be 'auto' or 'strict'")
        self._precision = precision
        self._config = config
        self._iterations = iterations
        self._tolerance = tolerance
        self._correction = correction
        self._scheme = scheme
        self._device_id = device_id

    def clone(self):
        return WeightsCalculatorFactory(self._config.clone(), self._iterations, self._tolerance, self._correction, self._scheme,
                                        self._device_id, self._precision)

    def apply_precision(self, handle):
        """``precision="strict"``: never fewer than SEVEN base-256 digit planes per pair product (>= 54 significant bits of every column
        maximum, exact integer accumulation; include/plspm_hip.h "i8_min_slices") -- the automatic choice takes six when the uploaded data
        keep the representation error below a quarter of the a-priori bound of the fp64 accumulation the reference performs
        (weights.py:43,60-61), and eight by itself when one gross outlier carries a pair column.  Costs ~20 % of the bootstrap rate."""
        if self._precision == "strict":
            handle.set_option("i8_min_slices", 7)
        return handle

    def config(self):
        return self._config

    def scheme(self):
        return self._scheme

    def _nonmetric(self):
        """0 metric, 1 Scale.NUM / RAW only (correlation-matrix solver), 2 Scale.ORD / NOM present (categorical solver)."""
        if self._config.metric():
            return 0
        self._config.promote_scales()
        kinds = set(self._config.all_scales())
        return 1 if kinds.issubset({Scale.NUM, Scale.RAW}) else 2

    def _incomplete_rows(self, compiled, values):
        """Non-metric (Scale.NUM) data with NaNs: the rows with missing cells and their masks in device column order; the NaNs
        themselves become column means for the upload (their rows are taken out of the resident matrix, include/plspm_hip.h)."""
        missing = np.isnan(values[:, compiled.col_index])                  # device column order
        for l, lv in enumerate(compiled.lvs):
            block = missing[:, compiled.block_offset[l]:compiled.block_offset[l + 1]]
            if block.any() and compiled.modes[l] == 1:
                raise Exception("Missing nonmetric data is not supported in mode B. LV with missing data: " + lv)          # mode.py:55-56
            if block.all(axis=1).any():
                raise ValueError("All mvs for lv " + lv + " in row " + str(int(np.flatnonzero(block.all(axis=1))[0])) + " are NaN.")   # weights.py:94-95
        rows = np.flatnonzero(missing.any(axis=1))
        with np.errstate(invalid="ignore"):
            means = np.nanmean(values, axis=0)
        filled = np.where(np.isnan(values), np.where(np.isnan(means), 0.0, means), values)
        return np.ascontiguousarray(filled), (rows, ~missing[rows], set(self._config.all_scales()) == {Scale.RAW})

    def run(self, data: pd.DataFrame, path: pd.DataFrame, scaled: bool, want_scores=True, want_cov=False, prepare_bootstrap=False) -> SolverResult:
        """Compile, upload ``data`` (raw or treated) and run one device fit.  Raises the reference's
        ``Exception("Could not converge ...")`` (weights.py:185-186) on non-convergence.  ``prepare_bootstrap``: a bootstrap on this
        handle follows (plspm.py:78-82) -- its per-data-set preparation is enqueued beside the fit."""
        nonmetric = self._nonmetric()
        n = data.shape[0]
        expected = np.sqrt(n / (n - 1))
        if abs(self._correction - expected) > 1e-12 * expected:
            raise ValueError("correction must be sqrt(N / (N - 1)) of the data handed to the solver")
        compiled = compile_model(self._config, path, list(data.columns))
        values = data.values
        incomplete = None
        if nonmetric == 2:
            xaug, aug_offset, mv_off, mv_kind = augment(compiled, self._config, values)
        else:
            values = values if values.dtype == np.float64 else values.astype(np.float64)
            col_index, ind_of = compiled.col_index, None
            if self._config.nan_columns(data).any():          # (Config.filter's scan when `data` is the frame it returned)
                if not nonmetric:
                    values, col_index, ind_of = with_missing_indicators(compiled, values)
                else:
                    values, incomplete = self._incomplete_rows(compiled, values)

        def build(device_id):
            """One handle of this model with the data resident on ``device_id``."""
            if nonmetric == 2:
                handle = _native.NativeModel(aug_offset, compiled.path, compiled.modes, self._scheme.value.code, scaled, self._iterations,
                                             self._tolerance, device_id, nonmetric=True, categorical=(mv_off, mv_kind))
                handle.upload(xaug)
                return self.apply_precision(handle)
            handle = _native.NativeModel(compiled.block_offset, compiled.path, compiled.modes, self._scheme.value.code, scaled,
                                         self._iterations, self._tolerance, device_id, nonmetric=bool(nonmetric), missing=ind_of)
            handle.upload(values, col_index)
            if incomplete is not None:
                handle.set_incomplete_rows(*incomplete)
            return self.apply_precision(handle)

        native = build(self._device_id)
        if prepare_bootstrap:
            native.prepare_bootstrap()
        raw = native.fit(want_scores=want_scores, want_cov=want_cov)
        if raw["status"] == _native.STATUS_NOT_CONVERGED:
            raise ConvergenceError("Could not converge after " + str(raw["iterations"]) + " iterations")
        if raw["status"] == _native.STATUS_NONFINITE:
            raise NumericalConditionError("The solver met a zero-variance manifest variable or non-finite data (status PLSPM_NONFINITE)")
        if raw["status"] == _native.STATUS_SINGULAR:
            raise NumericalConditionError("The solver met a regression it cannot solve (status PLSPM_SINGULAR)")
        return SolverResult(compiled, native, raw, data.index, builder=build)

    def calculate(self, data: pd.DataFrame, path: pd.DataFrame):
        """Reference seam: ``data`` is already treated; returns (final_data, scores, weights).  (Treating is idempotent for
        both branches: centring a centred matrix / standardising a standardised one changes nothing.)"""
        result = self.run(data, path, scaled=False)
        return data, result.scores(), result.weights()
