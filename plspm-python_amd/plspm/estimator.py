"""Estimation orchestration (reference plspm/estimator.py:24-74) for the MI355X backend.

The reference treats the data on the host and runs the solver twice (estimator.py:39,52 -- the second run
is only different when higher-order constructs exist).  Here the raw filtered data are uploaded once, the
treatment is part of the device moments stage, and the solver runs once.  Higher-order constructs
(two-stage approach) are not built yet: SURVEY.md 8(f) rank 2.
"""
from typing import Tuple

import pandas as pd

from plspm.weights import SolverResult, WeightsCalculatorFactory


class Estimator:
    """Estimates the model.  Thread-safe the same way the reference is: it works on a cloned calculator."""

    def __init__(self, config):
        if config.hoc():
            raise NotImplementedError("higher order constructs are not part of the MI355X hot path yet; see SURVEY.md 8(f)")
        self._config = config

    def run(self, calculator: WeightsCalculatorFactory, data: pd.DataFrame, want_scores=True, want_cov=False) -> SolverResult:
        calculator = calculator.clone()
        config = calculator.config()
        if config.missing():
            raise NotImplementedError("missing values (mean imputation) are not part of the MI355X hot path yet; see SURVEY.md 8(f)")
        self._config = config
        return calculator.run(data, config.path(), scaled=config.scaled(), want_scores=want_scores, want_cov=want_cov)

    def estimate(self, calculator: WeightsCalculatorFactory, data: pd.DataFrame) -> Tuple[pd.DataFrame, pd.DataFrame, pd.DataFrame]:
        """API parity with the reference: (final_data, scores, weights).  ``final_data`` (the treated frame) is
        rebuilt on the host for callers that want it; nothing in the estimator consumes it."""
        result = self.run(calculator, data)
        return self._config.treat(data), result.scores(), result.weights()

    def config(self):
        return self._config
