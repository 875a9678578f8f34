"""``Plspm`` -- the user-facing estimator (reference plspm/plspm.py:26-169), MI355X backend.

Same constructor arguments, clamps, assertions and accessors as the reference.  ``processes`` is accepted for
compatibility (bootstrap replicates run batched on the GPU of this process; with one process per GPU they are
sharded by ``plspm.parallel``).  Two keyword extensions: ``seed`` (reproducible bootstrap) and ``device_id``.
"""
import numpy as np
import pandas as pd

import plspm.config as c
import plspm.inner_model as im
import plspm.inner_summary as pis
import plspm.outer_model as om
import plspm.weights as w
from plspm.bootstrap import Bootstrap
from plspm.estimator import Estimator
from plspm.scheme import Scheme
from plspm.unidimensionality import Unidimensionality


class Plspm:
    """Estimates path models with latent variables using the partial least squares algorithm."""

    def __init__(self, data: pd.DataFrame, config: c.Config, scheme: Scheme = Scheme.CENTROID, iterations: int = 100,
                 tolerance: float = 0.000001, bootstrap: bool = False, bootstrap_iterations: int = 100, processes: int = 2,
                 seed: int = None, device_id: int = 0):
        if iterations < 100:
            iterations = 100
        assert tolerance > 0
        assert scheme in Scheme
        if bootstrap_iterations < 10:
            bootstrap_iterations = 100
        assert processes > 0
        assert bootstrap_iterations % processes == 0

        estimator = Estimator(config)
        filtered = config.filter(data)
        n = filtered.shape[0]
        correction = np.sqrt(n / (n - 1))
        calculator = w.WeightsCalculatorFactory(config, iterations, tolerance, correction, scheme, device_id)
        result = estimator.run(calculator, filtered, want_scores=True, want_cov=True)
        config = estimator.config()

        self._result = result
        self._scores = result.scores()
        self._inner_model = im.InnerModel.from_device(config.path(), result)
        self._outer_model = om.OuterModel(result, self._inner_model.r_squared())
        self._inner_summary = pis.InnerSummary(config, self._inner_model.r_squared(), self._inner_model.r_squared_adj(),
                                               self._outer_model.model())
        self._unidimensionality = Unidimensionality(config, result)
        self._bootstrap = None
        if bootstrap:
            if n < 10:
                raise Exception("Bootstrapping could not be performed, at least 10 observations are required.")
            self._bootstrap = Bootstrap(config, filtered, self._inner_model, self._outer_model, calculator, bootstrap_iterations,
                                        processes, result=result, seed=seed)

    def scores(self) -> pd.DataFrame:
        """Latent variable scores: one column per latent variable, index = the data's index."""
        return self._scores

    def outer_model(self) -> pd.DataFrame:
        """weight, loading, communality and redundancy for each manifest variable."""
        return self._outer_model.model()

    def inner_model(self) -> pd.DataFrame:
        """estimate, std error, t and p>|t| for every structural path."""
        return self._inner_model.inner_model()

    def path_coefficients(self) -> pd.DataFrame:
        return self._inner_model.path_coefficients()

    def crossloadings(self) -> pd.DataFrame:
        return self._outer_model.crossloadings()

    def inner_summary(self) -> pd.DataFrame:
        return self._inner_summary.summary()

    def goodness_of_fit(self) -> float:
        return self._inner_summary.goodness_of_fit()

    def effects(self) -> pd.DataFrame:
        return self._inner_model.effects()

    def unidimensionality(self) -> pd.DataFrame:
        return self._unidimensionality.summary()

    def bootstrap(self) -> Bootstrap:
        if self._bootstrap is None:
            raise Exception("To perform bootstrap validation, set the parameter bootstrap to True when calling Plspm")
        return self._bootstrap

    # --- extension: solver diagnostics -------------------------------------------------------------------
    def iterations(self) -> int:
        """Value of the reference's iteration counter when the solver stopped (weights.py:179-184)."""
        return self._result.raw["iterations"]
