"""``Plspm`` -- the user-facing estimator (reference plspm/plspm.py:26-169), MI355X backend.

Same constructor arguments, clamps, assertions and accessors as the reference.  ``processes`` -- the reference's number of forked
bootstrap workers (plspm.py:35-37, bootstrap.py:89-94) -- caps the number of GPUs of this process the replicates are sharded
over once the caller NAMES GPUs (``devices=[...]`` or the ``PLSPM_DEVICES`` allow-list; also capped by
``parallel.MIN_REPLICATES_PER_GPU`` replicates per GPU; ONE RCCL all-gather merges the shards; the rows do not depend on it).
Without named GPUs everything runs on ``device_id``.  Keyword extensions: ``seed`` (reproducible bootstrap), ``device_id``, ``devices``,
``precision`` ("auto": the bootstrap's moment sums on the fewest exact-integer digit planes that stay inside the error bound of the
reference's own fp64 accumulation on the data at hand; "strict": always seven planes = correctly rounded sums, ~20 % slower).
"""
import time

import numpy as np
import pandas as pd

import plspm.config as c
import plspm.inner_model as im
import plspm.inner_summary as pis
import plspm.outer_model as om
import plspm.weights as w
from plspm.bootstrap import Bootstrap
from plspm.bootstrap import launch as launch_bootstrap
from plspm.estimator import Estimator
from plspm.scale import Scale
from plspm.scheme import Scheme
from plspm.unidimensionality import Unidimensionality


def _normalise_arguments(scheme, iterations, tolerance, bootstrap_iterations, processes):
    """The argument clamps and assertions of reference plspm/plspm.py:54-61, in one place."""
    assert tolerance > 0
    assert scheme in Scheme
    assert processes > 0
    iterations = max(iterations, 100)                   # "default and minimum 100"
    if bootstrap_iterations < 10:                       # plspm.py:58-59: anything below 10 silently becomes 100
        bootstrap_iterations = 100
    assert bootstrap_iterations % processes == 0
    return iterations, bootstrap_iterations


class _Lazy:
    """Builds its object on first use and then stands in for it (attribute access is forwarded)."""

    def __init__(self, build):
        self._build, self._obj = build, None

    def get(self):
        if self._obj is None:
            self._obj = self._build()
            self._build = None
        return self._obj

    def __getattr__(self, name):
        return getattr(self.get(), name)


class Plspm:
    """PLS path model estimator.

    ``Plspm(data, config, scheme=Scheme.CENTROID, iterations=100, tolerance=1e-6, bootstrap=False,
    bootstrap_iterations=100, processes=2)`` -- argument meaning as in the reference; additionally ``seed`` makes the
    bootstrap reproducible (the reference is unseeded) and ``device_id`` picks the GPU.  Raises ``Exception`` when the
    solver does not converge, ``NotImplementedError`` for model features outside the MI355X hot path, and
    ``plspm._native.NativeBackendError`` when no GPU / library is available.
    """

    def __init__(self, data: pd.DataFrame, config: c.Config, scheme: Scheme = Scheme.CENTROID, iterations: int = 100,
                 tolerance: float = 0.000001, bootstrap: bool = False, bootstrap_iterations: int = 100, processes: int = 2,
                 seed: int = None, device_id: int = 0, devices=None, precision: str = "auto"):
        iterations, bootstrap_iterations = _normalise_arguments(scheme, iterations, tolerance, bootstrap_iterations, processes)
        t_start = time.perf_counter()
        estimator = Estimator(config)
        observations = config.filter(data)
        n_obs = observations.shape[0]
        calculator = w.WeightsCalculatorFactory(config, iterations, tolerance, np.sqrt(n_obs / (n_obs - 1)), scheme, device_id, precision)

        # one device fit: Gram -> LDS solver -> scores; everything below only re-labels / post-processes its outputs
        fit = estimator.run(calculator, observations, want_scores=True, want_cov=True, prepare_bootstrap=bool(bootstrap) and n_obs >= 10)
        model_spec = estimator.config()
        pending = None
        if bootstrap:
            if n_obs < 10:
                raise Exception("Bootstrapping could not be performed, at least 10 observations are required.")
            # the handle of the fit already holds the data in HBM: the replicates are enqueued on it NOW (HOC models: on a two-stage
            # handle pair), so that the GPU resamples and solves while the host does whatever is left to do
            # (HOC models: on a two-stage handle pair -- Scale.NUM / RAW and, since round 3, Scale.ORD / NOM data alike: both stages of
            #  every replicate run on the device, estimator.two_stage_bootstrap_handles)
            boot_on = estimator.two_stage_bootstrap_handles(calculator, observations) if config.hoc() else fit
            pending = launch_bootstrap(boot_on, bootstrap_iterations, processes, seed, devices=devices)
        self._result = fit
        # The report frames only re-label / post-process the device outputs already on the host (fit.raw); they are built on first
        # access (the reference builds them eagerly, plspm.py:69-77 -- same objects, same values, ~4 ms of pandas work per call that a
        # caller who wants two of the nine accessors does not pay).
        path = model_spec.path()
        incomplete = list(observations.columns[config.nan_columns(observations)])                 # (the scan Config.filter made)
        # (the builders close over LOCAL names, never over `self`: a Plspm object then holds no reference cycle, and dropping it releases its device
        #  handle and frames at once instead of whenever the cycle collector next runs -- inside somebody else's timed Plspm() call, as a rule)
        scores = _Lazy(fit.scores)
        inner_model = _Lazy(lambda: im.InnerModel.from_device(path, fit))
        outer_model = _Lazy(lambda: om.OuterModel(fit, inner_model.r_squared()))
        inner_summary = _Lazy(lambda: pis.InnerSummary(model_spec, inner_model.r_squared(), inner_model.r_squared_adj(), outer_model.model()))
        # (the reference's block diagnostics standardise the filtered DATA, unidimensionality.py:40: for Scale.ORD / NOM columns those are the category codes, not the quantified
        #  MVs whose covariance the device returns -- such models hand the observations in)
        categorical = (not config.metric()) and any(config.scale(mv) in (Scale.ORD, Scale.NOM) for mv in observations.columns)
        raw = observations if categorical else None
        unidimensionality = _Lazy(lambda: Unidimensionality(model_spec, fit, incomplete, raw))
        self._scores, self._inner_model, self._outer_model = scores, inner_model, outer_model
        self._inner_summary, self._unidimensionality = inner_summary, unidimensionality
        self._bootstrap = None
        t_fit = time.perf_counter()
        if bootstrap:
            self._bootstrap = Bootstrap(model_spec, observations, self._inner_model, self._outer_model, calculator,
                                        bootstrap_iterations, processes, pending=pending)
        # fit_s: filter + upload + device fit; bootstrap_s: what the bootstrap adds (the wait for the replicates + the device summaries);
        # bootstrap_latency_s: enqueue of the replicates -> summaries on the host
        self._timings = {"fit_s": t_fit - t_start, "bootstrap_s": time.perf_counter() - t_fit,
                         "bootstrap_latency_s": self._bootstrap.latency_s if self._bootstrap is not None else 0.0}

    # ---- accessors (names and return shapes of reference plspm/plspm.py:84-169) -------------------------------
    def scores(self) -> pd.DataFrame:
        """LV scores, N x L: index = the input data's index, columns = LVs in path-matrix order."""
        return self._scores.get()

    def outer_model(self) -> pd.DataFrame:
        """Per MV (alphabetical index): weight, loading, communality, redundancy."""
        return self._outer_model.model()

    def inner_model(self) -> pd.DataFrame:
        """Per structural path "A -> B": from, to, estimate, std error, t, p>|t|."""
        return self._inner_model.inner_model()

    def path_coefficients(self) -> pd.DataFrame:
        """L x L matrix shaped like the path matrix given to Config, holding the estimated coefficients."""
        return self._inner_model.path_coefficients()

    def crossloadings(self) -> pd.DataFrame:
        """Correlation of every MV (rows, data-column order) with every LV score (columns)."""
        return self._outer_model.crossloadings()

    def inner_summary(self) -> pd.DataFrame:
        """Per LV: type, r_squared, r_squared_adj, block_communality, mean_redundancy, ave."""
        return self._inner_summary.summary()

    def goodness_of_fit(self) -> float:
        return self._inner_summary.goodness_of_fit()

    def effects(self) -> pd.DataFrame:
        """Per connected LV pair: from, to, direct, indirect, total."""
        return self._inner_model.effects()

    def unidimensionality(self) -> pd.DataFrame:
        """Per block: mode, mvs, cronbach_alpha, dillon_goldstein_rho, eig_1st, eig_2nd."""
        return self._unidimensionality.summary()

    @staticmethod
    def _replicate_runner(config, calculator, observations):
        """run_one(idx): ONE complete device estimate (two stages for a HOC model) of the resampled observations in the device row
        layout (weights | r2 | total | direct | loadings) -- what a reference worker computes per replicate (bootstrap.py:56-64).  The
        bootstrap itself is batched; this is the per-replicate cross-check the tests use."""
        def run_one(idx):
            result = Estimator(config).run(calculator, observations.iloc[idx, :], want_scores=False)
            raw = result.raw
            row = np.concatenate((raw["weights"], raw["r2"], raw["total"], raw["direct"], raw["loadings"])).astype(np.float64)
            result.native.close()
            return row, int(raw["iterations"])
        return run_one

    def bootstrap(self) -> Bootstrap:
        """The :class:`plspm.bootstrap.Bootstrap` results; raises when bootstrap=True was not requested."""
        if self._bootstrap is None:
            raise Exception("To perform bootstrap validation, set the parameter bootstrap to True when calling Plspm")
        return self._bootstrap

    # --- extension: solver diagnostics -------------------------------------------------------------------
    def timings(self) -> dict:
        """Wall seconds of the two phases of the constructor: ``fit_s`` (filter, upload, device fit, result frames) and
        ``bootstrap_s`` (replicates + device summaries; 0 without bootstrap)."""
        return dict(self._timings)

    def iterations(self) -> int:
        """Value of the reference's iteration counter when the solver stopped (weights.py:179-184)."""
        return self._result.raw["iterations"]
