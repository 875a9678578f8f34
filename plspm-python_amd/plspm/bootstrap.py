"""Bootstrap validation (reference plspm/bootstrap.py:76-137) on the GPU.

The reference forks ``processes`` workers that each loop over resample -> estimate -> inner model -> loadings
(bootstrap.py:54-66) and merges five DataFrames through a Queue.  Here all replicates of this process run as
three batched kernels on the data already resident in HBM (resample/compact, fp64-MFMA Gram, LDS solver);
with several processes (one per GPU) the replicate range is sharded and merged by ``plspm.parallel``.
Replicates whose status is not OK are dropped, as the reference's bare ``except`` drops them
(bootstrap.py:65-66).  The summaries of ``_create_summary`` (bootstrap.py:24-32) are computed on the device as well
(``plspm_bootstrap_summary``: one workgroup per result column, LDS bitonic sort for the quantiles); the host
``_create_summary`` below is the same statistic in NumPy, kept for API parity and as the checker of the kernel.
"""
import os

import numpy as np
import pandas as pd

from plspm import parallel

SUMMARY_COLUMNS = ["original", "mean", "std.error", "perc.025", "perc.975", "t stat."]


def _create_summary(samples: pd.DataFrame, original) -> pd.DataFrame:
    """original / mean / std (ddof 1) / 2.5 % and 97.5 % quantiles (linear interpolation) / t = original / std."""
    v = samples.values.astype(np.float64)
    summary = pd.DataFrame(0.0, index=samples.columns, columns=SUMMARY_COLUMNS)
    summary["original"] = original
    if v.shape[0]:
        sd = v.std(axis=0, ddof=1) if v.shape[0] > 1 else np.full(v.shape[1], np.nan)
        summary["mean"] = v.mean(axis=0)
        summary["std.error"] = sd
        summary["perc.025"] = np.quantile(v, 0.025, axis=0)
        summary["perc.975"] = np.quantile(v, 0.975, axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            summary["t stat."] = summary["original"].values / sd
    return summary


class Bootstrap:
    """Bootstrap results; constructed by :class:`plspm.plspm.Plspm` when ``bootstrap=True``."""

    def __init__(self, config, data: pd.DataFrame, inner_model, outer_model, calculator, iterations: int, num_processes: int,
                 result=None, seed=None, group=None):
        if result is None:
            from plspm.estimator import Estimator
            result = Estimator(config).run(calculator, data, want_scores=False)
        native, cm = result.native, result.compiled
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")      # the reference is unseeded as well (bootstrap.py:56)
        self._seed = seed
        P, L, ne, R = cm.P, cm.L, native.n_eff, native.row_width
        cols = cm.used_data_cols()                  # == data.columns unless HOC constituents' MVs sit unused in stage 2
        eff_index = list(inner_model.effects().index)
        om = outer_model.model()
        # full-sample estimates in the device row layout: weights | r2 | total | direct | loadings (device column order)
        original = np.concatenate((om.loc[cm.dev_mvs, "weight"].values, inner_model.r_squared().loc[cm.lvs].values,
                                   inner_model.effects().loc[:, "total"].values, inner_model.effects().loc[:, "direct"].values,
                                   om.loc[cm.dev_mvs, "loading"].values)).astype(np.float64)
        dist, _, world = parallel._world(group)
        if dist is None:
            rows, status, iters = native.bootstrap(iterations, seed, 0)                 # resample + Gram + solver on this GPU
            table, used = native.summary(iterations, original)                          # _create_summary, on the rows still in HBM
        else:
            def run_shard(count, first):
                d_rows, _, _ = native.bootstrap_device(count, seed, first)
                return d_rows, native.sync
            records = parallel.sharded_bootstrap(run_shard, iterations, R, group=group, on_device=True, to_host=False)
            import torch
            torch.cuda.synchronize()
            table, used = native.summary(iterations, original, d_rows=records.data_ptr(), stride=R + 2)
            rows, status, iters = parallel.split_records(records.cpu().numpy(), R)
        self._status, self._iterations, self._used = status, iters, used
        self._replicates = rows[status == 0]
        inv = cm.inv_index[cm.inv_index >= 0]

        def frame(block, index):
            return pd.DataFrame(block, index=index, columns=SUMMARY_COLUMNS)
        self._weights = frame(table[:P][inv], cols)                                      # data-column order (bootstrap.py:83)
        self._r_squared = frame(table[P:P + L], cm.lvs).loc[inner_model.endogenous(), :]
        self._total_effects = frame(table[P + L:P + L + ne], eff_index)
        self._paths = frame(table[P + L + ne:P + L + 2 * ne], eff_index)
        self._loading = frame(table[P + L + 2 * ne:][inv], cols)

    def weights(self) -> pd.DataFrame:
        """Outer weights calculated from bootstrap validation."""
        return self._weights

    def r_squared(self) -> pd.DataFrame:
        """R squared for the endogenous latent variables."""
        return self._r_squared

    def total_effects(self) -> pd.DataFrame:
        return self._total_effects

    def paths(self) -> pd.DataFrame:
        """Direct effects; rows whose bootstrap mean is exactly zero (indirect-only pairs) are hidden (bootstrap.py:133)."""
        return self._paths[self._paths["mean"] != 0]

    def loading(self) -> pd.DataFrame:
        return self._loading

    # --- extensions (not in the reference) -----------------------------------------------------------
    def seed(self):
        return self._seed

    def status(self):
        """Per-replicate status codes (0 = used; others were dropped like the reference's failed replicates)."""
        return self._status

    def replicate_iterations(self):
        return self._iterations
