"""Bootstrap validation (reference plspm/bootstrap.py:76-137) on the GPU.

The reference forks ``processes`` workers that each loop over resample -> estimate -> inner model -> loadings
(bootstrap.py:54-66) and merges five DataFrames through a Queue.  Here all replicates of this process run as
three batched kernels on the data already resident in HBM (resample to int8 multiplicities, the batch's moment matrices as one
exact int8 MFMA product on the digit planes of the pair products, one-wave-per-replicate solver);
``processes`` maps to GPUs: the replicate range is sharded over them and merged by ONE RCCL all-gather inside
libplspm_hip.so (``plspm_group_*``), in one process or with one process per GPU (``plspm.parallel``).
Replicates whose status is not OK are dropped, as the reference's bare ``except`` drops them
(bootstrap.py:65-66).  The summaries of ``_create_summary`` (bootstrap.py:24-32) are computed on the device as well
(``plspm_bootstrap_summary``: one workgroup per result column, radix select for the quantiles); the host
``_create_summary`` below is the same statistic in NumPy, kept for API parity and as the checker of the kernel.
"""
import os
import time

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


class _Pending:
    """Replicates enqueued on the device(s); ``Bootstrap`` turns them into summaries."""

    def __init__(self, result, iterations, seed, source, group, helpers):
        self.result, self.iterations, self.seed = result, iterations, seed
        self.source, self.group, self.helpers = source, group, helpers
        self.t_launch = time.perf_counter()


def launch(result, iterations: int, num_processes: int, seed=None, comm=None, devices=None) -> _Pending:
    """Enqueue the bootstrap replicates of a fitted model (``result``: the SolverResult whose handle holds the data) and return at
    once -- for metric models nothing here waits for the device, so the caller's host work (``Plspm`` builds its result frames)
    runs while the GPU resamples and solves.  Where the replicates run: see :class:`Bootstrap`."""
    native = result.native
    if seed is None:
        seed = int.from_bytes(os.urandom(8), "little")      # the reference is unseeded as well (bootstrap.py:56)
    R = native.row_width
    group = helpers = None
    ctx = parallel.context()
    if comm is not None:
        records = parallel.sharded_bootstrap(lambda count, first: native.bootstrap(count, seed, first), iterations, R, comm)
        native.store(records)                                                       # merged records back to HBM for the summary
        source = native
    elif ctx is not None:
        from plspm import _native
        if native.device_id != ctx.local_rank:
            raise _native.NativeBackendError("the handle lives on device %d but this rank's GPU is %d: pass device_id=LOCAL_RANK"
                                             % (native.device_id, ctx.local_rank))
        group = _native.NativeGroup(ctx.comm, [native])
        if parallel.gather_to_root():
            group.set_option("gather_root", 1)
        group.bootstrap(iterations, seed, 0)
        source = group
    else:
        devices = parallel.devices_for(num_processes, iterations, native.device_id, devices)
        if len(devices) > 1 and result.builder is not None:
            from plspm import _native
            helpers = [result.builder(dev) for dev in devices[1:]]                  # the same model + data on the other GPUs
            group = _native.NativeGroup(parallel.local_comm(devices), [native] + helpers)
            if parallel.gather_to_root():
                group.set_option("gather_root", 1)
            group.bootstrap(iterations, seed, 0)
            source = group
        else:
            native.bootstrap_device(iterations, seed, 0)                            # resample + Gram + solver, rows stay in HBM
            source = native
    return _Pending(result, iterations, seed, source, group, helpers)


class Bootstrap:
    """Bootstrap results; constructed by :class:`plspm.plspm.Plspm` when ``bootstrap=True``.

    Everything heavy stays in HBM: the constructor enqueues the replicates, runs the device summary and copies back the
    ``R x 6`` table; the reference-shaped frames are built on first access and the per-replicate rows cross PCIe only when
    ``replicates()`` / ``status()`` / ``replicate_iterations()`` are asked for.

    Where the replicates run (results are bit-identical for every choice -- Philox stream keyed by (seed, replicate id)):
      * ``comm`` given (any object with rank / world / all_gather): the caller's host-side transport (``parallel.sharded_bootstrap``);
      * a one-process-per-GPU job (``parallel.init_process_group()`` was called): this rank's shard + ONE RCCL all-gather (``PLSPM_GATHER=root``:
        a gather to rank 0 -- the reference's own merge, bootstrap.py:96-111 -- and one small broadcast of the summary table; ``replicates()`` /
        ``status()`` then exist on rank 0 only);
      * otherwise the handle's own GPU -- or, when the caller names GPUs (``Plspm(devices=[...])`` / ``PLSPM_DEVICES``), up to
        ``num_processes`` (the reference's worker count) of them, capped by ``parallel.MIN_REPLICATES_PER_GPU``: one handle per GPU
        + ONE RCCL all-gather; a single GPU needs no collective.
    """

    def __init__(self, config, data: pd.DataFrame, inner_model, outer_model, calculator, iterations: int, num_processes: int,
                 result=None, seed=None, comm=None, pending=None, devices=None):
        if pending is None:
            if result is None:
                from plspm.estimator import Estimator
                result = Estimator(config).run(calculator, data, want_scores=False)
            pending = launch(result, iterations, num_processes, seed, comm, devices)
        result = pending.result
        native, cm = result.native, result.compiled
        self._seed = pending.seed
        self._cm, self._native, self._iterations_requested = cm, native, pending.iterations
        self._inner_model = inner_model
        self._group, self._helpers, self._source = pending.group, pending.helpers, pending.source
        self._ranks = pending.group.nranks if pending.group is not None else 1
        self._via_group = pending.group is not None              # the records came through plspm_group_* (RCCL or the same-device route)
        original = self._original(result, inner_model, outer_model)
        # _create_summary (bootstrap.py:24-32) on the records still in HBM (this is where the host waits for the replicates)
        if self._source is native:
            self._table, self._used = native.summary(pending.iterations, original)
        else:
            self._table, self._used = self._group.summary(original)
            # The gathered records move into this fit's own handle and the group goes away: a communicator serves ONE group at a
            # time, so a second live Plspm(bootstrap=True) of the job (or a rebinding loop) must not find it taken; the lazy
            # rows() / status() accessors then read the handle like after a single-GPU bootstrap.
            self._records_here = not (parallel.gather_to_root() and self._group.first_rank != 0)
            if self._records_here:                   # (gather_root: the records live on rank 0 only; the summary table came back to every rank)
                self._group.adopt()
            self._group.close()
            for helper in (self._helpers or ()):
                helper.close()
            self._group, self._helpers, self._source = None, None, native
        self._rows = None
        self._frames = None
        self.latency_s = time.perf_counter() - pending.t_launch      # enqueue -> summaries on the host (host work in between overlaps)

    @staticmethod
    def _original(result, inner_model, outer_model):
        """Full-sample estimates in the device row layout: weights | r2 | total | direct | loadings (device column order)."""
        cm, raw = result.compiled, result.raw
        if raw is not None:
            return np.concatenate((raw["weights"], raw["r2"], raw["total"], raw["direct"], raw["loadings"])).astype(np.float64)
        om = outer_model.model()                    # two-stage handles carry no raw fit: take the estimates from the frames
        return np.concatenate((om.loc[cm.dev_mvs, "weight"].values, inner_model.r_squared().loc[cm.lvs].values,
                               inner_model.effects().loc[:, "total"].values, inner_model.effects().loc[:, "direct"].values,
                               om.loc[cm.dev_mvs, "loading"].values)).astype(np.float64)

    def _build_frames(self):
        if self._frames is None:
            cm, table, inner_model = self._cm, self._table, self._inner_model
            P, L, ne = cm.P, cm.L, self._native.n_eff
            cols = cm.used_data_cols()              # == data.columns unless HOC constituents' MVs sit unused in stage 2
            eff_index = list(inner_model.effects().index)
            inv = cm.inv_index[cm.inv_index >= 0]

            def frame(block, index):
                return pd.DataFrame(block, index=index, columns=SUMMARY_COLUMNS)
            self._frames = {
                "weights": frame(table[:P][inv], cols),                                  # data-column order (bootstrap.py:83)
                "r_squared": frame(table[P:P + L], cm.lvs).loc[inner_model.endogenous(), :],
                "total_effects": frame(table[P + L:P + L + ne], eff_index),
                "paths": frame(table[P + L + ne:P + L + 2 * ne], eff_index),
                "loading": frame(table[P + L + 2 * ne:][inv], cols),
            }
        return self._frames

    def _fetch(self):
        if not getattr(self, "_records_here", True):
            raise RuntimeError("the replicate records of this bootstrap were gathered to rank 0 only (PLSPM_GATHER=root); the summaries are on every rank")
        if self._rows is None:
            self._rows = self._native.fetch(0, self._iterations_requested)
        return self._rows

    def weights(self) -> pd.DataFrame:
        """Outer weights calculated from bootstrap validation."""
        return self._build_frames()["weights"]

    def r_squared(self) -> pd.DataFrame:
        """R squared for the endogenous latent variables."""
        return self._build_frames()["r_squared"]

    def total_effects(self) -> pd.DataFrame:
        return self._build_frames()["total_effects"]

    def paths(self) -> pd.DataFrame:
        """Direct effects; rows whose bootstrap mean is exactly zero (indirect-only pairs) are hidden (bootstrap.py:133)."""
        paths = self._build_frames()["paths"]
        return paths[paths["mean"] != 0]

    def loading(self) -> pd.DataFrame:
        return self._build_frames()["loading"]

    # --- extensions (not in the reference) -----------------------------------------------------------
    def seed(self):
        return self._seed

    def ranks(self):
        """Number of ranks (GPUs) the replicates were sharded over (1: no collective)."""
        return self._ranks

    def used(self):
        """Number of replicates that entered the summaries (the others failed and were dropped, bootstrap.py:65-66)."""
        return self._used

    def status(self):
        """Per-replicate status codes (0 = used; others were dropped like the reference's failed replicates)."""
        return self._fetch()[1]

    def replicate_iterations(self):
        return self._fetch()[2]

    def replicates(self):
        """The OK replicates' result rows [n_used, R] in device layout (weights | r2 | total | direct | loadings), fetched from
        HBM on first use."""
        rows, status, _ = self._fetch()
        return rows[status == 0]
