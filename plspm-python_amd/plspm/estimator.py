"""Estimation orchestration (reference plspm/estimator.py:24-74) for the MI355X backend.

The reference treats the data on the host and runs the solver twice (estimator.py:39,52 -- the second run is only
different when higher-order constructs exist).  Here the raw filtered data are uploaded once, the treatment is part of
the device moments stage, and the solver runs once.

Missing values (rows whose whole block is missing were already dropped by Config.filter) travel to
``WeightsCalculatorFactory.run``: metric data are mean-imputed on the moments (reference config.py:300, util.py:61-68;
every bootstrap replicate with its own means), Scale.NUM / RAW data use the NaN-aware solver (weights.py:88-98).

Higher-order constructs (two-stage approach, estimator.py:43-52): stage 1 fits the model in which every HOC is replaced
by its constituent LVs (``hoc_path_first_stage``); the constituents' device scores are appended to the data as the
HOC's manifest variables and stage 2 fits the user's path matrix on that -- two device fits, host orchestration only.
As in the reference this works for non-metric (Scale.NUM / RAW) models; its metric branch cannot run a HOC model
(``data.dot(odm)`` misaligns, weights.py:30), so that combination raises here too.  The bootstrap of a HOC model runs both
stages per replicate on the device (``two_stage_bootstrap_handles``, include/plspm_hip.h plspm_model_attach_second_stage).
"""
from typing import Tuple

import numpy as np
import pandas as pd

import plspm.config as c
from plspm.scale import Scale
from plspm.util import MissingDataError
from plspm.weights import SolverResult, WeightsCalculatorFactory


class Estimator:
    """Estimates the model.  Thread-safe the same way the reference is: it works on a cloned calculator."""

    def __init__(self, config):
        self._config = config
        self._first_stage_path = self.hoc_path_first_stage(config)

    def hoc_path_first_stage(self, config) -> pd.DataFrame:
        """Path matrix of stage 1: every exogenous LV of a HOC points at each of its constituent LVs, each constituent
        points at the HOC's endogenous LVs, and the HOC itself is removed (reference estimator.py:60-74)."""
        path = config.path()
        for hoc, members in config.hoc().items():
            structure = c.Structure(path)
            incoming = path.loc[hoc]
            outgoing = path.loc[:, hoc]
            for lv in list(incoming[incoming == 1].index):
                structure.add_path([lv], members)
            for lv in list(outgoing[outgoing == 1].index):
                structure.add_path(members, [lv])
            path = structure.path().drop(hoc).drop(hoc, axis=1)
        return path

    def run(self, calculator: WeightsCalculatorFactory, data: pd.DataFrame, want_scores=True, want_cov=False, prepare_bootstrap=False) -> SolverResult:
        calculator = calculator.clone()
        config = calculator.config()
        if config.missing() and not config.metric() and calculator._nonmetric() == 2:
            holes = [mv for mv in data.columns[data.isnull().any()] if mv in config._mv_scales and config.scale(mv) in (Scale.ORD, Scale.NOM)]
            if holes:
                # the reference's own (most frequent) failure on such data, by name and message -- it has no defined estimate there (util.MissingDataError)
                raise MissingDataError("exog contains inf or nans (missing cells in Scale.ORD / Scale.NOM column(s) %s: the reference raises this from statsmodels, "
                                       "fails to converge, or returns a row-order dependent estimate; impute or drop those rows)" % ", ".join(map(str, holes)))
            # NaNs only in the NUM / RAW columns of a model that also has ORD / NOM columns: the reference estimates this (NaN-aware Mode A on the NUM
            # columns, weights.py:88-98); the device's categorical solver has no incomplete-row form yet
            raise NotImplementedError("missing values in the Scale.NUM / RAW columns of a model with Scale.ORD / NOM columns are not part of the MI355X hot path; see SURVEY.md 8(f)")
        # NaNs otherwise travel to WeightsCalculatorFactory.run: metric -> mean imputation on the moments (util.py:61-68, config.py:300),
        # Scale.NUM -> incomplete rows handled explicitly by the device solver (weights.py:88-98, mode.py:35-41)
        hocs = config.hoc()
        if not hocs:
            self._config = config
            return calculator.run(data, config.path(), scaled=config.scaled(), want_scores=want_scores, want_cov=want_cov, prepare_bootstrap=prepare_bootstrap)
        if config.metric():
            raise NotImplementedError("higher order constructs need Scale.NUM / Scale.RAW data (the reference's metric solver cannot run them either)")
        first = calculator.run(data, self._first_stage_path, scaled=config.scaled(), want_scores=True)
        stage1_scores = first.scores()
        first.native.close()
        extended = data.copy()
        for hoc, members in hocs.items():
            for lv in members:
                extended[lv] = stage1_scores[lv]                              # the constituent's scores become an MV of the HOC
            config.add_lv(hoc, config.mode(hoc), *[c.MV(lv, Scale.NUM) for lv in members])
        self._config = config
        return calculator.run(extended, config.path(), scaled=config.scaled(), want_scores=want_scores, want_cov=want_cov)

    @staticmethod
    def expanded_first_stage_path(config) -> pd.DataFrame:
        """The first-stage path in the LV order the device's two-stage bootstrap needs: the original order with every HOC replaced
        IN PLACE by its constituents (same edges as ``hoc_path_first_stage``; only the order differs, and no result depends on it)."""
        path = config.path()
        hocs = config.hoc() or {}
        members = {lv: list(hocs.get(lv, [lv])) for lv in path.index}
        order = [m for lv in path.index for m in members[lv]]
        expanded = pd.DataFrame(0, index=order, columns=order, dtype=np.int64)
        for to in path.index:
            for frm in path.columns:
                if path.loc[to, frm] == 1:
                    expanded.loc[members[to], members[frm]] = 1
        return expanded

    def two_stage_bootstrap_handles(self, calculator: WeightsCalculatorFactory, data: pd.DataFrame) -> SolverResult:
        """Device handles for bootstrapping a HOC model: stage 1 (expanded path, holds the data) with the stage-2 model attached
        (include/plspm_hip.h, plspm_model_attach_second_stage).  The returned result carries the stage-2 labels."""
        from plspm import _native
        from plspm._compile import compile_model
        calculator = calculator.clone()
        config = calculator.config()
        hocs = config.hoc()
        nonmetric = calculator._nonmetric()
        if config.metric() or not nonmetric or data.isnull().values.any():
            # (NaNs: the reference cannot estimate a HOC model on incomplete data either -- its Config.filter raises KeyError for the
            #  HOC, config.py:279)
            raise NotImplementedError("the batched two-stage bootstrap needs complete non-metric data")
        path1 = self.expanded_first_stage_path(config)
        compiled1 = compile_model(config, path1, list(data.columns))
        values = data.values
        values = values if values.dtype == np.float64 else values.astype(np.float64)
        lv_first, count = [0], 0
        for lv in config.path().index:
            count += len(hocs[lv]) if lv in hocs else 1
            lv_first.append(count)
        for hoc, parts in hocs.items():
            config.add_lv(hoc, config.mode(hoc), *[c.MV(lv, Scale.NUM) for lv in parts])
        compiled2 = compile_model(config, config.path(), list(data.columns) + [lv for parts in hocs.values() for lv in parts])
        scheme_code = calculator.scheme().value.code
        categorical1 = categorical2 = None
        offsets1, offsets2 = compiled1.block_offset, compiled2.block_offset
        if nonmetric == 2:
            # Scale.ORD / NOM: both stages live on aug columns (one indicator column per category; _compile.augment).  A plain LV of
            # stage 2 keeps the aug columns of its stage-1 twin, a HOC's MVs -- the constituents' stage-1 scores -- are one NUM column each.
            from plspm._compile import augment
            xaug, offsets1, mv_off1, mv_kind1 = augment(compiled1, config, values)
            categorical1 = (mv_off1, mv_kind1)
            widths2, kinds2, offsets2 = [], [], [0]
            for l, lv in enumerate(compiled2.lvs):
                if lv in hocs:
                    widths2 += [1] * len(hocs[lv]); kinds2 += [0] * len(hocs[lv])
                else:
                    j = lv_first[l]
                    for p in range(compiled1.block_offset[j], compiled1.block_offset[j + 1]):
                        widths2.append(int(mv_off1[p + 1] - mv_off1[p])); kinds2.append(int(mv_kind1[p]))
                offsets2.append(int(sum(widths2)))
            categorical2 = (np.concatenate(([0], np.cumsum(widths2))).astype(np.int32), np.array(kinds2, dtype=np.int32))
            offsets2 = np.array(offsets2, dtype=np.int32)

        def build(device_id):
            """The handle pair on ``device_id``: the data-holding first stage with the second stage attached."""
            first = _native.NativeModel(offsets1, compiled1.path, compiled1.modes, scheme_code, config.scaled(),
                                        calculator._iterations, calculator._tolerance, device_id, nonmetric=True, categorical=categorical1)
            if categorical1 is not None:
                first.upload(xaug)
            else:
                first.upload(values, compiled1.col_index)
            second = _native.NativeModel(offsets2, compiled2.path, compiled2.modes, scheme_code, config.scaled(),
                                         calculator._iterations, calculator._tolerance, device_id, nonmetric=True, categorical=categorical2)
            first.attach_second_stage(second, lv_first)
            return calculator.apply_precision(first)

        return SolverResult(compiled2, build(calculator._device_id), None, data.index, builder=build)

    def estimate(self, calculator: WeightsCalculatorFactory, data: pd.DataFrame) -> Tuple[pd.DataFrame, pd.DataFrame, pd.DataFrame]:
        """API parity with the reference: (final_data, scores, weights)."""
        result = self.run(calculator, data)
        used = set(result.compiled.dev_mvs)
        columns = [col for col in data.columns if col in used]           # the treated data keep the data's own column order (estimator.py:33)
        final = self._config.treat(data[columns]) if len(columns) == len(used) and self._config.metric() else None
        return final, result.scores(), result.weights()

    def config(self):
        return self._config
