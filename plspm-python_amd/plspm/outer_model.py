"""Outer (measurement) model table (reference plspm/outer_model.py:24-42).

Weights, loadings and cross-loadings are device outputs (cor(x_p, score_l) = (S W)_pl / sqrt(S_pp),
csrc/solver_core.h finalize); communality = loading^2 and redundancy = communality * R^2 of the MV's LV are
O(P) host arithmetic.  Index orders follow the reference: ``model()`` alphabetical (pd.concat(sort=True),
outer_model.py:34), ``crossloadings()`` in data-column (add_lv) order."""
import pandas as pd


class OuterModel:
    def __init__(self, result, r_squared: pd.Series):
        cm = result.compiled
        self._crossloadings = pd.DataFrame(result.by_data_column("crossloadings"), index=cm.used_data_cols(), columns=cm.lvs)
        weights = result.weights()["weight"]
        loading = pd.Series(result.raw["loadings"], index=cm.dev_mvs, name="loading")
        communality = (loading ** 2).rename("communality")
        lv_of = pd.Series([lv for lv, a, b in zip(cm.lvs, cm.block_offset[:-1], cm.block_offset[1:]) for _ in range(b - a)], index=cm.dev_mvs)
        redundancy = (communality * lv_of.map(r_squared)).rename("redundancy")
        self._model = pd.concat([weights, loading, communality, redundancy], axis=1).sort_index()

    def model(self) -> pd.DataFrame:
        return self._model

    def crossloadings(self) -> pd.DataFrame:
        return self._crossloadings
