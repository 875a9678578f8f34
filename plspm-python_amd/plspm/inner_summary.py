"""Inner-model summary and goodness-of-fit (reference plspm/inner_summary.py:26-65): O(P) host arithmetic on
the outer-model table -- not on the GPU hot path."""
import math

import numpy as np
import pandas as pd

from plspm.mode import Mode


class InnerSummary:
    def __init__(self, config, r_squared: pd.Series, r_squared_adj: pd.Series, outer_model: pd.DataFrame):
        path = config.path()
        lvs = list(path)
        endogenous = path.sum(axis=1).astype(bool)
        kind = endogenous.map({False: "Exogenous", True: "Endogenous"}).rename("type")
        block_communality = pd.Series(np.nan, index=lvs, name="block_communality")
        mean_redundancy = pd.Series(np.nan, index=lvs, name="mean_redundancy")
        ave = pd.Series(np.nan, index=lvs, name="ave")
        weighted, sizes = [], []
        for lv in lvs:
            mvs = config.mvs(lv)
            comm = outer_model.loc[mvs, "communality"]
            block_communality[lv] = comm.mean()
            mean_redundancy[lv] = outer_model.loc[mvs, "redundancy"].mean()
            if config.mode(lv) == Mode.A:
                ave[lv] = comm.sum() / (comm.sum() + (1 - comm).sum())
            if len(mvs) > 1:
                sizes.append(len(mvs))
                weighted.append(block_communality[lv] * len(mvs))
        self._summary = pd.concat([kind, r_squared, r_squared_adj, block_communality, mean_redundancy, ave], axis=1).sort_index()
        if sum(sizes) > 0:
            r2_endo = (r_squared * endogenous)
            self._gof = float(np.sqrt(sum(weighted) / sum(sizes) * r2_endo[r2_endo != 0].mean()))
        else:
            self._gof = float("nan")        # only single-item constructs

    def summary(self) -> pd.DataFrame:
        return self._summary

    def goodness_of_fit(self) -> float:
        if math.isnan(self._gof):
            raise ValueError("Cannot calculate goodness-of-fit if all constructs are single-item.")
        return self._gof
