"""Model compiler: Config + path matrix -> the dense descriptors of the C-ABI, built ONCE per fit.

The reference re-queries ``Config.odm / mvs / mode`` by label inside every iteration (weights.py:48-49,
config.py:140-160); here the labels are resolved up front into
  * ``lvs``          LV names in path-matrix order,
  * ``block_offset`` [L+1] ranges of the LV blocks in *device column order* (MVs grouped by LV, path order),
  * ``dev_mvs``      MV names in device column order (== row order of the reference's ``weights`` frame),
  * ``col_index``    for every device column, its position in the filtered data frame (add_lv order),
  * ``path`` (uint8 L x L), ``modes`` (int32 L).
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import pandas as pd


@dataclass
class CompiledModel:
    lvs: List[str]
    dev_mvs: List[str]
    data_cols: List[str]
    block_offset: np.ndarray
    col_index: np.ndarray
    inv_index: np.ndarray          # data column -> device column
    path: np.ndarray
    modes: np.ndarray

    @property
    def P(self):
        return len(self.dev_mvs)

    @property
    def L(self):
        return len(self.lvs)

    def used_data_cols(self):
        """Data columns that belong to an LV of the compiled path (others, e.g. the MVs of HOC constituents in stage 2, are ignored)."""
        return [col for col, dev in zip(self.data_cols, self.inv_index) if dev >= 0]

    def endogenous(self):
        return [lv for lv, row in zip(self.lvs, self.path) if row.sum() > 0]


def compile_model(config, path: pd.DataFrame, data_cols) -> CompiledModel:
    lvs = list(path)
    data_cols = list(data_cols)
    position = {name: i for i, name in enumerate(data_cols)}
    dev_mvs, offsets = [], [0]
    for lv in lvs:
        block = list(config.mvs(lv))
        if not block:
            raise ValueError("Latent variable " + lv + " has no manifest variables")
        dev_mvs.extend(block)
        offsets.append(len(dev_mvs))
    missing = [mv for mv in dev_mvs if mv not in position]
    if missing:
        raise ValueError("The following manifest variables you configured are not present in the data set: " + ", ".join(missing))
    col_index = np.array([position[mv] for mv in dev_mvs], dtype=np.int32)
    inv = np.full(len(data_cols), -1, dtype=np.int64)
    inv[col_index] = np.arange(len(dev_mvs))
    pmat = np.ascontiguousarray(path.values.astype(np.uint8))
    if np.any(np.triu(pmat) != 0):
        raise ValueError("Path argument must be a strictly lower triangular matrix for the MI355X backend")
    modes = np.array([config.mode(lv).value.code for lv in lvs], dtype=np.int32)
    return CompiledModel(lvs, dev_mvs, data_cols, np.array(offsets, dtype=np.int32), col_index, inv, pmat, modes)


KIND_NUM, KIND_ORD, KIND_NOM = 0, 1, 2


def augment(compiled: CompiledModel, config, values: np.ndarray):
    """Categorical (Scale.ORD / NOM) models: the device works on *augmented* columns (include/plspm_hip.h,
    plspm_model_set_categorical).  A NUM / RAW MV keeps its data column; an ORD / NOM MV becomes one 0/1 indicator column per
    distinct value in ascending order -- the dummy matrix of its rank codes, which the reference rebuilds inside every
    quantification (scale.py:44,54 with util.py rank / dummy).  Returns (Xaug [N, Q] float64, aug block offsets [L+1],
    mv_off [Pm+1], mv_kind [Pm])."""
    from plspm.scale import Scale
    kinds = {Scale.NUM: KIND_NUM, Scale.RAW: KIND_NUM, Scale.ORD: KIND_ORD, Scale.NOM: KIND_NOM}
    n = values.shape[0]
    cols, mv_off, mv_kind, boff = [], [0], [], [0]
    for l, lv in enumerate(compiled.lvs):
        for p in range(compiled.block_offset[l], compiled.block_offset[l + 1]):
            kind = kinds[config.scale(compiled.dev_mvs[p])]
            column = values[:, compiled.col_index[p]]
            if kind == KIND_NUM:
                cols.append(column.astype(np.float64)[:, None])
            else:
                _, codes = np.unique(column, return_inverse=True)
                indicator = np.zeros((n, int(codes.max()) + 1))
                indicator[np.arange(n), codes] = 1.0
                cols.append(indicator)
            mv_off.append(mv_off[-1] + cols[-1].shape[1])
            mv_kind.append(kind)
        boff.append(mv_off[-1])
    return (np.ascontiguousarray(np.concatenate(cols, axis=1)), np.array(boff, dtype=np.int32), np.array(mv_off, dtype=np.int32),
            np.array(mv_kind, dtype=np.int32))


def with_missing_indicators(compiled: CompiledModel, values: np.ndarray):
    """Metric data with NaNs (include/plspm_hip.h, plspm_model_set_missing): every NaN becomes its column's mean over the present
    cells (util.impute, reference util.py:61-68) and one 0/1 "missing" column per incomplete model column is appended, so that
    the device can re-impute every bootstrap replicate with the replicate's own means.  Returns (matrix, col_index incl. the
    indicator columns, ind_of [P])."""
    missing = np.isnan(values)
    means = np.nanmean(np.where(missing.all(axis=0), 0.0, values), axis=0)
    filled = np.where(missing, means, values)
    incomplete = [p for p in range(compiled.P) if missing[:, compiled.col_index[p]].any()]
    ind_of = np.full(compiled.P, -1, dtype=np.int32)
    ind_of[incomplete] = compiled.P + np.arange(len(incomplete))
    indicators = missing[:, compiled.col_index[incomplete]].astype(np.float64)
    col_index = np.concatenate([compiled.col_index, values.shape[1] + np.arange(len(incomplete))]).astype(np.int32)
    return np.ascontiguousarray(np.concatenate([filled, indicators], axis=1)), col_index, ind_of
