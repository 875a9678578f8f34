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
