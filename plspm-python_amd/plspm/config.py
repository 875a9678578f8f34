"""Model specification: ``Structure`` (inner model builder), ``MV`` and ``Config``.

API-compatible with reference plspm/config.py (Structure 24-58, MV 60-81, Config 84-319): same constructor
arguments, same method names, same exceptions for the same mistakes.  What differs is what happens after
specification: ``plspm._compile.compile_model`` lowers a Config + path matrix ONCE into the dense descriptors
of the C-ABI (block offsets, path bitmap, mode ids) instead of being queried by label inside the iteration.
"""
import itertools
import weakref

import numpy as np
import pandas as pd

from plspm.mode import Mode
from plspm.scale import Scale
from plspm.util import TopoSort


class Structure:
    """Collects directed relationships between constructs and renders them as a path matrix."""

    def __init__(self, path: pd.DataFrame = None):
        self._graph = TopoSort()
        if path is not None:
            rows, cols = np.nonzero(path.values == 1)
            for r, c in zip(rows, cols):
                self.add_path([path.columns[c]], [path.index[r]])

    def add_path(self, source: list, target: list):
        """Declare that every construct in ``source`` affects every construct in ``target``; one of the two
        lists must have exactly one entry."""
        if len(source) != 1 and len(target) != 1:
            raise ValueError("Either source or target must be a list containing a single entry")
        if len(source) == 0 or len(target) == 0:
            raise ValueError("Both source and target must contain at least one entry")
        for src, dst in itertools.product(source, target):
            self._graph.append(src, dst)

    def path(self) -> pd.DataFrame:
        """Lower-triangular 0/1 matrix, constructs in topological order; cell [row, col] = 1 iff col -> row."""
        order = self._graph.order()
        matrix = pd.DataFrame(np.zeros((len(order), len(order)), dtype=int), index=order, columns=order)
        for src, dst in self._graph.elements():
            matrix.at[dst, src] = 1
        return matrix


class MV:
    """A manifest variable: a column name of the data set plus (for non-metric data) its measurement scale."""

    def __init__(self, name: str, scale: Scale = None):
        self._name = name
        self._scale = scale

    def name(self):
        return self._name

    def scale(self):
        return self._scale


class Config:
    """The model to estimate: path matrix + the manifest variables and mode of every latent variable.

    Args:
        path: square lower-triangular 0/1 DataFrame with identical index and columns (the LV names).
        scaled: standardise the manifest variables (metric data only).
        default_scale: measurement scale for MVs that do not set one; ``None`` means metric data.
    """

    def __init__(self, path: pd.DataFrame, scaled: bool = True, default_scale: Scale = None):
        if not isinstance(path, pd.DataFrame):
            raise TypeError("Path argument must be a Pandas DataFrame")
        if path.shape[0] != path.shape[1]:
            raise ValueError("Path argument must be a square matrix")
        values = np.asarray(path.values)
        if not np.array_equal(values, np.tril(values)):
            raise ValueError("Path argument must be a lower triangular matrix")
        if not np.isin(values, (0, 1)).all():
            raise ValueError("Path matrix element values may only be in [0, 1]")
        if list(path.columns) != list(path.index):
            raise ValueError("Path matrix must have matching row and column index names")
        self._path = path
        self._scaled = scaled
        self._default_scale = default_scale
        self._modes = {}
        self._mvs = {}           # LV -> [MV names], insertion order == add_lv call order
        self._hoc = {}
        self._mv_scales = {}     # MV -> Scale or None, insertion order == data column order after filter()
        self._dummies = {}
        self._metric = True
        self._missing = False

    # ------------------------------------------------------------------ specification
    def add_lv(self, lv_name: str, mode: Mode, *mvs: MV):
        """Add a latent variable with its manifest variables."""
        assert mode in Mode
        hoc_members = [lv for members in self._hoc.values() for lv in members]
        if lv_name not in self._path and lv_name not in hoc_members:
            raise ValueError("Latent variable " + lv_name + " is not listed in the outer model paths or higher order constructs.")
        self._modes[lv_name] = mode
        self._mvs[lv_name] = []
        for mv in mvs:
            if mv.name() in self._mv_scales:
                raise ValueError("You can only specify a column once. You can specify a higher order construct with `add_higher_order(...)`")
            if mv.name() in list(self._path):
                raise ValueError("You cannot specify MVs with the same name as LVs.")
            self._mvs[lv_name].append(mv.name())
            scale = mv.scale() if mv.scale() is not None else self._default_scale
            self._mv_scales[mv.name()] = scale
            if scale is not None:
                self._metric = False

    def add_lv_with_columns_named(self, lv_name: str, mode: Mode, data: pd.DataFrame, col_name_starts_with: str,
                                  default_scale: Scale = None):
        """Add a latent variable whose manifest variables are all data columns sharing a name prefix."""
        chosen = [MV(col, default_scale) for col in list(data) if col.startswith(col_name_starts_with)]
        if not chosen:
            raise ValueError("No columns were found in the data starting with " + col_name_starts_with)
        self.add_lv(lv_name, mode, *chosen)

    def add_higher_order(self, hoc_name: str, mode: Mode, lvs: list):
        """Declare a higher order construct made of first order constructs."""
        assert mode in Mode
        if hoc_name not in self._path:
            raise ValueError("Path matrix does not contain reference to higher order construct " + hoc_name)
        self._modes[hoc_name] = mode
        self._hoc[hoc_name] = lvs

    def remove_lv(self, lv_name: str):
        self._mvs.pop(lv_name)
        self._modes.pop(lv_name)

    def clone(self):
        twin = Config(self._path, self._scaled, self._default_scale)
        twin._modes = dict(self._modes)
        twin._mvs = dict(self._mvs)
        twin._hoc = dict(self._hoc)
        twin._dummies = dict(self._dummies)
        twin._mv_scales = dict(self._mv_scales)
        twin._metric = self._metric
        twin._missing = self._missing
        twin._nan_seen = getattr(self, "_nan_seen", None)   # (keyed by a weak reference to the frame itself: valid for the twin as long as it is asked about that very object)
        return twin

    # ------------------------------------------------------------------ queries
    def path(self):
        return self._path

    def odm(self, path: pd.DataFrame) -> pd.DataFrame:
        """Outer design matrix: rows = MVs in path-LV order, columns = LVs, 1 where the MV belongs to the LV."""
        lvs = list(path)
        rows = [mv for lv in lvs for mv in self._mvs[lv]]
        matrix = pd.DataFrame(0.0, index=rows, columns=lvs)
        for lv in lvs:
            matrix.loc[self._mvs[lv], lv] = 1.0
        return matrix

    def mv_index(self, lv, mv):
        return self._mvs[lv].index(mv)

    def mvs(self, lv):
        return self._mvs[lv]

    def hoc(self):
        return self._hoc

    def mode(self, lv: str):
        return self._modes[lv]

    def metric(self):
        return self._metric

    def missing(self):
        return self._missing

    def scaled(self):
        return self._scaled

    def scale(self, mv: str):
        return self._mv_scales[mv]

    def all_scales(self):
        """Scales of every configured MV (including the MVs of HOC constituents that are not in the path matrix)."""
        return list(self._mv_scales.values())

    def dummies(self, mv: str):
        return self._dummies[mv]

    def promote_scales(self):
        """The scale bookkeeping the reference performs inside treat() (config.py:309-313): RAW-only models are unscaled,
        a RAW + NUM mix becomes all-NUM.  Raises TypeError when some MV has no scale (config.py:307-308)."""
        if self._metric:
            return
        if None in self._mv_scales.values():
            raise TypeError("If you supply a scale for any MV, you must either supply a scale for all of them or specify a default scale.")
        kinds = set(self._mv_scales.values())
        if kinds == {Scale.RAW}:
            self._scaled = False
        if kinds == {Scale.RAW, Scale.NUM}:
            self._scaled = True
            self._mv_scales = dict.fromkeys(self._mv_scales, Scale.NUM)

    # ------------------------------------------------------------------ data preparation
    def filter(self, data: pd.DataFrame) -> pd.DataFrame:
        """Keep only the configured MV columns (in add_lv order) and validate them (reference config.py:247-285)."""
        hoc_members = [lv for members in self._hoc.values() for lv in members]
        expected_lvs = [lv for lv in list(self.path()) + hoc_members if lv not in self._hoc]
        if set(self._mvs) != set(expected_lvs):
            raise ValueError(
                "The Path matrix supplied does not specify the same latent variables as you added when configuring manifest variables." +
                " Path: " + ", ".join(expected_lvs) + " LVs: " + ", ".join(set(self._mvs)))
        wanted = list(self._mv_scales)
        absent = set(wanted).difference(set(data))
        if absent:
            raise ValueError("The following manifest variables you configured are not present in the data set: " + ", ".join(absent))
        if list(data.columns) != wanted:                     # (the frame itself when it already holds exactly the configured columns in order: nothing
            data = data[wanted]                              #  downstream writes to it, and the copy of a 10k x 60 frame is a third of a millisecond)
        if not all(np.issubdtype(dtype, np.number) for dtype in data.dtypes):
            raise ValueError("Data must only contain numeric values. Please convert any categorical data into numerical values.")
        # ONE pass over the values tells whether -- and in which columns -- cells are missing; plspm.py / weights.py ask nan_columns(data)
        # instead of scanning the same matrix again (three scans were 0.6 ms of a 2.3 ms Plspm() call at 10k x 60)
        nan_cols = self._scan_nan(data)
        self._missing = bool(nan_cols.any())
        if self._missing:
            drop = np.zeros(len(data.index), dtype=bool)
            for lv in list(self.path()):
                block = data[self.mvs(lv)].values.astype(np.float64)
                drop |= np.isnan(block).all(axis=1)
            if drop.any():
                data = data.loc[~drop]
                nan_cols = self._scan_nan(data)
        self._nan_seen = (weakref.ref(data), data.shape, nan_cols)      # (a weak reference: an id() could be recycled by another frame of the same shape)
        return data

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_nan_seen"] = None                            # never travels to another process: the frame it describes does not
        return state

    @staticmethod
    def _scan_nan(data: pd.DataFrame) -> np.ndarray:
        values = data.to_numpy()
        if values.dtype.kind != "f":
            values = values.astype(np.float64) if values.dtype == object else None
        return np.isnan(values).any(axis=0) if values is not None else np.zeros(data.shape[1], dtype=bool)

    def nan_columns(self, data: pd.DataFrame) -> np.ndarray:
        """Per column of ``data``: does it hold a NaN?  Served from the scan ``filter`` made when ``data`` is the frame it returned."""
        seen = getattr(self, "_nan_seen", None)
        if seen is not None and seen[0]() is data and seen[1] == data.shape:
            return seen[2]
        return self._scan_nan(data)

    def treat(self, data: pd.DataFrame) -> pd.DataFrame:
        """Host-side restatement of the data pre-treatment (reference config.py:299-318).  Metric data: centre, and
        when ``scaled`` divide by ONE scalar, std(ddof=1) of all values times sqrt((N-1)/N).  Non-metric data:
        standardise by the population std, rank ORD/NOM columns and build their dummy matrices.  The estimator does
        not call this -- the metric treatment is folded into the device moments stage -- it exists for API parity."""
        if not self._metric:
            self.promote_scales()
            n = data.shape[0]
            out = ((data - data.mean()) / data.std()) / np.sqrt((n - 1) / n)
            for mv, kind in self._mv_scales.items():
                if kind in (Scale.ORD, Scale.NOM):
                    levels = pd.Series(out[mv].unique())
                    ranks = dict(zip(levels, levels.rank()))
                    out[mv] = out[mv].map(ranks).astype(float)
                    codes = out[mv].values
                    self._dummies[mv] = (codes[:, None] == np.arange(1, len(levels) + 1)[None, :]).astype(int)
            return out
        values = data.fillna(data.mean()) if self._missing else data
        centred = values - values.mean()
        if self._scaled:
            n = values.shape[0]
            g = np.std(values.values.reshape(-1), ddof=1) * np.sqrt((n - 1) / n)
            centred = centred / g
        return centred
