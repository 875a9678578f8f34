"""Small host-side helpers (graph ordering, enum payloads)."""
from collections import OrderedDict


class MissingDataError(Exception):
    """Raised for missing cells in a Scale.ORD / Scale.NOM column.  The reference has no defined behaviour there: depending on mode, scheme and on
    WHERE the NaN sits it raises ``statsmodels.tools.sm_exceptions.MissingDataError("exog contains inf or nans")`` (every Mode-B block, the PATH
    scheme: weights.py:141, scheme.py:50), "Could not converge", a shape error (two NaNs in one column, scale.py:88), returns all-zero weights
    (ORD), or an estimate that changes when the rows of the data set are permuted (NOM: util.groupby_mean sorts one dict key per NaN cell in
    between the category keys, scale.py:83-88) -- tests/golden/g16_ordnom_missing_probe.json records all of it.  This backend raises the reference's
    most frequent failure, by name and message, for every such data set."""


class Value:
    """Payload type of the Mode / Scheme / Scale enums: compares by its tag (reference util.py:115-124)."""

    def __init__(self, tag):
        self._tag = tag

    def tag(self):
        return self._tag

    def __eq__(self, other):
        return isinstance(other, Value) and self._tag == other._tag

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._tag)


class TopoSort:
    """Kahn's algorithm with the visiting order the reference produces (util.py:127-160): vertices are
    registered target-first per edge, and ready vertices are taken LIFO.  Unlike the reference,
    ``order()`` may be called repeatedly."""

    def __init__(self):
        self._indegree = OrderedDict()
        self._children = {}
        self._edges = []

    def append(self, src, dest):
        self._edges.append((src, dest))
        self._indegree[dest] = self._indegree.get(dest, 0) + 1
        self._indegree.setdefault(src, 0)
        self._children.setdefault(src, [])
        self._children.setdefault(dest, [])
        self._children[src].append(dest)

    def order(self):
        remaining = OrderedDict(self._indegree)
        ready = [v for v, d in remaining.items() if d == 0]
        out = []
        while ready:
            v = ready.pop()
            out.append(v)
            for child in self._children[v]:
                remaining[child] -= 1
                if remaining[child] == 0:
                    ready.append(child)
        if any(d != 0 for d in remaining.values()):
            raise ValueError("Structural graph contains cycles.")
        return out

    def elements(self):
        return list(self._edges)


def sort_cols(frame):
    """Columns in sorted order (used by the tests the same way the reference's helper is)."""
    return frame.reindex(sorted(frame.columns), axis=1)
