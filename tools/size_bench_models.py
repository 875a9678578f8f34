"""Inner models of the size benches (tools/size_bench.py, tools/solver_quad_ab.py): a chain with a few extra paths."""
import numpy as np


def chain_C(L):
    C = np.zeros((L, L), dtype=np.int64)
    for j in range(L):
        if j - 1 >= 0: C[j, j - 1] = 1
        if j - 3 >= 0: C[j, j - 3] = 1
    return C
