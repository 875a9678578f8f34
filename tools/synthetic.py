"""Synthetic workload generator of SURVEY.md 8(d) for bench.py and the tools (no PLS arithmetic in here: data and the
satisfaction path structure only).  The test-only oracle keeps an identical generator for its own fixtures;
tests/test_host_api.py::test_workload_generator_matches_the_oracle_copy pins the two together."""
import numpy as np

SAT_LVS = ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]
SAT_EDGES = [("IMAG", "EXPE"), ("IMAG", "SAT"), ("IMAG", "LOY"), ("EXPE", "QUAL"), ("EXPE", "VAL"), ("EXPE", "SAT"),
             ("QUAL", "VAL"), ("QUAL", "SAT"), ("VAL", "SAT"), ("SAT", "LOY")]


def satisfaction_C():
    """6 x 6 path matrix of the satisfaction model, C[i, j] = 1 iff LV j -> LV i (reference README / tests)."""
    C = np.zeros((6, 6), dtype=np.int64)
    for frm, to in SAT_EDGES:
        C[SAT_LVS.index(to), SAT_LVS.index(frm)] = 1
    return C


def chain_C(L):
    """Structure of BASELINE.json configs[4]: edges j-1 -> j and j-3 -> j."""
    C = np.zeros((L, L), dtype=np.int64)
    for j in range(L):
        if j - 1 >= 0:
            C[j, j - 1] = 1
        if j - 3 >= 0:
            C[j, j - 3] = 1
    return C


def synth(n, C, mvs_per_lv=10, seed=0, dtype=np.float64):
    """eta_j = sum_i 0.4 C[j,i] eta_i + eps;  x_jk = lambda_k eta_j + delta, lambda = linspace(0.5, 0.9, k), delta ~ N(0, 0.6^2).
    Draw order: all eta noise first, then MV noise block by block."""
    rng = np.random.default_rng(seed)
    L = C.shape[0]
    eps = rng.standard_normal((n, L))
    eta = np.zeros((n, L))
    for j in range(L):
        eta[:, j] = eps[:, j] + 0.4 * (eta[:, C[j, :] == 1]).sum(axis=1)
    lam = np.linspace(0.5, 0.9, mvs_per_lv)
    X = np.empty((n, L * mvs_per_lv), dtype=dtype)
    for j in range(L):
        delta = 0.6 * rng.standard_normal((n, mvs_per_lv))
        X[:, j * mvs_per_lv:(j + 1) * mvs_per_lv] = eta[:, [j]] * lam[None, :] + delta
    blocks = [np.arange(j * mvs_per_lv, (j + 1) * mvs_per_lv) for j in range(L)]
    return X, blocks
