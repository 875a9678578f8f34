"""Seeded random models for the differential fuzz tests (tests/test_gpu_fuzz.py, tools/experiments/*_fuzz_sweep.py) and for the oracle-against-the-real-reference sweep
(tests/golden/sweep_oracle_vs_reference.py, build container only).  NumPy + the oracle's Model only: importable beside the reference's `plspm` package."""
import numpy as np

import plspm_oracle as orc


def _random_dag(L, rng, density=0.5):
    C = np.zeros((L, L), dtype=np.int64)
    for i in range(1, L):
        for j in range(i):
            if rng.random() < density:
                C[i, j] = 1
        if C[i].sum() == 0 and C[:, i].sum() == 0:
            C[i, rng.integers(0, i)] = 1
    return C


def _ragged(n, C, sizes, seed):
    rng = np.random.default_rng(seed)
    L = C.shape[0]
    eta = np.zeros((n, L))
    for j in range(L):
        eta[:, j] = rng.standard_normal(n) + 0.4 * eta[:, C[j] == 1].sum(axis=1)
    cols, blocks, at = [], [], 0
    for j, k in enumerate(sizes):
        lam = np.linspace(0.6, 0.9, k)
        cols.append(eta[:, [j]] * lam[None, :] + 0.6 * rng.standard_normal((n, k)) + 3.0 * (j + 1))     # non-zero means on purpose
        blocks.append(np.arange(at, at + k)); at += k
    return np.column_stack(cols), blocks


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(2, 9))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 9)) for _ in range(L)]
    n = int(rng.integers(30, 700))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    modes = "".join("AB"[int(rng.integers(0, 2))] if sizes[l] > 1 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    nonmetric = bool(rng.integers(0, 3) == 0)
    scaled = bool(rng.integers(0, 2))
    model = orc.Model(blocks, C, modes, scheme, scaled, tol=1e-6 if not nonmetric else 1e-7,
                      scales=(["NUM"] * X.shape[1]) if nonmetric else None)
    return X, model, nonmetric


def make_cat_case(seed):
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.integers(2, 7))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 6)) for _ in range(L)]
    n = int(rng.integers(60, 900))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    P = X.shape[1]
    kind = int(rng.integers(0, 3))            # 0 all ORD, 1 ORD / NOM mix, 2 with NUM columns
    scales, data = [], X.copy()
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    for p in range(P):
        s = "ORD" if kind == 0 else ("ORD", "NOM")[int(rng.integers(0, 2))] if kind == 1 else ("ORD", "NOM", "NUM")[int(rng.integers(0, 3))]
        scales.append(s)
        if s != "NUM":
            c = int(rng.integers(2, 13))
            data[:, p] = np.clip(np.round((c + 1) / 2.0 + float(rng.uniform(0.6, 1.4)) * c / 5.0 * Z[:, p]), 1, c)
    all_a = bool(rng.integers(0, 2))
    modes = "".join("A" if all_a or sizes[l] == 1 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    model = orc.Model(blocks, C, modes, scheme, True, tol=1e-6, scales=scales)
    return data, model


def make_missing_case(seed):
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.integers(2, 8))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 8)) for _ in range(L)]
    n = int(rng.integers(50, 1200))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    P = X.shape[1]
    Xn = X.copy()
    ncols = int(rng.integers(1, P + 1))
    for col in rng.choice(P, size=ncols, replace=False):
        k = int(rng.integers(1, max(2, int(0.15 * n))))
        Xn[rng.choice(n, size=k, replace=False), col] = np.nan
    modes = "".join("AB"[int(rng.integers(0, 2))] if sizes[l] > 1 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    model = orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2)))
    return Xn, model


def make_nmx_case(seed):
    """Scale.NUM / RAW data with NaN cells: the NaN-aware Mode-A products of mode.py:35-41 (blocks with a missing cell must be Mode A: mode.py:55-56)."""
    rng = np.random.default_rng(10000 + seed)
    L = int(rng.integers(2, 7))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 7)) for _ in range(L)]
    n = int(rng.integers(40, 800))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    Xn = X.copy()
    nrows = int(rng.integers(1, max(2, int(0.1 * n))))
    miss_blocks = set()
    for r in rng.choice(n, size=nrows, replace=False):
        l = int(rng.integers(0, L))
        if sizes[l] == 1:
            continue                                                       # (an LV whose only MV is missing in a row has no score there: the reference raises)
        k = int(rng.integers(1, sizes[l]))
        Xn[r, rng.choice(blocks[l], size=k, replace=False)] = np.nan
        miss_blocks.add(l)
    modes = "".join("A" if (l in miss_blocks or sizes[l] == 1) else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    raw = bool(rng.integers(0, 4) == 0)
    model = orc.Model(blocks, C, modes, scheme, True, tol=1e-7, scales=["RAW" if raw else "NUM"] * X.shape[1])
    return Xn, model


def make_hoc_case(seed):
    rng = np.random.default_rng(11000 + seed)
    L2 = int(rng.integers(3, 7))
    C2 = _random_dag(L2, rng, density=float(rng.uniform(0.4, 0.9)))
    h = int(rng.integers(0, L2))                               # the HOC
    k = int(rng.integers(2, 4))                                # its constituents
    expand = [[l] for l in range(L2)]
    first_of = []
    at = 0
    for l in range(L2):
        first_of.append(at)
        at += k if l == h else 1
    L1 = at
    first_of.append(L1)
    owner = []                                                 # stage-1 LV -> stage-2 LV
    for l in range(L2):
        owner += [l] * (k if l == h else 1)
    C1 = np.zeros((L1, L1), dtype=int)
    for i in range(L1):
        for j in range(L1):
            if owner[i] != owner[j]:
                C1[i, j] = C2[owner[i], owner[j]]
    sizes = [int(rng.integers(1, 6)) for _ in range(L1)]
    n = int(rng.integers(80, 700))
    X, blocks = _ragged(n, C1, sizes, seed=seed)
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    modes1 = "".join("AB"[int(rng.integers(0, 2))] if sizes[i] > 1 else "A" for i in range(L1))
    # stage 2: a plain LV keeps its mode, the HOC block (k score columns) takes a mode of its own
    modes2 = "".join(("AB"[int(rng.integers(0, 2))] if l == h else modes1[first_of[l]]) for l in range(L2))
    stage2 = [("hoc", list(range(first_of[l], first_of[l + 1]))) if l == h else ("lv", first_of[l]) for l in range(L2)]
    model1 = orc.Model(blocks, C1, modes1, scheme, True, tol=1e-7, scales=["NUM"] * X.shape[1])
    return X, model1, stage2, C2, modes2, first_of


def make_hoc_ord_case(seed):
    """The HOC model of make_hoc_case on ordinal items (3 .. 7 categories per MV, all Mode A: the reference's Mode-B correction of ordinal blocks is exercised by make_cat_case)."""
    X, model1, stage2, C2, modes2, first_of = make_hoc_case(seed)
    rng = np.random.default_rng(12000 + seed)
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    data = X.copy()
    for p in range(X.shape[1]):
        c = int(rng.integers(3, 8))
        data[:, p] = np.clip(np.round((c + 1) / 2.0 + float(rng.uniform(0.7, 1.3)) * c / 5.0 * Z[:, p]), 1, c)
    model1 = orc.Model(model1.blocks, model1.C, "A" * model1.L, model1.scheme, True, tol=1e-6, scales=["ORD"] * X.shape[1])
    return data, model1, stage2, C2, "A" * len(stage2), first_of


def make_hostile_case(seed):
    """Metric models whose DATA are unkind to a fixed-point evaluation of the moments (the int8 digit-plane Gram): columns on scales from 1e-6 to 1e6,
    offsets of up to 1e7 standard deviations, heavy tails, a handful of cells a thousand times their column's scale, five- / seven-point items read as numbers.
    At most 40 MVs in 2 ... 8 blocks, 300 ... 3,000 rows."""
    rng = np.random.default_rng(13000 + seed)
    L = int(rng.integers(2, 9))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 6)) for _ in range(L)]
    n = int(rng.integers(300, 3000))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    P = X.shape[1]
    kind = int(rng.integers(0, 6))
    allA = bool(rng.integers(0, 2))
    modes = "".join("A" if allA or sizes[l] == 1 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    # (Mode-B weights are inverse to the data's scale and the stop rule is an ABSOLUTE tolerance on them (weights.py:179-186): with a block of columns twelve orders of magnitude
    #  apart -- a covariance matrix of condition 1e24 -- or a global scale factor of 1e12 the reference's own trip count is rounding noise; such models keep three orders of magnitude)
    span, off = (6.0, 7.0) if "B" not in modes else (1.5, 3.0)
    X = X - X.mean(axis=0)
    sd = X.std(axis=0)
    if kind in (0, 5):                                         # scales
        X = X * (10.0 ** rng.uniform(-span, span, size=P))[None, :]
    if kind in (1, 5):                                         # offsets in units of the column's own spread
        X = X + (X.std(axis=0) * 10.0 ** rng.uniform(0, off, size=P) * rng.choice([-1.0, 1.0], size=P))[None, :]
    if kind == 2:                                              # heavy tails
        X = X + sd[None, :] * rng.standard_t(1.5, size=(n, P)) * 0.3
    if kind == 3:                                              # items read as numbers
        c = int(rng.choice([5, 7]))
        X = np.clip(np.round((c + 1) / 2.0 + c / 5.0 * X / sd[None, :]), 1, c)
    if kind == 4:                                              # a few wild cells
        k = max(1, int(0.004 * n * P))
        X[rng.integers(0, n, size=k), rng.integers(0, P, size=k)] *= 1.0e3
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    return X, orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2))), sizes, kind


def make_degenerate_case(seed):
    """Metric / Scale.NUM models at the edges the reference's own tests skirt: tiny samples (8 ... 40 rows: resamples with a handful of distinct rows), a column that is an
    exact copy or an exact multiple of its neighbour (rank-deficient Mode-B blocks and path regressions take the minimum-norm answers), a constant column (zero variance: the
    reference's loadings / Mode-B systems turn NaN or singular), iteration caps of 1 ... 4 and tolerances of 1e-2 / 1e-12."""
    rng = np.random.default_rng(14000 + seed)
    L = int(rng.integers(2, 6))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 5)) for _ in range(L)]
    kind = int(rng.integers(0, 5))
    n = int(rng.integers(8, 41)) if kind == 0 else int(rng.integers(60, 400))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    big = [l for l in range(L) if sizes[l] > 1]
    nonmetric_first = bool(np.random.default_rng(15000 + seed).integers(0, 3) == 0)      # (drawn ahead of the data edits: the model kind decides which duplicate is fair)
    if kind == 1 and big:                                      # an exact copy / multiple of a neighbouring column
        b = blocks[big[int(rng.integers(0, len(big)))]]
        # (a NEGATIVE multiple under Scale.NUM is left out: both columns are standardised, the initial score (x + x') / sqrt(2) is then zero in exact arithmetic, and the
        #  reference iterates on from its rounding residue -- DESIGN 6, "what the sweeps found")
        X[:, b[1]] = X[:, b[0]] * (1.0 if rng.integers(0, 2) else (2.5 if nonmetric_first else -2.5))
    if kind == 2:                                              # a constant column
        X[:, int(rng.integers(0, X.shape[1]))] = 3.25
    modes = "".join("AB"[int(rng.integers(0, 2))] if sizes[l] > 1 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    nonmetric = nonmetric_first
    max_iter, tol = 100, (1e-6 if not nonmetric else 1e-7)
    if kind == 3:
        max_iter = int(rng.integers(1, 5))
    if kind == 4:
        tol = float(rng.choice([1e-2, 1e-12]))
    model = orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2)) or nonmetric, max_iter=max_iter, tol=tol, scales=(["NUM"] * X.shape[1]) if nonmetric else None)
    return X, model, nonmetric, kind


def make_cat_big_case(seed):
    """Categorical models beyond the wave step's class and at its limits: up to 10 LVs, blocks of up to 8 items (up to 80 MVs: more than 64 take the workgroup step), items of
    up to 16 categories (the sixteen-category instantiation), 300 ... 2,500 rows; all Mode A in two of three cases."""
    rng = np.random.default_rng(16000 + seed)
    L = int(rng.integers(3, 11))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.8)))
    sizes = [int(rng.integers(1, 9)) for _ in range(L)]
    n = int(rng.integers(300, 2500))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    P = X.shape[1]
    Z = (X - X.mean(axis=0)) / X.std(axis=0)
    cmax = int(rng.choice([5, 8, 10, 13, 16]))
    data, scales = X.copy(), []
    for p in range(P):
        s_ = ("ORD", "ORD", "NOM")[int(rng.integers(0, 3))]
        scales.append(s_)
        c = int(rng.integers(2, cmax + 1))
        data[:, p] = np.clip(np.round((c + 1) / 2.0 + float(rng.uniform(0.6, 1.4)) * c / 5.0 * Z[:, p]), 1, c)
    all_a = bool(rng.integers(0, 3) != 0)
    modes = "".join("A" if all_a or sizes[l] == 1 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    return data, orc.Model(blocks, C, modes, scheme, True, tol=1e-6, scales=scales)


def make_huge_case(seed):
    """Metric / Scale.NUM models beyond the narrow and wide fuzz classes: 9 ... 48 LVs and up to 230 MVs (the 17 ... 32-LV wave solver, the rows solver of 33 ... 64 LVs, the LDS
    solver beyond 128 MVs, Gram tile counts up to 15), sparse inner models (an LV has at most six predecessors), 400 ... 2,000 rows."""
    rng = np.random.default_rng(17000 + seed)
    L = int(rng.integers(9, 49))
    C = np.zeros((L, L), dtype=np.int64)
    for i in range(1, L):
        k = int(rng.integers(1, min(i, 6) + 1))
        C[i, rng.choice(i, size=k, replace=False)] = 1
    P = int(rng.integers(L, min(230, 8 * L) + 1))
    cuts = np.sort(rng.choice(np.arange(1, P), size=L - 1, replace=False))
    sizes = np.diff(np.concatenate(([0], cuts, [P]))).tolist()
    n = int(rng.integers(400, 2000))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    allA = bool(rng.integers(0, 2))
    modes = "".join("A" if allA or sizes[l] == 1 or sizes[l] > 12 else "AB"[int(rng.integers(0, 2))] for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    nonmetric = bool(rng.integers(0, 4) == 0)
    model = orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2)) or nonmetric, tol=(1e-7 if nonmetric else 1e-6), scales=(["NUM"] * P) if nonmetric else None)
    return X, model, sizes, nonmetric


def make_rare_indicator_case(seed):
    """A metric model of make_case with one column replaced by a rare 0/1 indicator (2 ... 6 ones) and six explicit index lists: the data themselves, two ordinary resamples
    and three resamples that miss every one -- the indicator is CONSTANT there: the reference centres it to zeros (weight 0, loading 0) and counts the replicate.
    Returns (X, model, idx [6, n]) or None for the Scale.NUM seeds."""
    X, model, nonmetric = make_case(seed)
    if nonmetric:
        return None
    rng = np.random.default_rng(18000 + seed)
    n, P = X.shape
    col = int(rng.integers(0, P))
    ones = rng.choice(n, size=int(rng.integers(2, 7)), replace=False)
    X = X.copy(); X[:, col] = 0.0; X[ones, col] = 1.0
    idx = [np.arange(n), rng.integers(0, n, size=n), rng.integers(0, n, size=n)]
    rest = np.setdiff1d(np.arange(n), ones)
    idx += [rng.choice(rest, size=n, replace=True) for _ in range(3)]
    return X, model, np.asarray(idx, dtype=np.int32)


def make_raw_case(seed):
    """Non-metric models on Scale.RAW / Scale.NUM columns: all RAW (Config.treat switches scaling off: config.py:309-310) or a RAW / NUM mix (every column becomes NUM:
    config.py:311-313) -- rules of the host-side Config, so these cases are for the API-level fixtures (golden g17)."""
    X, model, _ = make_case(seed)
    rng = np.random.default_rng(20000 + seed)
    all_raw = bool(rng.integers(0, 2))
    scales = ["RAW"] * X.shape[1] if all_raw else [("RAW", "NUM")[int(rng.integers(0, 2))] for _ in range(X.shape[1])]
    if not all_raw and len(set(scales)) == 1:
        scales[0] = "NUM" if scales[0] == "RAW" else "RAW"
    return X, orc.Model(model.blocks, model.C, model.modes, model.scheme, True, tol=1e-7, scales=scales)


def make_cat_small_case(seed):
    """make_cat_case cut down to 30 ... 90 rows: nearly every resample loses categories (util.rank re-ranks the present ones), some items keep one category only (a constant
    quantification: the estimate fails as the reference's does)."""
    X, model = make_cat_case(seed)
    rng = np.random.default_rng(21000 + seed)
    n = min(X.shape[0], int(rng.integers(30, 91)))
    return X[rng.choice(X.shape[0], size=n, replace=False)], model


def make_small_int_case(seed):
    """Metric / Scale.NUM models on five- / seven-point items read as numbers, 20 ... 80 rows: small samples of integers, where covariances can be EXACTLY zero."""
    rng = np.random.default_rng(22000 + seed)
    L = int(rng.integers(2, 6))
    C = _random_dag(L, rng, density=float(rng.uniform(0.3, 0.9)))
    sizes = [int(rng.integers(1, 5)) for _ in range(L)]
    n = int(rng.integers(20, 81))
    X, blocks = _ragged(n, C, sizes, seed=seed)
    c = int(rng.choice([5, 7]))
    X = np.clip(np.round((c + 1) / 2.0 + c / 5.0 * (X - X.mean(axis=0)) / X.std(axis=0)), 1, c)
    modes = "".join("AB"[int(rng.integers(0, 2))] if sizes[l] > 1 else "A" for l in range(L))
    scheme = ["centroid", "factorial", "path"][int(rng.integers(0, 3))]
    nonmetric = bool(rng.integers(0, 3) == 0)
    model = orc.Model(blocks, C, modes, scheme, bool(rng.integers(0, 2)) or nonmetric, tol=(1e-7 if nonmetric else 1e-6), scales=(["NUM"] * X.shape[1]) if nonmetric else None)
    return X, model, nonmetric
