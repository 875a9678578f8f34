"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the PLS-PM metric hot path.

A plain-NumPy, data-level restatement of the reference algorithm (GoogleCloudPlatform/plspm-python
v0.5.6).  It exists to CHECK the HIP path; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product package (plspm-python_amd/plspm) never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
  (a) the reference's own golden CSVs for this path (tests/golden/ref_data/satisfaction*.csv,
      originally reference tests/data/, R `plspm` output), and
  (b) tests/golden/*.npz, produced by importing the real reference in the build container
      (tests/golden/make_golden.py, interpreter /opt/conda/bin/python3.9 + oracle/refshim.py).

Every function cites the reference file:line it follows (paths relative to /root/reference).
The arithmetic works on the N x P observation matrix exactly like the reference does (no
second-moment shortcut), so that the GPU's Gram formulation is checked against an independent
formulation.

Conventions: X is the filtered raw data, N x P float64, columns in "data order" (the order of
Config.add_lv calls, reference config.py:269).  `blocks[l]` lists the columns of LV l, LVs in path
order.  `C[i, j] = 1` iff LV j -> LV i (reference config.py:95).
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

CENTROID, FACTORIAL, PATH = "centroid", "factorial", "path"


@dataclass
class Model:
    blocks: List[np.ndarray]          # per LV (path order): column indices into X
    C: np.ndarray                     # L x L 0/1 lower-triangular path matrix
    modes: Sequence[str]              # "A" / "B" per LV
    scheme: str = CENTROID
    scaled: bool = True
    max_iter: int = 100
    tol: float = 1e-6
    scales: Optional[Sequence[str]] = None   # non-metric data: "NUM" / "RAW" per data column (None = metric)

    def __post_init__(self):
        self.blocks = [np.asarray(b, dtype=np.int64) for b in self.blocks]
        self.C = np.asarray(self.C, dtype=np.int64)
        self.L = len(self.blocks)
        self.P = int(sum(len(b) for b in self.blocks))
        # MVs in path-LV order == row order of the reference's `weights` frame (weights.py:31,69)
        self.mv_order = np.concatenate(self.blocks)


class NotConverged(Exception):
    pass


# ----------------------------------------------------------------------------- pre-treatment
def treat_metric(X: np.ndarray, scaled: bool) -> np.ndarray:
    """Config.treat metric branch (config.py:299-305) + util.treat (util.py:33-39).

    scaled=True divides the centred data by ONE scalar: std(ddof=1) over all N*P raw values times
    sqrt((N-1)/N) (config.py:302-303); scaled=False only centres (config.py:305).
    """
    n = X.shape[0]
    Xc = X - X.mean(axis=0)
    if scaled:
        g = np.std(X.reshape(-1), ddof=1) * math.sqrt((n - 1) / n)
        Xc = Xc / g
    return Xc


def filter_missing(X: np.ndarray, model) -> np.ndarray:
    """Config.filter (config.py:273-285): drop the rows in which ALL MVs of some LV are missing."""
    drop = np.zeros(X.shape[0], dtype=bool)
    for b in model.blocks:
        drop |= np.isnan(X[:, b]).all(axis=1)
    return X[~drop]


def impute(X: np.ndarray) -> np.ndarray:
    """util.impute (util.py:61-68): column-mean imputation of the remaining missing values (metric data only)."""
    means = np.nanmean(X, axis=0)
    return np.where(np.isnan(X), means[None, :], X)


def correction(n: int) -> float:
    """plspm.py:65."""
    return math.sqrt(n / (n - 1))


# ----------------------------------------------------------------------------- inner weights
def scheme_centroid(C, Y):
    """scheme.py:27-28."""
    return np.sign(np.corrcoef(Y, rowvar=False) * (C + C.T))


def scheme_factorial(C, Y):
    """scheme.py:36-37 (np.cov default ddof=1)."""
    return np.cov(Y, rowvar=False) * (C + C.T)


def scheme_path(C, Y):
    """scheme.py:45-54.  OLS without intercept via pinv (statsmodels OLS.fit default method)."""
    E = C.astype(np.float64)
    L = C.shape[0]
    for i in range(L):
        follow = C[i, :] == 1
        if C[i, :].sum() > 0:
            E[follow, i] = np.linalg.pinv(Y[:, follow]) @ Y[:, i]
        predec = C[:, i] == 1
        if C[:, i].sum() > 0:
            E[predec, i] = np.corrcoef(np.column_stack((Y[:, predec], Y[:, i])), rowvar=False)[:, -1][:-1]
    return E


_SCHEMES = {CENTROID: scheme_centroid, FACTORIAL: scheme_factorial, PATH: scheme_path}


# ----------------------------------------------------------------------------- outer weights
def mode_a(Xk, z):
    """mode.py:28-29: (1/N) X_k' z."""
    return (Xk.T @ z) / Xk.shape[0]


def mode_b(Xk, z):
    """mode.py:50-52: least squares of z on X_k (LAPACK gelsd, like scipy.linalg.lstsq)."""
    return np.linalg.lstsq(Xk, z, rcond=None)[0]


# ----------------------------------------------------------------------------- the solver
def init_weights(Xt, model: Model, corr: float) -> np.ndarray:
    """_MetricWeights.__init__ (weights.py:28-39): W = odm * diag(corr / std1(X odm))."""
    W = np.zeros((Xt.shape[1], model.L))
    for l, b in enumerate(model.blocks):
        W[b, l] = corr / np.std(Xt[:, b].sum(axis=1), ddof=1)
    return W


def iterate(Xt, W, model: Model, corr: float):
    """_MetricWeights.iterate (weights.py:41-54).  Returns (W_new, convergence)."""
    w_old = W.sum(axis=1)
    Y = Xt @ W                                                    # weights.py:43
    Y = (Y - Y.mean(axis=0)) / np.std(Y, axis=0, ddof=1) / corr   # weights.py:44 (util.treat)
    E = _SCHEMES[model.scheme](model.C, Y)                        # weights.py:45
    Z = Y @ E                                                     # weights.py:46
    if DIAG is not None:                                          # (test diagnostics, as in solve_nonmetric: scores of linked LVs that are uncorrelated to rounding)
        with np.errstate(all="ignore"):
            Rs = np.abs(np.corrcoef(Y, rowvar=False))
            zs = float(np.min(np.std(Z, axis=0)) / max(float(np.max(np.std(Y, axis=0))), 1e-300))
        link = (np.asarray(model.C) + np.asarray(model.C).T) > 0
        if link.any():
            rmin = float(np.nanmin(np.where(link, Rs, np.nan)))
            DIAG["min_linked_score_corr"] = min(DIAG.get("min_linked_score_corr", 1.0), rmin if rmin == rmin else 0.0)
        DIAG["min_z_scale"] = min(DIAG.get("min_z_scale", 1.0), zs if zs == zs else 0.0)
    Wn = W.copy()
    for l, b in enumerate(model.blocks):                          # weights.py:47-50
        Xk = Xt[:, b]
        Wn[b, l] = mode_a(Xk, Z[:, l]) if model.modes[l] == "A" else mode_b(Xk, Z[:, l])
    w_new = Wn.sum(axis=1)
    conv = float(np.sum((np.abs(w_old) - np.abs(w_new)) ** 2))    # weights.py:51-52
    return Wn, conv


def finalize(Xt, W, model: Model, corr: float):
    """_MetricWeights.calculate (weights.py:56-70).

    Returns (scores N x L, weights P [data-column order], cor P x L, sign L).  The sign rule counts
    the sign of the correlation of EVERY MV with the LV (the `cor * odm` product yields -0.0 off
    block and copysign(1, -0.0) = -1, weights.py:62-64); scores are flipped, weights are not.
    """
    wf = 1.0 / (np.std(Xt @ W, axis=0, ddof=1) / corr)            # weights.py:57
    W = W * wf                                                    # weights.py:58-59
    scores = Xt @ W                                               # weights.py:60
    xs = np.std(Xt, axis=0, ddof=1)
    ss = np.std(scores, axis=0, ddof=1)
    n = Xt.shape[0]
    cov = (Xt - Xt.mean(axis=0)).T @ (scores - scores.mean(axis=0)) / (n - 1)
    cor = cov / np.outer(xs, ss)                                  # weights.py:61
    odm = (W != 0).astype(int)                                    # weights.py:62
    prod = cor * odm                                              # -0.0 where cor<0 and odm==0
    sgn_mv = np.where(np.isnan(prod), 1.0, np.copysign(1.0, prod))      # a zero-variance column: pandas' corr() returns np.nan -- sign bit CLEAR -- for it, so math.copysign
                                                                  # (weights.py:63) makes its vote +1 in every LV (NumPy's own 0 / 0 would carry the sign bit: -1)
    sign = np.copysign(1.0, sgn_mv.sum(axis=0))                   # weights.py:64
    if DIAG is not None:
        # a vote is DECISIVE when turning it alone turns the LV: sums of 0 / +1 lean on their +1 votes, sums of -1 / -2 on their -1 votes.  The smallest |correlation| behind a
        # decisive vote tells a sign rule settled by the sign bit of rounding residue (an off-block item that is exactly uncorrelated with the score in a sample of integers)
        tot = sgn_mv.sum(axis=0)
        decisive = ((tot >= 0) & (tot <= 1))[None, :] & (sgn_mv > 0) | ((tot < 0) & (tot >= -2))[None, :] & (sgn_mv < 0)
        mag = np.where(decisive & ~np.isnan(cor), np.abs(cor), 1.0)
        DIAG["min_decisive_vote_corr"] = min(DIAG.get("min_decisive_vote_corr", 1.0), float(mag.min()) if mag.size else 1.0)
    if -1 in sign:                                                # weights.py:65-68
        scores = scores * sign
    return scores, W.sum(axis=1), cor, sign


def solve(Xt, model: Model, corr: float):
    """WeightsCalculatorFactory.calculate (weights.py:172-187), metric branch.

    Returns dict(scores, weights, iterations, cor, sign).  Raises NotConverged exactly where the
    reference raises (iteration counter exceeds max_iter, weights.py:183-186).
    """
    W = init_weights(Xt, model, corr)
    iteration = 0
    while True:
        iteration += 1
        W, conv = iterate(Xt, W, model, corr)
        if conv < model.tol or iteration > model.max_iter:
            break
    if iteration > model.max_iter:
        raise NotConverged("Could not converge after %d iterations" % iteration)
    scores, weights, cor, sign = finalize(Xt, W, model, corr)
    return dict(scores=scores, weights=weights, iterations=iteration, cor=cor, sign=sign)


# ----------------------------------------------------------------------------- non-metric (NUM / RAW) solver
def treat_numpy(v):
    """util.treat_numpy (util.py:43-53)."""
    v = v - np.nanmean(v)
    return v / np.nanstd(v, axis=0, ddof=1)


def treat_nonmetric(X):
    """Config.treat non-metric branch (config.py:314): util.treat(data) / sqrt((N-1)/N), i.e. population-standardised."""
    n = X.shape[0]
    if np.isnan(X).any():                                         # pandas mean / std skip NaN; N stays the row count
        return (X - np.nanmean(X, axis=0)) / np.nanstd(X, axis=0, ddof=1) / math.sqrt((n - 1) / n)
    return (X - X.mean(axis=0)) / np.std(X, axis=0, ddof=1) / math.sqrt((n - 1) / n)


def rank_column(col):
    """util.rank (util.py:80-86): every value replaced by the rank (1..C) of its unique value."""
    uniq = np.unique(col)
    return (np.searchsorted(uniq, col) + 1).astype(np.float64)


def dummy_matrix(ranked):
    """util.dummy (util.py:89-95): N x C indicator matrix of the rank codes 1..C."""
    c = int(ranked.max())
    return (ranked[:, None] == np.arange(1, c + 1)[None, :]).astype(np.float64)


def _quantify(dummies, z):
    """scale.py:47-52: mean of z inside every (possibly merged) category."""
    return (dummies * z[:, None]).sum(axis=0) / dummies.sum(axis=0)


def _ordinalize(scaling, dummies, z, sign):
    """scale.py:54-66: pool adjacent categories until the category means are monotone (non-decreasing for sign=+1)."""
    scaling = np.array(scaling, dtype=np.float64)
    while True:
        ncols = dummies.shape[1]
        for n in range(ncols - 1):
            if np.sign(scaling[n] - scaling[n + 1]) == sign:
                dummies[:, n + 1] = dummies[:, n] + dummies[:, n + 1]
                dummies = np.delete(dummies, n, axis=1)
                scaling = _quantify(dummies, z)
                break
        if dummies.shape[1] == 1 or dummies.shape[1] == ncols:
            break
    x_new = dummies @ scaling
    return x_new, np.var(x_new)


def treat_nonmetric_full(X, scales):
    """Config.treat non-metric branch (config.py:314-318): standardise; ORD / NOM columns become rank codes + dummy matrices."""
    X0 = treat_nonmetric(X)
    dummies = {}
    for p, kind in enumerate(scales):
        if kind in ("ORD", "NOM"):
            X0[:, p] = rank_column(X0[:, p])
            dummies[p] = dummy_matrix(X0[:, p])
    return X0, dummies


def nm_init_scores(X0, model: Model):
    """_NonmetricWeights.__init__ (weights.py:82-98, no missing data): equal weights 1/sqrt(k) per block, on the treated values."""
    Y = np.zeros((X0.shape[0], model.L))
    for l, b in enumerate(model.blocks):
        w = np.ones(len(b)) / math.sqrt(len(b))
        Xb = X0[:, b]
        if np.isnan(Xb).any():                                    # weights.py:88-96: NaN-aware, per-row normaliser
            present = ~np.isnan(Xb)
            den = ((w[None, :] * present) ** 2).sum(axis=1)
            if np.any(den == 0):
                raise ValueError("All mvs for an lv are NaN in some row")
            Y[:, l] = np.nansum(Xb * w[None, :], axis=1) / den
        else:
            Y[:, l] = Xb @ w
    return Y


# Test diagnostics (tests/test_gpu_fuzz.py): a dict that solve_nonmetric fills when it is not None.  "min_direction_margin": the smallest relative gap
# |var_incr - var_decr| / max(...) over all (trip, ordinal MV) decisions of scale.py:74.  The two variances are EQUAL in exact arithmetic whenever the category
# means are mirror-symmetric (e.g. three categories with means a, b, a and equal outer counts -- coarse data, small samples, the first trips); the reference then takes the
# direction np.var's rounding happens to favour, and two correct evaluations of the same formulas may walk different trajectories to the same fixed point.
# "min_quant_var_over_z": the smallest var(quantified MV) / var(z) -- category means that tie leave a quantification of rounding residue (1e-34), which the reference
# standardises into a +-1 pattern of that residue's signs (or, when the tie is exact in floating point too, into NaN: the estimate fails).  "min_z_scale": the smallest
# std(z_l) / std(scores) -- an inner estimate that is residue itself (two initial scores whose covariance is exactly zero: z = -8e-18 y); "min_linked_score_corr": the smallest
# |correlation| between the scores of two linked LVs (the same coincidence under the centroid scheme, which keeps only the residue's sign).  An evaluation on integer counts
# meets these ties EXACTLY (0 / 0: the estimate fails); which of the two the reference does is decided by its own rounding.
DIAG = None


def solve_nonmetric(X0, model: Model, corr: float, dummies=None):
    """WeightsCalculatorFactory.calculate, non-metric branch: _NonmetricWeights.__init__ / iterate / calculate
    (weights.py:73-154) with the four Scale operators (scale.py) and the Mode-B correction get_Z_for_mode_b (weights.py:135-145).
    X0 is the treated data (ranks in the ORD / NOM columns).  No sign rule here."""
    n = X0.shape[0]
    dummies = dummies or {}
    cur = X0.copy()                                               # self.__mv_grouped_by_lv (updated column by column)
    Y = nm_init_scores(X0, model)
    iteration = 0
    while True:
        iteration += 1
        Y_old = Y.copy()
        E = _SCHEMES[model.scheme](model.C, Y)
        Z = Y @ E
        W = np.zeros((X0.shape[1], model.L))
        if DIAG is not None:                                      # (an inner estimate that is rounding residue: e.g. two initial scores with a covariance of exactly zero)
            with np.errstate(all="ignore"):
                zs = float(np.min(np.std(Z, axis=0)) / max(float(np.max(np.std(Y, axis=0))), 1e-300))
            DIAG["min_z_scale"] = min(DIAG.get("min_z_scale", 1.0), zs if zs == zs else 0.0)
            with np.errstate(all="ignore"):                          # (linked LVs whose scores are uncorrelated to rounding: the centroid scheme takes the SIGN of that residue)
                Rs = np.abs(np.corrcoef(Y, rowvar=False))
            link = (np.asarray(model.C) + np.asarray(model.C).T) > 0
            if link.any():
                rmin = float(np.nanmin(np.where(link, Rs, np.nan)))
                DIAG["min_linked_score_corr"] = min(DIAG.get("min_linked_score_corr", 1.0), rmin if rmin == rmin else 0.0)
        for l, b in enumerate(model.blocks):
            z = Z[:, l]
            betas = None
            for j, p in enumerate(b):
                kind = model.scales[p]
                if kind == "NUM":
                    f = int(np.isfinite(X0[:, p]).sum())                                # scale.py:27-30 (finite cells only)
                    cur[:, p] = treat_numpy(X0[:, p]) * math.sqrt(f / (f - 1))
                elif kind == "RAW":
                    cur[:, p] = X0[:, p]                                                # scale.py:38-39
                else:
                    zc = z
                    if model.modes[l] == "B" and len(b) > 1:                            # weights.py:135-145
                        if betas is None:
                            A = np.column_stack((np.ones(n), cur[:, b]))
                            betas = (np.linalg.pinv(A) @ z)[1:]
                        others = np.delete(np.arange(len(b)), j)
                        zc = (1.0 / betas[j]) * (z - cur[:, b[others]] @ betas[others])
                    d = dummies[p]
                    means = _quantify(d, zc)                                            # util.groupby_mean on [codes; z] (scale.py:70-71, 85-86)
                    if kind == "ORD":
                        x_inc, v_inc = _ordinalize(means, d.copy(), zc, 1)
                        x_dec, v_dec = _ordinalize(means, d.copy(), zc, -1)
                        xq = -x_dec if v_inc < v_dec else x_inc                         # scale.py:74
                        if DIAG is not None:                                            # (test diagnostics: how close this run came to a coin toss)
                            big = max(v_inc, v_dec)
                            margin = abs(v_inc - v_dec) / big if big > 0 else 1.0
                            DIAG["min_direction_margin"] = min(DIAG.get("min_direction_margin", 1.0), margin)
                    else:
                        xq = d @ means                                                  # scale.py:87
                    if DIAG is not None:                                                # (a quantification that is rounding residue: category means that tie)
                        with np.errstate(all="ignore"):
                            q = float(np.var(xq) / max(float(np.var(zc)), 1e-300))
                        DIAG["min_quant_var_over_z"] = min(DIAG.get("min_quant_var_over_z", 1.0), q if q == q else 0.0)
                    cur[:, p] = treat_numpy(xq) * corr
            Xk = cur[:, b]
            if np.isnan(X0[:, b]).any():                                                # "lv in mv_grouped_by_lv_missing" (weights.py:88-89)
                if model.modes[l] != "A":
                    raise Exception("Missing nonmetric data is not supported in mode B")  # mode.py:55-56
                present = (~np.isnan(X0[:, b])).astype(np.float64)
                w = np.nansum(Xk * z[:, None], axis=0) / ((present * z[:, None]) ** 2).sum(axis=0)      # mode.py:35-36
                y = np.nansum(Xk * w[None, :], axis=1) / ((present * w[None, :]) ** 2).sum(axis=1)      # mode.py:37-38
            else:
                if model.modes[l] == "A":
                    w = (Xk.T @ z) / np.sum(z ** 2)
                else:
                    w = np.linalg.lstsq(Xk, z, rcond=None)[0]
                y = Xk @ w
            W[b, l] = w
            Y[:, l] = treat_numpy(y) * corr
        conv = float(np.sum((np.abs(Y_old) - np.abs(Y)) ** 2))                          # weights.py:120
        if conv < model.tol or iteration > model.max_iter:
            break
    if iteration > model.max_iter:
        raise NotConverged("Could not converge after %d iterations" % iteration)
    wf = 1.0 / (np.nanstd(cur @ W, axis=0, ddof=1) / corr)         # weights.py:130 (a NaN anywhere in a row drops the row: NaN * 0)
    weights = (W * wf).sum(axis=1)                                # weights.py:131-132
    return dict(scores=Y, weights=weights, iterations=iteration, data=cur)


# ----------------------------------------------------------------------------- inner model
def inner_model(C, scores):
    """InnerModel.__init__ (inner_model.py:58-75): OLS with intercept per endogenous LV.

    Returns (B L x L path coefficients, r2 L, r2_adj L).
    """
    L = C.shape[0]
    n = scores.shape[0]
    B = np.zeros((L, L))
    r2 = np.zeros(L)
    r2_adj = np.zeros(L)
    for i in range(L):
        if C[i, :].sum() == 0:
            continue
        ivs = np.where(C[i, :] == 1)[0]
        A = np.column_stack((np.ones(n), scores[:, ivs]))
        y = scores[:, i]
        params = np.linalg.pinv(A) @ y
        resid = y - A @ params
        yc = y - y.mean()
        r2[i] = 1.0 - (resid @ resid) / (yc @ yc)
        B[i, ivs] = params[1:]
        r2_adj[i] = 1 - (1 - r2[i]) * (n - 1) / (n - C[i, :].sum() - 1)
    return B, r2, r2_adj


def effects(B):
    """_effects (inner_model.py:33-53).  Returns (pairs [(from, to)], direct, indirect, total)."""
    L = B.shape[0]
    indirect = np.zeros_like(B)
    if L == 2:
        total = B.copy()
    else:
        power = B.copy()
        for _ in range(1, L):
            power = power @ B
            indirect = indirect + power
        total = B + indirect
    pairs, d, ind, tot = [], [], [], []
    for f in range(L):
        for t in range(L):
            if f != t and total[t, f] != 0:
                pairs.append((f, t)); d.append(B[t, f]); ind.append(indirect[t, f]); tot.append(total[t, f])
    return pairs, np.array(d), np.array(ind), np.array(tot)


def crossloadings(Xt, scores):
    """OuterModel.__init__ (outer_model.py:26): Pearson correlation of every MV with every LV score."""
    if np.isnan(Xt).any():                                        # DataFrame.corrwith: pairwise-complete observations per MV
        out = np.zeros((Xt.shape[1], scores.shape[1]))
        for p in range(Xt.shape[1]):
            ok = ~np.isnan(Xt[:, p])
            for l in range(scores.shape[1]):
                out[p, l] = np.corrcoef(Xt[ok, p], scores[ok, l])[0, 1]
        return out
    n = Xt.shape[0]
    Xc = Xt - Xt.mean(axis=0)
    Sc = scores - scores.mean(axis=0)
    cov = Xc.T @ Sc / (n - 1)
    return cov / np.outer(np.std(Xt, axis=0, ddof=1), np.std(scores, axis=0, ddof=1))


def loadings(Xt, scores, model: Model):
    """bootstrap.py:63-64 / outer_model.py:27: own-block crossloading per MV (data-column order)."""
    cl = crossloadings(Xt, scores)
    out = np.zeros(Xt.shape[1])
    for l, b in enumerate(model.blocks):
        out[b] = np.nan_to_num(cl[b, l], nan=0.0)        # (crossloadings * odm).sum(axis=1): pandas' sum skips the NaN a zero-variance column's corrwith gives -- loading 0
    return out


# ----------------------------------------------------------------------------- whole fit / bootstrap
def fit(X, model: Model, corr: Optional[float] = None, _treated=None):
    """Estimator.estimate + the statistics the hot path feeds (estimator.py:29-55, plspm.py:63-72).
    `_treated` (fit_two_stage only): (treated data, dummy matrices) of a non-metric model that Config.treat produced earlier -- the second stage of a HOC
    estimate runs on the treatment of the FIRST (estimator.py:33,52: `treated_data` is treated once).

    `corr` defaults to sqrt(N/(N-1)) of X; the bootstrap passes the ORIGINAL fit's value
    (the calculator is built once, plspm.py:65-67, and cloned per replicate).
    """
    n = X.shape[0]
    if corr is None:
        corr = correction(n)
    if model.scales is not None:
        X0, dummies = _treated if _treated is not None else treat_nonmetric_full(X, model.scales)       # estimator.py:33 (config.py:306-318)
        s = solve_nonmetric(X0, model, corr, dummies)             # estimator.py:39 non-metric
        Xt = s["data"]
        s["sign"] = np.ones(model.L)
    else:
        if np.isnan(X).any():
            X = impute(X)                                         # config.py:300
        Xt = treat_metric(X, model.scaled)                        # estimator.py:33
        s = solve(Xt, model, corr)                                # estimator.py:39 (== :52, no HOC)
    B, r2, r2_adj = inner_model(model.C, s["scores"])             # plspm.py:71
    pairs, d, ind, tot = effects(B)
    cl = crossloadings(Xt, s["scores"])
    ld = np.zeros(X.shape[1])
    for l, b in enumerate(model.blocks):
        ld[b] = np.nan_to_num(cl[b, l], nan=0.0)         # (as loadings(): pandas' sum skips NaN -- a zero-variance column has loading 0, outer_model.py:27 / bootstrap.py:62)
    return dict(treated=Xt, scores=s["scores"], weights=s["weights"], iterations=s["iterations"],
                sign=s["sign"], path_coef=B, r2=r2, r2_adj=r2_adj, effect_pairs=pairs, direct=d,
                indirect=ind, total=tot, crossloadings=cl, loadings=ld)


def fit_two_stage(X, model1: Model, stage2, C2, modes2, corr: Optional[float] = None):
    """Estimator.estimate with higher order constructs (estimator.py:29-55): stage 1 = `model1` (every HOC replaced by its
    constituent LVs, estimator.py:60-74); stage 2 = the original path `C2` where LV l is `stage2[l]`: ("lv", j) -- stage-1
    LV j with its own MVs -- or ("hoc", [j, ...]) -- a HOC whose MVs are the stage-1 SCORES of those LVs, Scale.NUM
    (estimator.py:43-52).  Non-metric models (NUM / RAW / ORD / NOM columns).  Returns the stage-2 fit (its MV order:
    stage-2 LV by LV) plus `iterations1`."""
    if corr is None:
        corr = correction(X.shape[0])
    # Config.treat runs ONCE per estimate (estimator.py:33); the second stage is calculate(treated_data, config.path()) on that same frame with one score
    # column per constituent added (estimator.py:43-52): a plain MV enters stage 2 with its TREATED values -- rank codes for ORD / NOM, with the dummy
    # matrices of the first treatment (config.dummies) -- not with the quantification stage 1 ended on.  (Ranking the quantified column instead merges every
    # pair of categories that stage 1 had pooled: found by tests/golden/sweep_oracle_vs_reference.py `hocord`, round 6; NUM / RAW columns are standardised
    # values either way.)
    treated1 = treat_nonmetric_full(X, model1.scales) if model1.scales is not None else None
    r1 = fit(X, model1, corr, _treated=treated1)
    cols, blocks2, dummies2 = [], [], {}
    for kind, ref in stage2:
        start = len(cols)
        if kind == "lv":
            for p in model1.blocks[ref]:
                if treated1 is not None and p in treated1[1]:
                    dummies2[len(cols)] = treated1[1][p]
                cols.append(treated1[0][:, p] if treated1 is not None else r1["treated"][:, p])
        else:
            cols.extend(r1["scores"][:, j] for j in ref)
        blocks2.append(np.arange(start, len(cols)))
    X2 = np.column_stack(cols)
    # stage-2 scales: a plain MV keeps its own (an ordinal MV is quantified again in stage 2), a HOC's score columns are Scale.NUM
    scales2 = []
    for kind, ref in stage2:
        scales2.extend([model1.scales[p] for p in model1.blocks[ref]] if (kind == "lv" and model1.scales is not None) else ["NUM"] * (len(model1.blocks[ref]) if kind == "lv" else len(ref)))
    model2 = Model(blocks2, np.asarray(C2), modes2, model1.scheme, model1.scaled, max_iter=model1.max_iter, tol=model1.tol, scales=scales2)
    r2 = fit(X2, model2, corr, _treated=(X2, dummies2) if treated1 is not None else None)
    r2["iterations1"] = r1["iterations"]
    return r2


def bootstrap_replicate(X, model: Model, idx, corr: float):
    """One pass of BootstrapProcess.run's loop body (bootstrap.py:54-66) on explicit indices.

    Returns the row [weights (P, data order) | r2 (L) | total (n_eff) | direct (n_eff) |
    loadings (P)] plus the iteration count; raises NotConverged where the reference would drop
    the replicate (bare except, bootstrap.py:65-66).
    """
    r = fit(X[np.asarray(idx), :], model, corr)
    row = np.concatenate((r["weights"], r["r2"], r["total"], r["direct"], r["loadings"]))
    return row, r["iterations"]


def summary(samples, original):
    """_create_summary (bootstrap.py:24-32): columns original, mean, std.error, perc.025, perc.975,
    t stat.  pandas std = ddof 1; pandas quantile = linear interpolation."""
    samples = np.asarray(samples, dtype=np.float64)
    mean = samples.mean(axis=0)
    sd = samples.std(axis=0, ddof=1)
    lo = np.quantile(samples, 0.025, axis=0)
    hi = np.quantile(samples, 0.975, axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.asarray(original, dtype=np.float64) / sd
    return np.column_stack((original, mean, sd, lo, hi, t))


# ----------------------------------------------------------------------------- synthetic data
SAT_LVS = ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]
SAT_EDGES = [("IMAG", "EXPE"), ("IMAG", "SAT"), ("IMAG", "LOY"), ("EXPE", "QUAL"), ("EXPE", "VAL"),
             ("EXPE", "SAT"), ("QUAL", "VAL"), ("QUAL", "SAT"), ("VAL", "SAT"), ("SAT", "LOY")]


def satisfaction_C():
    """Path matrix of the 6-LV satisfaction structure (tests/test_regression_metric.py:23-30) in the
    topological order Structure.path() yields: IMAG, EXPE, QUAL, VAL, SAT, LOY."""
    C = np.zeros((6, 6), dtype=np.int64)
    for s, t in SAT_EDGES:
        C[SAT_LVS.index(t), SAT_LVS.index(s)] = 1
    return C


def chain_C(L):
    """C5 structure: edges j-1 -> j and j-3 -> j."""
    C = np.zeros((L, L), dtype=np.int64)
    for j in range(L):
        if j - 1 >= 0:
            C[j, j - 1] = 1
        if j - 3 >= 0:
            C[j, j - 3] = 1
    return C


def synth(n, C, mvs_per_lv=10, seed=0, dtype=np.float64):
    """Synthetic generator of SURVEY.md 8(d): eta_j = sum_i 0.4 C[j,i] eta_i + eps; x_jk = lambda_k
    eta_j + delta, lambda = linspace(0.5, 0.9, k), delta ~ N(0, 0.6^2).  Draw order: all eta noise
    first, then MV noise block by block."""
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
