// solver_nmx.h -- non-metric (Scale.NUM) data with MISSING VALUES.
//
// TEST NOTE: like solver_core.h this source is compiled twice -- for the GPU (plspm_nonmetric.hip / plspm_fit.hip) and for the std::thread emulation
// build that the CPU tests drive (tests/hostemu).
//
// Reference: _NonmetricWeights with NaNs (plspm/weights.py:88-98, 107-133), Mode A NaN-aware products (mode.py:35-41),
// Scale.NUM on the finite cells (scale.py:27-30), Config.treat (config.py:314), pairwise-complete loadings
// (outer_model.py:26 via DataFrame.corrwith).  Mode B blocks must be complete (mode.py:55-56).
//
// A score of an observation with missing cells is  sum_p m_p xh_p w_p / sum_p m_p w_p^2  -- the normaliser depends on the
// row's missing pattern, so such a row is not an affine function of the columns and its contribution to a sum over the
// observations is not a second moment.  The incomplete rows are few, though: they are taken out of the uploaded matrix
// (all-zero rows incl. the ones column, so the MFMA Gram is the moment matrix of the COMPLETE rows) and kept in a side
// table [K x P] with their masks.  Every sum over the observations then is
//       (a moment expression on the complete rows)  +  (an explicit weighted sum over the K incomplete rows),
// with the bootstrap weight of row j (its count in the replicate) in ck[j].  Scores of complete rows stay affine in the
// columns (score maps c / k, the layout nm_conv_kernel streams); scores of the K rows are carried explicitly (Yo / Yn).
//
// Notation: xh = standardised MV (mean / population sd over the PRESENT cells of the column, all rows); Rn = moments of xh over
// the complete rows divided by the total weight n (ws.S, ones row/col at index P); om_j = ck_j / n.
#pragma once
#include "solver_core.h"

namespace plspm {

struct MissDesc {
    int raw;                // Scale.RAW-only model: the MVs keep the treated values (scale.py:38-39) = xh * kappa_p
    int K;                  // incomplete rows
    const double* Xk;       // [K*P] stored (upload-shifted) values, 0 in the missing cells
    const double* Mk;       // [K*P] 1 = present, 0 = missing
};

struct NmxExtra {
    double *ck;             // [K]   bootstrap weight of every incomplete row (1 for a plain fit)
    double *alpha, *beta;   // [P]   xh_p = alpha_p * stored_p + beta_p
    double *Xh;             // [K*P] standardised incomplete rows (0 in the missing cells)
    double *Yo, *Yn;        // [K*L] scores of the incomplete rows, previous / current iteration
    double *g, *h;          // [P], [L] current scores of the complete rows: y_l = sum_{p in block l} xh_p g_p + h_l
    double *V;              // [(P+1)*L] Rn . (g, h): V[p,l] = <xh_p, y_l>_c / n, row P: <1, y_l>_c / n
    double *YYc, *YY;       // [L*L] raw second moments of the scores / n: complete rows only, all rows
    double *sy;             // [L]   sum of the scores / n (all rows)
    double *flag;           // [L]   1 when block l has a missing cell in THIS data set (weights.py:88-89)
    double *zz;             // [L]   <z_l, z_l>_c / n
    double *t1, *t2;        // [P]   scratch
};
PLSPM_HD long nmx_extra_doubles(int P, int L, int K) { return (long)K + 2L * P + (long)K * P + 2L * K * L + P + L + (long)(P + 1) * L + 2L * L * L + 3L * L + 2L * P; }
PLSPM_HD long nmx_state_doubles(int P, int L, int n_chol, int K) { return nm_state_doubles(P, L, n_chol) + nmx_extra_doubles(P, L, K); }
PLSPM_HD void nmx_carve(NmxExtra& x, double* base, int P, int L, int K) {
    double* p = base;
    x.ck = p; p += K; x.alpha = p; p += P; x.beta = p; p += P; x.Xh = p; p += (long)K * P; x.Yo = p; p += (long)K * L; x.Yn = p; p += (long)K * L;
    x.g = p; p += P; x.h = p; p += L; x.V = p; p += (long)(P + 1) * L; x.YYc = p; p += L * L; x.YY = p; p += L * L; x.sy = p; p += L;
    x.flag = p; p += L; x.zz = p; p += L; x.t1 = p; p += P; x.t2 = p; p += P;
}

// score maps on the uploaded columns from (g, h): y = sum_p stored_p c_p + k_l
template <class Ex>
PLSPM_HD void nmx_score_map(Ex& ex, const ModelDesc& md, const NmxExtra& x, double* c, double* k) {
    ex.par(md.P, [&](int p) { c[p] = x.alpha[p] * x.g[p]; });
    ex.par(md.L, [&](int l) {
        double s = x.h[l];
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += x.beta[p] * x.g[p];
        k[l] = s;
    });
}

// V, YYc, YY, sy of the CURRENT scores (g, h, Yn)
template <class Ex>
PLSPM_HD void nmx_score_moments(Ex& ex, const ModelDesc& md, const MissDesc& xd, Workspace& ws, NmState& st, NmxExtra& x) {
    const int P = md.P, L = md.L, PS = ws.PS, K = xd.K;
    const double inv_n = 1.0 / st.scal[0];
    ex.par2(P + 1, L, [&](int p, int l) {
        double s = ws.S[P * PS + p] * x.h[l];
        for (int q = md.boff[l]; q < md.boff[l + 1]; ++q) s += ws.S[q * PS + p] * x.g[q];
        x.V[p * L + l] = s;
    });
    ex.par(L * L, [&](int e) {
        const int l = e / L, m = e - l * L;
        double s = x.h[l] * x.V[P * L + m];
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += x.g[p] * x.V[p * L + m];
        x.YYc[e] = s;
        double t = 0.0;
        for (int j = 0; j < K; ++j) t += x.ck[j] * x.Yn[j * L + l] * x.Yn[j * L + m];
        x.YY[e] = s + t * inv_n;
    });
    ex.par(L, [&](int l) {
        double t = 0.0;
        for (int j = 0; j < K; ++j) t += x.ck[j] * x.Yn[j * L + l];
        x.sy[l] = x.V[P * L + l] + t * inv_n;
    });
}

// raw scores of all rows, sum_p m_p xh_p wnum_p / sum_p m_p wden_p^2, then population standardisation over ALL rows -> g, h, Yn
// (mode.py:37-41: wnum == wden == the block weights).  The initial scores (weights.py:86-98) are not standardised and are
// formed on the TREATED columns (config.py:314), which differ from xh by kappa_p = sqrt((f_p-1)/f_p) / sqrt((n-1)/n) where cells
// are missing: wnum = kappa / sqrt(k), wden = 1 / sqrt(k).
template <class Ex>
PLSPM_HD void nmx_new_scores(Ex& ex, const ModelDesc& md, const MissDesc& xd, Workspace& ws, NmState& st, NmxExtra& x, const double* wnum, const double* wden,
                             bool standardise) {
    const int P = md.P, L = md.L, PS = ws.PS, K = xd.K;
    const double inv_n = 1.0 / st.scal[0];
    ex.par2(K > 0 ? K : 1, L, [&](int j, int l) {
        if (j >= K) return;
        double num = 0.0, den = 0.0;
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) { num += x.Xh[j * P + p] * wnum[p]; den += xd.Mk[j * P + p] * wden[p] * wden[p]; }
        x.Yn[j * L + l] = (x.flag[l] != 0.0) ? num / den : num;
    });
    ex.par(L, [&](int l) {
        const int b0 = md.boff[l], b1 = md.boff[l + 1];
        double D = 1.0;
        if (x.flag[l] != 0.0) { D = 0.0; for (int p = b0; p < b1; ++p) D += wden[p] * wden[p]; }
        double s1 = 0.0, s2 = 0.0;
        for (int p = b0; p < b1; ++p) {
            s1 += ws.S[P * PS + p] * wnum[p];
            s2 += wnum[p] * dot_col(ws.S, PS, p, wnum, b0, b1);
        }
        s1 /= D; s2 /= D * D;
        double mean = 0.0, sd = 1.0;
        if (standardise) {
            double k1 = 0.0, k2 = 0.0;
            for (int j = 0; j < K; ++j) { const double y = x.Yn[j * L + l]; k1 += x.ck[j] * y; k2 += x.ck[j] * y * y; }
            mean = s1 + k1 * inv_n;
            sd = sqrt(s2 + k2 * inv_n - mean * mean);
        }
        for (int p = b0; p < b1; ++p) x.g[p] = wnum[p] / (D * sd);
        x.h[l] = -mean / sd;
        ws.dv[l] = mean; ws.wf[l] = sd;
    });
    if (standardise) ex.par2(K > 0 ? K : 1, L, [&](int j, int l) { if (j < K) x.Yn[j * L + l] = (x.Yn[j * L + l] - ws.dv[l]) / ws.wf[l]; });
}

// Mp: packed Gram of the uploaded matrix (incomplete rows zeroed).  x.ck must be filled by the caller.
template <class Ex>
PLSPM_HD void nmx_prepare(Ex& ex, const ModelDesc& md, const MissDesc& xd, Workspace& ws, NmState& st, NmxExtra& x, const double* Mp) {
    const int P = md.P, L = md.L, PS = ws.PS, T = md.T, K = xd.K;
    const int ntile = T * (T + 1) / 2;
    ex.par_chunks64(ntile * 4, Mp, [&](int chunk, int lane, double m) {
        const int tile = chunk >> 2, r = chunk & 3;
        int t, u;
        if (md.tile_tu) { const int tu = md.tile_tu[tile]; t = tu & 255; u = tu >> 8; }
        else { t = 0; int rem = tile; while (rem >= T - t) { rem -= T - t; ++t; } u = t + rem; }
        const int p = 32 * (t >> 1) + (t & 1) + 8 * r + 2 * (lane >> 4);
        const int q = 32 * (u >> 1) + (u & 1) + 2 * (lane & 15);
        if ((t != u || p <= q) && p <= P && q <= P) { ws.S[q * PS + p] = m; ws.S[p * PS + q] = m; }
    });
    const double nc = ws.S[P * PS + P];
    const double nk = ex.sum(K, [&](int j) { return x.ck[j]; });
    const double n = nc + nk, inv_n = 1.0 / n;
    ex.one([&]() { st.scal[0] = n; st.scal[1] = (double)ST_OK; st.scal[2] = 0.0; st.scal[3] = 1.0; st.scal[4] = 0.0; st.scal[5] = nc; st.scal[6] = nk; });
    // column statistics over the present cells of ALL rows (config.py:314 + scale.py:27-30 == population standardisation)
    ex.par(P, [&](int p) {
        double f = nc, s1 = ws.S[P * PS + p], s2 = ws.S[p * PS + p];
        for (int j = 0; j < K; ++j) {
            const double w = x.ck[j] * xd.Mk[j * P + p], v = xd.Xk[j * P + p];
            f += w; s1 += w * v; s2 += w * v * v;
        }
        const double mu = s1 / f, sd = nm_column_sd(s2 / f, mu);                  // (NaN for a column that is constant in this replicate: solver_core.h)
        st.mu[p] = mu; st.sd[p] = sd;
        const double kappa = sqrt((f - 1.0) / f) / sqrt((n - 1.0) / n);   // treated column (config.py:314) / xh; 1 for a complete column
        const double unit = xd.raw ? kappa : 1.0;                       // Scale.RAW iterates on the treated values themselves
        x.alpha[p] = unit / sd; x.beta[p] = -unit * mu / sd;
        x.t1[p] = ws.S[P * PS + p];                                   // raw column sums of the complete rows
        x.t2[p] = xd.raw ? 1.0 : kappa;                               // initial scores use the treated columns
        if (!(f > 1.0) || !(sd > 0.0) || !isfinite(sd)) st.scal[1] = (double)ST_NONFINITE;
    });
    ex.par(K * P, [&](int e) { const int p = e % P; x.Xh[e] = xd.Mk[e] * (x.alpha[p] * xd.Xk[e] + x.beta[p]); });
    // Rn = moments of xh over the complete rows / n (in place; the ones row last)
    ex.par(P, [&](int p) {
        const double ap = x.alpha[p], bp = x.beta[p], sp = x.t1[p];
        for (int q = 0; q < P; ++q) {
            const double v = ap * x.alpha[q] * ws.S[q * PS + p] + ap * x.beta[q] * sp + bp * x.alpha[q] * x.t1[q] + bp * x.beta[q] * nc;
            ws.S[q * PS + p] = v * inv_n;
        }
    });
    ex.par(P, [&](int p) { const double v = (x.alpha[p] * x.t1[p] + x.beta[p] * nc) * inv_n; ws.S[P * PS + p] = v; ws.S[p * PS + P] = v; });
    ex.one([&]() { ws.S[P * PS + P] = nc * inv_n; });
    // which blocks have a missing cell in this data set; Mode B needs a complete block (mode.py:55-56)
    ex.par(L, [&](int l) {
        double miss = 0.0;
        for (int j = 0; j < K; ++j)
            if (x.ck[j] > 0.0) for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) miss += 1.0 - xd.Mk[j * P + p];
        x.flag[l] = miss > 0.0 ? 1.0 : 0.0;
        if (miss > 0.0 && md.mode[l] == MODE_B) st.scal[1] = (double)ST_SINGULAR;
    });
    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {
            if (md.mode[l] == MODE_B) {
                const int b0 = md.boff[l], k = md.boff[l + 1] - b0;
                const bool ok = psd_factor(st.chol + md.chol_off[l], k, [&](double* R) {
                    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) {
                        double s = ws.S[(b0 + r) * PS + b0 + c];
                        for (int j = 0; j < K; ++j) s += x.ck[j] * inv_n * x.Xh[j * P + b0 + r] * x.Xh[j * P + b0 + c];
                        R[r * k + c] = s;
                    }
                });
                if (!ok) st.scal[1] = (double)ST_SINGULAR;
            }
        });
    }
    // initial scores: equal weights 1 / sqrt(k), not standardised (weights.py:86-98)
    ex.par(P, [&](int p) {
        const int l = md.lvof[p];
        ws.w[p] = 1.0 / sqrt((double)(md.boff[l + 1] - md.boff[l]));
        ws.wn[p] = ws.w[p] * x.t2[p];
        st.a_new[p] = ws.w[p]; st.a_old[p] = ws.w[p];
    });
    nmx_new_scores(ex, md, xd, ws, st, x, ws.wn, ws.w, false);
    ex.par(K * L, [&](int e) { x.Yo[e] = x.Yn[e]; });
    nmx_score_map(ex, md, x, st.c_new, st.k_new);
    ex.par(P, [&](int p) { st.c_old[p] = st.c_new[p]; });
    ex.par(L, [&](int l) { st.k_old[l] = st.k_new[l]; });
}

// One iteration (weights.py:107-120); same launch protocol as nm_step.  `partial`: the streaming pass over the uploaded matrix,
// in which every incomplete row is an all-zero row and therefore contributed count * sum_l (|k_old_l| - |k_new_l|)^2.
template <class Ex>
PLSPM_HD bool nmx_step(Ex& ex, const ModelDesc& md, const MissDesc& xd, Workspace& ws, NmState& st, NmxExtra& x, const double* partial, int nparts) {
    const int P = md.P, L = md.L, K = xd.K;
    if (st.scal[3] == 0.0) return false;
    const int iteration = (int)st.scal[2];
    const double n = st.scal[0], inv_n = 1.0 / n, corr2 = n / (n - 1.0);
    if (iteration > 0) {
        const double streamed = ex.sum(nparts, [&](int c) { return partial[c]; });
        const double artefact = ex.sum(L, [&](int l) { const double d = fabs(st.k_old[l]) - fabs(st.k_new[l]); return d * d; });
        const double explicit_rows = ex.sum(K * L, [&](int e) { const double d = fabs(x.Yo[e]) - fabs(x.Yn[e]); return x.ck[e / L] * d * d; });
        const double conv = streamed - st.scal[6] * artefact + explicit_rows;
        // (a NaN criterion is absorbing -- solver_nmg.h nmg_step --: the problem leaves with the record its max_iter + 1 trips would end in)
        const bool never = conv != conv;
        const bool stop = (conv < md.tol) || (iteration > md.max_iter) || never;
        ex.one([&]() {
            st.scal[4] = conv;
            if (stop) { st.scal[3] = 0.0; if ((iteration > md.max_iter || never) && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED; }
            if (never) st.scal[2] = (double)(md.max_iter + 1);
        });
        if (stop) return false;
        ex.par(P, [&](int p) { st.c_old[p] = st.c_new[p]; st.a_old[p] = st.a_new[p]; });
        ex.par(L, [&](int l) { st.k_old[l] = st.k_new[l]; });
        ex.par(K * L, [&](int e) { x.Yo[e] = x.Yn[e]; });
    }
    ex.one([&]() { ws.scal[3] = (double)ST_OK; });
    nmx_score_moments(ex, md, xd, ws, st, x);                                     // of the current scores (Yn == Yo here)
    ex.par(L * L, [&](int e) { const int l = e / L, m = e - l * L; ws.G[e] = x.YY[e] - x.sy[l] * x.sy[m]; });
    inner_weights(ex, md, ws, corr2, x.YY);
    // inner estimates: complete rows through V . E, incomplete rows explicitly (Yn <- Z_k; Yo keeps the scores)
    ex.par2(K > 0 ? K : 1, L, [&](int j, int l) {
        if (j >= K) return;
        double s = 0.0;
        for (int m = 0; m < L; ++m) s += x.Yo[j * L + m] * ws.E[m * L + l];
        x.Yn[j * L + l] = s;
    });
    ex.par(L, [&](int l) {                                                         // <z_l, z_l>_c / n
        double s = 0.0;
        for (int m = 0; m < L; ++m) {
            const double em = ws.E[m * L + l];
            if (em == 0.0) continue;
            double t = 0.0;
            for (int m2 = 0; m2 < L; ++m2) t += x.YYc[m * L + m2] * ws.E[m2 * L + l];
            s += em * t;
        }
        x.zz[l] = s;
    });
    ex.par(P, [&](int p) {
        const int l = md.lvof[p];
        double num = 0.0;
        for (int m = 0; m < L; ++m) num += x.V[p * L + m] * ws.E[m * L + l];
        double kn = 0.0, kd = 0.0;
        for (int j = 0; j < K; ++j) {
            const double z = x.Yn[j * L + l], w = x.ck[j];
            kn += w * x.Xh[j * P + p] * z;
            kd += w * xd.Mk[j * P + p] * z * z;
        }
        num += kn * inv_n;
        ws.wn[p] = (md.mode[l] == MODE_A) ? num / (x.zz[l] + kd * inv_n) : num;   // mode.py:35-36 / 38
    });
    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {                                                     // Mode B: lstsq(X_b, z) on the all-row normal equations (mode.py:58)
            if (md.mode[l] == MODE_B) { const int b0 = md.boff[l]; psd_solve(st.chol + md.chol_off[l], md.boff[l + 1] - b0, ws.wn + b0); }
        });
    }
    ex.par(P, [&](int p) { st.a_new[p] = ws.wn[p]; });
    nmx_new_scores(ex, md, xd, ws, st, x, ws.wn, ws.wn, true);
    nmx_score_map(ex, md, x, st.c_new, st.k_new);
    ex.one([&]() { st.scal[2] = (double)(iteration + 1); if (ws.scal[3] != (double)ST_OK && st.scal[1] == (double)ST_OK) st.scal[1] = ws.scal[3]; });
    // The stop-rule value of THIS step bounded without a pass over the observations (solver_core.h nm_step has the argument): the
    // complete rows contribute at most  sum_i c_i (y_old - y_new)^2 = n sum_l [d_g' Rn_bb d_g + 2 d_h (Rn_1b . d_g) + d_h^2 Rn_11]  with
    // (d_g, d_h) the change of the score map on the standardised columns (the old map recovered from c_old / k_old), the incomplete
    // rows their exact term.  Below the tolerance the problem stops here, with the reference's iteration count.
    if (iteration >= 1) {
        const int PS = ws.PS;
        ex.par(P, [&](int p) { ws.cv[p] = x.g[p] - st.c_old[p] / x.alpha[p]; });
        ex.par(L, [&](int l) {
            double h_old = st.k_old[l];
            for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) h_old -= x.beta[p] * (st.c_old[p] / x.alpha[p]);
            ws.a[l] = x.h[l] - h_old;
        });
        ex.par(P, [&](int p) {
            const int l = md.lvof[p];
            ws.dv[p] = ws.cv[p] * (dot_col(ws.S, PS, p, ws.cv, md.boff[l], md.boff[l + 1]) + 2.0 * ws.a[l] * ws.S[P * PS + p]);
        });
        const double quad = ex.sum(P, [&](int p) { return ws.dv[p]; }) + ex.sum(L, [&](int l) { return ws.a[l] * ws.a[l] * ws.S[P * PS + P]; });
        const double explicit_rows = ex.sum(K * L, [&](int e) { const double d = fabs(x.Yo[e]) - fabs(x.Yn[e]); return x.ck[e / L] * d * d; });
        const double ub = n * quad + explicit_rows;
        if (ub < md.tol * (1.0 - 1e-9)) {
            ex.one([&]() {
                st.scal[4] = ub; st.scal[3] = 0.0;
                if (iteration + 1 > md.max_iter && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED;
            });
            return false;
        }
    }
    return true;
}

// After the loop (weights.py:122-133, inner_model.py, outer_model.py:26-27): weight factors from the rows that are complete in
// EVERY block (a NaN anywhere makes DataFrame.dot return NaN for the whole row), inner model on the all-row score covariance,
// pairwise-complete correlations for the (cross-)loadings.  out.score_w / score_c describe the complete rows only; the caller
// patches the incomplete rows' scores from `Yn`.
template <class Ex>
PLSPM_HD void nmx_finish(Ex& ex, const ModelDesc& md, const MissDesc& xd, Workspace& ws, NmState& st, NmxExtra& x, const FitOutputs& out) {
    const int P = md.P, L = md.L, PS = ws.PS, K = xd.K, ne = md.n_eff;
    const double n = st.scal[0], nc = st.scal[5];
    ex.one([&]() { ws.scal[3] = st.scal[1]; ws.scal[1] = n; });
    nmx_score_moments(ex, md, xd, ws, st, x);
    ex.par(L * L, [&](int e) { const int l = e / L, m = e - l * L; ws.Cs[e] = x.YY[e] - x.sy[l] * x.sy[m]; });
    inner_model_effects(ex, md, ws);
    ex.par(L, [&](int l) {                                                         // 1 / (std1(X_b w over the complete rows) / correction)
        const int b0 = md.boff[l], b1 = md.boff[l + 1];
        double s1 = 0.0, s2 = 0.0;
        for (int p = b0; p < b1; ++p) { s1 += ws.S[P * PS + p] * st.a_new[p]; s2 += st.a_new[p] * dot_col(ws.S, PS, p, st.a_new, b0, b1); }
        const double var1 = (s2 * n - (s1 * n) * (s1 * n) / nc) / (nc - 1.0);
        ws.wf[l] = sqrt(n / (n - 1.0)) / sqrt(var1);
    });
    ex.par(P, [&](int p) { ws.w[p] = st.a_new[p] * ws.wf[md.lvof[p]]; });
    // pairwise-complete Pearson correlation of MV p with score l
    ex.par2(P, L, [&](int p, int l) {
        double np_ = nc, sx = ws.S[P * PS + p] * n, sxx = ws.S[p * PS + p] * n, sy = x.V[P * L + l] * n, syy = x.YYc[l * L + l] * n, sxy = x.V[p * L + l] * n;
        for (int j = 0; j < K; ++j) {
            const double w = x.ck[j] * xd.Mk[j * P + p];
            if (w == 0.0) continue;
            const double xv = x.Xh[j * P + p], yv = x.Yn[j * L + l];
            np_ += w; sx += w * xv; sxx += w * xv * xv; sy += w * yv; syy += w * yv * yv; sxy += w * xv * yv;
        }
        ws.V[p * L + l] = (sxy - sx * sy / np_) / sqrt((sxx - sx * sx / np_) * (syy - sy * sy / np_));
    });
    ex.par(P, [&](int p) {
        const int l = md.lvof[p];
        const double ld = ws.V[p * L + l];
        if (out.row) { out.row[p] = ws.w[p]; out.row[P + L + 2 * ne + p] = ld; }
        if (out.weights) out.weights[p] = ws.w[p];
        if (out.loadings) out.loadings[p] = ld;
        if (out.crossloadings) for (int m = 0; m < L; ++m) out.crossloadings[p * L + m] = ws.V[p * L + m];
        if (out.score_w) out.score_w[p] = st.c_new[p];
        if (out.mean) out.mean[p] = 0.0;
        if (out.cov) for (int q = 0; q < P; ++q) out.cov[p * P + q] = (ws.S[q * PS + p] - ws.S[P * PS + p] * ws.S[P * PS + q] * n / nc) * n / nc;
    });
    ex.par(L, [&](int l) {
        if (out.row) out.row[P + l] = ws.r2[l];
        if (out.r2) out.r2[l] = ws.r2[l];
        if (out.sign) out.sign[l] = 1;
        if (out.score_c) out.score_c[l] = st.k_new[l];
    });
    ex.par(L * L, [&](int e) {
        if (out.path_coef) out.path_coef[e] = ws.Bm[e];
        if (out.lv_cov) out.lv_cov[e] = ws.Cs[e];
    });
    ex.par(ne, [&](int e) {
        const int idx = md.eff_to[e] * L + md.eff_from[e];
        if (out.row) { out.row[P + L + e] = ws.Bm[idx] + ws.Ind[idx]; out.row[P + L + ne + e] = ws.Bm[idx]; }
        if (out.indirect) out.indirect[e] = ws.Ind[idx];
    });
    const bool bad = ex.any(P + L, [&](int e) {
        if (e < P) return !(isfinite(ws.w[e]) && isfinite(ws.V[e * L + md.lvof[e]]));
        return !isfinite(ws.r2[e - P]);
    });
    const int iteration = (int)st.scal[2];
    ex.one([&]() {
        int s = (int)ws.scal[3];
        if (s == ST_OK && bad) s = ST_NONFINITE;
        if (out.status) *out.status = s;
        if (out.iters) *out.iters = iteration;
        if (out.row) { out.row[2 * P + L + 2 * ne] = (double)s; out.row[2 * P + L + 2 * ne + 1] = (double)iteration; }
    });
}

}  // namespace plspm
