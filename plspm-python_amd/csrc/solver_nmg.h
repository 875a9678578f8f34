// Non-metric solver with optimal scaling (Scale.ORD / Scale.NOM mixed with NUM / RAW) on second moments.
//
// Reference: _NonmetricWeights (plspm/weights.py:73-154), the Scale operators (plspm/scale.py:22-89: NUM, RAW, ORD with the
// two-direction monotone pooling `_ordinalize`, NOM), the Mode-B correction get_Z_for_mode_b (weights.py:135-145), the
// non-metric outer steps (mode.py:31-42, 54-61) and Config.treat's rank / dummy coding (config.py:314-318, util.py:80-95).
//
// Device columns ("aug columns", Q of them + a ones column): a NUM / RAW manifest variable is one column (its raw values);
// an ORD / NOM manifest variable with C categories is C indicator columns (category codes in rank order).  Categorical
// models are uploaded UNSHIFTED, so Mn = M / n holds raw second moments and Mn[.., Q] the column means.  Every quantified
// manifest variable is an affine function of its own columns,  MV_p = sum_j tq_j col_j + tc_p,  every LV score an affine
// function of its MVs,  y_l = sum_p a_p MV_p + kk_l;  group means of z by category, the pooled quantifications, the Mode-B
// betas, outer weights and normalisations are all inner products through Mn.  As in the NUM / RAW solver (solver_core.h,
// nm_*) only the score-based stop rule (weights.py:120) needs the observations: nm_conv_kernel evaluates it from the two
// score maps over the aug columns that every step leaves in the NmState-compatible head of the state.
//
// State = [NmState layout for P := Q, n_chol := 0] ++ NmgExtra.  One cooperating group per problem; serial sections
// (pooling over <= cmax categories, k x k block solves) run on thread 0.
#pragma once
#include "solver_core.h"

namespace plspm {

enum { KIND_NUM = 0, KIND_ORD = 1, KIND_NOM = 2 };

struct CatDesc {
    int Pm, cmax, kmv;          // logical MVs, max categories of an MV, max MVs of a block
    const int* mv_off;          // [Pm+1] aug-column range of every MV
    const int* mv_kind;         // [Pm]
    const int* lmv_off;         // [L+1]  MV range of every LV (MVs are grouped by LV, path order)
};

struct NmgExtra {
    double *tq, *tc;            // [Q], [Pm]   current quantification: MV_p = sum_j tq_j col_j + tc_p
    double *akk;                // [L]         score constants kk_l of the CURRENT (old) scores
    double *V, *MZ;             // [(Q+1)*L]   Mn . score-coefficients, Mn . z-coefficients (row Q: means)
    double *YY;                 // [L*L]       raw second moments of the scores
    double *mvm;                // [Pm*L + Pm] <MV_p, z_l> raw for the own LV (first Pm) + scratch
    double *cm, *cf, *cs;       // [cmax] category means of z, category frequencies, chosen quantification
    double *inc, *dec, *gsum, *gf;   // [cmax] pooled values of the two directions; pooling scratch (group sums / weights)
    double *Bm, *Fm, *Vm, *beta, *rhs;   // [kmv*kmv] block moment matrix, its factor copy, eigenvector scratch of the minimum-norm fallback; [kmv], [kmv]
    int* grp;                   // [cmax] (stored in a double-aligned slot)
    double* pq;                 // [Pm * 8 * cmax] per-MV scratch of the parallel quantification (Mode A blocks)
    // Models whose device columns are ALL 0/1 indicators (every MV ORD / NOM): the moment matrix n Mn is a matrix of co-occurrence counts
    // <= N.  With N <= 65,535 the caller may supply room for a uint16 copy (row pitch ld16, a multiple of 4): the product V = Mn c of every
    // step -- the one pass that streams the whole matrix -- then moves a quarter of the bytes.  (double)count * (1/n) IS the entry of Mn
    // (nmg_prepare forms it the same way), so the step is bitwise the same.  Not part of the carved state.
    unsigned short* k16 = nullptr;
    int ld16 = 0;
};
PLSPM_HD long nmg_extra_doubles(int Q, int Pm, int L, int cmax, int kmv) {
    return (long)Q + Pm + L + 2L * (Q + 1) * L + (long)L * L + ((long)Pm * L + Pm) + 7L * cmax + 3L * kmv * kmv + 2L * kmv + cmax + 8 + 8L * Pm * cmax;
}
PLSPM_HD void nmg_carve(NmgExtra& x, double* base, int Q, int Pm, int L, int cmax, int kmv) {
    double* p = base;
    x.tq = p; p += Q; x.tc = p; p += Pm; x.akk = p; p += L;
    x.V = p; p += (long)(Q + 1) * L; x.MZ = p; p += (long)(Q + 1) * L; x.YY = p; p += (long)L * L;
    x.mvm = p; p += (long)Pm * L + Pm;
    x.cm = p; p += cmax; x.cf = p; p += cmax; x.cs = p; p += cmax;
    x.inc = p; p += cmax; x.dec = p; p += cmax; x.gsum = p; p += cmax; x.gf = p; p += cmax;
    x.Bm = p; p += (long)kmv * kmv; x.Fm = p; p += (long)kmv * kmv; x.Vm = p; p += (long)kmv * kmv; x.beta = p; p += kmv; x.rhs = p; p += kmv;
    x.grp = reinterpret_cast<int*>(p); p += cmax + 8;
    x.pq = p;
}
PLSPM_HD long nmg_state_doubles(int Q, int Pm, int L, int cmax, int kmv) { return nm_state_doubles(Q, L, 0) + nmg_extra_doubles(Q, Pm, L, cmax, kmv); }
// The same arrays with everything but the two (Q+1) x L products in a caller-supplied FAST area (the device kernel: LDS).  The quantifications
// and score constants persist between launches: the caller copies them in from / out to their slots of the global layout (`g`).
PLSPM_HD long nmg_fast_doubles(int Q, int Pm, int L, int cmax, int kmv) { return nmg_extra_doubles(Q, Pm, L, cmax, kmv) - 2L * (Q + 1) * L; }
PLSPM_HD long nmg_persistent_doubles(int Q, int Pm, int L) { return (long)Q + Pm + L; }          // tq | tc | akk, contiguous at the head of both layouts
PLSPM_HD void nmg_carve_fast(NmgExtra& x, NmgExtra& g, double* gbase, double* fast, int Q, int Pm, int L, int cmax, int kmv) {
    nmg_carve(g, gbase, Q, Pm, L, cmax, kmv);
    double* p = fast;
    x.tq = p; p += Q; x.tc = p; p += Pm; x.akk = p; p += L;
    x.V = g.V; x.MZ = g.MZ;
    x.YY = p; p += (long)L * L;
    x.mvm = p; p += (long)Pm * L + Pm;
    x.cm = p; p += cmax; x.cf = p; p += cmax; x.cs = p; p += cmax;
    x.inc = p; p += cmax; x.dec = p; p += cmax; x.gsum = p; p += cmax; x.gf = p; p += cmax;
    x.Bm = p; p += (long)kmv * kmv; x.Fm = p; p += (long)kmv * kmv; x.Vm = p; p += (long)kmv * kmv; x.beta = p; p += kmv; x.rhs = p; p += kmv;
    x.grp = reinterpret_cast<int*>(p); p += cmax + 8;
    x.pq = p;
}

// score map over the aug columns of coefficient set (a, kk) with the current quantification: c_j = a_mv(j) tq_j, k_l = kk_l + sum a_p tc_p
template <class Ex>
PLSPM_HD void nmg_score_map(Ex& ex, const ModelDesc& md, const CatDesc& cd, const NmgExtra& x, const double* a, const double* kk, double* c, double* k) {
    ex.par(cd.Pm, [&](int p) { for (int j = cd.mv_off[p]; j < cd.mv_off[p + 1]; ++j) c[j] = a[p] * x.tq[j]; });
    ex.par(md.L, [&](int l) {
        double s = kk ? kk[l] : 0.0;
        for (int p = cd.lmv_off[l]; p < cd.lmv_off[l + 1]; ++p) s += a[p] * x.tc[p];
        k[l] = s;
    });
}

// V[j,m] = <col_j, y_m> for j = 0..Q (row Q: mean of y_m), from the score map (c, k)
template <class Ex>
PLSPM_HD void nmg_apply(Ex& ex, const ModelDesc& md, const double* Mn, int LD, const double* c, const double* k, double* V, const unsigned short* k16 = nullptr, int ld16 = 0,
                        double inv_n = 0.0) {
    const int Q = md.P, L = md.L;
    if (k16) {
        // four columns per item (one 8-byte load per row of the block), eight rows in flight
        ex.par2((Q + 4) / 4, L, [&](int j4, int m) {
            const int j = 4 * j4;
            double s[4];
            for (int u = 0; u < 4; ++u) s[u] = (j + u <= Q) ? Mn[Q * LD + j + u] * k[m] : 0.0;
            const int q1 = md.boff[m + 1];
            for (int q = md.boff[m]; q < q1; q += 8) {
                unsigned long long w[8];
                for (int t = 0; t < 8; ++t) w[t] = (q + t < q1) ? *reinterpret_cast<const unsigned long long*>(k16 + (long)(q + t) * ld16 + j) : 0ull;
                for (int t = 0; t < 8 && q + t < q1; ++t) {
                    const double cq = c[q + t];
                    for (int u = 0; u < 4; ++u) s[u] += ((double)(unsigned)((w[t] >> (16 * u)) & 0xffffull) * inv_n) * cq;
                }
            }
            for (int u = 0; u < 4; ++u) if (j + u <= Q) V[(j + u) * L + m] = s[u];
        });
        return;
    }
    ex.par2(Q + 1, L, [&](int j, int m) {
        double s = Mn[Q * LD + j] * k[m];                               // <col_j, 1> k_m   (Mn[Q][Q] = 1)
        // (eight loads in flight per thread: the pass streams all of Mn once and is bound by memory latency x concurrency)
        const int q1 = md.boff[m + 1];
        int q = md.boff[m];
        for (; q + 8 <= q1; q += 8) {
            double v[8];
            for (int u = 0; u < 8; ++u) v[u] = Mn[(q + u) * LD + j];
            for (int u = 0; u < 8; ++u) s += v[u] * c[q + u];
        }
        for (; q < q1; ++q) s += Mn[q * LD + j] * c[q];
        V[j * L + m] = s;
    });
}

// raw moment <MV_p, u> for a variable u given by its column moments Mu[j] = <col_j, u> and its mean
PLSPM_HD double nmg_mv_moment(const CatDesc& cd, const NmgExtra& x, int p, const double* Mu, int stride, double mean_u) {
    double s = x.tc[p] * mean_u;
    const int j1 = cd.mv_off[p + 1];
    int j = cd.mv_off[p];
    while (j < j1) {                                                    // (up to eight moment loads in flight, see nmg_mv_mv)
        double v[8];
        for (int u = 0; u < 8; ++u) v[u] = (j + u < j1) ? Mu[(j + u) * stride] : 0.0;
        for (int u = 0; u < 8 && j + u < j1; ++u) s += x.tq[j + u] * v[u];
        j += 8;
    }
    return s;
}
// raw moment <MV_p, MV_q>
// (the moment loads of a row of the sub-block are issued together, up to eight at a time: the entries come from global memory and a
//  dependent chain of single loads costs a memory round trip each; same terms in the same order)
PLSPM_HD double nmg_mv_mv(const CatDesc& cd, const NmgExtra& x, const double* Mn, int LD, int Q, int p, int q) {
    double s = 0.0, mq = x.tc[q];
    const int j0 = cd.mv_off[q], j1 = cd.mv_off[q + 1];
    for (int j = j0; j < j1; ++j) mq += x.tq[j] * Mn[Q * LD + j];          // mean of MV_q
    for (int i = cd.mv_off[p]; i < cd.mv_off[p + 1]; ++i) {
        double t = 0.0;
        const double mi = Mn[Q * LD + i];
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
            double v[8];
            for (int u = 0; u < 8; ++u) v[u] = Mn[(j + u) * LD + i];
            for (int u = 0; u < 8; ++u) t += v[u] * x.tq[j + u];
        }
        if (j < j1) {
            double v[8];
            for (int u = 0; u < 8; ++u) v[u] = (j + u < j1) ? Mn[(j + u) * LD + i] : 0.0;
            for (int u = 0; u < 8; ++u) if (j + u < j1) t += v[u] * x.tq[j + u];
        }
        s += x.tq[i] * (t + mi * x.tc[q]);
    }
    return s + x.tc[p] * mq;
}

// scale.py:54-66 on (category mean, frequency) pairs: pool adjacent categories (first violation from the left, restart) until
// the means are monotone for `sign`; out[c] = pooled value of category c; returns the population variance of the result.
PLSPM_HD double nmg_ordinalize(const double* m, const double* f, int C, double sign, double* out, int* grp, double* gsum, double* gf) {
    // groups are runs of categories; gsum / gf: weighted sum and weight per group (arrays of length C, in place of `out` scratch)
    int ng = C;
    for (int c = 0; c < C; ++c) { grp[c] = c; gsum[c] = m[c] * f[c]; gf[c] = f[c]; }
    while (true) {
        const int before = ng;
        for (int g = 0; g + 1 < ng; ++g) {
            const double a = gsum[g] / gf[g], b = gsum[g + 1] / gf[g + 1], d = a - b;
            const double sg = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0);
            if (sg == sign) {
                gsum[g + 1] += gsum[g]; gf[g + 1] += gf[g];
                for (int h = g; h + 1 < ng; ++h) { gsum[h] = gsum[h + 1]; gf[h] = gf[h + 1]; }
                for (int c = 0; c < C; ++c) if (grp[c] > g) --grp[c];
                --ng;
                break;
            }
        }
        if (ng == 1 || ng == before) break;
    }
    double mean = 0.0, ss = 0.0;
    for (int c = 0; c < C; ++c) { const double v = gsum[grp[c]] / gf[grp[c]]; out[c] = v; mean += f[c] * v; ss += f[c] * v * v; }
    return ss - mean * mean;
}

// Standard deviation of a quantified MV from its second moment and mean.  A quantification whose categories all carry the SAME value (every category mean of z
// equal -- e.g. a two-category item whose category means tie exactly in the first trip of the centroid scheme; or ordinal pooling down to one group in both
// directions) is a constant column in the reference: util.treat_numpy divides 0 by 0, the scores are NaN from there on, the stop rule never holds and the
// estimate raises (weights.py:120-127) -- a bootstrap replicate is dropped.  On second moments the same variance is `ss - mean^2` of two equal numbers, i.e.
// rounding noise of either sign: a negative one gave NaN as well, a positive one a standardisation of noise and a replicate that "converged".  A variance below
// 1e-12 of the second moment (no quantification of real data comes near: the category means of a standardised z spread by percents) is that constant column.
PLSPM_HD double nmg_quant_sd(double ss, double mean) {
    const double var = ss - mean * mean;
    return sqrt((var > 1e-12 * ss) ? var : -1.0);
}

// Quantification of one ORD / NOM manifest variable from the category means of (corrected) z (scale.py:42-89): compact to the
// categories present in this problem (frequency > 0), pool monotonically (ORD, both directions) or keep the means (NOM),
// population-standardise, scatter back into tq.  cm / cf: [C] category means and frequencies (overwritten); scratch: 5 arrays of C
// doubles + C ints.
PLSPM_HD void nmg_quantify_mv(int kind, int C, const double* freq_all, double* cm, double* cf, double* cs, double* inc, double* dec, double* gsum, double* gf,
                              int* grp, double* tq_out) {
    int Cp = 0;
    for (int c = 0; c < C; ++c) if (cf[c] > 0.0) { cm[Cp] = cm[c]; cf[Cp] = cf[c]; ++Cp; }
    double mean = 0.0, ss = 0.0;
    if (kind == KIND_ORD) {
        const double v_inc = nmg_ordinalize(cm, cf, Cp, 1.0, inc, grp, gsum, gf);
        const double v_dec = nmg_ordinalize(cm, cf, Cp, -1.0, dec, grp, gsum, gf);
        if (v_inc < v_dec) { for (int c = 0; c < Cp; ++c) cs[c] = -dec[c]; }                      // -x_quant_decr (scale.py:74)
        else { for (int c = 0; c < Cp; ++c) cs[c] = inc[c]; }
    } else {
        for (int c = 0; c < Cp; ++c) cs[c] = cm[c];                                               // NOM (scale.py:87)
    }
    for (int c = 0; c < Cp; ++c) { mean += cf[c] * cs[c]; ss += cf[c] * cs[c] * cs[c]; }
    const double sd = nmg_quant_sd(ss, mean);                          // treat_numpy(.) * correction == population standardisation
    int at = 0;
    for (int c = 0; c < C; ++c) {
        if (freq_all[c] > 0.0) { tq_out[c] = (cs[at] - mean) / sd; ++at; }
        else tq_out[c] = 0.0;
    }
}

// packed scatter -> Mn (raw second moments / n, ones row/column = column means), initial quantification and scores (weights.py:82-98)
template <class Ex>
PLSPM_HD void nmg_prepare(Ex& ex, const ModelDesc& md, const CatDesc& cd, Workspace& ws, NmState& st, NmgExtra& x, const double* Mp) {
    const int Q = md.P, L = md.L, LD = ws.PS, T = md.T;
    const int ntile = T * (T + 1) / 2;
    // n = <1, 1> first (one uniform load): the scatter then writes the scaled entries -- and the uint16 copy of the counts -- in the same
    // pass (a separate scaling pass re-read and re-wrote the whole square: 1.4 MB of traffic per problem)
    // (Mp null: the uint16 count matrix is complete already -- the int8 product wrote it, nmg_kernel<4> mirrored it: only the pitch padding and the state are left)
    const double n = Mp ? Mp[packed_index(T, Q, Q)] : (double)x.k16[(long)Q * x.ld16 + Q], inv_n = 1.0 / n;
    if (Mp) ex.par_chunks64(ntile * 4, Mp, [&](int chunk, int lane, double m) {
        const int tile = chunk >> 2, r = chunk & 3;
        int t, u;
        if (md.tile_tu) { const int tu = md.tile_tu[tile]; t = tu & 255; u = tu >> 8; }
        else { t = 0; int rem = tile; while (rem >= T - t) { rem -= T - t; ++t; } u = t + rem; }
        const int p = 32 * (t >> 1) + (t & 1) + 8 * r + 2 * (lane >> 4);
        const int q = 32 * (u >> 1) + (u & 1) + 2 * (lane & 15);
        if ((t != u || p <= q) && p <= Q && q <= Q) {
            const double v = m * inv_n;
            if (ws.S) { ws.S[q * LD + p] = v; ws.S[p * LD + q] = v; }       // (null: the uint16 counts are all the caller's iteration reads -- the wave step, kernels_nmw.h)
            if (x.k16) { const unsigned short h = (unsigned short)(m + 0.5); x.k16[(long)q * x.ld16 + p] = h; x.k16[(long)p * x.ld16 + q] = h; }       // (integers: exact)
        }
    });
    if (x.k16 && x.ld16 > Q + 1) {                         // pitch padding zeroed: the product reads whole 8-byte groups
        const int pad = x.ld16 - (Q + 1);
        ex.par((Q + 1) * pad, [&](int e) { const int q = e / pad, c = e - q * pad; x.k16[(long)q * x.ld16 + Q + 1 + c] = (unsigned short)0; });
    }
    ex.one([&]() { st.scal[0] = n; st.scal[1] = (double)ST_OK; st.scal[2] = 0.0; st.scal[3] = 1.0; st.scal[4] = 0.0; });
    const double* Mn = ws.S;
    // initial "treated" values (config.py:314-318): NUM / RAW population-standardised, ORD / NOM rank codes 1..C
    ex.par(cd.Pm, [&](int p) {
        const int j0 = cd.mv_off[p], C = cd.mv_off[p + 1] - j0;
        if (cd.mv_kind[p] == KIND_NUM) {
            const double mu = Mn[Q * LD + j0], sd = nm_column_sd(Mn[j0 * LD + j0], mu);      // (NaN for a column that is constant in this replicate: solver_core.h)
            x.tq[j0] = 1.0 / sd; x.tc[p] = -mu / sd;
        } else {
            // rank codes 1..C' over the categories PRESENT in this problem (a bootstrap replicate may miss some: util.rank ranks
            // the values that occur, util.py:80-86); absent categories have an all-zero indicator column, any coefficient does
            int code = 0;
            for (int c = 0; c < C; ++c) {
                const bool present = Mn ? Mn[Q * LD + j0 + c] > 0.0 : x.k16[(long)Q * x.ld16 + j0 + c] != 0;      // (category count > 0)
                if (present) ++code;
                x.tq[j0 + c] = (double)code;
            }
            x.tc[p] = 0.0;
        }
    });
    ex.par(cd.Pm, [&](int p) {
        int l = 0;
        while (p >= cd.lmv_off[l + 1]) ++l;
        st.a_old[p] = 1.0 / sqrt((double)(cd.lmv_off[l + 1] - cd.lmv_off[l]));
        st.a_new[p] = st.a_old[p];
    });
    ex.par(L, [&](int l) { x.akk[l] = 0.0; });
    nmg_score_map(ex, md, cd, x, st.a_old, x.akk, st.c_old, st.k_old);
    nmg_score_map(ex, md, cd, x, st.a_new, x.akk, st.c_new, st.k_new);
}

// One iteration (weights.py:107-120).  Same protocol as nm_step: decide on the previous convergence value first.
template <class Ex>
PLSPM_HD bool nmg_step(Ex& ex, const ModelDesc& md, const CatDesc& cd, Workspace& ws, NmState& st, NmgExtra& x, const double* partial, int nparts) {
    const int Q = md.P, L = md.L, LD = ws.PS, Pm = cd.Pm;
    const double* Mn = ws.S;
    if (st.scal[3] == 0.0) return false;
    const int iteration = (int)st.scal[2];
    if (iteration > 0) {
        const double conv = ex.sum(nparts, [&](int c) { return partial[c]; });
        // A criterion that is NaN stays NaN: a score y_l that is not finite makes its own inner weights e_l. (a sign, a correlation, a regression on y_l) and with them z_l, the
        // block's quantifications and the next y_l not finite again, in the reference as here -- "could not converge after max_iter + 1 iterations" (weights.py:183-186) is
        // certain, so the problem leaves now with that record (status, iteration count) instead of spinning max_iter more trips: the second stage of a higher order construct
        // whose first stage failed arrives with NaN moments (solver_hoc.h) -- five of 5,000 ordinal mobi replicates kept ~90 trips of 0.14 ms alive, 13 of the call's 46 ms.
        const bool never = conv != conv;
        const bool stop = (conv < md.tol) || (iteration > md.max_iter) || never;
        ex.one([&]() {
            st.scal[4] = conv;
            if (stop) { st.scal[3] = 0.0; if ((iteration > md.max_iter || never) && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED; }
            if (never) st.scal[2] = (double)(md.max_iter + 1);
        });
        if (stop) return false;
        ex.par(Pm, [&](int p) { st.a_old[p] = st.a_new[p]; });
        ex.par(Q, [&](int j) { st.c_old[j] = st.c_new[j]; });
        ex.par(L, [&](int l) { st.k_old[l] = st.k_new[l]; });
    }
    const double n = st.scal[0], corr2 = n / (n - 1.0);
    ex.one([&]() { ws.scal[3] = (double)ST_OK; });                                // the small workspace does not survive between launches
    ex.mark(20);
    // scores' moments: V = Mn . score maps, YY raw, means, covariance
    nmg_apply(ex, md, Mn, LD, st.c_old, st.k_old, x.V, x.k16, x.ld16, 1.0 / n);
    ex.mark(21);
    ex.par(L * L, [&](int e) {
        const int l = e / L, m = e - l * L;
        double s = st.k_old[l] * x.V[Q * L + m];
        const int j1 = md.boff[l + 1];
        int j = md.boff[l];
        for (; j + 8 <= j1; j += 8) {                                             // (eight independent loads per trip, see nmg_apply)
            double v[8], c[8];
            for (int u = 0; u < 8; ++u) { v[u] = x.V[(j + u) * L + m]; c[u] = st.c_old[j + u]; }
            for (int u = 0; u < 8; ++u) s += c[u] * v[u];
        }
        for (; j < j1; ++j) s += st.c_old[j] * x.V[j * L + m];
        x.YY[e] = s;
    });
    ex.par(L * L, [&](int e) { const int l = e / L, m = e - l * L; ws.G[e] = x.YY[e] - x.V[Q * L + l] * x.V[Q * L + m]; });
    ex.mark(22);
    inner_weights(ex, md, ws, corr2, x.YY);
    ex.mark(23);
    // MZ[j,l] = <col_j, z_l>, row Q = mean(z_l);  ws.a[l] = <z_l, z_l> raw
    ex.par2(Q + 1, L, [&](int j, int l) {
        double s = 0.0;
        for (int m = 0; m < L; m += 8) {                                          // (the row of V in one go)
            double v[8];
            for (int u = 0; u < 8; ++u) v[u] = (m + u < L) ? x.V[j * L + m + u] : 0.0;
            for (int u = 0; u < 8 && m + u < L; ++u) s += v[u] * ws.E[(m + u) * L + l];
        }
        x.MZ[j * L + l] = s;
    });
    ex.par(L, [&](int l) {
        double s = 0.0;
        for (int m = 0; m < L; ++m) {
            const double em = ws.E[m * L + l];
            if (em == 0.0) continue;
            double t = 0.0;
            for (int m2 = 0; m2 < L; ++m2) t += x.YY[m * L + m2] * ws.E[m2 * L + l];
            s += em * t;
        }
        ws.a[l] = s;
    });
    ex.mark(24);
    // Quantification (weights.py:112-115).  Without the Mode-B correction (Mode A blocks, single-MV blocks) the MVs of a block
    // only read z_l, so ALL of them are quantified at once, one thread per MV with its own scratch; Mode-B blocks go MV by MV
    // below (Gauss-Seidel: the correction of MV j uses the block's MVs as updated so far, weights.py:143-145).
    ex.par(Pm, [&](int p) {
        const int kind = cd.mv_kind[p];
        if (kind == KIND_NUM) return;
        int l = 0;
        while (p >= cd.lmv_off[l + 1]) ++l;
        if (md.mode[l] == MODE_B && cd.lmv_off[l + 1] - cd.lmv_off[l] > 1) return;
        const int j0 = cd.mv_off[p], C = cd.mv_off[p + 1] - j0, cm_ = cd.cmax;
        double* w = x.pq + (long)p * 8 * cm_;
        double *cm = w, *cf = w + cm_, *cs = w + 2 * cm_, *inc = w + 3 * cm_, *dec = w + 4 * cm_, *gsum = w + 5 * cm_, *gf = w + 6 * cm_;
        int* grp = reinterpret_cast<int*>(w + 7 * cm_);
        for (int c0 = 0; c0 < C; c0 += 8) {                                       // (category counts and sums: the loads first, then the divisions)
            double f[8], z[8];
            for (int u = 0; u < 8; ++u) { const bool in = c0 + u < C; f[u] = in ? Mn[Q * LD + j0 + c0 + u] : 0.0; z[u] = in ? x.MZ[(j0 + c0 + u) * L + l] : 0.0; }
            for (int u = 0; u < 8 && c0 + u < C; ++u) { cf[c0 + u] = f[u]; cm[c0 + u] = (f[u] > 0.0) ? z[u] / f[u] : 0.0; }
        }
        nmg_quantify_mv(kind, C, Mn + Q * LD + j0, cm, cf, cs, inc, dec, gsum, gf, grp, x.tq + j0);
        x.tc[p] = 0.0;
    });
    ex.mark(25);
    // Outer weights and normalisation.  When every block is Mode A the LVs do not depend on each other inside the step:
    // all of them go through the same four phases at once (the per-LV loop below costs four dependent memory round trips per LV).
    // Same expressions, same summation order as the loop: bitwise the same a_new / akk.
    bool fused = true;
    long npairs = 0;
    for (int l = 0; l < L; ++l) {
        const int k = cd.lmv_off[l + 1] - cd.lmv_off[l];
        if (md.mode[l] == MODE_B) fused = false;                                  // (block solves and the Gauss-Seidel correction: the loop below)
        npairs += (long)k * k;
    }
    if (npairs > 8L * Pm * cd.cmax) fused = false;                                // the pair products borrow the quantification scratch
    if (fused) {
        double* pairs = x.pq;                                                     // [sum_l k_l^2]: <MV_r, MV_c> of the block, row-major per LV
        double* mv_mean = x.mvm + Pm;                                             // [Pm] means of the quantified MVs
        ex.par(Pm, [&](int p) {
            int l = 0;
            while (p >= cd.lmv_off[l + 1]) ++l;
            const double m = nmg_mv_moment(cd, x, p, x.MZ + l, L, x.MZ[Q * L + l]);
            x.mvm[p] = m;
            ws.wn[p] = m / ws.a[l];            // Mode A (mode.py:38)
            double mr = x.tc[p];
            for (int j = cd.mv_off[p]; j < cd.mv_off[p + 1]; ++j) mr += x.tq[j] * Mn[Q * LD + j];
            mv_mean[p] = mr;
        });
        ex.par((int)npairs, [&](int e) {
            int l = 0, base = 0;
            for (;;) { const int k = cd.lmv_off[l + 1] - cd.lmv_off[l]; if (e < base + k * k) break; base += k * k; ++l; }
            const int p0 = cd.lmv_off[l], k = cd.lmv_off[l + 1] - p0;
            const int r = (e - base) / k, c = (e - base) - r * k;
            if (r <= c) { const double v = nmg_mv_mv(cd, x, Mn, LD, Q, p0 + r, p0 + c); pairs[base + r * k + c] = v; pairs[base + c * k + r] = v; }
        });
        ex.par(L, [&](int l) {
            const int p0 = cd.lmv_off[l], k = cd.lmv_off[l + 1] - p0;
            int base = 0;
            for (int m = 0; m < l; ++m) { const int km = cd.lmv_off[m + 1] - cd.lmv_off[m]; base += km * km; }
            double q = 0.0, mw = 0.0;
            for (int r = 0; r < k; ++r) {
                mw += ws.wn[p0 + r] * mv_mean[p0 + r];
                for (int c = 0; c < k; ++c) q += ws.wn[p0 + r] * pairs[base + r * k + c] * ws.wn[p0 + c];
            }
            const double sd = sqrt(q - mw * mw);
            x.akk[l] = -mw / sd;
            ws.wf[l] = sd;                     // (an L-vector of the workspace that only the finish uses)
        });
        ex.par(Pm, [&](int p) {
            int l = 0;
            while (p >= cd.lmv_off[l + 1]) ++l;
            st.a_new[p] = ws.wn[p] / ws.wf[l];
        });
    }
    for (int l = 0; l < L && !fused; ++l) {
        const int p0 = cd.lmv_off[l], p1 = cd.lmv_off[l + 1], k = p1 - p0;
        const double mean_z = x.MZ[Q * L + l];
        bool have_beta = false;
        const bool modeb = (md.mode[l] == MODE_B) && (k > 1);
        for (int p = p0; modeb && p < p1; ++p) {
            const int kind = cd.mv_kind[p];
            if (kind == KIND_NUM) continue;                                       // NUM / RAW: constant quantification
            const int j0 = cd.mv_off[p], C = cd.mv_off[p + 1] - j0;
            if (modeb && !have_beta) {
                // betas of OLS(z ~ 1 + current block) = Cov_bb^-1 cov_bz, once per LV and iteration (weights.py:139-142)
                ex.par(k * k, [&](int e) {
                    const int r = e / k, c = e - r * k;
                    if (r <= c) {
                        const double v = nmg_mv_mv(cd, x, Mn, LD, Q, p0 + r, p0 + c);
                        x.Bm[r * k + c] = v; x.Bm[c * k + r] = v;
                    }
                });
                ex.par(k, [&](int r) { x.mvm[Pm * L + r] = nmg_mv_moment(cd, x, p0 + r, Mn + Q * LD, 1, 1.0); });       // means of the block MVs
                ex.par(k, [&](int r) { x.rhs[r] = nmg_mv_moment(cd, x, p0 + r, x.MZ + l, L, mean_z) - x.mvm[Pm * L + r] * mean_z; });
                ex.one([&]() {
                    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) x.Bm[r * k + c] -= x.mvm[Pm * L + r] * x.mvm[Pm * L + c];
                    for (int r = 0; r < k; ++r) x.beta[r] = x.rhs[r];
                    if (!psd_solve_once(x.Bm, k, x.Fm, x.Vm, x.beta)) st.scal[1] = (double)ST_SINGULAR;       // pinv like sm.OLS(...).fit() (weights.py:141)
                });
                have_beta = true;
            }
            // category means of (corrected) z: <D_c, zc> / <D_c, 1>
            ex.par(C, [&](int c) {
                const int j = j0 + c;
                double s = x.MZ[j * L + l];
                if (modeb) {
                    for (int r = 0; r < k; ++r) {
                        if (p0 + r == p) continue;
                        // <D_c, MV_q> = sum_i tq_i Mn[j][i] + tc_q * mean(D_c)
                        const int q = p0 + r;
                        double t = x.tc[q] * Mn[Q * LD + j];
                        for (int i = cd.mv_off[q]; i < cd.mv_off[q + 1]; ++i) t += x.tq[i] * Mn[i * LD + j];
                        s -= x.beta[r] * t;
                    }
                    s /= x.beta[p - p0];
                }
                x.cf[c] = Mn[Q * LD + j];
                x.cm[c] = (x.cf[c] > 0.0) ? s / x.cf[c] : 0.0;
            });
            ex.one([&]() {
                nmg_quantify_mv(kind, C, Mn + Q * LD + j0, x.cm, x.cf, x.cs, x.inc, x.dec, x.gsum, x.gf, x.grp, x.tq + j0);
                x.tc[p] = 0.0;
            });
        }
        // outer weights of the block with the updated MVs (mode.py:38, 58), then Y = treat_numpy(X w) * correction (mode.py:41, 60)
        ex.par(k, [&](int r) { x.mvm[p0 + r] = nmg_mv_moment(cd, x, p0 + r, x.MZ + l, L, mean_z); });
        if (md.mode[l] == MODE_A) {
            ex.par(k, [&](int r) { ws.wn[p0 + r] = x.mvm[p0 + r] / ws.a[l]; });
            ex.par(k * k, [&](int e) {
                const int r = e / k, c = e - r * k;
                if (r <= c) { const double v = nmg_mv_mv(cd, x, Mn, LD, Q, p0 + r, p0 + c); x.Bm[r * k + c] = v; x.Bm[c * k + r] = v; }
            });
        } else {
            ex.par(k * k, [&](int e) {
                const int r = e / k, c = e - r * k;
                if (r <= c) { const double v = nmg_mv_mv(cd, x, Mn, LD, Q, p0 + r, p0 + c); x.Bm[r * k + c] = v; x.Bm[c * k + r] = v; }
            });
            ex.one([&]() {
                // lstsq(X_b, z) without intercept, on raw moments
                for (int r = 0; r < k; ++r) x.rhs[r] = x.mvm[p0 + r];
                if (!psd_solve_once(x.Bm, k, x.Fm, x.Vm, x.rhs)) st.scal[1] = (double)ST_SINGULAR;   // Bm itself is needed for the normalisation below
                for (int r = 0; r < k; ++r) ws.wn[p0 + r] = x.rhs[r];
            });
        }
        ex.one([&]() {
            // variance of X_b w: w' Cov_b w (means of the block MVs removed: treat_numpy centres)
            double q = 0.0, mw = 0.0;
            for (int r = 0; r < k; ++r) {
                double mr = x.tc[p0 + r];
                for (int j = cd.mv_off[p0 + r]; j < cd.mv_off[p0 + r + 1]; ++j) mr += x.tq[j] * Mn[Q * LD + j];
                mw += ws.wn[p0 + r] * mr;
                for (int c = 0; c < k; ++c) q += ws.wn[p0 + r] * x.Bm[r * k + c] * ws.wn[p0 + c];
            }
            const double sd = sqrt(q - mw * mw);
            for (int r = 0; r < k; ++r) st.a_new[p0 + r] = ws.wn[p0 + r] / sd;
            x.akk[l] = -mw / sd;                                                   // constant of the NEW score (zero for centred MVs)
        });
    }
    ex.mark(26);
    nmg_score_map(ex, md, cd, x, st.a_new, x.akk, st.c_new, st.k_new);
    ex.mark(27);
    ex.one([&]() { st.scal[2] = (double)(iteration + 1); if (ws.scal[3] != (double)ST_OK && st.scal[1] == (double)ST_OK) st.scal[1] = ws.scal[3]; });
    return true;
}

// Collapse to the MV level -- correlation matrix of the final quantified MVs, weights a_new -- and run the shared tail.
// `mdm` is the MV-level descriptor (P = Pm, boff = lmv_off, ...); wsm its workspace (S: Pm x PSm).
template <class Ex>
PLSPM_HD void nmg_finish(Ex& ex, const ModelDesc& md, const CatDesc& cd, const ModelDesc& mdm, Workspace& ws, Workspace& wsm, NmState& st, NmgExtra& x,
                         const FitOutputs& out) {
    const int Q = md.P, LD = ws.PS, Pm = cd.Pm, PSm = wsm.PS;
    const double* Mn = ws.S;
    ex.par(Pm, [&](int p) { x.mvm[p] = nmg_mv_moment(cd, x, p, Mn + Q * LD, 1, 1.0); });                         // MV means
    ex.par(Pm * Pm, [&](int e) {
        const int p = e / Pm, q = e - p * Pm;
        if (p <= q) {
            const double v = nmg_mv_mv(cd, x, Mn, LD, Q, p, q) - x.mvm[p] * x.mvm[q];
            wsm.S[q * PSm + p] = v; wsm.S[p * PSm + q] = v;
        }
    });
    ex.par(Pm, [&](int p) { wsm.w[p] = st.a_new[p]; wsm.mu[p] = 0.0; wsm.cs[p] = 1.0; wsm.sd[p] = sqrt(wsm.S[p * PSm + p]); });
    ex.one([&]() { wsm.scal[1] = st.scal[0]; wsm.scal[2] = 1.0 / st.scal[0]; wsm.scal[3] = st.scal[1]; });
    FitOutputs o2 = out;
    o2.score_w = nullptr; o2.score_c = nullptr; o2.mean = nullptr;          // the score map lives on the aug columns (below)
    finish_problem(ex, mdm, wsm, o2, (int)st.scal[2], false);
    if (out.score_w) ex.par(Q, [&](int j) { out.score_w[j] = st.c_new[j]; });
    if (out.score_c) ex.par(md.L, [&](int l) { out.score_c[l] = st.k_new[l]; });
}

}  // namespace plspm
