// solver_quad.h -- the wave solver's fixed lane roles (solver_wave.h) on FOUR waves per problem: the batched solver of bootstrap replicates of
// metric Mode-A models with 65 .. 128 MVs and at most LMAX = 16 LVs (round 5; VERDICT r4 item 5: "one solver form for every model size").
// The split rows solver (solver_core.h solve_problem_rows<64, true>) covered that class with the GENERIC thread-group code -- ~50 `par`
// phases of a few dozen flops each, every one an LDS round trip + loops with run-time bounds (130 k clocks per problem at 120 x 12, of
// which the O(P^2) products are ~12 k); this is the same arithmetic with the roles of solver_wave.h:
//   MV thread  t = 128 side + p   the columns q0 .. q0 + nq of row p of the treated covariance that lie on `side` of the block boundary
//                                 rows_split_block (64 register pairs: the 128 x 128 matrix lives in the registers of the four waves), the MV's
//                                 weight, and the entries V[p, m] of V = S W for the LVs m of its side
//   pair thread e = 16 l + m      entry (l, m) of every L x L matrix (Q, G, E, score covariance) -- all 256 threads when L = 16
//   LV thread  i < L              the small regressions of LV i (PATH scheme, inner model) and column i of (I - B)^-1
// Reference: the same lines solver_wave.h restates --
//   Config.treat plspm/config.py:299-305 + util.treat plspm/util.py:33-39; _MetricWeights.__init__ plspm/weights.py:28-39; .iterate
//   plspm/weights.py:41-54; Scheme.*.calculate plspm/scheme.py:27-28, 36-37, 45-54; _ModeA.outer_weights_metric plspm/mode.py:28-29;
//   WeightsCalculatorFactory.calculate plspm/weights.py:172-187 (stop rule); _MetricWeights.calculate plspm/weights.py:56-70 (sign rule);
//   InnerModel / _effects plspm/inner_model.py:58-75, 33-53; bootstrap row plspm/bootstrap.py:58-64.
// An iteration is four workgroup barriers (PATH: five): S W -> [T = w V transposed] -> Q ->
// [G, E, a] -> (PATH: the regressions) -> outer step on both threads of an MV alike, w + the stop-rule sum (wave butterflies + one LDS slot per wave).
// Mode-B blocks, blocks that straddle every boundary, L > 16 and inner models whose regression scratch does not fit the staging area keep
// the split rows / LDS solvers (quad_solver_covers).
//
// Executor: the wave executor (kernels_solver.h DevWaveExec; tests/hostemu HostExec) with nt = 256: ex.sync() = workgroup barrier,
// ex.allsum = butterfly sum over the lanes of the caller's WAVE, ex.wave_vote_count = ballot over the caller's wave, ex.load_cov_block /
// ex.seg_products = the split rows solver's loader and segmented multiply-add stream.
#pragma once
#include "solver_wave.h"

namespace plspm {

constexpr int QUAD_VP = 17;                  // pitch of a row of V (doubles): 34 dwords -- the 64 lanes' 8-byte stores of one column spread over all banks
constexpr int QUAD_STAGE = 2 * 128 * QUAD_VP;                 // 4,352 doubles >= 4 x 16 x 66 (the loader's four transposition tiles): V and its copy w V

template <int LMAX>
struct QuadWs {
    double* stage;   // [QUAD_STAGE]  the loader's tiles; then V and T
    double* V;       // [128 * QUAD_VP]   V[p * QUAD_VP + m]
    double* T;       // [128 * QUAD_VP]   w_p V[p, m], the product stream's second store (device_exec.h seg_products2): the Q sums read one value per term
    double* w;       // [128]
    double* mu;      // [128]  column sums
    double *Qm, *Gm, *Em, *Cs, *Bm, *Ind;      // [LMAX * LMAX], entry (l, m) at l * LMAX + m
    double* a;       // [LMAX]
    double* red;     // [8]  two alternating sets of one slot per wave (group sums)
    double* votes;   // [2 * LMAX]  negative votes of the sign rule per wave of side 0 and LV
    double* scr;     // [L * regression_scratch_doubles(kmax)]  the LV threads' normal equations (five and more predecessors, minimum-norm fallback) and solutions
};
template <int LMAX> PLSPM_HD constexpr long quad_ws_doubles(int L, int kmax) { return QUAD_STAGE + 128 + 128 + 6 * LMAX * LMAX + LMAX + 8 + 2 * LMAX + (long)L * regression_scratch_doubles(kmax); }
template <int LMAX> PLSPM_HD void quad_carve(QuadWs<LMAX>& ws, double* base) {
    static_assert(QUAD_STAGE >= 4 * 16 * 66, "the loader's four tiles fit the staging area");
    static_assert(LMAX == 16, "pair thread e = 16 l + m; 256 threads");
    double* p = base;
    ws.stage = p; ws.V = p; ws.T = p + 128 * QUAD_VP; p += QUAD_STAGE;
    ws.w = p; p += 128; ws.mu = p; p += 128;
    ws.Qm = p; p += LMAX * LMAX; ws.Gm = p; p += LMAX * LMAX; ws.Em = p; p += LMAX * LMAX;
    ws.Cs = p; p += LMAX * LMAX; ws.Bm = p; p += LMAX * LMAX; ws.Ind = p; p += LMAX * LMAX;
    ws.a = p; p += LMAX;
    ws.red = p; p += 8;
    ws.votes = p; p += 2 * LMAX;
    ws.scr = p;
}
// What the quad solver covers (the host asks before it launches): two problems per CU, i.e. at most 80 KB of LDS each -- the workspace is ~47 KB + the
// regression scratch (16 LVs with ten predecessors each: 27 KB).
template <int LMAX> PLSPM_HD bool quad_solver_covers(int P, int L, int n_chol, int kmax, const int* boff) {
    return P > 64 && P <= 128 && L >= 2 && L <= LMAX && n_chol == 0 && rows_split_block(boff, L, 64) > 0 &&
           quad_ws_doubles<LMAX>(L, kmax) * (long)sizeof(double) <= 80 * 1024;
}

// Md: the DENSE moment matrix [(P+1) x cov_ld(P)] of the mean-shifted columns + ones, upper triangle (entry (r, c >= r) at r * PS + c),
// as the int8 digit-plane Gram writes it.  Outputs: out.row / out.status / out.iters (a bootstrap record).
template <int LMAX, class Ex>
PLSPM_HD void solve_problem_quad(Ex& ex, const ModelDesc& md, const QuadWs<LMAX>& ws, const double* Md, const FitOutputs& out) {
    constexpr int PMAX = 64;
    const int P = md.P, L = md.L, PS = cov_ld(P), t = ex.tid;
    // (side and wave are the same on all lanes of a wave: scalar registers, and with them the window [q0, q0 + nq), the LV range [l0, l1) and every
    //  test against them -- as vector values they cost the registers that sent the column sum to scratch and a reload per column of the treat loop)
    const int p = t & 127, side = ex.uniform_i(t >> 7), wave = ex.uniform_i(t >> 6);
    const bool valid = p < P;                                    // MV role: a thread with an MV
    const bool owner = valid && side == 0;                       // ... and the one that owns the MV's entries of the per-MV arrays
    const int pc = valid ? p : P - 1;
    const int lp = md.lvof[pc];                                  // MV role: my LV
    const int ms = rows_split_block(md.boff, L, PMAX);           // LVs [0, ms) on side 0, [ms, L) on side 1
    const int l0 = side ? ms : 0, l1 = side ? L : ms;
    const int q0 = md.boff[l0], nq = md.boff[l1] - q0;           // my window of columns
    const int el = t / LMAX, em = t % LMAX;                      // pair role: entry (el, em)
    const bool pair = el < L && em < L;
    const int elc = pair ? el : 0;
    const int pb0 = md.boff[elc], pk = pair ? md.boff[elc + 1] - pb0 : 0;      // pair role: the block of row el
    const bool lvlane = t < L;                                   // LV role
    const bool c_lm = pair && md.C[el * L + em] != 0;             // LV em -> LV el
    const int d_lm = pair ? (int)md.C[el * L + em] + (int)md.C[em * L + el] : 0;
    int nk = 0;                                                  // LV role: my predecessors, the first four one byte each
    unsigned fpack = 0u;
    if (lvlane) {
        const int o = md.pred_off[t];
        nk = md.pred_off[t + 1] - o;
        for (int r = 0; r < 4; ++r) if (r < nk) fpack |= (unsigned)md.pred_idx[o + r] << (8 * r);
    }
    const int ne = md.n_eff;
    const int eidx = (t < ne) ? md.eff_to[t] * LMAX + md.eff_from[t] : 0;
    const double shp = md.scaled ? md.shift[pc] : 0.0;
    int kbmax = 0;
    unsigned long long ends = 0ull;                              // bit j: column q0 + j closes its block
    for (int l = 0; l < L; ++l) {
        const int k = md.boff[l + 1] - md.boff[l];
        kbmax = k > kbmax ? k : kbmax;
        if (l >= l0 && l < l1) ends |= 1ull << (md.boff[l + 1] - 1 - q0);
    }
    ends = ex.uniform(ends);
    bool singular = false;
    // group sum of one value per thread: butterfly inside the wave, one LDS slot per wave (two alternating sets: a thread may enter the next
    // sum while another still reads this one's slots, never the one after)
    auto groupsum = [&](double v, int set) {
        const double pw = ex.allsum(v);
        if ((t & 63) == 0) ws.red[4 * set + wave] = pw;
        ex.sync();
        return (ws.red[4 * set] + ws.red[4 * set + 1]) + (ws.red[4 * set + 2] + ws.red[4 * set + 3]);
    };

    // 1. moments -> treated covariance (config.py:299-305, util.py:33-39): my window of row p in registers
    ex.mark(0);
    double s[PMAX];
    // (the two strided entries of my row first: their round trip runs under the loader's)
    double dpp = 0.0, mup = 0.0;                                 // raw M[p][p] and the column sum M[p][P] (ones column)
    if (valid) { mup = Md[(long)p * PS + P]; dpp = Md[(long)p * PS + p]; }
    const double n_raw = Md[(long)P * PS + P];
    ex.template load_cov_block<PMAX>(Md, PS, P, pc, q0, nq, s);
    const double n = ex.uniform_d(n_raw);
    ex.mark(1);
    if (side == 0) { ws.mu[p] = mup; ws.w[p] = 1.0; }            // init: block products with w = 1
    ex.sync();
    const double inv_n = ex.uniform_d(1.0 / n);
    double fac = inv_n;
    if (md.scaled) {
        // g = std1(all N*P raw values) * sqrt((N-1)/N)   (config.py:302), evaluated around the grand mean
        const double tot = groupsum(owner ? mup + n * shp : 0.0, 0);
        const double np_ = n * (double)P, grand = tot / np_;
        const double d = shp - grand;
        const double ss = groupsum(owner ? dpp + 2.0 * d * mup + n * d * d : 0.0, 1);
        const double g2 = ss / (np_ - 1.0) * ((n - 1.0) / n);
        fac = ex.uniform_d(1.0 / (n * g2));
    }
    // (n, 1 / n and the scale factor are the same on every thread: scalar registers.  Every thread treats its 64 registers, a thread without an MV
    //  clears them afterwards: one branch instead of a predicate -- and an exec-masked block -- per column)
    ex.fence();
#pragma unroll
    for (int qb = 0; qb < PMAX; qb += 8) {
#pragma unroll
        for (int q = qb; q < qb + 8; ++q) {
            const double v = (s[q] - (mup * ws.mu[q0 + ((q < nq) ? q : nq - 1)]) * inv_n) * fac;      // (mu_p mu_q) first: bitwise symmetric in (p, q)
            s[q] = (q < nq) ? v : 0.0;
        }
        ex.pin8(s[qb], s[qb + 1], s[qb + 2], s[qb + 3], s[qb + 4], s[qb + 5], s[qb + 6], s[qb + 7]);
    }
    if (!valid) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) s[q] = 0.0;
    }
    const double sdp = treated_sd(dpp, mup, inv_n, fac);           // (0: a column that is constant in this replicate -- solver_core.h)
    // (its row of the covariance is rounding residue: exact zeros instead -- an LV whose only item is that column then has a score variance of exactly 0 and fails as the
    //  reference's does, instead of normalising the residue; rare: one ballot, the loop runs for the waves that hold such a column)
    if (ex.wave_vote_count(valid && sdp == 0.0) > 0) {
        const double keep = (sdp == 0.0) ? 0.0 : 1.0;
#pragma unroll
        for (int q = 0; q < PMAX; ++q) s[q] *= keep;
    }
    const double corr2 = ex.uniform_d(n / (n - 1.0));
    ex.mark(2);                                                  // (the loader's last barrier stands behind its last tile read: the staging area is free)

    // LV role: normal equations M[f, f] x = M[f, t] over my predecessors f -- up to four in registers (wave_ldl4), more by Cholesky in LDS
    // scratch; a rank-deficient system takes the minimum-norm answer of the reference's pinv / gelsd (solver_core.h pinv_solve).  Scratch and
    // x live in an area of their own (ws.scr): V stays in LDS through the trip, and the outer step, the sign rule and the loadings read it there --
    // sixteen register pairs per thread that the regressions' straight-line code would otherwise send to scratch.
    const int* fglob = md.pred_idx + (lvlane ? md.pred_off[t] : 0);
    const long rscr = regression_scratch_doubles(md.kmax);
    const int km = md.kmax;
    auto pred = [&](int r) { return r < 4 ? (int)((fpack >> (8 * r)) & 255u) : fglob[r]; };
    auto regress = [&](const double* M) {
        double* scratch = ws.scr + t * rscr;
        double* x = scratch + 2 * km * km;
        bool ok;
        if (nk <= 4) {
            unsigned fp = fpack;
            ex.opaque(fp);
            ok = wave_ldl4(M, LMAX, fp, nk, t, x);
        } else {
            for (int r = 0; r < nk; ++r) {
                for (int c = 0; c < nk; ++c) scratch[r * nk + c] = M[fglob[r] * LMAX + fglob[c]];
                x[r] = M[fglob[r] * LMAX + t];
            }
            ok = chol_factor(scratch, nk);
            if (ok) chol_solve(scratch, nk, x);
        }
        if (!ok && !(nk > 1 && pinv_solve(M, LMAX, fglob, nk, t, x, scratch))) singular = true;
        return x;
    };

    // ONE loop carries init, the iterations and the finalisation (as solve_problem_wave): each trip starts with V = S W and Q = W' S W for
    // the weights in ws.w / wp.   phase 0 init (weights.py:28-39), 1 iterations (weights.py:41-54, 179-186), 2 the product of the final weights
    const double icorr2 = ex.uniform_d((n - 1.0) / n);
    double Qe = 0.0, wp = valid ? 1.0 : 0.0;
    int iteration = 0, phase = 0;
    while (true) {
        int pl = p, lpl = lp, pb0l = pb0, tl = t;               // opaque copies: LDS addresses recomputed per trip instead of hoisted and spilled
        ex.opaque(pl); ex.opaque(lpl); ex.opaque(pb0l); ex.opaque(tl);
        const int l0l = l0;
        const int ell = tl / LMAX, eml = tl % LMAX;
        ex.mark(16);
        ex.template seg_products2<PMAX, 128 * QUAD_VP * 8>(s, ws.w + q0, nq, ends, ws.V + pl * QUAD_VP + l0l, wp);      // (own row: no exchange; a thread without an MV stores zeros in a row >= P)
        ex.mark(17);
        ex.sync();
        ex.mark(18);
        {
            // Q[el, em] = sum over the MVs p of block el of w_p V[p, em] -- out of the product stream's second copy: eight loads in flight per trip.  (The wave
            // solver stores T = w V transposed in a pass of its own for this sum -- an LDS round trip per LV on every MV thread; reading w_p and V[p, m] here
            // cost two loads per term: 2.0 k clocks per trip against 1.2 k.)
            double s0 = 0.0, s1 = 0.0;
            const double* vv = ws.T + pb0l * QUAD_VP + eml;
            for (int i0 = 0; i0 < kbmax; i0 += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (i0 + j < pk) ? vv[(i0 + j) * QUAD_VP] : 0.0;
                s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3]; s0 += v[4]; s1 += v[5]; s0 += v[6]; s1 += v[7];
            }
            Qe = s0 + s1;
        }
        ws.Qm[tl] = pair ? Qe : 1.0;
        ex.sync();
        ex.mark(19);
        if (phase == 2) break;
        if (phase == 0) {
            wp = valid ? wave_rsqrt(ws.Qm[lpl * LMAX + lpl]) : 0.0;
            if (side == 0) ws.w[pl] = valid ? wp : 1.0;
            ex.sync();
            ex.mark(3);
            phase = 1;
            continue;
        }
        ++iteration;
        ex.mark(9);
        // Yhat_l = Y_l / std1 / corr:  a_l = 1 / (corr2 sqrt(Q_ll)),  G = cov0(Yhat) = a a' o Q   (weights.py:43-44)
        const double rl = wave_rsqrt(ws.Qm[ell * LMAX + ell]), rm = wave_rsqrt(ws.Qm[eml * LMAX + eml]);
        const double al = rl * icorr2, am = rm * icorr2;
        const double Ge = al * am * Qe;
        double Ee = 0.0;
        if (pair) {
            if (md.scheme == SCHEME_PATH) {
                if (c_lm) Ee = Qe * rl * rm;                     // column em of E: correlations with the successors of em (scheme.py:51-53)
            } else if (d_lm) {
                Ee = (md.scheme == SCHEME_CENTROID) ? ((Ge > 0.0) ? 1.0 : ((Ge < 0.0) ? -1.0 : 0.0)) : Ge * corr2 * (double)d_lm;   // cov1 = cov0 N/(N-1)
            }
            ws.Gm[tl] = Ge; ws.Em[tl] = al * Ee;                 // (row el of E carries a_el: the outer step multiplies V with a E)
            if (el == em) ws.a[ell] = al;
        }
        ex.sync();
        ex.mark(10);
        if (md.scheme == SCHEME_PATH) {
            if (lvlane && nk > 0) {                              // regression of Yhat_t on its predecessors, no intercept (scheme.py:48-50)
                const double* x = regress(ws.Gm);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nk) ws.Em[pred(r) * LMAX + tl] = ws.a[pred(r)] * x[r];
                for (int r = 4; r < nk; ++r) ws.Em[pred(r) * LMAX + tl] = ws.a[pred(r)] * x[r];
            }
            ex.sync();
        }
        ex.mark(11);
        // outer step, Mode A: w = (S Wn E)[p, lv(p)] == X'Z/N  (mode.py:29): row p of V out of LDS (the two threads of an MV wrote its halves), both
        // threads of an MV add the same L terms in the same order
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < LMAX; m += 2) {
            if (m < L) c0 += ws.V[pl * QUAD_VP + m] * ws.Em[m * LMAX + lpl];
            if (m + 1 < L) c1 += ws.V[pl * QUAD_VP + m + 1] * ws.Em[(m + 1) * LMAX + lpl];
        }
        const double wn = valid ? c0 + c1 : 0.0;
        const double dd = fabs(wp) - fabs(wn);
        wp = wn;
        if (side == 0) ws.w[pl] = valid ? wp : 1.0;              // (the next reader is the next trip's seg_products, behind the barrier of the sum)
        const double conv = groupsum(owner ? dd * dd : 0.0, iteration & 1);
        ex.mark(12);
        if (conv < md.tol || iteration > md.max_iter) phase = 2;
    }
    const bool not_converged = iteration > md.max_iter;
    ex.mark(4);

    // finalize (weights.py:56-70): wf_l = 1 / sqrt(Q_ll); returned weights never sign-flipped
    const double wfp = wave_rsqrt(ws.Qm[lp * LMAX + lp]);
    wp *= wfp;
    // sign rule: EVERY MV votes (weights.py:62-64); sign(cor[p,l]) == sign(V[p,l]); a zero-variance column votes +1 everywhere (pandas' NaN correlation has its sign bit clear: solver_core.h).  The owners sit in waves 0 and 1; every wave casts the same number of
    // ballots (the CPU emulation's ballot is a barrier).
    {
        double vr[LMAX];                                         // (my row of V in one batch of loads: a load inside every ballot's block waits out its own LDS round trip)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) vr[l] = ws.V[p * QUAD_VP + (l < L ? l : 0)];
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
            if (l < L) { const int neg = ex.wave_vote_count(owner && vr[l] < 0.0 && sdp != 0.0); if ((t & 63) == 0 && side == 0) ws.votes[wave * LMAX + l] = (double)neg; }
    }
    ex.sync();
    // (a thread looks up the three signs it needs -- its pair's two LVs, its MV's LV -- instead of walking all L tallies)
    auto sign_of = [&](int l) { return ((double)P - 2.0 * (ws.votes[l] + ws.votes[LMAX + l]) < 0.0) ? -1.0 : 1.0; };
    const double sgl = sign_of(lp);
    {
        const double wfl = wave_rsqrt(ws.Qm[el * LMAX + el]), wfm = wave_rsqrt(ws.Qm[em * LMAX + em]);
        const double sl = sign_of(pair ? el : 0), sm = sign_of(pair ? em : 0);
        if (pair) ws.Cs[t] = sl * sm * wfl * wfm * Qe;           // population covariance of the sign-corrected scores
        ws.Bm[t] = 0.0;                                          // (all 256 entries: the effects below then run without a test per row)
    }
    ex.sync();
    ex.mark(5);
    // inner model (inner_model.py:58-75): OLS with intercept == centred normal equations on the score covariance
    double r2p = 0.0;
    if (lvlane) {
        if (nk > 0) {
            const double* x = regress(ws.Cs);
            double expl = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk) { const int fr = pred(r); ws.Bm[t * LMAX + fr] = x[r]; expl += x[r] * ws.Cs[fr * LMAX + t]; }
            for (int r = 4; r < nk; ++r) { const int fr = pred(r); ws.Bm[t * LMAX + fr] = x[r]; expl += x[r] * ws.Cs[fr * LMAX + t]; }
            r2p = expl / ws.Cs[t * LMAX + t];
        }
    }
    ex.sync();
    ex.mark(6);
    // effects (inner_model.py:33-53): indirect = B^2 + B^3 + ... = (I - B)^-1 - I - B; B is strictly lower triangular in path order, so column t of
    // (I - B)^-1 follows by forward substitution in the registers of LV thread t.  colx = the column without its unit entry: rows above t are
    // zero, so the sum over ALL k < i of B[i][k] colx[k] is the sum over the paths of length >= 2 -- no predicate per term.  Rows >= L of B are zero
    // (cleared above), so all 120 terms run as one block: the loads of B do not depend on the arithmetic and go out in batches, where a branch per
    // row held every row's loads back behind the previous row's chain (3.4 k -> ~1.5 k clocks).
    if (lvlane) {
        double colx[LMAX];
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            double i0 = 0.0, i1 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < i; k += 2) { i0 += ws.Bm[i * LMAX + k] * colx[k]; i1 += ws.Bm[i * LMAX + k + 1] * colx[k + 1]; }
            if (i & 1) i0 += ws.Bm[i * LMAX + i - 1] * colx[i - 1];
            const double ind = (i > t) ? i0 + i1 : 0.0;
            colx[i] = (i > t) ? ws.Bm[i * LMAX + t] + ind : 0.0;
            ws.Ind[i * LMAX + t] = ind;
        }
    }
    ex.sync();
    ex.mark(7);
    // outputs: the bootstrap record  weights | r2 | total | direct | loadings | status | iterations  (bootstrap.py:58-64)
    if (out.row) {
        if (owner) {
            out.row[p] = wp;
            out.row[P + L + 2 * ne + p] = (sdp > 0.0) ? sgl * ws.V[p * QUAD_VP + lp] * wfp / sdp : 0.0;      // (zero variance: loading 0, solver_core.h treated_sd)
        }
        if (lvlane) out.row[P + t] = r2p;
        if (t < ne) {
            out.row[P + L + t] = ws.Bm[eidx] + ws.Ind[eidx];
            out.row[P + L + ne + t] = ws.Bm[eidx];
        }
    }
    const int nbad = ex.wave_vote_count((owner && !(isfinite(wp) && isfinite(sdp) && sdp >= 0.0)) || (lvlane && !isfinite(r2p)));
    const bool sing = ex.wave_vote_count(singular) > 0;         // (the LV threads all sit in wave 0, with thread 0)
    if ((t & 63) == 0) ws.red[wave] = (double)nbad;
    ex.sync();
    if (t == 0) {
        const bool bad = (ws.red[0] + ws.red[1]) + (ws.red[2] + ws.red[3]) > 0.0;
        int st = sing ? ST_SINGULAR : (not_converged ? ST_NOT_CONVERGED : ST_OK);
        if (st == ST_OK && bad) st = ST_NONFINITE;
        if (out.status) *out.status = st;
        if (out.iters) *out.iters = iteration;
        if (out.row) { out.row[2 * P + L + 2 * ne] = (double)st; out.row[2 * P + L + 2 * ne + 1] = (double)iteration; }
    }
    ex.mark(13);
}

}  // namespace plspm
