// solver_wave16.h -- the wave solver (solver_wave.h: ONE 64-lane wave per problem, fixed lane roles) in the arrangement round 5 arrived at, as a template on the
// largest number of LVs: metric models with at most 64 MVs behind the int8 Gram's dense moment matrices.
//   LMAX = 8    the headline's class (at most 8 LVs; Mode A and -- second instantiation -- Mode-B blocks): the default since the second half of round 5
//               (solver_wave_kernel<8> of rounds 3 / 4 stays behind set_option("solver_wave", 3))
//   LMAX = 16   9 .. 16 LVs -- "many LVs, few indicators each" (twelve LVs of three to five items) -- which the rows solver's generic thread-group code served
//               before (0.16 / 0.18 / 0.32 ms per 5,000 replicates at 10 / 12 / 16 LVs beside the wave solver's 0.10 at 8); Mode A and Mode-B blocks
//   LMAX = 32   17 .. 32 LVs, all Mode A (the LDS solver's class before): three problems per CU by LDS, one wave per SIMD
// Same arithmetic, same reference lines as solver_wave.h / solver_quad.h:
//   Config.treat plspm/config.py:299-305 + util.treat plspm/util.py:33-39; _MetricWeights.__init__ plspm/weights.py:28-39; .iterate
//   plspm/weights.py:41-54; Scheme.*.calculate plspm/scheme.py:27-28, 36-37, 45-54; _ModeA.outer_weights_metric plspm/mode.py:28-29; _ModeB.outer_weights_metric
//   plspm/mode.py:50-52; WeightsCalculatorFactory.calculate plspm/weights.py:172-187 (stop rule); _MetricWeights.calculate plspm/weights.py:56-70 (sign rule);
//   InnerModel / _effects plspm/inner_model.py:58-75, 33-53; bootstrap row plspm/bootstrap.py:58-64.
// Roles:
//   MV lane p < P          column p of the treated covariance in 64 register pairs, its weight
//   pair lane t            NE = LMAX^2 / 64 entries of every L x L matrix: column m = t mod LMAX, rows l = t / LMAX + (64 / LMAX) u, u < NE (one entry at LMAX = 8,
//                          four at 16, sixteen at 32)
//   LV lane i < L          the small regressions of LV i and column i of (I - B)^-1
// and the quad solver's arrangement of the small arrays: V = S W stays in LDS (odd pitch: conflict-free) and is read there by the Q sums, the outer step (a folded
// into E), the sign rule and the loadings; the regressions have a scratch area of their own; Q / G / E are re-used as indirect effects / score covariance / path
// matrix after the loop.  LDS per problem 12 KB (LMAX = 8), 16 KB (16), 43 KB (32) + the regression scratch (+ the Mode-B inverses).
#pragma once
#include "solver_wave.h"

namespace plspm {

// VP: pitch of a row of V (doubles): odd -- conflict-free 8-byte accesses.  TCOPY (LMAX = 8): the product stream stores w_p V[p, m] as a second copy behind V
// (device_exec.h seg_products2: one multiply + one store per block close) and the Q sums read that -- one load per term; at LMAX = 16 the copy would cost two of the
// eight problems a CU holds, and the sums multiply w_p V[p, m] term by term.
template <int LMAX> struct W16 { static constexpr int VP = LMAX + 1; static constexpr bool TCOPY = LMAX <= 8; };

template <int LMAX>
struct Wave16Ws {
    double* stage;   // [64 * W16<LMAX>::VP]  the column loader's 16 x 66 transposition tile; then V[p * W16<LMAX>::VP + m]
    double* V;
    double* w;       // [64]
    double* mu;      // [64]
    double *Qm, *Gm, *Em;      // [LMAX * LMAX], entry (l, m) at l * LMAX + m;  after the loop: Ind (= Qm), Cs (= Gm), Bm (= Em)
    double* a;       // [LMAX]
    double* sink;    // [LMAX] where the idle lanes of seg_products store (LMAX = 16)
    double* scr;     // [L * regression_scratch_doubles(kmax)]
    double* inv;     // [n_chol / 2]  Mode-B blocks: (S_bb)^-1 (or the pseudo-inverse of a rank-deficient block), full k x k, block l at chol_off[l] / 2
};
template <int LMAX> PLSPM_HD constexpr long wave16_v_doubles() {                                                                             // (the loader's tile fits the V area)
    return (W16<LMAX>::TCOPY ? 2 : 1) * 64 * W16<LMAX>::VP > 16 * 66 ? (W16<LMAX>::TCOPY ? 2 : 1) * 64 * W16<LMAX>::VP : 16 * 66;
}
template <int LMAX> PLSPM_HD constexpr long wave16_ws_doubles(int L, int kmax, int n_chol = 0) { return wave16_v_doubles<LMAX>() + 64 + 64 + 3 * LMAX * LMAX + 2 * LMAX + (long)L * regression_scratch_doubles(kmax) + n_chol / 2; }
template <int LMAX> PLSPM_HD void wave16_carve(Wave16Ws<LMAX>& ws, double* base, int L, int kmax) {
    static_assert(LMAX == 32 || LMAX == 16 || LMAX == 8, "pair lane: column t mod LMAX, rows t / LMAX + (64 / LMAX) u");
    double* p = base;
    ws.stage = p; ws.V = p; p += wave16_v_doubles<LMAX>();
    ws.w = p; p += 64; ws.mu = p; p += 64;
    ws.Qm = p; p += LMAX * LMAX; ws.Gm = p; p += LMAX * LMAX; ws.Em = p; p += LMAX * LMAX;
    ws.a = p; p += LMAX; ws.sink = p; p += LMAX;
    ws.scr = p; p += (long)L * regression_scratch_doubles(kmax);
    ws.inv = p;
}
// What this solver covers (the host asks before it launches): at least four problems per CU.
// Mode-B blocks (round 5, last part): their inverses are formed in the V area before the first S W (solver_wave.h: the sweep ping-pongs between ws.inv and the staging area).
template <int LMAX> PLSPM_HD bool wave16_solver_covers(int P, int L, int n_chol, int kmax) {
    return P >= 1 && P <= 64 && L > LMAX / 2 && L <= LMAX && n_chol / 2 <= 16 * 66 && wave16_ws_doubles<LMAX>(L, kmax, n_chol) * (long)sizeof(double) <= (LMAX > 16 ? 53 : 40) * 1024;
}

// NM (round 6): the non-metric iteration on Scale.NUM / RAW data (reference _NonmetricWeights, plspm/weights.py:73-133; mode.py:31-42, 54-61; scale.py:22-39;
// Config.treat config.py:306-318) on the same lane roles -- solver_core.h nm_prepare / nm_step / nm_finish restated.  Every MV is population-standardised, so
// the column registers hold the CORRELATION matrix R; the scores y_l = Xs a_l are population-standardised after every step (treat_numpy(.) * correction,
// mode.py:41,60), which in this loop's terms is the metric iteration with a_l = 1 / sqrt(Q_ll) in place of 1 / (corr2 sqrt(Q_ll)) -- except in the first step,
// whose scores X a_0, a_0 = 1 / sqrt(k_l), are used as they are (weights.py:82-98).  Mode A's division by sum z^2 (mode.py:38) is a positive factor the
// normalisation removes again.  The reference's stop rule is on the SCORES, sum_il c_i (|y_old| - |y_new|)^2 < tol (weights.py:120): not a function of second
// moments.  Here a step STOPS on the upper bound n sum_l d_l' R_bb d_l (d = a_new - a_old; ||a| - |b|| <= |a - b|: the problem stops exactly where the
// reference stops whenever the bound says so) and CONTINUES speculatively otherwise, leaving the score map of every step it continued behind in `io.maps`:
// the host's verification pass (plspm_nonmetric.hip run_nonmetric_wave) evaluates the exact criterion of those steps on the observations and replays a
// problem whose exact value was below the tolerance although its bound was not, with `io.force_T` = the step the reference stops at.
struct NmWaveIo {
    double* maps;      // [max_iter + 2][P + L + 1]: step j's score map (c_p per uploaded column | k_l per LV | the bound of step j), j = 0 .. steps - 1; null: nothing stored (replay)
    int force_T;       // > 0: stop behind exactly this many steps, whatever the bound says
    int* steps;        // steps taken (== the record's iteration count)
    double bound_scale = 1.0;      // test seam (set_option "nm_bound_shift"): the bound times 2^k, k >= 0 -- still an upper bound, only a worse one: the problem runs on behind
                                   // the reference's stop and the verification has to move it back
};

// Md: the DENSE moment matrix [(P+1) x cov_ld(P)] of the mean-shifted columns + ones, upper triangle.  Outputs: out.row / out.status / out.iters.
template <int LMAX, bool MODEB = false, bool NM = false, class Ex>
PLSPM_HD void solve_problem_wave16(Ex& ex, const ModelDesc& md, const Wave16Ws<LMAX>& ws, const double* Md, const FitOutputs& out, const NmWaveIo* io = nullptr, double* nmk = nullptr) {
    constexpr int PMAX = 64, NE = LMAX * LMAX / 64;              // (LMAX = 8: one entry per lane -- an A/B form of the wave solver's own class, option solver_wave 2)
    const int P = md.P, L = md.L, PS = cov_ld(P), t = ex.tid, p = t;
    const bool valid = p < P;
    const int pc = valid ? p : P - 1;
    const int lp = md.lvof[pc];                                  // MV role: my LV
    const int em = t % LMAX, er0 = t / LMAX;                     // pair role: column em, rows er0 + 4 u
    const bool lvlane = t < L;                                   // LV role
    // pair role, entry u, ONE register: first MV of its row's block | MVs of that block << 8 | (C[el, em] + 2 C[em, el]) << 16 | "the entry exists" << 24
    // (four registers per field of the <16> form were the values the allocator sent to scratch: twelve reloads per trip)
    unsigned dsc[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int el = er0 + (64 / LMAX) * u;
        const bool pr = el < L && em < L;
        const int elc = pr ? el : 0, emc = pr ? em : 0;
        const int b0 = md.boff[elc];
        dsc[u] = (unsigned)b0 | (unsigned)(pr ? md.boff[elc + 1] - b0 : 0) << 8 | (unsigned)(pr ? (int)md.C[elc * L + emc] + 2 * (int)md.C[emc * L + elc] : 0) << 16 |
                 (pr ? 1u : 0u) << 24;                                   // bit 16: LV em -> LV el
    }
    unsigned dl[NE];                                             // (inside the loop: opaque copies, unpacked where they are used instead of hoisted and spilled)
#pragma unroll
    for (int u = 0; u < NE; ++u) dl[u] = dsc[u];
    auto pb0 = [&](int u) { return (int)(dl[u] & 255u); };
    auto pk = [&](int u) { return (int)((dl[u] >> 8) & 255u); };
    auto dlm = [&](int u) { return (int)((dl[u] >> 16) & 3u); };
    auto pairu = [&](int u) { return (dl[u] >> 24) != 0u; };
    int nk = 0;                                                  // LV role: my predecessors, the first four one byte each
    unsigned fpack = 0u;
    if (lvlane) {
        const int o = md.pred_off[t];
        nk = md.pred_off[t + 1] - o;
        for (int r = 0; r < 4; ++r) if (r < nk) fpack |= (unsigned)md.pred_idx[o + r] << (8 * r);
    }
    const int ne = md.n_eff;
    const double shp = md.scaled ? md.shift[pc] : 0.0;
    int kbmax = 0, kbB = 0;                                      // widest block; widest Mode-B block
    unsigned long long ends = 0ull;
    for (int l = 0; l < L; ++l) {
        const int k = md.boff[l + 1] - md.boff[l];
        kbmax = k > kbmax ? k : kbmax;
        if (MODEB && md.mode[l] == MODE_B) kbB = k > kbB ? k : kbB;
        ends |= 1ull << (md.boff[l + 1] - 1);
    }
    ends = ex.uniform(ends);
    bool singular = false;

    // 1. moments -> treated covariance (config.py:299-305, util.py:33-39): column p in registers
    ex.mark(0);
    double s[PMAX];
    double dpp = 0.0, mup = 0.0;                                 // raw M[p][p] and the column sum M[p][P] (ones column)
    const double n = ex.uniform_d(Md[(long)P * PS + P]);
    ex.template load_cov<PMAX>(Md, PS, P, s, ws.stage, mup, dpp);
    if (!valid) { mup = 0.0; dpp = 0.0; }
    ex.mark(1);
    const double inv_n = ex.uniform_d(1.0 / n);
    // NM: population std of the uploaded column (config.py:314) -- the expression of solver_core.h nm_prepare
    const double sdraw = NM ? nm_column_sd(dpp * inv_n, mup * inv_n) : 1.0;      // (NaN for a column that is constant in this replicate: solver_core.h)
    ws.mu[p] = mup;
    ws.w[p] = NM ? (valid ? sdraw : 1.0) : 1.0;                  // init: block products with w = 1  (NM: sigma_q published for the loop below; the initial weights follow it)
    ex.sync();
    double fac = inv_n;
    if (!NM && md.scaled) {
        // g = std1(all N*P raw values) * sqrt((N-1)/N)   (config.py:302), evaluated around the grand mean
        const double tot = ex.allsum(valid ? mup + n * shp : 0.0);
        const double np_ = n * (double)P, grand = tot / np_;
        const double d = shp - grand;
        const double ss = ex.allsum(valid ? dpp + 2.0 * d * mup + n * d * d : 0.0);
        const double g2 = ss / (np_ - 1.0) * ((n - 1.0) / n);
        fac = ex.uniform_d(1.0 / (n * g2));
    }
    ex.fence();
#pragma unroll
    for (int qb = 0; qb < PMAX; qb += 8) {
#pragma unroll
        for (int q = qb; q < qb + 8; ++q) {
            double v;
            if constexpr (NM) v = ((s[q] - (mup * ex.bcast(mup, (q < P) ? q : P - 1, ws.mu)) * inv_n) * inv_n) / (sdraw * ex.bcast(sdraw, (q < P) ? q : P - 1, ws.w));      // R_pq (nm_prepare)
            else v = (s[q] - (mup * ex.bcast(mup, (q < P) ? q : P - 1, ws.mu)) * inv_n) * fac;      // (mu_p mu_q) first: bitwise symmetric in (p, q); mu_q: a lane broadcast
            s[q] = (q < P) ? v : 0.0;
        }
        ex.pin8(s[qb], s[qb + 1], s[qb + 2], s[qb + 3], s[qb + 4], s[qb + 5], s[qb + 6], s[qb + 7]);
    }
    if (!valid) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) s[q] = 0.0;
    }
    const double sdp = NM ? sqrt(((dpp - (mup * mup) * inv_n) * inv_n) / (sdraw * sdraw)) : treated_sd(dpp, mup, inv_n, fac);      // (NM: sqrt(R_pp), 1 to rounding)
    if constexpr (!NM) {
        // (a zero-variance column's row of the covariance is rounding residue: exact zeros instead -- an LV whose only item is that column then has a score variance of exactly
        //  0 and fails as the reference's does; rare: one ballot, the loop runs for the waves that hold such a column)
        if (ex.vote_any(valid && sdp == 0.0)) {
            const double keep = (sdp == 0.0) ? 0.0 : 1.0;
#pragma unroll
            for (int q = 0; q < PMAX; ++q) s[q] *= keep;
        }
    }
    const double corr2 = ex.uniform_d(n / (n - 1.0));
    ex.mark(2);                                                  // (the loader's last barrier stands behind its last tile read: V takes the tile's place)

    // Mode-B blocks (solver_wave.h, round 4: the same code on this workspace -- the staging area is the V area, free until the first S W) (mode.py:50-52: w_b = argmin |X_b w - z| = S_bb^-1 (X_b' z / N)): S_bb does not change over the iterations, so its
    // inverse is formed once.  Gauss-Jordan sweep without pivoting (S_bb is positive definite; pivot j is the same Schur complement the
    // Cholesky factorisation of solver_core.h tests, so a rank-deficient block is recognised by the same rule), every MV lane of a Mode-B
    // block owning ROW i of its k x k matrix, all blocks at once, out of place: step j reads A, writes A' -- no lane reads what another
    // rewrites in the same step, ONE exchange per step -- ping-ponging between ws.inv and the staging area (free until the first S W).
    //     i == j:  A'[j][c] = A[j][c] / piv  (c != j),  A'[j][j] = 1 / piv
    //     i != j:  A'[i][c] = A[i][c] - A[i][j] A[j][c] / piv  (c != j),  A'[i][j] = -A[i][j] / piv
    // A block whose sweep meets a pivot that is not safely positive takes the minimum-norm route of the reference's gelsd (jacobi_pinv, on
    // its LV lane, one such block at a time in the staging area) -- the outer step multiplies with the full symmetric matrix either way.
    const bool modeb = MODEB && valid && md.mode[lp] == MODE_B;
    const int bb0 = md.boff[lp], bk = md.boff[lp + 1] - bb0, bi = p - bb0;
    const long boffB = modeb ? md.chol_off[lp] / 2 : 0;
    if constexpr (MODEB) {
        ex.sync();                                              // (every lane is done with the column sums ws.mu held)
        ws.mu[p] = (sdp > 0.0) ? sdp * sdp : 1e300;             // the treated diagonal S_pp: the scale a pivot is measured against (a zero-variance column: no pivot passes -- the block takes the minimum-norm route, weight 0 for it)
        double* A = (kbB & 1) ? ws.stage : ws.inv;              // an odd number of steps ends in ws.inv
        double* An = (kbB & 1) ? ws.inv : ws.stage;
        auto fill_block = [&](double* dst) {                    // row bi of S_bb out of the column registers (S is symmetric)
#pragma unroll
            for (int q = 0; q < PMAX; ++q)
                if (modeb && q >= bb0 && q < bb0 + bk) dst[boffB + bi * bk + (q - bb0)] = s[q];
        };
        // the same with every lane storing every column -- the ones outside its block into a slot of its own behind the pivot rows of the
        // sweep below (an address select instead of an exec-masked branch per column: 7 k -> 2 k clocks)
        auto fill_block_all = [&](double* dst) {
            double* mine_row = dst + boffB + bi * bk - bb0;      // column q of my block lands at mine_row[q]
            double* junk = ws.stage + 2 * LMAX * 16 + p;
            const unsigned bkm = modeb ? (unsigned)bk : 0u;
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                double* d = ((unsigned)(q - bb0) < bkm) ? mine_row + q : junk;
                *d = s[q];
            }
        };
        bool okrow = true;
        ex.mark(20);
        // blocks of at most 16 MVs (round 4): every lane keeps ITS row of the block in registers; step j = the pivot lane publishes its row
        // (double-buffered at the head of the staging area), one exchange, every lane updates its row with straight-line code (compile-time
        // column index, `c == j` a scalar test) -- where the loop over LDS-resident matrices below pays two dependent LDS round trips per
        // column and step (5 k clocks per step at k = 10 against ~1 k).  Same formulas, same pivot test.  KR = registers of a row: 8 / 12 / 16.
        auto sweep_rows = [&](auto krc) {
            constexpr int KR = decltype(krc)::value;
            fill_block_all(ws.inv);
            ex.sync();
            ex.mark(21);
            double row[KR];
            {
                const double* Ib0 = ws.inv + boffB + bi * bk;
#pragma unroll
                for (int c = 0; c < KR; ++c) row[c] = (modeb && c < bk) ? Ib0[c] : 0.0;
            }
            ex.mark(22);
            for (int j = 0; j < kbB; ++j) {
                double* Pj = ws.stage + ((j & 1) * LMAX + lp) * 16;
                if (modeb && bi == j) {
#pragma unroll
                    for (int c = 0; c < KR; ++c) Pj[c] = row[c];
                }
                ex.sync();
                if (modeb && j < bk) {
                    double f = 0.0;
#pragma unroll
                    for (int c = 0; c < KR; ++c) f = (c == j) ? row[c] : f;
                    const double piv = Pj[j];
                    okrow = okrow && (piv > PLSPM_PIVOT_RTOL * ws.mu[bb0 + j]);
                    const double ip = wave_rcp(piv);
                    const bool pivlane = bi == j;
                    const double fip = f * ip;
#pragma unroll
                    for (int c = 0; c < KR; ++c) {
                        const double rjc = Pj[c] * ip;
                        const double off = pivlane ? rjc : row[c] - f * rjc;
                        const double dia = pivlane ? ip : -fip;
                        row[c] = (c == j) ? dia : off;
                    }
                }
            }
            ex.mark(23);
#pragma unroll
            for (int c = 0; c < KR; ++c) if (modeb && c < bk) ws.inv[boffB + bi * bk + c] = row[c];
            ex.sync();
            ex.mark(24);
        };
        if (kbB <= 8) sweep_rows(std::integral_constant<int, 8>{});
        else if (kbB <= 12) sweep_rows(std::integral_constant<int, 12>{});
        else if (kbB <= 16) sweep_rows(std::integral_constant<int, 16>{});
        else {
        fill_block(A);
        ex.sync();
        for (int j = 0; j < kbB; ++j) {
            if (modeb) {
                const double* Ab = A + boffB;
                double* Ob = An + boffB;
                if (j < bk) {
                    const double piv = Ab[j * bk + j];
                    okrow = okrow && (piv > PLSPM_PIVOT_RTOL * ws.mu[bb0 + j]);
                    const double ip = 1.0 / piv;
                    const double f = Ab[bi * bk + j];
                    for (int c = 0; c < bk; ++c) {
                        const double rjc = Ab[j * bk + c] * ip;
                        double v;
                        if (bi == j) v = (c == j) ? ip : rjc;
                        else v = (c == j) ? -f * ip : Ab[bi * bk + c] - f * rjc;
                        Ob[bi * bk + c] = v;
                    }
                } else {
                    for (int c = 0; c < bk; ++c) Ob[bi * bk + c] = Ab[bi * bk + c];      // a smaller block waits out the larger ones' steps
                }
            }
            ex.sync();
            double* t = A; A = An; An = t;
        }
        }
        // rank-deficient blocks, one at a time: S_bb once more from the registers, pseudo-inverse on the block's LV lane (scratch: the staging area)
        const bool any_bad = ex.vote_any(modeb && !okrow);      // (one ballot instead of a descriptor walk when every sweep went through)
        for (int l = 0; any_bad && l < L; ++l) {
            if (md.mode[l] != MODE_B) continue;
            const int k = md.boff[l + 1] - md.boff[l];
            const bool bad_here = ex.vote_any(modeb && lp == l && !okrow);
            if (!bad_here) continue;
            if (lp == l) fill_block(ws.inv);
            ex.sync();
            if (p == l) {
                double* F = ws.inv + md.chol_off[l] / 2;
                bool ok = k > 1 && jacobi_pinv(F, k, ws.stage);
                if (!ok) singular = true;
                for (int r = 0; r < k; ++r) for (int c = 0; c < r; ++c) F[r * k + c] = F[c * k + r];      // jacobi_pinv leaves the upper triangle
            }
            ex.sync();
        }
        ex.mark(25);
    }


    // LV role: normal equations M[f, f] x = M[f, t] over my predecessors f (solver_quad.h: the same lambda)
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

    // ONE loop carries init, the iterations and the finalisation (as solve_problem_wave)
    const double icorr2 = ex.uniform_d((n - 1.0) / n);
    double Qe[NE], wp = valid ? 1.0 : 0.0;
#pragma unroll
    for (int u = 0; u < NE; ++u) Qe[u] = 0.0;
    int iteration = 0, phase = 0;
    double a_prev = 0.0, Ra_prev = 0.0;                          // NM: my entry of the previous step's normalised weights a and of R a (own block)
    if constexpr (NM) {
        // initial weights 1 / sqrt(k_l), scores X a_0 used as they are (weights.py:82-98): no normalising trip
        ex.sync();                                               // (every lane is done with the sigma published in ws.w)
        wp = valid ? 1.0 / sqrt((double)(md.boff[lp + 1] - md.boff[lp])) : 0.0;
        ws.w[p] = valid ? wp : 1.0;
        ex.sync();
        phase = 1;
    }
    while (true) {
        int pl = p, lpl = lp;                                    // opaque copies: LDS addresses recomputed per trip instead of hoisted and spilled
        ex.opaque(pl); ex.opaque(lpl);
#pragma unroll
        for (int u = 0; u < NE; ++u) { dl[u] = dsc[u]; ex.opaque(dl[u]); }
        const int eml = pl % LMAX, er0l = pl / LMAX;
        ex.mark(16);
        if constexpr (W16<LMAX>::TCOPY) ex.template seg_products2<PMAX, 64 * W16<LMAX>::VP * 8>(s, ws.w, P, ends, ws.V + pl * W16<LMAX>::VP, wp);      // (idle lanes: zeros into rows >= P)
        else ex.template seg_products<PMAX>(s, ws.w, P, ends, valid ? ws.V + pl * W16<LMAX>::VP : ex.sink(ws.sink));
        ex.mark(17);
        ex.sync();
        ex.mark(18);
        if constexpr (NE == 1) {
            // Q[el, em] = sum over the MVs p of block el of w_p V[p, em]: eight terms in flight per trip
            double s0 = 0.0, s1 = 0.0;
            int pb = pb0(0);
            ex.opaque(pb);
            const double* vv = ws.V + pb * W16<LMAX>::VP + eml;
            const double* ww = ws.w + pb;
            for (int i0 = 0; i0 < kbmax; i0 += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if constexpr (W16<LMAX>::TCOPY) v[j] = (i0 + j < pk(0)) ? vv[(64 + i0 + j) * W16<LMAX>::VP] : 0.0;      // (the copy: 64 rows behind V)
                    else v[j] = (i0 + j < pk(0)) ? ww[i0 + j] * vv[(i0 + j) * W16<LMAX>::VP] : 0.0;
                }
                s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3]; s0 += v[4]; s1 += v[5]; s0 += v[6]; s1 += v[7];
            }
            Qe[0] = s0 + s1;
            ws.Qm[pl] = pairu(0) ? Qe[0] : 1.0;
        } else {
            // several entries per lane: the trips run over the TERMS, all of the lane's entries side by side -- TT terms of every entry in flight per trip (models of
            // many LVs have small blocks: a trip of eight terms per entry issued 2 x 8 loads per entry for three live terms; 60 x 20: 9.8 k -> 3 k clocks per product).
            // Even terms into one chain, odd terms into the other: the sums of the eight-term form, bit for bit.
            double s0[NE], s1[NE];
#pragma unroll
            for (int u = 0; u < NE; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
            auto terms = [&](auto ttc) {                         // TT terms of every entry per trip
                constexpr int TT = decltype(ttc)::value;
                for (int i0 = 0; i0 < kbmax; i0 += TT) {
#pragma unroll
                    for (int u = 0; u < NE; ++u) {
                        if ((64 / LMAX) * u < L) {               // (uniform: the rows of this entry group exist)
                            const int pb = pb0(u), pkk = pk(u);
                            const double* vv = ws.V + pb * W16<LMAX>::VP + eml;
                            const double* ww = ws.w + pb;
                            double v[TT];
#pragma unroll
                            for (int j = 0; j < TT; ++j) v[j] = (i0 + j < pkk) ? ww[i0 + j] * vv[(i0 + j) * W16<LMAX>::VP] : 0.0;
#pragma unroll
                            for (int j = 0; j < TT; j += 2) { s0[u] += v[j]; s1[u] += v[j + 1]; }
                        }
                    }
                }
            };
            // (by the widest block: a model whose blocks fit one short trip takes the short trip; four entries per lane with wider blocks: one entry after the other,
            //  eight terms per trip -- measured best at 60 x 10 / 60 x 12: 0.115 / 0.117 ms against 0.118 / 0.120 with four and 0.121 / 0.122 with eight terms side by side)
            if (kbmax <= (NE <= 4 ? 4 : 2)) terms(std::integral_constant<int, (NE <= 4 ? 4 : 2)>{});
            else if (NE > 4) terms(std::integral_constant<int, 4>{});
            else {
#pragma unroll
                for (int u = 0; u < NE; ++u) {
                    if ((64 / LMAX) * u < L) {
                        int pb = pb0(u);
                        ex.opaque(pb);
                        const int pkk = pk(u);
                        const double* vv = ws.V + pb * W16<LMAX>::VP + eml;
                        const double* ww = ws.w + pb;
                        for (int i0 = 0; i0 < kbmax; i0 += 8) {
                            double v[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = (i0 + j < pkk) ? ww[i0 + j] * vv[(i0 + j) * W16<LMAX>::VP] : 0.0;
                            s0[u] += v[0]; s1[u] += v[1]; s0[u] += v[2]; s1[u] += v[3]; s0[u] += v[4]; s1[u] += v[5]; s0[u] += v[6]; s1[u] += v[7];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                if ((64 / LMAX) * u < L) {
                    Qe[u] = s0[u] + s1[u];
                    ws.Qm[pl + 64 * u] = pairu(u) ? Qe[u] : 1.0;
                }
            }
        }
        ex.sync();
        ex.mark(19);
        bool nm_first = false;
        if constexpr (NM) {
            // the products above belong to the weights of step `iteration` (unnormalised; step 0: a_0): normalise, bound the criterion of that step, stop or go on
            nm_first = iteration == 0;
            const double wfl = nm_first ? 1.0 : wave_rsqrt(ws.Qm[lpl * LMAX + lpl]);
            const double a_cur = valid ? wp * wfl : 0.0, Ra_cur = valid ? ws.V[pl * W16<LMAX>::VP + lpl] * wfl : 0.0;
            bool stop = false;
            double ub = 0.0;
            if (!nm_first) {
                const double d = a_cur - a_prev, Rd = Ra_cur - Ra_prev;
                ub = n * ex.allsum(d * Rd);                      // n sum_l d_l' R_bb d_l >= sum_il c_i (|y_old| - |y_new|)^2
                stop = (io && io->force_T > 0) ? (iteration >= io->force_T) : (ub * (io ? io->bound_scale : 1.0) < md.tol * (1.0 - 1e-9));
                if (iteration > md.max_iter) stop = true;        // (weights.py:183)
            }
            if (stop) break;
            a_prev = a_cur; Ra_prev = Ra_cur;
            // the map of this step's scores on the uploaded columns: y = sum_p x'_p c_p + k_l,  c_p = a_p / sigma_p,  k_l = -sum_p (mu_p / n) c_p
            const double cmap = a_cur / sdraw;
            if (io && io->maps && valid) io->maps[(long)iteration * (P + L + 1) + pl] = cmap;
            if (io && io->maps && pl == 0) io->maps[(long)iteration * (P + L + 1) + P + L] = ub;      // (how many rows the verification reads for this step)
            nmk[pl] = valid ? (mup * inv_n) * cmap : 0.0;        // (summed per block by the LV lanes behind the next barrier)
        }
        if (!NM && phase == 2) break;
        if (phase == 0) {
            wp = valid ? wave_rsqrt(ws.Qm[lpl * LMAX + lpl]) : 0.0;
            ws.w[pl] = valid ? wp : 1.0;
            ex.sync();
            ex.mark(3);
            phase = 1;
            continue;
        }
        ++iteration;
        ex.mark(9);
        // Yhat_l = Y_l / std1 / corr:  a_l = 1 / (corr2 sqrt(Q_ll)),  G = cov0(Yhat) = a a' o Q   (weights.py:43-44)
        const double rm = wave_rsqrt(ws.Qm[eml * LMAX + eml]), am = NM ? (nm_first ? 1.0 : rm) : rm * icorr2;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if ((64 / LMAX) * u < L) {
                const int ell = er0l + (64 / LMAX) * u;
                const double rl = wave_rsqrt(ws.Qm[ell * LMAX + ell]), al = NM ? (nm_first ? 1.0 : rl) : rl * icorr2;
                const double Ge = al * am * Qe[u];
                double Ee = 0.0;
                if (pairu(u)) {
                    if (md.scheme == SCHEME_PATH) {
                        if (dlm(u) & 1) Ee = Qe[u] * rl * rm;    // column em of E: correlations with the successors of em (scheme.py:51-53)
                    } else if (dlm(u)) {
                        const int d = (dlm(u) & 1) + (dlm(u) >> 1);
                        Ee = (md.scheme == SCHEME_CENTROID) ? ((Ge > 0.0) ? 1.0 : ((Ge < 0.0) ? -1.0 : 0.0)) : Ge * corr2 * (double)d;   // cov1 = cov0 N/(N-1)
                    }
                    ws.Gm[pl + 64 * u] = Ge; ws.Em[pl + 64 * u] = al * Ee;      // (row el of E carries a_el: the outer step multiplies V with a E)
                    if (ell == eml) ws.a[ell] = al;
                }
            }
        }
        ex.sync();
        ex.mark(10);
        if constexpr (NM) {
            if (io && io->maps && lvlane) {                      // k_l of the map stored above (iteration counts the steps taken: this map is index iteration - 1)
                double k0 = 0.0;
                for (int q = md.boff[pl]; q < md.boff[pl + 1]; ++q) k0 += nmk[q];
                io->maps[(long)(iteration - 1) * (P + L + 1) + P + pl] = -k0;
            }
        }
        if (md.scheme == SCHEME_PATH) {
            if (lvlane && nk > 0) {                              // regression of Yhat_t on its predecessors, no intercept (scheme.py:48-50)
                const double* x = regress(ws.Gm);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nk) ws.Em[pred(r) * LMAX + pl] = ws.a[pred(r)] * x[r];
                for (int r = 4; r < nk; ++r) ws.Em[pred(r) * LMAX + pl] = ws.a[pred(r)] * x[r];
            }
            ex.sync();
        }
        ex.mark(11);
        // outer step, Mode A: w = (S Wn E)[p, lv(p)] == X'Z/N  (mode.py:29): row p of V out of LDS
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < LMAX; m += 2) {
            if (m < L) c0 += ws.V[pl * W16<LMAX>::VP + m] * ws.Em[m * LMAX + lpl];
            if (m + 1 < L) c1 += ws.V[pl * W16<LMAX>::VP + m + 1] * ws.Em[(m + 1) * LMAX + lpl];
        }
        double wn = valid ? c0 + c1 : 0.0;
        if constexpr (MODEB) {                                   // Mode B: w_b = S_bb^-1 c_b  (mode.py:51)
            ws.mu[pl] = wn;
            ex.sync();
            if (modeb) {
                const double* Ib = ws.inv + boffB + bi * bk;
                const double* cb = ws.mu + bb0;
                double a0 = 0.0, a1 = 0.0;
                int q = 0;
                for (; q + 1 < bk; q += 2) { a0 += Ib[q] * cb[q]; a1 += Ib[q + 1] * cb[q + 1]; }
                if (q < bk) a0 += Ib[q] * cb[q];
                wn = a0 + a1;
            }
        }
        if constexpr (NM) {
            wp = wn;                                             // (the stop rule of this step: behind the next products, above)
            ws.w[pl] = valid ? wp : 1.0;
            ex.sync();
            ex.mark(12);
        } else {
        const double dd = fabs(wp) - fabs(wn);
        const double conv = ex.allsum(dd * dd);
        wp = wn;
        ws.w[pl] = valid ? wp : 1.0;
        ex.sync();
        ex.mark(12);
        if (conv < md.tol || iteration > md.max_iter) phase = 2;
        }
    }
    const bool not_converged = iteration > md.max_iter;
    ex.mark(4);

    // finalize (weights.py:56-70): wf_l = 1 / sqrt(Q_ll); returned weights never sign-flipped
    const double wfp = wave_rsqrt(ws.Qm[lp * LMAX + lp]);
    wp *= wfp;
    // sign rule: EVERY MV votes (weights.py:62-64); sign(cor[p,l]) == sign(V[p,l]); a zero-variance column votes +1 everywhere (pandas' NaN correlation has its sign bit clear: solver_core.h)
    unsigned negmask = 0u;
    {
        double vr[LMAX];                                         // (my row of V in one batch of loads)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) vr[l] = ws.V[p * W16<LMAX>::VP + (l < L ? l : 0)];
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
            if (!NM && l < L) { const int neg = ex.vote_count(valid && vr[l] < 0.0 && sdp != 0.0); if (P - 2 * neg < 0) negmask |= 1u << l; }      // (non-metric: no sign rule, weights.py:122-133)
    }
    const double sgl = ((negmask >> lp) & 1u) ? -1.0 : 1.0;
    const double vlp = ws.V[p * W16<LMAX>::VP + lp];                    // V[p, lv(p)] for the loading
    double* const Cs = ws.Gm;                                    // (the iteration's G and E are dead: the score covariance and the path matrix take their places,
    double* const Bm = ws.Em;                                    //  the indirect effects the place of Q once the covariance is formed)
    double* const Ind = ws.Qm;
    {
        const double wfm = wave_rsqrt(ws.Qm[em * LMAX + em]), sm = ((negmask >> em) & 1u) ? -1.0 : 1.0;
        double cs[NE];
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int el = er0 + (64 / LMAX) * u;
            const double wfl = wave_rsqrt(ws.Qm[el * LMAX + el]), sl = ((negmask >> el) & 1u) ? -1.0 : 1.0;
            cs[u] = sl * sm * wfl * wfm * Qe[u];                 // population covariance of the sign-corrected scores
        }
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if (pairu(u)) Cs[t + 64 * u] = cs[u];
            Bm[t + 64 * u] = 0.0;                                // (all 256 entries: the effects below run without a test per row)
        }
    }
    ex.sync();
    ex.mark(5);
    // inner model (inner_model.py:58-75): OLS with intercept == centred normal equations on the score covariance
    double r2p = 0.0;
    if (lvlane) {
        if (nk > 0) {
            const double* x = regress(Cs);
            double expl = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk) { const int fr = pred(r); Bm[t * LMAX + fr] = x[r]; expl += x[r] * Cs[fr * LMAX + t]; }
            for (int r = 4; r < nk; ++r) { const int fr = pred(r); Bm[t * LMAX + fr] = x[r]; expl += x[r] * Cs[fr * LMAX + t]; }
            r2p = expl / Cs[t * LMAX + t];
        }
    }
    ex.sync();
    ex.mark(6);
    // effects (inner_model.py:33-53): column t of (I - B)^-1 by forward substitution in the registers of LV lane t (solver_quad.h: colx = the column
    // without its unit entry; rows >= L of B are zero)
    if (lvlane) {
        double colx[LMAX];
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            double i0 = 0.0, i1 = 0.0;
#pragma unroll
            for (int k = 0; k + 1 < i; k += 2) { i0 += Bm[i * LMAX + k] * colx[k]; i1 += Bm[i * LMAX + k + 1] * colx[k + 1]; }
            if (i & 1) i0 += Bm[i * LMAX + i - 1] * colx[i - 1];
            const double ind = (i > t) ? i0 + i1 : 0.0;
            colx[i] = (i > t) ? Bm[i * LMAX + t] + ind : 0.0;
            Ind[i * LMAX + t] = ind;
        }
    }
    ex.sync();
    ex.mark(7);
    // outputs: the bootstrap record  weights | r2 | total | direct | loadings | status | iterations  (bootstrap.py:58-64)
    if (out.row) {
        if (valid) {
            out.row[p] = wp;
            out.row[P + L + 2 * ne + p] = (sdp > 0.0) ? sgl * vlp * wfp / sdp : 0.0;      // (a zero-variance column: the reference's loading is 0, solver_core.h treated_sd)
        }
        if (lvlane) out.row[P + t] = r2p;
        for (int e = t; e < ne; e += 64) {                       // (up to 120 effects at 16 LVs)
            const int idx = md.eff_to[e] * LMAX + md.eff_from[e];
            out.row[P + L + e] = Bm[idx] + Ind[idx];
            out.row[P + L + ne + e] = Bm[idx];
        }
    }
    const bool bad = ex.vote_any((valid && !(isfinite(wp) && isfinite(sdp) && sdp >= 0.0)) || (lvlane && !isfinite(r2p)));
    const bool sing = ex.vote_any(singular);
    if (t == 0) {
        int st = sing ? ST_SINGULAR : (not_converged ? ST_NOT_CONVERGED : ST_OK);
        if (st == ST_OK && bad) st = ST_NONFINITE;
        if (out.status) *out.status = st;
        if (out.iters) *out.iters = iteration;
        if (out.row) { out.row[2 * P + L + 2 * ne] = (double)st; out.row[2 * P + L + 2 * ne + 1] = (double)iteration; }
        if constexpr (NM) { if (io && io->steps) *io->steps = iteration; }
    }
    ex.mark(13);
}

}  // namespace plspm
