// solver_wave.h -- the metric PLS-PM solver as ONE 64-lane wave per problem, written for the wave instead of for a generic
// thread group: the batched solver of bootstrap replicates for the model class the reference's own
// examples and BASELINE.json's headline live in -- at most 64 MVs, at most LMAX = 8 LVs; Mode A blocks, and (round 4) Mode B blocks whose
// k x k inverses fit the staging area together (sum k^2 <= 1,056: e.g. six blocks of 13, one of 32).
// Everything else keeps solve_problem_rows / solve_problem (solver_core.h), whose arithmetic this restates:
//   Config.treat              plspm/config.py:299-305, util.treat plspm/util.py:33-39   -> treat block
//   _MetricWeights.__init__   plspm/weights.py:28-39                                     -> init (block products with w = 1)
//   _MetricWeights.iterate    plspm/weights.py:41-54                                     -> iteration loop
//   Scheme.*.calculate        plspm/scheme.py:27-28, 36-37, 45-54                        -> inner weights on the pair lanes / LV lanes
//   _ModeA.outer_weights_metric plspm/mode.py:28-29                                      -> outer step
//   _ModeB.outer_weights_metric plspm/mode.py:50-52 (lstsq of z on the block)            -> outer step: w_b = S_bb^-1 (S Wn E)_b with the inverse
//                                                                                           of every Mode-B block formed ONCE per problem
//   WeightsCalculatorFactory.calculate plspm/weights.py:172-187                          -> stop rule
//   _MetricWeights.calculate  plspm/weights.py:56-70                                     -> finalize, sign rule
//   InnerModel / _effects     plspm/inner_model.py:58-75, 33-53                          -> inner model, effects
//   bootstrap row             plspm/bootstrap.py:58-64                                   -> outputs
//
// Why a second formulation.  solve_problem_rows spends ~75 k clocks per problem, of which the O(P^2) products are 12 k: the rest are
// ~50 `par` phases of a few dozen flops each, every one an LDS round trip + a loop with run-time bounds + index divisions by the
// run-time L (profiles/r02c_pmc.md: 46 % of the wave cycles parked at s_waitcnt, 828 scalar branches per problem).  Here the lanes
// of the wave take three fixed roles, all indices are shifts by the compile-time LMAX, and a phase boundary exists only where data
// really change lanes:
//   MV lane p  < P         column p of the treated covariance in 64 register pairs, its weight, its row of V = S W
//   pair lane e = 8 l + m  entry (l, m) of every L x L matrix (Q, G, E, score covariance): the "par(L*L)" phases are straight-line
//   LV lane i  < L         the small regressions of LV i (PATH scheme, inner model) and column i of (I - B)^-1
// An iteration is six exchanges through LDS (rows solver: fourteen): S W -> [T = w V transposed] -> Q -> [G, E, a] -> (PATH: the
// regressions) -> outer step + stop-rule sum (a butterfly, no LDS) -> w.
//
// Executor (device: kernels_solver.h DevWaveExec; CPU emulation: tests/hostemu HostExec): ex.tid = lane, ex.sync() = all lanes have
// passed, LDS writes visible; ex.allsum(v) = butterfly sum, bitwise identical on every lane; ex.vote_count / ex.vote_any = ballots;
// ex.load_cov = the column loader; ex.seg_products = the segmented multiply-add stream of solve_problem_rows.
#pragma once
#include "solver_core.h"
#include <type_traits>

namespace plspm {

template <int LMAX>
struct WaveWs {
    double* V;       // [64 * LMAX]   V[p * L + m] (run-time pitch L: the rows seg_products writes); head of `stage`
    double* T;       // [LMAX * 64]   T[m * 64 + p] = w_p V[p, m]
    double* stage;   // [16 * 66]     aliases V and T (+ 32 doubles of pad): transpose staging of the column loader; scratch of the regressions
    double* w;       // [64]
    double* mu;      // [64]
    double *Qm, *Gm, *Em, *Cs, *Bm, *Ind;      // [LMAX * LMAX], entry (l, m) at l * LMAX + m
    double *a, *r2;  // [LMAX]
    double* sink;    // [LMAX] where the idle lanes of seg_products store
    double* inv;     // [sum over the Mode-B blocks of k^2] (S_bb)^-1 (or the pseudo-inverse of a rank-deficient block), full k x k, block l at chol_off[l] / 2
};
// (+ n_chol / 2 doubles behind it for the Mode-B inverses: ModelDesc::n_chol counts chol_block_doubles(k) = 2 k^2 per Mode-B block)
template <int LMAX> PLSPM_HD constexpr long wave_ws_doubles() { return 16 * 66 + 64 + 64 + 6 * LMAX * LMAX + 2 * LMAX + LMAX; }
template <int LMAX> PLSPM_HD constexpr long wave_ws_doubles(int n_chol) { return wave_ws_doubles<LMAX>() + n_chol / 2; }
template <int LMAX> PLSPM_HD void wave_carve(WaveWs<LMAX>& ws, double* base) {
    static_assert(2 * 64 * LMAX <= 16 * 66, "V and T fit the staging area");
    static_assert(LMAX * (2 * (LMAX - 1) * (LMAX - 1) + (LMAX - 1)) <= 16 * 66, "so does the scratch of LMAX regressions on LMAX - 1 predecessors");
    double* p = base;
    ws.stage = p; ws.V = p; ws.T = p + 64 * LMAX; p += 16 * 66;
    ws.w = p; p += 64; ws.mu = p; p += 64;
    ws.Qm = p; p += LMAX * LMAX; ws.Gm = p; p += LMAX * LMAX; ws.Em = p; p += LMAX * LMAX;
    ws.Cs = p; p += LMAX * LMAX; ws.Bm = p; p += LMAX * LMAX; ws.Ind = p; p += LMAX * LMAX;
    ws.a = p; p += LMAX; ws.r2 = p; p += LMAX;
    ws.sink = p; p += LMAX;
    ws.inv = p;
}
// What the wave solver covers (the host asks before it launches; everything else takes solve_problem_rows / solve_problem).
// Mode-B blocks: their inverses are formed by an out-of-place Gauss-Jordan sweep that ping-pongs between ws.inv and the staging area.
template <int LMAX> PLSPM_HD bool wave_solver_covers(int P, int L, int n_chol) { return P >= 1 && P <= 64 && L >= 1 && L <= LMAX && n_chol / 2 <= 16 * 66; }

// 1 / sqrt(x) and 1 / x to the last bit or two (not correctly rounded): v_rsq_f64 / v_rcp_f64 seeds + Newton steps on the device -- a
// fraction of the dependent instructions of the IEEE sqrt + divide sequences, which sit on the critical path of every small phase.
// The CPU emulation runs the SAME refinement on a seed of the hardware's accuracy (the exact value rounded to fp32's 24 bits, exponent kept
// apart so that any double is in range): what the emulation tests then hold against the oracle is this arithmetic, not the IEEE divide.
PLSPM_HD double wave_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);                           // ~2^-26 relative
#else
    double r;
    if (!(x > 0.0) || x > 1.7976931348623157e308) r = 1.0 / sqrt(x);      // 0, negative, inf, NaN: what the instruction returns
    else {
        int e;
        double m = frexp(x, &e);                                  // x = m 2^e, m in [1/2, 1)
        if (e & 1) { m *= 2.0; --e; }                             // even exponent: sqrt(2^e) is a power of two
        r = ldexp((double)(float)(1.0 / sqrt(m)), -e / 2);
    }
#endif
    const double hx = 0.5 * x;
    r = fma(r, fma(-hx * r, r, 0.5), r);                          // r (1 + (1/2 - x r^2 / 2))
    r = fma(r, fma(-hx * r, r, 0.5), r);
    return r;
}
PLSPM_HD double wave_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
#else
    double r;
    if (!(fabs(x) > 0.0) || fabs(x) > 1.7976931348623157e308) r = 1.0 / x;
    else {
        int e;
        const double m = frexp(x, &e);
        r = ldexp((double)(float)(1.0 / m), -e);
    }
#endif
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
// Normal equations M[f, f] x = M[f, col], f = the k <= 4 indices packed one per byte in `fpack`, M an L x L matrix with pitch ld in the
// workspace: square-root-free factorisation A = U' D U in registers (as solver_core.h spd_solve_fixed; identity padding beyond k),
// reciprocals by wave_rcp.  Returns false at a pivot that is not safely positive (the caller takes the minimum-norm route).
PLSPM_HD bool wave_ldl4(const double* M, int ld, unsigned fpack, int k, int col, double* x) {
    constexpr int K = 4;
    double A[K][K], b[K];
    int f[K];
#pragma unroll
    for (int r = 0; r < K; ++r) f[r] = (r < k) ? (int)((fpack >> (8 * r)) & 255u) : 0;
#pragma unroll
    for (int r = 0; r < K; ++r) {
#pragma unroll
        for (int c = r; c < K; ++c) A[r][c] = (r < k && c < k) ? M[f[r] * ld + f[c]] : ((r == c) ? 1.0 : 0.0);
        b[r] = (r < k) ? M[f[r] * ld + col] : 0.0;
    }
    bool ok = true;
    double T[K][K], invd[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double ajj = A[j][j];
        double d = ajj;
#pragma unroll
        for (int r = 0; r < j; ++r) d -= A[r][j] * T[r][j];
        ok = ok && (d > PLSPM_PIVOT_RTOL * ajj);
        invd[j] = wave_rcp(d);
#pragma unroll
        for (int c = j + 1; c < K; ++c) {
            double t = A[j][c];
#pragma unroll
            for (int r = 0; r < j; ++r) t -= A[r][j] * T[r][c];
            T[j][c] = t;
            A[j][c] = t * invd[j];
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double t = b[i];
#pragma unroll
        for (int r = 0; r < i; ++r) t -= A[r][i] * b[r];
        b[i] = t;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) b[i] *= invd[i];
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        double t = b[i];
#pragma unroll
        for (int c = i + 1; c < K; ++c) t -= A[i][c] * b[c];
        b[i] = t;
    }
#pragma unroll
    for (int r = 0; r < K; ++r) if (r < k) x[r] = b[r];
    return ok;
}

// Md: the DENSE moment matrix [(P+1) x cov_ld(P)] of the mean-shifted columns + ones, upper triangle (entry (r, c >= r) at r * PS + c),
// as the int8 digit-plane Gram writes it.  Outputs: out.row / out.status / out.iters (a bootstrap record).
// MODEB: the model has Mode-B blocks (an instantiation of its own: the all-Mode-A code -- the headline's -- stays as it was, register for register)
template <int LMAX, bool MODEB = false, class Ex>
PLSPM_HD void solve_problem_wave(Ex& ex, const ModelDesc& md, const WaveWs<LMAX>& ws, const double* Md, const FitOutputs& out) {
    constexpr int PMAX = 64, LL = LMAX * LMAX, WAVE_REG_SCRATCH = 2 * (LMAX - 1) * (LMAX - 1) + (LMAX - 1);
    static_assert(LL <= 64, "one pair lane per entry of an L x L matrix");
    const int P = md.P, L = md.L, PS = cov_ld(P), p = ex.tid;
    const bool mine = p < P;
    const int pc = mine ? p : P - 1;
    const int lp = md.lvof[pc];                                  // MV role: my LV
    const int el = p / LMAX, em = p % LMAX;                      // pair role: entry (el, em)
    const bool pair = el < L && em < L && p < LL;
    const int elc = pair ? el : 0;
    const int pb0 = md.boff[elc], pk = pair ? md.boff[elc + 1] - pb0 : 0;      // pair role: the block of row el
    const bool lvlane = p < L;                                   // LV role
    // model descriptors this lane consults inside the loop: fetched once (they live in global memory), kept in registers
    const bool c_lm = pair && md.C[el * L + em] != 0;             // LV em -> LV el: el is a successor of em
    const int d_lm = pair ? (int)md.C[el * L + em] + (int)md.C[em * L + el] : 0;
    int nk = 0;                                                  // LV role: my predecessors, one byte each
    unsigned fpack = 0u;
    if (lvlane) {
        const int o = md.pred_off[p];
        nk = md.pred_off[p + 1] - o;
        for (int r = 0; r < 4; ++r) if (r < nk) fpack |= (unsigned)md.pred_idx[o + r] << (8 * r);
    }
    const int ne = md.n_eff;
    const int eidx = (p < ne) ? md.eff_to[p] * LMAX + md.eff_from[p] : 0;
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
    double dpp = 0.0, mup = 0.0;                                              // raw M[p][p] and the column sum M[p][P] (ones column)
    const double n = Md[(long)P * PS + P];
    ex.template load_cov<PMAX>(Md, PS, P, s, ws.stage, mup, dpp);
    if (!mine) { mup = 0.0; dpp = 0.0; }
    ex.mark(1);
    ws.mu[p] = mup;
    ws.w[p] = 1.0;                                               // init: block products with w = 1
    ex.sync();
    const double inv_n = 1.0 / n;
    double fac = inv_n;
    if (md.scaled) {
        // g = std1(all N*P raw values) * sqrt((N-1)/N)   (config.py:302), evaluated around the grand mean
        const double tot = ex.allsum(mine ? mup + n * shp : 0.0);
        const double np_ = n * (double)P, grand = tot / np_;
        const double d = shp - grand;
        const double ss = ex.allsum(mine ? dpp + 2.0 * d * mup + n * d * d : 0.0);
        const double g2 = ss / (np_ - 1.0) * ((n - 1.0) / n);
        fac = 1.0 / (n * g2);
    }
    // mu_q = the column sum lane q holds: a lane broadcast (v_readlane into scalar registers; the CPU emulation reads the published copy).
    // (eight columns per trip, a scheduling fence in front of the loop and between trips: left alone the compiler hoists the 64
    // broadcasts above the reductions that produce `fac` and spills them)
    ex.fence();
#pragma unroll
    for (int q0 = 0; q0 < PMAX; q0 += 8) {
#pragma unroll
        for (int q = q0; q < q0 + 8; ++q) {
            const double v = (s[q] - (mup * ex.bcast(mup, (q < P) ? q : P - 1, ws.mu)) * inv_n) * fac;      // (mu_p mu_q) first: bitwise symmetric in (p, q)
            s[q] = (q < P) ? v : 0.0;
        }
        ex.pin8(s[q0], s[q0 + 1], s[q0 + 2], s[q0 + 3], s[q0 + 4], s[q0 + 5], s[q0 + 6], s[q0 + 7]);     // results final before the next loads issue
    }
    const double sdp = treated_sd(dpp, mup, inv_n, fac);           // (0: a column that is constant in this replicate -- solver_core.h)
    // (its row of the covariance is rounding residue: exact zeros instead -- an LV whose only item is that column then has a score variance of exactly 0 and fails as the
    //  reference's does, instead of normalising the residue; rare: one ballot, the loop runs for the waves that hold such a column)
    if (ex.vote_any(mine && sdp == 0.0)) {
        const double keep = (sdp == 0.0) ? 0.0 : 1.0;
#pragma unroll
        for (int q = 0; q < PMAX; ++q) s[q] *= keep;
    }
    // (loop constants in SCALAR registers: every lane holds the same value; as vector registers they were spilled around the loop)
    const double corr2 = ex.uniform_d(n / (n - 1.0));
    ex.mark(2);


    // Mode-B blocks (mode.py:50-52: w_b = argmin |X_b w - z| = S_bb^-1 (X_b' z / N)): S_bb does not change over the iterations, so its
    // inverse is formed once.  Gauss-Jordan sweep without pivoting (S_bb is positive definite; pivot j is the same Schur complement the
    // Cholesky factorisation of solver_core.h tests, so a rank-deficient block is recognised by the same rule), every MV lane of a Mode-B
    // block owning ROW i of its k x k matrix, all blocks at once, out of place: step j reads A, writes A' -- no lane reads what another
    // rewrites in the same step, ONE exchange per step -- ping-ponging between ws.inv and the staging area (free until the first S W).
    //     i == j:  A'[j][c] = A[j][c] / piv  (c != j),  A'[j][j] = 1 / piv
    //     i != j:  A'[i][c] = A[i][c] - A[i][j] A[j][c] / piv  (c != j),  A'[i][j] = -A[i][j] / piv
    // A block whose sweep meets a pivot that is not safely positive takes the minimum-norm route of the reference's gelsd (jacobi_pinv, on
    // its LV lane, one such block at a time in the staging area) -- the outer step multiplies with the full symmetric matrix either way.
    const bool modeb = MODEB && mine && md.mode[lp] == MODE_B;
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

    // LV role: normal equations M[f, f] x = M[f, p] over my predecessors f -- up to four in registers (wave_ldl4: every LV lane runs the
    // same straight-line code, identity-padded), five to LMAX - 1 by Cholesky in LDS scratch; a rank-deficient system takes the
    // minimum-norm answer of the reference's pinv / gelsd (solver_core.h pinv_solve).  Scratch and x live in the staging area: V and T
    // are dead between the Q exchange and the next S W.
    const int* fglob = md.pred_idx + (lvlane ? md.pred_off[p] : 0);
    auto pred = [&](int r) { return r < 4 ? (int)((fpack >> (8 * r)) & 255u) : fglob[r]; };
    auto regress = [&](const double* M) {
        double* scratch = ws.stage + p * WAVE_REG_SCRATCH;
        double* x = scratch + 2 * (LMAX - 1) * (LMAX - 1);
        bool ok;
        if (nk <= 4) {
            unsigned fp = fpack;
            ex.opaque(fp);            // the 14 gather addresses are recomputed here (a few integer operations) instead of living -- spilled -- across the loop
            ok = wave_ldl4(M, LMAX, fp, nk, p, x);
        } else {
            for (int r = 0; r < nk; ++r) {
                for (int c = 0; c < nk; ++c) scratch[r * nk + c] = M[fglob[r] * LMAX + fglob[c]];
                x[r] = M[fglob[r] * LMAX + p];
            }
            ok = chol_factor(scratch, nk);
            if (ok) chol_solve(scratch, nk, x);
        }
        if (!ok && !(nk > 1 && pinv_solve(M, LMAX, fglob, nk, p, x, scratch))) singular = true;
        return x;
    };

    // ONE loop carries init, the iterations and the finalisation: each trip starts with V = S W (row p in Vp) and Q = W' S W (entry
    // (el, em) in Qe and in ws.Qm) for the weights in ws.w / wp -- a single copy of the product code in the instruction stream.
    //   phase 0  init (weights.py:28-39): w_p = corr / std1(sum of the block's MVs) = 1 / sqrt(sum(S_bb)) = 1 / sqrt(Q_ll) at w = 1
    //   phase 1  iterations (weights.py:41-54, stop rule weights.py:179-186)
    //   phase 2  the product of the final weights (weights.py:56-70), then out of the loop
    const double icorr2 = ex.uniform_d((n - 1.0) / n);
    double Vp[LMAX], Qe = 0.0, wp = mine ? 1.0 : 0.0;
    int iteration = 0, phase = 0;
    while (true) {
        // the lane's coordinates once more, opaque to the optimiser: every LDS address of the loop body is recomputed from them (a
        // few integer operations per trip) instead of being hoisted out of the loop and -- 256 registers hold s[] and little else --
        // spilled to scratch, whose reloads inside the loop cost an L2 round trip each
        int pl = p, lpl = lp, pb0l = pb0;
        ex.opaque(pl); ex.opaque(lpl); ex.opaque(pb0l);
        const int ell = pl / LMAX, eml = pl % LMAX;
        ex.mark(16);
        ex.template seg_products<PMAX>(s, ws.w, P, ends, mine ? ws.V + pl * L : ex.sink(ws.sink));      // (own row: no exchange)
#pragma unroll
        for (int m = 0; m < LMAX; ++m) {
            Vp[m] = (mine && m < L) ? ws.V[pl * L + m] : 0.0;
            if (m < L) ws.T[m * 64 + pl] = wp * Vp[m];
        }
        ex.mark(17);
        ex.sync();
        ex.mark(18);
        {
            // Q[el, em] = sum over the MVs of block el of T[em, .]: eight loads in flight per trip (one LDS latency per eight terms)
            double s0 = 0.0, s1 = 0.0;
            const double* t = ws.T + eml * 64 + pb0l;
            for (int i0 = 0; i0 < kbmax; i0 += 8) {
                double v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (i0 + j < pk) ? t[i0 + j] : 0.0;
                s0 += v[0]; s1 += v[1]; s0 += v[2]; s1 += v[3]; s0 += v[4]; s1 += v[5]; s0 += v[6]; s1 += v[7];
            }
            Qe = s0 + s1;
        }
        if (p < LL) ws.Qm[pl] = pair ? Qe : 1.0;
        ex.sync();
        ex.mark(19);
        if (phase == 2) break;
        if (phase == 0) {
            wp = mine ? wave_rsqrt(ws.Qm[lpl * LMAX + lpl]) : 0.0;
            ws.w[pl] = mine ? wp : 1.0;
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
                // column em of E: correlations with the successors of em (scheme.py:51-53) -- cor(Yhat) = cor(Y) = Q o r r'; the
                // regression entries follow below
                if (c_lm) Ee = Qe * rl * rm;
            } else if (d_lm) {
                Ee = (md.scheme == SCHEME_CENTROID) ? ((Ge > 0.0) ? 1.0 : ((Ge < 0.0) ? -1.0 : 0.0)) : Ge * corr2 * (double)d_lm;   // cov1 = cov0 N/(N-1)
            }
            ws.Gm[pl] = Ge; ws.Em[pl] = Ee;
            if (el == em) ws.a[ell] = al;
        }
        ex.sync();
        ex.mark(10);
        if (md.scheme == SCHEME_PATH) {
            if (lvlane && nk > 0) {                              // regression of Yhat_p on its predecessors, no intercept (scheme.py:48-50)
                const double* x = regress(ws.Gm);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < nk) ws.Em[pred(r) * LMAX + pl] = x[r];
                for (int r = 4; r < nk; ++r) ws.Em[pred(r) * LMAX + pl] = x[r];
            }
            ex.sync();
        }
        ex.mark(11);
        // outer step, Mode A: w = (S Wn E)[p, lv(p)] == X'Z/N  (mode.py:29)
        double c0 = 0.0, c1 = 0.0;
#pragma unroll
        for (int m = 0; m + 1 < LMAX; m += 2) {
            if (m < L) c0 += ws.a[m] * Vp[m] * ws.Em[m * LMAX + lpl];
            if (m + 1 < L) c1 += ws.a[m + 1] * Vp[m + 1] * ws.Em[(m + 1) * LMAX + lpl];
        }
        if (LMAX & 1) { if (LMAX - 1 < L) c0 += ws.a[LMAX - 1] * Vp[LMAX - 1] * ws.Em[(LMAX - 1) * LMAX + lpl]; }
        double wn = mine ? c0 + c1 : 0.0;
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
        const double dd = fabs(wp) - fabs(wn);
        const double conv = ex.allsum(dd * dd);
        wp = wn;
        ws.w[pl] = mine ? wp : 1.0;
        ex.sync();
        ex.mark(12);
        if (conv < md.tol || iteration > md.max_iter) phase = 2;
    }
    const bool not_converged = iteration > md.max_iter;
    ex.mark(4);

    // finalize (weights.py:56-70): wf_l = 1 / (std1(X w_l) / corr) = 1 / sqrt(Q_ll); returned weights never sign-flipped
    const double wfp = wave_rsqrt(ws.Qm[lp * LMAX + lp]);
    wp *= wfp;
    // sign rule: EVERY MV votes (weights.py:62-64); sign(cor[p,l]) == sign(V[p,l]); a zero-variance column votes +1 everywhere (pandas' NaN correlation has its sign bit clear: solver_core.h)
    unsigned negmask = 0u;
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
        if (l < L) { const int neg = ex.vote_count(mine && Vp[l] < 0.0 && sdp != 0.0); if (P - 2 * neg < 0) negmask |= 1u << l; }
    {
        const double wfl = wave_rsqrt(ws.Qm[el * LMAX + el]), wfm = wave_rsqrt(ws.Qm[em * LMAX + em]);
        const double sl = ((negmask >> el) & 1u) ? -1.0 : 1.0, sm = ((negmask >> em) & 1u) ? -1.0 : 1.0;
        if (pair) ws.Cs[p] = sl * sm * wfl * wfm * Qe;           // population covariance of the sign-corrected scores
    }
    ex.sync();
    ex.mark(5);
    // inner model (inner_model.py:58-75): OLS with intercept == centred normal equations on the score covariance
    double r2p = 0.0;
    if (lvlane) {
#pragma unroll
        for (int j = 0; j < LMAX; ++j) ws.Bm[p * LMAX + j] = 0.0;
        if (nk > 0) {
            const double* x = regress(ws.Cs);
            double expl = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nk) { const int fr = pred(r); ws.Bm[p * LMAX + fr] = x[r]; expl += x[r] * ws.Cs[fr * LMAX + p]; }
            for (int r = 4; r < nk; ++r) { const int fr = pred(r); ws.Bm[p * LMAX + fr] = x[r]; expl += x[r] * ws.Cs[fr * LMAX + p]; }
            r2p = expl / ws.Cs[p * LMAX + p];
        }
    }
    ex.sync();
    ex.mark(6);
    // effects (inner_model.py:33-53): indirect = B^2 + B^3 + ... = (I - B)^-1 - I - B; B is strictly lower triangular in path order, so
    // column j of (I - B)^-1 follows by forward substitution (the column stays in the registers of LV lane j)
    if (lvlane) {
        double col[LMAX];
#pragma unroll
        for (int i = 0; i < LMAX; ++i) {
            double ind = 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k)
                if (k > p && i < L) ind += ws.Bm[i * LMAX + k] * col[k];
            col[i] = (i == p) ? 1.0 : ((i > p && i < L) ? ws.Bm[i * LMAX + p] + ind : 0.0);
            if (i < L) ws.Ind[i * LMAX + p] = ind;
        }
    }
    ex.sync();
    ex.mark(7);
    // outputs: the bootstrap record  weights | r2 | total | direct | loadings | status | iterations  (bootstrap.py:58-64)
    if (out.row) {
        if (mine) {
            const double sgl = ((negmask >> lp) & 1u) ? -1.0 : 1.0;
            out.row[p] = wp;
            double vl = 0.0;                                     // V[p, lv(p)]: the LDS copy may have served as regression scratch
#pragma unroll
            for (int m = 0; m < LMAX; ++m) vl = (m == lp) ? Vp[m] : vl;
            out.row[P + L + 2 * ne + p] = (sdp > 0.0) ? sgl * vl * wfp / sdp : 0.0;       // (a zero-variance column: the reference's loading is 0, solver_core.h treated_sd)
        }
        if (lvlane) out.row[P + p] = r2p;
        if (p < ne) {
            out.row[P + L + p] = ws.Bm[eidx] + ws.Ind[eidx];
            out.row[P + L + ne + p] = ws.Bm[eidx];
        }
    }
    const bool bad = ex.vote_any((mine && !(isfinite(wp) && isfinite(sdp) && sdp >= 0.0)) || (lvlane && !isfinite(r2p)));
    const bool sing = ex.vote_any(singular);
    if (p == 0) {
        int st = sing ? ST_SINGULAR : (not_converged ? ST_NOT_CONVERGED : ST_OK);
        if (st == ST_OK && bad) st = ST_NONFINITE;
        if (out.status) *out.status = st;
        if (out.iters) *out.iters = iteration;
        if (out.row) { out.row[2 * P + L + 2 * ne] = (double)st; out.row[2 * P + L + 2 * ne + 1] = (double)iteration; }
    }
    ex.mark(13);
}

}  // namespace plspm
