// kernels_nmw.h -- Device kernels, part 4b (round 5): one iteration of the categorical (Scale.ORD / NOM, optimal scaling) solver as ONE 64-lane WAVE per
// problem.  Included by plspm_nonmetric.hip behind kernels_nonmetric.h; device code only.
//
// Reference: _NonmetricWeights.iterate (plspm/weights.py:107-120) with the Scale operators (plspm/scale.py:41-89: ORD with the two-direction monotone
// pooling `_ordinalize`, NOM), the Mode-A non-metric outer step (plspm/mode.py:31-42) and the inner schemes (plspm/scheme.py) -- the arithmetic of
// solver_nmg.h nmg_step, which this restates for the model class the categorical bootstrap lives in: every MV ORD / NOM (all device columns 0/1
// indicators: the moment matrix is a matrix of co-occurrence counts, kept as uint16), every block Mode A, at most 64 MVs of at most CMAX categories
// (instantiated for 8 and for 16 = CMAX_MAX), at most LMAX = 8 LVs, at most 511 indicator columns.  Everything else keeps nmg_kernel<1>.
//
// Why.  nmg_kernel<1> runs ~50 dependent small phases per step on a 256-thread workgroup whose LDS footprint (54 KB) leaves two workgroups per CU:
// 78 % of its wave cycles are waits, its VALUs are 14 % busy (profiles/r04); the one phase that moves data -- V = Mn c, the whole
// count matrix once per iteration -- reaches ~2 TB/s in aggregate.  Here the lanes of a wave take fixed roles and the state of a step lives in
// registers and ~12 KB of LDS, so that eight and more problems share a CU and one problem's round trips hide under its neighbours' streaming:
//   column lane g       aug columns 8 g .. 8 g + 7 of every matrix-vector product: V[., m] and MZ[., l] are 8 x LMAX registers of the lane; a row of
//                       the count matrix reaches the wave as ONE 16-byte load per lane (rows are contiguous: coalesced)
//   MV lane p           manifest variable p: its <= 8 category frequencies, category means, pooled quantification -- registers, fully unrolled
//                       (the per-MV scratch of nmg_step: 19 KB of LDS at 60 five-point items)
//   pair lane e         entry (l, m) of the L x L matrices (YY, G, E): through solver_core.h inner_weights, unchanged
// A step is: V = Mn c (stream 1) -> YY (segmented wave sums) -> inner weights -> category sums of z (the own-LV column of V E) -> category means -> quantification
// (pooling in registers) -> outer weights, where the block quadratic form w' <MV, MV'> w is ONE more matrix-vector product on the BLOCK DIAGONAL of
// the count matrix (stream 2: U = Mn_ll d, d = w_p tq_j; q_l = d' U) instead of k^2 pair moments of C^2 scattered loads each -> score map.
// The prepare (first launch) and the finish stay nmg_kernel<0> / <2>; the state between launches is the same NmState / NmgExtra layout.
// Same expressions as nmg_step up to the order of the sums that cross lanes (segmented wave sums; the quadratic form): records agree to ~1e-13,
// iteration counts are equal (tests/test_gpu_categorical.py).
#pragma once
#include <type_traits>

namespace nmw {

constexpr int LMAX_MAX = 8, CMAX_MAX = 16, CPL8 = 8;      // LVs (the kernel is instantiated for LMAX = 2, 4, 6, 8), categories per MV (CMAX = 8; 16: ten-point items -- the reference's own
                                                         // mobi / ECSI example data -- at one wave per SIMD), columns per lane

// LDS of one problem (doubles): c | tq | mean | mzown (each QP = Q + 1 rounded up to 8), then the small arrays, then c_old (QP)
__host__ __device__ inline long lds_doubles(int Q, int Pm, int L, int kmax) {
    const long QP = (Q + 1 + 7) & ~7L;
    const long step = 3L * L * L + 6L * L + 2L * Pm + (long)L * regression_scratch_doubles(kmax) + 8 + 16;
    const long fin = workspace_small_doubles(Pm, L, kmax, 0) + Pm;           // the fused finish: MV-level workspace of finish_problem + one row of the MV moment matrix
    return 4 * QP + (step > fin ? step : fin) + QP;             // (+ QP, round 6: the old score map beside the new one for the step's own bound)
}

// CPL consecutive uint16 counts of a row as packed dwords: 8 columns per lane (one 16-byte load; up to 511 aug columns) or -- round 6 -- 6 (one 12-byte load; up to 383
// columns): the count streams are VALU-bound and 300 columns at 8 per lane leave 26 of the 64 lanes without work; at 6 per lane 51 lanes share it.
template <int CPL> struct CountRow;
template <> struct CountRow<8> {
    static constexpr int W = 4;
    unsigned w[4];
    __device__ __forceinline__ void load(const unsigned short* p) { const uint4 v = *reinterpret_cast<const uint4*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    __device__ __forceinline__ void zero() { w[0] = w[1] = w[2] = w[3] = 0u; }
};
template <> struct CountRow<6> {
    static constexpr int W = 3;
    unsigned w[3];
    __device__ __forceinline__ void load(const unsigned short* p) { const uint3 v = *reinterpret_cast<const uint3*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; }
    __device__ __forceinline__ void zero() { w[0] = w[1] = w[2] = 0u; }
};

// value of a[i] for a run-time i < CMAX out of a register array (static indices only)
template <int CMAX> __device__ __forceinline__ double pick(const double (&a)[CMAX], int i) {
    double v = a[0];
#pragma unroll
    for (int d = 1; d < CMAX; ++d) v = (i == d) ? a[d] : v;
    return v;
}

// scale.py:54-66 (solver_nmg.h nmg_ordinalize) on register arrays: pool adjacent categories -- first violation from the left, restart -- until the
// category means are monotone for `sign`; out[c] = pooled value of (compacted) category c < C; returns the population variance of the result.
// Every lane runs its own MV (C = 0: idle); the wave loops while any lane still merges.
template <int CMAX> __device__ __forceinline__ double ordinalize(const double (&m)[CMAX], const double (&f)[CMAX], int C, double sign, double (&out)[CMAX]) {
    double gs[CMAX], gw[CMAX];
    using GrpT = typename std::conditional<(CMAX > 8), unsigned long long, unsigned>::type;
    GrpT grp = (GrpT)0xfedcba9876543210ull;                       // group of category c in nibble c
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { gs[c] = (c < C) ? m[c] * f[c] : 0.0; gw[c] = (c < C) ? f[c] : 1.0; }
    int ng = C;
    bool active = ng > 1;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        double gm[CMAX];
#pragma unroll
        for (int g = 0; g < CMAX; ++g) gm[g] = gs[g] / gw[g];
        int first = -1;
#pragma unroll
        for (int g = CMAX - 2; g >= 0; --g) {
            const double d = gm[g] - gm[g + 1];
            const double sg = (d > 0.0) ? 1.0 : ((d < 0.0) ? -1.0 : 0.0);
            if (g + 1 < ng && sg == sign) first = g;              // (descending: the smallest violating g wins)
        }
        const bool merge = active && first >= 0;
        if (merge) {
            double ns[CMAX], nw[CMAX];
#pragma unroll
            for (int h = 0; h < CMAX; ++h) {
                const double s_next = (h + 1 < CMAX) ? gs[h + 1] : 0.0, w_next = (h + 1 < CMAX) ? gw[h + 1] : 1.0;
                ns[h] = (h < first) ? gs[h] : ((h == first) ? s_next + gs[h] : s_next);
                nw[h] = (h < first) ? gw[h] : ((h == first) ? w_next + gw[h] : w_next);
            }
#pragma unroll
            for (int h = 0; h < CMAX; ++h) { gs[h] = ns[h]; gw[h] = nw[h]; }
            GrpT ngrp = 0;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) { const unsigned gc = (unsigned)(grp >> (4 * c)) & 15u; ngrp |= (GrpT)(((int)gc > first) ? gc - 1u : gc) << (4 * c); }
            grp = ngrp;
            --ng;
        }
        active = merge && ng > 1;
    }
    double gm[CMAX];
#pragma unroll
    for (int g = 0; g < CMAX; ++g) gm[g] = gs[g] / gw[g];
    double mean = 0.0, ss = 0.0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const double v = pick<CMAX>(gm, (int)((unsigned)(grp >> (4 * c)) & 15u));
        out[c] = v;
        if (c < C) { mean += f[c] * v; ss += f[c] * v * v; }
    }
    return ss - mean * mean;
}

// sum over the lanes whose key equals l, for l = 0 .. L-1, of v[l] -- every lane gets all L totals (bitwise the same on every lane)
template <int N> __device__ __forceinline__ void allsum_each(double (&v)[N], int n) {
#pragma unroll
    for (int i = 0; i < N; ++i) if (i < n) v[i] = wv::allsum(v[i]);
}

// The finish of a problem whose stop has just been decided (solver_nmg.h nmg_finish): the correlation matrix of the final quantified MVs --
// <MV_r, MV_c> = tq_r' N_rc tq_c / n, one more pass over the count matrix: for every MV c the wave forms y = Mn[:, cols(c)] tq_c on its column lanes
// (the rows of MV c: one 16-byte load per lane and row) and folds y with tq over the columns of every MV r <= c -- then the shared tail
// (finish_problem: loadings, cross-loadings, path regressions, effects, the record).  `fast`: the problem's LDS; MV r's columns sit in at most
// three neighbouring lanes (at most CMAX <= 16 categories, 8 columns per lane): the lane that holds its first column stores, the other one adds.
template <int CMAX, int CPL = CPL8>
__device__ inline void finish(const ModelDesc& md, const CatDesc& cd, const ModelDesc& mdm, const SolverOut& so, double* gSm, NmState& st, NmgExtra& xg,
                              const unsigned short* k16, int ld16, double* fast, long b) {
    const int lane = threadIdx.x, Q = md.P, L = md.L, Pm = cd.Pm;
    const int QP = (Q + 1 + 7) & ~7;
    double* lp = fast;
    double* mvm_s = lp; lp += QP;        // means of the quantified MVs [Pm]
    double* tq_s = lp; lp += QP;
    double* mean_s = lp; lp += QP;
    int* mvcol = reinterpret_cast<int*>(lp); lp += QP;      // MV of every aug column
    Workspace wsm{};
    wsm.PS = cov_ld(Pm);
    wsm.S = gSm + b * cov_doubles(Pm);
    carve_small(wsm, lp, Pm, L, md.kmax, 0);
    lp += workspace_small_doubles(Pm, L, md.kmax, 0);
    double* srow = lp;                   // [Pm] <MV_r, MV_c> of the current c
    DevExec ex{lane, 64, wsm.red, nullptr};
    const double n = st.scal[0], inv_n = 1.0 / n;
    for (int j = lane; j < QP; j += 64) {
        tq_s[j] = (j < Q) ? xg.tq[j] : 0.0;
        mean_s[j] = (j <= Q) ? (double)k16[(long)Q * ld16 + j] * inv_n : 0.0;
    }
    __syncthreads();
    if (lane < Pm) {
        const int j0 = cd.mv_off[lane], C = cd.mv_off[lane + 1] - j0;
        double mv = xg.tc[lane];                                  // nmg_mv_moment(.., Mn + Q LD, 1, 1.0) with tc = 0
        for (int c = 0; c < C; ++c) { mv += tq_s[j0 + c] * mean_s[j0 + c]; mvcol[j0 + c] = lane; }
        mvm_s[lane] = mv;
    }
    __syncthreads();
    const int i0 = CPL * lane;
    const bool lane_on = i0 < Q;
    int mvc[CPL];
    double tqi[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) { mvc[u] = (i0 + u < Q) ? mvcol[i0 + u] : -1; tqi[u] = (i0 + u < Q) ? tq_s[i0 + u] : 0.0; }
    for (int c = 0; c < Pm; ++c) {
        const int jc0 = cd.mv_off[c], Cc = cd.mv_off[c + 1] - jc0;                       // (uniform)
        double y[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) y[u] = 0.0;
        {
            const unsigned short* row = k16 + (long)jc0 * ld16 + (lane_on ? i0 : 0);
            CountRow<CPL> w[CMAX];
#pragma unroll
            for (int t = 0; t < CMAX; ++t) { if (lane_on && t < Cc) w[t].load(row + (long)t * ld16); else w[t].zero(); }
#pragma unroll
            for (int t = 0; t < CMAX; ++t) {
                if (t < Cc) {
                    const double tj = tq_s[jc0 + t] * inv_n;
                    const unsigned* ww = w[t].w;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) y[u] = fma((double)((ww[u >> 1] >> (16 * (u & 1))) & 0xffffu), tj, y[u]);
                }
            }
        }
        // fold with tq over the runs of equal MV among this lane's columns: a run whose MV starts in this lane is stored, the head of an MV that
        // started in the previous lane is added behind the barrier
        int head_r = -1;
        double head_v = 0.0;
        {
            int cur = -1;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                if (mvc[u] >= 0) {
                    if (mvc[u] != cur) {
                        if (cur >= 0) { if (cd.mv_off[cur] >= i0) srow[cur] = acc; else { head_r = cur; head_v = acc; } }
                        cur = mvc[u]; acc = 0.0;
                    }
                    acc += tqi[u] * y[u];
                }
            }
            if (cur >= 0) { if (cd.mv_off[cur] >= i0) srow[cur] = acc; else { head_r = cur; head_v = acc; } }
        }
        __syncthreads();
        if constexpr (CMAX <= CPL) { if (head_r >= 0) srow[head_r] += head_v; }
        else {
            // up to 16 categories: an MV may lie across THREE lanes -- the lane next to its first one adds first, the one after that behind a second barrier
            const bool near = head_r >= 0 && cd.mv_off[head_r] >= i0 - CPL;
            if (near) srow[head_r] += head_v;
            __syncthreads();
            if (head_r >= 0 && !near) srow[head_r] += head_v;
        }
        __syncthreads();
        if (lane <= c && lane < Pm) {
            const double v = srow[lane] - mvm_s[lane] * mvm_s[c];
            wsm.S[(long)c * wsm.PS + lane] = v; wsm.S[(long)lane * wsm.PS + c] = v;
        }
        __syncthreads();
    }
    if (lane < Pm) { wsm.w[lane] = st.a_new[lane]; wsm.mu[lane] = 0.0; wsm.cs[lane] = 1.0; wsm.sd[lane] = sqrt(wsm.S[(long)lane * wsm.PS + lane]); }
    if (lane == 0) { wsm.scal[1] = st.scal[0]; wsm.scal[2] = 1.0 / st.scal[0]; wsm.scal[3] = st.scal[1]; }
    __syncthreads();
    FitOutputs out = so.fit;
    if (b != 0) out = FitOutputs{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    FitOutputs o2 = out;
    o2.score_w = nullptr; o2.score_c = nullptr; o2.mean = nullptr;          // the score map lives on the aug columns (below)
    finish_problem(ex, mdm, wsm, o2, (int)st.scal[2], false);
    if (out.score_w) for (int j = lane; j < Q; j += 64) out.score_w[j] = st.c_new[j];
    if (out.score_c && lane < L) out.score_c[lane] = st.k_new[lane];
}

// SUB (round 6): the step bounds its own criterion and stops on it, the pass behind it reads the rows each problem asks for (nsub > 0) -- an instantiation of its
// own so that the launches without it (data sets of a few hundred rows, the A/B option) keep the register allocation they had.
// ONE (round 6, needs SUB): prepare-to-finish of a problem in ONE launch -- the steps in a loop, each stopping on its own upper bound and going on speculatively
// otherwise, the score map (and the bound) of every step left in cmaps / kmaps for the verification pass on the observations (plspm_nonmetric.hip
// run_categorical_wave; the Scale.NUM solver of solver_wave16.h does the same); force[b] > 0: stop behind exactly that many steps (the replay of a problem
// whose stop the verification moved).  The state still travels through the problem's block in global memory between steps (the same loads and stores as the
// per-launch form, behind a device-scope fence): what goes away is the launch boundary -- and with it the passes over all rows and the host round trips.
struct NmwMaps { double* c; long cstride; double* k; long kstride; int* steps; const int* force; double bound_scale = 1.0; };      // (bound_scale: test seam, option nm_bound_shift -- the bound times 2^k is still an upper bound)
template <int LMAX, int CMAX = 8, bool SUB = false, bool ONE = false, int CPL = CPL8>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CMAX > 10 ? 1 : 2))) nmw_step_kernel(ModelDesc md, CatDesc cd, ModelDesc mdm, SolverOut so, double* __restrict__ gSm, double* __restrict__ gstate, long state_stride,
                                                      const double* __restrict__ partial, int nparts, int* __restrict__ nactive, const unsigned short* __restrict__ gK16, int ld16,
                                                      int fuse_finish, const int* __restrict__ live, int nsub, NmwMaps mp = NmwMaps{}) {
    static_assert(!ONE || SUB, "the one-launch form stops on the step's own bound");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long b = live ? live[blockIdx.x] : (long)blockIdx.x;      // (kernels_nonmetric.h nm_kernel: the live list of the previous step)
    const int lane = threadIdx.x;
    const int Q = md.P, L = md.L, Pm = cd.Pm;
    double* state = gstate + b * state_stride;
    NmState st;
    nm_carve(st, state, Q, L);
    if (st.scal[3] == 0.0) return;                               // finished problems cost nothing more
    NmgExtra xg;
    nmg_carve(xg, state + nm_state_doubles(Q, L, 0), Q, Pm, L, cd.cmax, cd.kmv);
    const unsigned short* k16 = gK16 + b * (long)(Q + 1) * ld16;
    // ---- LDS
    const int QP = (Q + 1 + 7) & ~7;
    double* lp = reinterpret_cast<double*>(smem_raw);
    double* const lp0 = lp;
    double* c_s = lp; lp += QP;          // score-map coefficients of the current scores (later: d = w_p tq_j)
    double* tq_s = lp; lp += QP;         // quantification values
    double* mean_s = lp; lp += QP;       // column means = category frequencies (row Q of Mn)
    double* mz_s = lp; lp += QP;         // <col_j, z_l(j)>: the inner estimate of the column's own LV
    Workspace ws{};
    ws.G = lp; lp += L * L; ws.E = lp; lp += L * L;
    double* YY = lp; lp += L * L;
    ws.a = lp; lp += L;
    double* kold = lp; lp += L;
    double* zmean = lp; lp += L;
    double* vmean = lp; lp += L;
    double* akk = lp; lp += L;
    double* sdl = lp; lp += L;
    lp += 2 * Pm;                        // (spare)
    ws.scr = lp; lp += (long)L * regression_scratch_doubles(md.kmax);
    ws.scal = lp; lp += 8;
    ws.red = lp;
    double* cold_s = lp0 + (lds_doubles(Q, Pm, L, md.kmax) - QP);      // the old score map (c_s is overwritten with the new one's numerators before the bound needs both)
    DevExec ex{lane, 64, ws.red, nullptr};
#ifdef PLSPM_DEBUG_MARKS
#define NMW_MARK(i) do { if (b == 0 && lane == 0 && so.marks && (int)st.scal[2] == 1) so.marks[i] = clock64(); } while (0)      // (the second iteration of problem 0)
#else
#define NMW_MARK(i) do { } while (0)
#endif

    // ---- decide on the previous convergence value (solver_nmg.h nmg_step: same protocol)
    // Round 6 (nsub > 0): the pass behind a step reads only the first row chunks of a problem -- as many as its own upper bound says are needed to carry the
    // sum over the tolerance (scal[7], below; nsub = the safety factor).  Every term of the criterion is non-negative, so that sum is a LOWER bound: at or above the tolerance it says "go on" as surely as the exact value would (the step itself
    // stops on the UPPER bound, below); below the tolerance it decides nothing -- the problem asks for the remaining chunks (scal[5] = 1: the host runs the
    // full pass for such problems), sits this launch out and decides on the exact value in the next one.
    int iteration = (int)st.scal[2];
    if (!ONE && SUB && iteration > 0 && st.scal[5] == 2.0) {                    // (words 5 .. 7 of a freshly prepared problem are whatever the buffer held: the first step writes them)
        // the previous launch stopped this problem on its own upper bound and left the finish to this one: behind the step in the same wave it lengthened the
        // launch by its whole duration (the steppers of a launch hide the finishers' time, not the other way round: 1.58 -> 1.96 ms of step launches per 1,000)
        if (lane == 0) { st.scal[3] = 0.0; st.scal[5] = 0.0; }
        if (fuse_finish) {
            __syncthreads();
            finish<CMAX, CPL>(md, cd, mdm, so, gSm, st, xg, k16, ld16, lp0, b);
        }
        return;
    }
    if (!ONE && iteration > 0) {
        const int need = (int)st.scal[7];                         // row chunks the pass behind the last step read for this problem (set at the end of that step)
        const bool exact = !SUB || nsub <= 0 || need >= nparts || st.scal[5] != 0.0;
        const int np = exact ? nparts : need;
        double s = 0.0;
        for (int i = lane; i < np; i += 64) s += partial[b * nparts + i];
        const double conv = wv::allsum(s);
        if (!exact && !(conv >= md.tol) && iteration <= md.max_iter) {      // (beyond max_iter the problem stops whatever the value: no exact pass needed)
            if (lane == 0) { st.scal[5] = 1.0; atomicAdd(nactive, 1); }
            return;
        }
        const bool stop = (exact && conv < md.tol) || (iteration > md.max_iter);
        if (lane == 0) {
            st.scal[4] = conv; st.scal[5] = 0.0;
            if (stop) { st.scal[3] = 0.0; if (iteration > md.max_iter && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED; }
        }
        if (stop) {
            if (fuse_finish) {
                __syncthreads();
                finish<CMAX, CPL>(md, cd, mdm, so, gSm, st, xg, k16, ld16, lp0, b);
            }
            return;
        }
    }
    NMW_MARK(20);
    const double n = st.scal[0], inv_n = 1.0 / n, corr2 = n / (n - 1.0);
    for (;;) {                                                    // (ONE: a trip per step; otherwise one trip)
    // new -> old (the stop-rule pass of this iteration compares them), and the step's inputs into LDS
    for (int j = lane; j < QP; j += 64) {
        double cj = 0.0, tqj = 0.0, mj = 0.0;
        if (j < Q) {
            cj = iteration > 0 ? st.c_new[j] : st.c_old[j];
            if (iteration > 0) st.c_old[j] = cj;
            tqj = xg.tq[j];
        }
        if (j <= Q) mj = (double)k16[(long)Q * ld16 + j] * inv_n;
        c_s[j] = cj; tq_s[j] = tqj; mean_s[j] = mj; mz_s[j] = 0.0;
        if (SUB) cold_s[j] = cj;
    }
    if (lane < Pm && iteration > 0) st.a_old[lane] = st.a_new[lane];
    if (lane < L) { const double k = iteration > 0 ? st.k_new[lane] : st.k_old[lane]; if (iteration > 0) st.k_old[lane] = k; kold[lane] = k; }
    if (lane == 0) ws.scal[3] = (double)ST_OK;
    __syncthreads();
    if (ONE && iteration == 0 && mp.c) {                         // the map of the initial scores: step 1 is verified against it
        for (int j = lane; j < Q; j += 64) mp.c[b * mp.cstride + j] = c_s[j];
        if (lane < L) mp.k[b * mp.kstride + lane] = kold[lane];
        if (lane == 0) mp.k[b * mp.kstride + L] = 0.0;
    }

    // ---- stream 1: V[j, m] = <col_j, y_m> = mean_j k_m + sum over the rows q of block m of Mn[q][j] c_q, for this lane's eight columns
    const int j0 = CPL * lane;
    const bool have_cols = j0 <= Q;
    double V[CPL][LMAX];
#pragma unroll
    for (int m = 0; m < LMAX; ++m) {
        double acc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) acc[u] = 0.0;
        if (m < L) {
            const double km = kold[m];
#pragma unroll
            for (int u = 0; u < CPL; ++u) acc[u] = (j0 + u <= Q) ? mean_s[j0 + u] * km : 0.0;
            const int q0 = md.boff[m], q1 = md.boff[m + 1];
            const unsigned short* row = k16 + (long)q0 * ld16 + (have_cols ? j0 : 0);
            constexpr int NR = 16;                                // rows in flight per lane: one problem's stream is latency-bound (a wave has its SIMD to itself at 1,000 problems)
            for (int q = q0; q < q1; q += NR) {
                CountRow<CPL> w[NR];
#pragma unroll
                for (int t = 0; t < NR; ++t) { if (have_cols && q + t < q1) w[t].load(row + (long)t * ld16); else w[t].zero(); }
                row += (long)NR * ld16;
#pragma unroll
                for (int t = 0; t < NR; ++t) {
                    if (q + t < q1) {
                        const double cq = c_s[q + t] * inv_n;                 // (1 / n once per row, not once per count: a quarter of the stream's arithmetic)
                        const unsigned* ww = w[t].w;
#pragma unroll
                        for (int u = 0; u < CPL; ++u) acc[u] = fma((double)((ww[u >> 1] >> (16 * (u & 1))) & 0xffffu), cq, acc[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CPL; ++u) V[u][m] = acc[u];
    }
    NMW_MARK(21);
    // LV of this lane's columns (columns >= Q: none)
    int lvc[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) lvc[u] = (j0 + u < Q) ? md.lvof[j0 + u] : -1;
    // mean of y_m = V[Q, m]: held by the lane that owns column Q
    {
        const int gq = Q / CPL, uq = Q % CPL;
#pragma unroll
        for (int m = 0; m < LMAX; ++m) {
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < CPL; ++u) v = (u == uq) ? V[u][m] : v;
            if (lane == gq && m < L) vmean[m] = v;
        }
    }
    // ---- YY[l][m] = k_l mean(y_m) + sum over the columns j of block l of c_j V[j, m]   (raw second moments of the scores)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        if (l < L) {                                              // (uniform)
            // (the upper triangle only: <y_l, y_m> is symmetric, and a wave-wide sum per entry is what this phase costs -- 21 instead of 36 at six LVs)
            double part[LMAX];
#pragma unroll
            for (int m = 0; m < LMAX; ++m) {
                double s = 0.0;
                if (m >= l) {
#pragma unroll
                    for (int u = 0; u < CPL; ++u) s += (lvc[u] == l) ? c_s[j0 + u] * V[u][m] : 0.0;
                }
                part[m] = s;
            }
#pragma unroll
            for (int m = 0; m < LMAX; ++m) if (m >= l && m < L) part[m] = wv::allsum(part[m]);
            if (lane < L && lane >= l) {
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < LMAX; ++m) v = (lane == m) ? part[m] : v;
                YY[l * L + lane] = v;                             // (+ k_l mean(y_m) below, once vmean is visible)
            }
        }
    }
    __syncthreads();
    double yy_up = 0.0;
    for (int e = lane; e < L * L; e += 64) {                      // (L <= 8: one trip)
        const int l = e / L, m = e - l * L;
        const int lo = l < m ? l : m, hi = l < m ? m : l;
        yy_up = kold[lo] * vmean[hi] + YY[lo * L + hi];
    }
    __syncthreads();
    for (int e = lane; e < L * L; e += 64) {
        const int l = e / L, m = e - l * L;
        YY[e] = yy_up;
        ws.G[e] = yy_up - vmean[l] * vmean[m];
    }
    __syncthreads();
    NMW_MARK(22);
    inner_weights(ex, md, ws, corr2, YY);
    NMW_MARK(23);
    // ---- a_l = <z_l, z_l>;  of MZ = V E only the column of the own LV is needed per aug column (the category sums of z_l) and the row of means
    if (lane < L) {
        const int l = lane;
        double s = 0.0;
        for (int m = 0; m < L; ++m) {
            const double em = ws.E[m * L + l];
            if (em == 0.0) continue;
            double t = 0.0;
            for (int m2 = 0; m2 < L; ++m2) t += YY[m * L + m2] * ws.E[m2 * L + l];
            s += em * t;
        }
        ws.a[l] = s;
        double zm = 0.0;
        for (int m = 0; m < L; ++m) zm += vmean[m] * ws.E[m * L + l];
        zmean[l] = zm;
    }
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
        if (lvc[u] >= 0) {
            double s = 0.0, vown = 0.0;
#pragma unroll
            for (int m = 0; m < LMAX; ++m) if (m < L) { s += V[u][m] * ws.E[m * L + lvc[u]]; if (SUB) vown = (lvc[u] == m) ? V[u][m] : vown; }
            mz_s[j0 + u] = s;
            // (round 6, for the step's own bound below: M_ll c_old of this column = V[., own LV] - mean k_old -- filed while V is still in registers; tq_s is free:
            //  the quantification works from the state's copy in registers)
            if (SUB) tq_s[j0 + u] = vown - mean_s[j0 + u] * kold[lvc[u]];
        }
    }
    __syncthreads();

    NMW_MARK(24);
    // ---- quantification (weights.py:112-115, scale.py:42-89) and the Mode-A outer weights: MV lane p
    const bool is_mv = lane < Pm;
    const int p = lane;
    int jm0 = 0, C = 0, kind = KIND_NOM, lv = 0;
    if (is_mv) { jm0 = cd.mv_off[p]; C = cd.mv_off[p + 1] - jm0; kind = cd.mv_kind[p]; lv = md.lvof[jm0]; }
    double tqn[CMAX];
    {
        double cf[CMAX], cm[CMAX], fall[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            const double f = (c < C) ? mean_s[jm0 + c] : 0.0, z = (c < C) ? mz_s[jm0 + c] : 0.0;
            fall[c] = f;
            cf[c] = f;
            cm[c] = (f > 0.0) ? z / f : 0.0;
        }
        // compact to the categories present in this problem (a replicate may miss some)
        double m2[CMAX], f2[CMAX];
#pragma unroll
        for (int d = 0; d < CMAX; ++d) { m2[d] = 0.0; f2[d] = 0.0; }
        int Cp = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C && cf[c] > 0.0) {
#pragma unroll
                for (int d = 0; d <= c; ++d) if (Cp == d) { m2[d] = cm[c]; f2[d] = cf[c]; }
                ++Cp;
            }
        }
        double cs[CMAX];
        // (every lane runs the pooling loops: C = 0 on the idle ones; NOM lanes take the category means)
        double inc[CMAX], dec[CMAX];
        const int Cord = (is_mv && kind == KIND_ORD) ? Cp : 0;
        const double v_inc = ordinalize<CMAX>(m2, f2, Cord, 1.0, inc);
        const double v_dec = ordinalize<CMAX>(m2, f2, Cord, -1.0, dec);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) cs[c] = (kind == KIND_ORD) ? ((v_inc < v_dec) ? -dec[c] : inc[c]) : m2[c];
        double mean = 0.0, ss = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < Cp) { mean += f2[c] * cs[c]; ss += f2[c] * cs[c] * cs[c]; }
        const double sd = nmg_quant_sd(ss, mean);                // (a constant quantification is NaN as in the reference: solver_nmg.h)
        int at = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            double v = 0.0;
            if (c < C && fall[c] > 0.0) { v = (pick<CMAX>(cs, at) - mean) / sd; ++at; }
            tqn[c] = v;
        }
    }
    // <MV_p, z_l>, Mode-A weight, mean of the quantified MV; d_j = w_p tq_j for the block quadratic form
    double wn = 0.0, mvmean = 0.0;
    if (is_mv) {
        double mvm = 0.0 * zmean[lv];                            // tc_p = 0: the constant term of nmg_mv_moment
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) mvm += tqn[c] * mz_s[jm0 + c];
        wn = mvm / ws.a[lv];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) mvmean += tqn[c] * mean_s[jm0 + c];
    }
    __syncthreads();                                              // (every lane is done reading c_s as the OLD score map)
    if (is_mv) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) c_s[jm0 + c] = wn * tqn[c];
    }
    __syncthreads();
    NMW_MARK(25);
    // ---- stream 2: U_i = sum over the rows j of the block of column i of Mn[j][i] d_j;  q_l = sum_i d_i U_i  (= w' <MV, MV'> w of block l)
    double qpart[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) qpart[l] = 0.0;
    {
        const int bA = (have_cols && j0 < Q) ? lvc[0] : 0;
        int bB = bA;
#pragma unroll
        for (int u = 0; u < CPL; ++u) if (lvc[u] >= 0) bB = lvc[u];
        const bool lane_on = have_cols && j0 < Q;
        double U[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) U[u] = 0.0;
        // the lane's columns lie in the consecutive blocks bA .. bB.  Two blocks per trip -- block bA + 2 s and its right neighbour, rows of both
        // in flight together: blocks of eight and more columns (the usual case) put at most two into a lane's eight columns, so ONE pass of
        // max(block length) trips serves every lane (a wave has its SIMD to itself at 1,000 problems: the stream is a chain of round trips)
        for (int s2 = 0; s2 < CPL; s2 += 2) {
            const int blkA = bA + s2, blkB = blkA + 1;
            const bool onA = lane_on && blkA <= bB, onB = lane_on && blkB <= bB;
            if (__builtin_amdgcn_ballot_w64(onA) == 0ull) break;
            const int a0 = onA ? md.boff[blkA] : 0, a1 = onA ? md.boff[blkA + 1] : 0;
            const int b0 = onB ? md.boff[blkB] : 0, b1 = onB ? md.boff[blkB + 1] : 0;
            const int len = max(a1 - a0, b1 - b0);
            const int trips = (int)wv::allmax((unsigned long long)(unsigned)len);
            const unsigned short* rowA = k16 + (long)a0 * ld16 + (lane_on ? j0 : 0);
            const unsigned short* rowB = k16 + (long)b0 * ld16 + (lane_on ? j0 : 0);
            constexpr int NU = 8;
            for (int t0 = 0; t0 < trips; t0 += NU) {
                CountRow<CPL> wa[NU], wb[NU];
#pragma unroll
                for (int t = 0; t < NU; ++t) {
                    if (onA && a0 + t0 + t < a1) wa[t].load(rowA + (long)(t0 + t) * ld16); else wa[t].zero();
                    if (onB && b0 + t0 + t < b1) wb[t].load(rowB + (long)(t0 + t) * ld16); else wb[t].zero();
                }
#pragma unroll
                for (int t = 0; t < NU; ++t) {
                    const int ja = a0 + t0 + t, jb = b0 + t0 + t;
                    const double da = (onA && ja < a1) ? c_s[ja] * inv_n : 0.0, db = (onB && jb < b1) ? c_s[jb] * inv_n : 0.0;
                    const unsigned* wwa = wa[t].w;
                    const unsigned* wwb = wb[t].w;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        if (lvc[u] == blkA) U[u] = fma((double)((wwa[u >> 1] >> (16 * (u & 1))) & 0xffffu), da, U[u]);
                        if (lvc[u] == blkB) U[u] = fma((double)((wwb[u >> 1] >> (16 * (u & 1))) & 0xffffu), db, U[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < CPL; ++u) s += (lvc[u] == l) ? c_s[j0 + u] * U[u] : 0.0;
            qpart[l] = s;
        }
        // for the upper bound of this step's criterion (below): U = M_ll d of the lane's columns (mz_s is dead from here on: the quantification read it)
#pragma unroll
        for (int u = 0; u < CPL; ++u) if (SUB && lvc[u] >= 0) mz_s[j0 + u] = U[u];
    }
    allsum_each(qpart, L);
    double mwpart[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) mwpart[l] = (is_mv && lv == l) ? wn * mvmean : 0.0;
    allsum_each(mwpart, L);
    if (lane < L) {
        double q = 0.0, mw = 0.0;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { q = (lane == l) ? qpart[l] : q; mw = (lane == l) ? mwpart[l] : mw; }
        const double sd = sqrt(q - mw * mw);
        sdl[lane] = sd;
        akk[lane] = -mw / sd;
    }
    __syncthreads();
    NMW_MARK(26);
    // ---- new coefficients, score map (tc = 0: k_new = akk), state back to global memory
    if (is_mv) {
        const double an = wn / sdl[lv];
        st.a_new[p] = an;
        xg.tc[p] = 0.0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) { st.c_new[jm0 + c] = an * tqn[c]; xg.tq[jm0 + c] = tqn[c]; }
    }
    if (lane < L) { st.k_new[lane] = akk[lane]; xg.akk[lane] = akk[lane]; }
    NMW_MARK(27);
    // ---- the criterion of THIS step from above (round 6; solver_core.h nm_step does the same for Scale.NUM): sum_il c_i (|y_old| - |y_new|)^2 <= sum_il c_i (y_old - y_new)^2
    //      = n sum_l [ delta' M_ll delta + 2 kappa_l mean' delta + kappa_l^2 ],  delta = c_new - c_old on the columns of block l, kappa = k_new - k_old, M = counts / n --
    //      and M_ll c_new = U / sd_l (stream 2), M_ll c_old = V[., l] - mean k_old (stream 1): no third stream.  Below the tolerance the problem stops HERE, with the
    //      reference's iteration count, and the pass of its last step is not needed.
    bool bound_stop = false;
    double need_chunks = (double)nparts;
    if (SUB && nsub > 0) {
        // mean' delta of block l = mean(new score) - k_new - (mean(old score) - k_old) = -2 akk_l ... in the step's own terms: mw_l / sd_l = -akk_l, and
        // mean(old score) = vmean[l]: no per-column work for it
        double t1 = 0.0;
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            if (lvc[u] >= 0) {
                const int j = j0 + u;
                const double isd = 1.0 / sdl[lvc[u]];
                const double delta = c_s[j] * isd - cold_s[j];
                t1 = fma(delta, mz_s[j] * isd - tq_s[j], t1);
            }
        }
        double ub = wv::allsum(t1);
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
                const double kap = akk[l] - kold[l];
                const double t2 = -akk[l] - (vmean[l] - kold[l]);
                ub += 2.0 * kap * t2 + kap * kap;
            }
        }
        ub *= n;
        bound_stop = ub * (ONE ? mp.bound_scale : 1.0) < md.tol * (1.0 - 1e-9);
        // rows the pass needs: the criterion is nearly always a few sign flips below its bound and the rows are exchangeable, so a fraction nsub tol / ub of them
        // carries the lower bound over the tolerance (nsub = 4: four times what the expectation asks for); all of them when that is most of them anyway
        const double want = (double)nsub * md.tol / ub * (double)nparts;
        if (want >= 0.0 && want < 0.75 * (double)nparts) need_chunks = floor(want) + 1.0;
        if (lane == 0) st.scal[6] = ub;
    }
    if constexpr (ONE) {
        // the map of this step beside the state (MV lanes: their columns' coefficients; LV lanes: the constants; lane 0: the bound), then stop or go on
        const long mrow = (long)(iteration + 1);
        if (mp.c) {
            if (is_mv) {
                const double an = wn / sdl[lv];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) if (c < C) mp.c[b * mp.cstride + mrow * Q + jm0 + c] = an * tqn[c];
            }
            if (lane < L) mp.k[b * mp.kstride + mrow * (L + 1) + lane] = akk[lane];
            if (lane == 0) mp.k[b * mp.kstride + mrow * (L + 1) + L] = st.scal[6];
        }
        ++iteration;
        const int forced = mp.force ? mp.force[b] : 0;
        const bool stop = (forced > 0 ? iteration >= forced : bound_stop) || iteration > md.max_iter;
        if (lane == 0) {
            st.scal[2] = (double)iteration;
            if (ws.scal[3] != (double)ST_OK && st.scal[1] == (double)ST_OK) st.scal[1] = ws.scal[3];
            if (stop) {
                st.scal[3] = 0.0; st.scal[4] = st.scal[6];
                if (iteration > md.max_iter && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED;      // (weights.py:183-186)
                if (mp.steps && !mp.force) mp.steps[b] = iteration;
            }
        }
        __threadfence();                                         // the state written above is read back by other lanes: at the top of the next trip, or by the finish
        __syncthreads();
        if (stop) {
            if (fuse_finish) finish<CMAX, CPL>(md, cd, mdm, so, gSm, st, xg, k16, ld16, lp0, b);
            return;
        }
        continue;
    }
    if (lane == 0) {
        st.scal[2] = (double)(iteration + 1);
        // (2: stopped; stays on the live list for the next launch, which finishes it -- its pass reads no rows.  Nothing to finish -- the first stage of a HOC
        //  pair --: off the list at once)
        st.scal[5] = (bound_stop && fuse_finish) ? 2.0 : 0.0;
        st.scal[7] = bound_stop ? 0.0 : need_chunks;
        if (ws.scal[3] != (double)ST_OK && st.scal[1] == (double)ST_OK) st.scal[1] = ws.scal[3];
        if (bound_stop) {
            st.scal[4] = st.scal[6];
            if (!fuse_finish) st.scal[3] = 0.0;
            if (iteration + 1 > md.max_iter && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED;      // (weights.py:183-186)
        }
        if (!bound_stop || fuse_finish) atomicAdd(nactive, 1);
    }
    return;
    }                                                             // (the step loop)
}

}  // namespace nmw
