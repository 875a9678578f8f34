// kernels_nmp.h -- Device kernels, part 4c (round 5): the score-based stop rule of all-indicator categorical models as an exact int8 matrix product.
// Included by plspm_nonmetric.hip behind kernels_nonmetric.h; device code only.
//
// Reference: _NonmetricWeights.iterate returns np.power(np.abs(scores_old) - np.abs(self.__scores), 2).sum() (plspm/weights.py:120) -- a sum over
// the ROWS of the (resampled) data, not a function of the moment matrix, so every iteration of every replicate walks the rows.  For ORD / NOM data
// the score of row n under LV l is  y_nl = k_l + sum over the MVs p of block l of c[column of p's category in row n]  (solver_nmg.h: the score map
// on the indicator columns), i.e. one row of the product
//         Y_l  =  I_l  C_l            I_l: N x kb, the 0/1 indicator columns of block l (the same for every replicate and iteration)
//                                     C_l: kb x (replicates x {old, new}), the score maps
// nm_conv_codes_kernel evaluates it as one 16-byte LDS read per (row, MV, replicate) -- 0.65 of the LDS roofline, 40 % of a categorical bootstrap
// step.  Here it IS a matrix product on the int8 matrix pipe, exact the way the digit-plane Gram is (plspm_gram_i8.hip): every score map is
// written as S = 7 signed base-256 digits of a 54-bit fixed-point number relative to the largest coefficient of its (replicate, LV, map) --
// |c - digits| <= 2^-55 max|c|, below the rounding of the fp64 additions it replaces -- the products with the 0/1 indicator bytes and their int32
// sums are exact, and the digits are put together again in the lane that owns (row, replicate).  One v_mfma_i32_16x16x64_i8 = 16 rows x 16
// replicates x one digit plane x a whole block of <= 64 columns; the epilogue (recombine, |a| - |b|, square, weight with the row's multiplicity in
// the replicate) runs on the 4 rows x 1 replicate a lane holds.  Rows beyond N and replicate slots beyond the live list carry weight 0.
//
// Layouts (16 bytes per lane, lane = 16 kg + i: the k-bytes 16 kg .. 16 kg + 15 of row / replicate i; A and B use the same k order, whatever
// order the instruction gives the 64 k inside):
//   ind8[((l * ntiles + t) * KS + ks) * 64 + lane]                     indicator bytes of rows 16 t + i under columns 64 ks .. of block l   (built once per upload)
//   tab8[(((((g * L + l) * 2 + m) * S + s) * KS + ks) * 64 + lane]     digit s of map m (0 old, 1 new) of live slots 16 g + i               (built once per pass)
//   KS = k-steps of a block: 1 for blocks of at most 64 indicator columns, 2 up to 128 (the two MFMAs of a plane chain through the accumulator)
//   scl [(((g * L + l) * 2 + m) * 16 + i]                  {2^(e - 54), k_l}: value of one unit of the last digit, constant term
#pragma once

namespace nmp {

constexpr int S = 7, FBITS = 8 * S - 2;
typedef int v4i __attribute__((ext_vector_type(4)));

// indicator bytes from the category codes (cat_codes_kernel: codes[(t * Pm + mv) * 16 + r] = block-relative column of MV mv's category in row
// 16 t + r, or `kb` for a row without one)
__global__ void __launch_bounds__(256) ind8_kernel(const unsigned short* __restrict__ codes, long ntiles, int Pm, int L, int KS, const int* __restrict__ lmv_off, uint4* __restrict__ ind8) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)L * ntiles * KS * 64) return;
    const int lane = (int)(e & 63), r = lane & 15;
    const long ltk = e >> 6;
    const int ks = (int)(ltk % KS);
    const long lt = ltk / KS;
    const int l = (int)(lt / ntiles);
    const long t = lt - (long)l * ntiles;
    const int kg = ks * 4 + (lane >> 4);                         // group of 16 block columns this lane's bytes stand for
    unsigned w[4] = {0u, 0u, 0u, 0u};
    for (int mv = lmv_off[l]; mv < lmv_off[l + 1]; ++mv) {
        const unsigned code = codes[(t * Pm + mv) * 16 + r];
        if ((int)(code >> 4) == kg) w[(code >> 2) & 3u] |= 1u << (8u * (code & 3u));
    }
    ind8[e] = make_uint4(w[0], w[1], w[2], w[3]);
}

// digit planes of the live problems' score maps.  One wave per live slot (grid: slots rounded up to whole groups of 16; the padding slots write
// zeros: weight 0 in the pass).  state_b[8 + 2 P ..]: c_old[P] | c_new[P] | k_old[L] | k_new[L] (coef_table_kernel reads the same run).
// Virtual problems (round 6, vj != null: the verification of the one-launch categorical solver, kernels_nmw.h ONE): slot = (replicate list[slot], step vj[slot]); its
// old / new maps are the rows vj - 1 and vj of the replicate's map store -- gstate = the coefficient rows [steps][P] (adjacent rows ARE c_old | c_new), kmaps = the
// constant rows [steps][L + 1] with the step's own bound behind them, from which the slot's row-chunk need is filed in vneed (nsub tol / bound of the chunks).
__global__ void __launch_bounds__(64) planes_kernel(const double* __restrict__ gstate, long state_stride, int P, int L, int KS, const int* __restrict__ boff, const int* __restrict__ list,
                                                    const int* __restrict__ count, uint4* __restrict__ tab8, double2* __restrict__ scl, const int* __restrict__ vj = nullptr,
                                                    const double* __restrict__ kmaps = nullptr, long kstride = 0, int* __restrict__ vneed = nullptr, double tol = 0.0, int nsub = 0,
                                                    int nparts = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char dig[S][64];
    const int n = *count;
    const long slot = blockIdx.x;
    if (slot >= ((long)(n + 15) & ~15L)) return;
    const long g = slot >> 4;
    const int i = (int)(slot & 15), lane = threadIdx.x;
    const bool live = slot < n;
    const double* st = live ? (vj ? gstate + (long)list[slot] * state_stride + (long)(vj[slot] - 1) * P : gstate + (long)list[slot] * state_stride + 8 + 2 * P) : nullptr;
    const double* kk = (live && vj) ? kmaps + (long)list[slot] * kstride + (long)(vj[slot] - 1) * (L + 1) : nullptr;
    if (vneed && live && lane == 0) {
        const double ub = kk[(L + 1) + L];                        // the bound of step vj
        int need = nparts;
        const double want = (double)nsub * tol / ub * (double)nparts;
        if (want >= 0.0 && want < 0.75 * (double)nparts) need = (int)want + 1;
        vneed[slot] = need;
    }
    for (int l = 0; l < L; ++l) {
        const int p0 = boff[l], nb = boff[l + 1] - p0;
        for (int m = 0; m < 2; ++m) {
            double cks[2];                                        // this lane's coefficient of either k-step (KS <= 2)
            double amax = 0.0;
            for (int ks = 0; ks < KS; ++ks) {
                const int k = ks * 64 + lane;
                cks[ks] = (live && k < nb) ? st[(long)m * P + p0 + k] : 0.0;
                const double a = fabs(cks[ks]);
                amax = (a > amax || a != a) ? a : amax;           // (NaN wins)
            }
            const double mx = __longlong_as_double((long long)wv::allmax((unsigned long long)__double_as_longlong(amax)));      // (non-negative doubles order like their bit patterns; NaN sorts above inf)
            int ex = 0;
            double unit = 0.0;
            const bool fin = mx > 0.0 && mx <= 1.7976931348623157e308;
            if (fin) {
                (void)frexp(mx, &ex);                             // mx = f 2^ex, 1/2 <= f < 1: every |c| < 2^ex
                unit = ldexp(1.0, ex - FBITS);
            } else if (!(mx == 0.0)) unit = mx - mx;              // inf / NaN coefficients: NaN scores, a NaN criterion (what the fp64 pass returns)
            for (int ks = 0; ks < KS; ++ks) {
                long long q = fin ? (long long)rint(ldexp(cks[ks], FBITS - ex)) : 0;      // |q| <= 2^54
#pragma unroll
                for (int s = S - 1; s >= 0; --s) {
                    const long long d = ((q + 128) & 255) - 128;   // balanced digit in [-128, 127]
                    dig[s][lane] = (unsigned char)(signed char)d;
                    q = (q - d) >> 8;
                }
                __syncthreads();
                if (lane < S * 4) {
                    const int s = lane >> 2, kg = lane & 3;
                    tab8[((((g * L + l) * 2 + m) * S + s) * KS + ks) * 64 + kg * 16 + i] = *reinterpret_cast<const uint4*>(&dig[s][16 * kg]);
                }
                __syncthreads();
            }
            if (lane == 0) scl[((g * L + l) * 2 + m) * 16 + i] = make_double2(unit, live ? (vj ? kk[(long)m * (L + 1) + l] : st[2L * P + (long)m * L + l]) : 0.0);
        }
    }
}

// value of the S digit sums of one (row, replicate) in units of the last digit: three planes at a time in int32 (|sum of <= 64 digits| <= 2^13, so
// ((d0 256) + d1) 256 + d2 stays below 2^30), then two fp64 multiply-adds -- the first exact (< 2^53), the second rounds once
__device__ __forceinline__ int shl8_add(int hi, int lo) {      // 256 hi + lo as ONE v_lshl_add_u32: the empty statement keeps the compiler from re-associating three planes into two
    int r = (int)(((unsigned)hi << 8) + (unsigned)lo);          // shifts + v_add3_u32.  (No instruction in the asm: the hazard recogniser does not see inline asm as a reader of
    asm("" : "+v"(r));                                         // MFMA results -- a real v_lshl_add_u32 written as asm read the digit sums too early.)
    return r;
}
__device__ __forceinline__ double digits_value(int d0, int d1, int d2, int d3, int d4, int d5, int d6) {
    const int i0 = shl8_add(shl8_add(d0, d1), d2), i1 = shl8_add(shl8_add(d3, d4), d5);
    return fma(fma((double)i0, 16777216.0, (double)i1), 256.0, (double)d6);
}

// The pass.  Workgroup = NW waves; wave w of workgroup (chunk, gq) takes the live slots 16 (gq NW + w) .. + 15 and the row tiles
// [chunk tpc, (chunk + 1) tpc): the waves of a workgroup walk the SAME indicator tiles (they meet in the CU's vector cache).  LV blocks outermost --
// the 2 S digit fragments of a block stay in registers (56) over the chunk's tiles.  partial[b * nparts + chunk]: summed by the step kernel in a fixed order.
template <int NW, int KS = 1>
__global__ void __launch_bounds__(64 * NW) conv_mfma_kernel(const uint4* __restrict__ ind8, long ntiles, int L, const unsigned* __restrict__ cd, long MT, const uint4* __restrict__ tab8,
                                                             const double2* __restrict__ scl, const int* __restrict__ list, const int* __restrict__ count, double* __restrict__ partial,
                                                             int nparts, int tpc, int nrun, const double* __restrict__ gstate, long state_stride, const int* __restrict__ vneed = nullptr,
                                                             double* __restrict__ vsum = nullptr, int by_slot = 0) {
    const int lane = threadIdx.x & 63, i = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nlive = *count;
    const long ng = (nlive + 15) >> 4;
    // (nrun <= nparts: the row chunks this launch covers)
    const int chunk = (int)(blockIdx.x % (unsigned)nrun);
    const long g = (long)(blockIdx.x / (unsigned)nrun) * NW + wave;
    if (g >= ng) return;
    const long slot = g * 16 + i;
    const bool live = slot < nlive;
    const long b = live ? (long)list[slot] : 0;
    if (gstate) {
        // round 6: every problem names the row chunks it needs (state word 7, kernels_nmw.h: enough for the lower bound to clear the tolerance); a wave whose
        // sixteen problems all need fewer than this chunk has nothing to do
        const int need = live ? (int)gstate[b * state_stride + 7] : 0;
        if (chunk >= wv::allreduce(need, [](int a, int c) { return a > c ? a : c; })) return;
    }
    if (vneed) {                                                 // (virtual problems: the need filed by planes_kernel)
        const int need = live ? vneed[slot] : 0;
        if (chunk >= wv::allreduce(need, [](int a, int c) { return a > c ? a : c; })) return;
    }
    const long t0 = (long)chunk * tpc, t1 = min(ntiles, t0 + tpc);
    // the lane's four rows of tile t in replicate b: dword kg of the 16-byte piece (k-block t / 4, replicate tile b / 16, piece t % 4, replicate b % 16)
    const unsigned* cdb = cd + ((b >> 4) * 64 + (b & 15)) * 4 + kg;
    double acc = 0.0;
    const v4i zero = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
        v4i B[2][S][KS];
        const uint4* tb = tab8 + ((g * L + l) * 2 * S * KS) * 64 + lane;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { const uint4 v = tb[((m * S + s) * KS + ks) * 64]; B[m][s][ks] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
        const double2 so = scl[((g * L + l) * 2 + 0) * 16 + i], sn = scl[((g * L + l) * 2 + 1) * 16 + i];
        const uint4* ia = ind8 + ((long)l * ntiles + t0) * KS * 64 + lane;
        for (long t = t0; t < t1; ++t, ia += KS * 64) {
            v4i A[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { const uint4 av = ia[ks * 64]; A[ks] = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w}; }
            const unsigned cw = live ? cdb[((t >> 2) * MT * 64 + (t & 3) * 16) * 4] : 0u;
            v4i D[2][S];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    D[m][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], B[m][s][0], zero, 0, 0, 0);
                    if constexpr (KS > 1) D[m][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], B[m][s][1], D[m][s], 0, 0, 0);      // (the second 64 columns of the block)
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a = fma(digits_value(D[0][0][r], D[0][1][r], D[0][2][r], D[0][3][r], D[0][4][r], D[0][5][r], D[0][6][r]), so.x, so.y);
                const double bb = fma(digits_value(D[1][0][r], D[1][1][r], D[1][2][r], D[1][3][r], D[1][4][r], D[1][5][r], D[1][6][r]), sn.x, sn.y);
                const double d = fabs(a) - fabs(bb);
                const double w = (double)((cw >> (8 * r)) & 0xffu);
                acc = fma(w * d, d, acc);
            }
        }
    }
    // the four row groups of a replicate: lanes i, i + 16, i + 32, i + 48 (fixed tree)
    double x, y;
    wv::swap16(acc, x, y); acc = x + y;
    wv::swap32(acc, x, y); acc = x + y;
    if (live && kg == 0) {
        // by_slot: filed under the (virtual) slot; vsum: the chunks a slot asked for added up atomically -- a lower bound that only has to clear the tolerance
        if (vsum) { if (!vneed || chunk < vneed[slot]) unsafeAtomicAdd(&vsum[slot], acc); }
        else partial[(by_slot ? slot : b) * nparts + chunk] = acc;
    }
}

}  // namespace nmp
