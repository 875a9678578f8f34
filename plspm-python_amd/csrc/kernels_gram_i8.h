// kernels_gram_i8.h -- Device kernels, part 2b: the weighted Gram of a bootstrap BATCH as one exact int8 MFMA product.
// Device code shared by the translation units of libplspm_hip.so (host_internal.h lists them); not a stand-alone header.
//
// Why.  Every replicate's moment matrix is M_b[p,q] = sum_i c_bi x_ip x_iq with the SAME data and small integer multiplicities
// c_bi (how often row i was drawn).  Over a batch that is a matrix product  M[b, (p,q)] = C[b, i] . Z[i, (p,q)],  Z[i,(p,q)] =
// x_ip x_iq (one column per unordered pair incl. the ones column, built once per data set).  C is EXACTLY int8.  Z is fp64 -- but
// cut into S balanced base-256 digits relative to the largest |z| of its column,
//     z_i 2^k = sum_s d_is 256^s,   d_is in [-128, 127],   k = 8S - 1 - exponent(max_i |z_i|)   (one less when the maximum fills its binade),
// every digit plane is int8 too, and  sum_i c_bi d_is  is an exact int32 dot product (|.| <= 128 N, N <= 2^24).  The fp64 matrix
// pipe of MI355X peaks at 78.6 TFLOP/s, the int8 one at ~5,000 TOP/s: S digit planes cost S int8 MACs per fp64 MAC and still run several
// times faster than v_mfma_f64.  Accuracy (measured against 80-bit sums, worst entry relative to sqrt(M_pp M_qq) on the 10k x 60 benchmark
// data; bench.py prints the three figures of its own run as digit_planes.moment_error_vs_80bit): S = 7 (>= 53 significant bits of the
// column maximum; the sum itself is exact, only the final int -> fp64 conversion rounds) 1e-16 -- below the fp64 MFMA accumulation chain's
// 1.6e-15, which rounds after every one of its ~6,300 additions; S = 6, the automatic choice where every column satisfies
// sum|z| >= 256 max|z| (plspm_gram_i8.hip choose_slices), 3e-15 -- about twice the fp64 chain's figure, inside the a-priori bound of an fp64
// accumulation of the same terms, nine orders below the 1e-6 the records are held to.  tools/experiments/slice_poc.py and
// tests/test_gpu_gram_i8.py compare all three with exact rational / extended-precision sums.
// This is the error-free-transformation (Ozaki-style) scheme with the simplification that one operand needs no splitting.
//
// Layout ("fragment-major", both operands).  v_mfma_i32_16x16x64_i8 takes, per lane l, 16 consecutive k of row (l & 15) for A and
// of column (l & 15) for B, k-group l >> 4.  Both matrices are stored as 1 KB blocks [k-block of 64][16-row tile][g 0..3][r 0..15]
// [16 B]: the block of (k-block kb, tile t) is exactly the 64 x 16 B a wave loads for one operand fragment, lane l reads its 16
// bytes at offset 16 l (conflict-free ds_read_b128), and the blocks a workgroup needs for one k-step are contiguous in memory
// (16 KB of counts, 2 S KB of digit planes), so global -> LDS is a linear copy.
//   Cd: counts,  block index kb * MT + mt   (mt = replicate / 16, r = replicate % 16)
//   Zs: digits,  block index kb * NT + nt   (nt = pair_group * S + s, r = pair % 16)
#pragma once
#include <type_traits>
#include "wave_ops.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define I8_SLACK_KB 6        // k-blocks of readable slack behind the counts and the digit planes (the DMA ring runs up to 5 k-steps ahead)
#ifndef I8_DEFAULT_VAR
#define I8_DEFAULT_VAR 3
#endif

// ---------------------------------------------------------------------------------------------- digit planes of the pair products
// Column maxima of the pair products, then k_j and 2^-k_j of every pair column j = (p, q).
// zs_max_kernel: one workgroup per block of RB <= 64 rows: the block's rows staged in LDS (row pitch PA + 1), thread t takes the pairs t, t + 256,
// ... and folds its 64 products into the pair's running maximum with an integer atomicMax on the bit pattern (non-negative doubles
// order like their bits; NaN / Inf compare above every finite value, which is what flags non-finite data).  (One workgroup per PAIR
// walking all rows from L2 took 0.14 ms at 10k x 60; this pass reads the matrix once.)
__global__ void __launch_bounds__(256) zs_max_kernel(const double* __restrict__ Xa, long N, int PA, int C, const int* __restrict__ pair_p, const int* __restrict__ pair_q,
                                                      int npair, int RB, unsigned long long* __restrict__ pair_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);              // [RB][C | 1]: odd pitch, conflict-free column walks (RB <= 64 rows, by LDS)
    const int pitch = C | 1;
    const long r0 = (long)blockIdx.x * RB;
    const int nr = (int)min((long)RB, N - r0);
    for (int e = threadIdx.x; e < RB * C; e += 256) {
        const int r = e / C, c = e - r * C;
        tile[r * pitch + c] = (r < nr) ? Xa[(r0 + r) * PA + c] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < npair; j += 256) {
        const int p = pair_p[j], q = pair_q[j];
        double mx = 0.0;
        bool bad = false;
#pragma unroll 8
        for (int r = 0; r < RB; ++r) {
            const double z = fabs(tile[r * pitch + p] * tile[r * pitch + q]);
            bad |= !(z <= 1.7976931348623157e308);
            mx = fmax(mx, z);
        }
        const unsigned long long bits = bad ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(mx);
        if (bits) atomicMax(pair_max + j, bits);
    }
}
// Second pass for the automatic plane count (plspm_gram_i8.hip choose_slices): sum_i |z_i| of every pair column in fixed point relative to the
// binade of the column maximum, 2^40 per unit of 2^e_max -- a thread's partial sum over its rows has one fixed order and integer adds
// commute, so the result is a deterministic function of the data.  Same staging as zs_max_kernel.
__global__ void __launch_bounds__(256) zs_abssum_kernel(const double* __restrict__ Xa, long N, int PA, int C, const int* __restrict__ pair_p, const int* __restrict__ pair_q,
                                                         int npair, int RB, const unsigned long long* __restrict__ pair_max, unsigned long long* __restrict__ pair_sum,
                                                         unsigned long long* __restrict__ pair_or) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);
    const int pitch = C | 1;
    const long r0 = (long)blockIdx.x * RB;
    const int nr = (int)min((long)RB, N - r0);
    for (int e = threadIdx.x; e < RB * C; e += 256) {
        const int r = e / C, c = e - r * C;
        tile[r * pitch + c] = (r < nr) ? Xa[(r0 + r) * PA + c] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < npair; j += 256) {
        const unsigned long long mb = pair_max[j];
        const int ef = (int)((mb >> 52) & 0x7ffull);                     // biased exponent of the column maximum
        if (ef < 64 || ef >= 0x7ff) continue;                            // zero / tiny / non-finite column: the host keeps the full plane count
        const double sc = __longlong_as_double((long long)((unsigned long long)(1023 + 40 - (ef - 1022)) << 52));     // 2^(40 - e), zmax = f 2^e
        // the integers the seven-plane decomposition would cut (zs_scale_kernel / zs_build_kernel): their bitwise OR tells how many low
        // planes are identically zero -- data whose products are short binary fractions of the column maximum (0/1 indicator columns:
        // one plane) are represented EXACTLY by fewer planes
        int e7;
        const double f7 = frexp(__longlong_as_double((long long)mb), &e7);
        const int k7 = 55 - e7 - (f7 >= 0.99 ? 1 : 0);
        const int p = pair_p[j], q = pair_q[j];
        double s = 0.0;
        unsigned long long bits = 0ull;
#pragma unroll 8
        for (int r = 0; r < RB; ++r) {
            const double z = fabs(tile[r * pitch + p] * tile[r * pitch + q]);
            s += z * sc;                                                 // each term < 2^40
            bits |= (unsigned long long)__double2ll_rn(ldexp(z, k7));
        }
        atomicAdd(pair_sum + j, (unsigned long long)s);
        if (bits) atomicOr(pair_or + j, bits);
    }
}
__global__ void __launch_bounds__(256) zs_scale_kernel(const unsigned long long* __restrict__ pair_max, int npair, int S, int* __restrict__ pair_k, double* __restrict__ pair_scale) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= npair) return;
    const double zmax = __longlong_as_double((long long)pair_max[j]);
    int k = 0;
    double sc = 0.0;
    if (!(zmax <= 1.7976931348623157e308)) sc = __builtin_nan("");       // non-finite data: the fp64 path would report NaN moments as well
    else if (zmax > 0.0) {
        int e;
        const double f = frexp(zmax, &e);                                // zmax = f 2^e, f in [0.5, 1)
        // balanced digits reach +-127.5 x 256^(S-1): |z| 2^k = f 2^(8S-1) fits while f <= 0.996 (one bit more resolution than a blanket
        // 2^(8S-2) bound); a maximum that close to the top of its binade gives that bit back
        // (eight planes: 2^63 would leave the int64 the digits are cut from)
        k = 8 * S - 1 - e - ((f >= 0.99 || S >= 8) ? 1 : 0);
        sc = ldexp(1.0, -k);
    }
    pair_k[j] = k;
    pair_scale[j] = sc;
}

// Digit planes in fragment-major layout.  Thread (pgl, g, r) of workgroup (kb, y): pair 16 (4y + pgl) + r, rows 64 kb + 16 g .. + 15;
// the 64 threads of one pair group write one contiguous 1 KB block per digit plane.
// shape 16: thread (pgl, g, r) of workgroup (kb, y): pair 16 (4y + pgl) + r, rows 64 kb + 16 g .. + 15 -- block (kb, pair group * S + plane),
// piece g * 16 + r.  shape 32: thread (pgl, sub, r): pair 32 (2y + pgl) + r, rows 64 kb + 16 sub .. -- block ((kb, group32 * S + plane), half
// sub / 2), piece (sub % 2) * 32 + r.
template <int S>
__global__ void __launch_bounds__(256) zs_build_kernel(const double* __restrict__ Xa, long N, int PA, const int* __restrict__ pair_p, const int* __restrict__ pair_q,
                                                        const int* __restrict__ pair_k, int npair, int npg, int NT, int shape, uint4* __restrict__ Zs) {
    const int tid = threadIdx.x;
    int r, sub, pgw, j;                                   // pair within its group, 16-row chunk of the k-block, group (of 16 / 32 pairs)
    if (shape == 16) { r = tid & 15; sub = (tid >> 4) & 3; pgw = (int)blockIdx.y * 4 + (tid >> 6); j = pgw * 16 + r; if (pgw >= npg) return; }
    else { r = tid & 31; sub = (tid >> 5) & 3; pgw = (int)blockIdx.y * 2 + (tid >> 7); j = pgw * 32 + r; if (2 * pgw >= npg) return; }
    const int kb = blockIdx.x;
    unsigned char dig[S][16];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < 16; ++t) dig[s][t] = 0;
    if (j < npair) {
        const int p = pair_p[j], q = pair_q[j], k = pair_k[j];
        const long i0 = (long)kb * 64 + sub * 16;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const long i = i0 + t;
            double z = 0.0;
            if (i < N) z = Xa[i * PA + p] * Xa[i * PA + q];
            long long v = (z == z && fabs(z) <= 1.7976931348623157e308) ? __double2ll_rn(ldexp(z, k)) : 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int d = (int)((v + 128) & 255) - 128;                   // balanced digit in [-128, 127]
                v = (v - d) >> 8;
                dig[s][t] = (unsigned char)(d & 255);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        uint4 w;
        unsigned u[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c] = (unsigned)dig[s][4 * c] | ((unsigned)dig[s][4 * c + 1] << 8) | ((unsigned)dig[s][4 * c + 2] << 16) | ((unsigned)dig[s][4 * c + 3] << 24);
        w.x = u[0]; w.y = u[1]; w.z = u[2]; w.w = u[3];
        if (shape == 16) Zs[((long)kb * NT + (long)pgw * S + s) * 64 + sub * 16 + r] = w;
        else Zs[((long)kb * NT + ((long)pgw * S + s) * 2 + (sub >> 1)) * 64 + (sub & 1) * 32 + r] = w;
    }
}

// ---------------------------------------------------------------------------------------------- resample -> dense int8 counts
// One workgroup per (replicate, window of I8_HIST_KB k-blocks = 65,536 rows), the same Philox draws / explicit indices and the same
// 16-bit LDS histogram as resample_kernel; the histogram leaves as bytes in fragment-major layout (16 consecutive rows = one 16-byte
// piece).  Data sets of more than 65,536 rows take several windows per replicate (blockIdx.y): every window's workgroup walks ALL N draws
// of the replicate and counts the ones that fall into its rows -- the draws are regenerated from the counter-based stream, nothing is
// exchanged between the windows (N = 100,000: two windows, the Philox work of the kernel doubles; the product behind it is 10 x larger
// than at N = 10,000 anyway).  err bit 0: index out of range, bit 1: a multiplicity above 127 (only possible with explicit indices; the
// host then falls back to the fp64 Gram).
// BYTES (Philox draws of data sets beyond one 16-bit window only): 8-bit counters, four rows per LDS word, 131,072 rows per window -- N =
// 100,000 takes ONE window again (one pass of Philox instead of two) and the histogram already is the output layout.  A Philox count
// cannot reach 128, let alone carry into its neighbour at 256 (N draws over N >= 65,536 rows: P < 1e-200); explicit index lists CAN carry
// any multiplicity, so they keep the 16-bit counters (where the overflow bit is reliable).
#define I8_HIST_KB 1024
#define I8_HIST_KB_BYTES 2048
template <bool BYTES>
__global__ void __launch_bounds__(1024) resample_i8_kernel(int N, int KB, int MT, int shape, const int* __restrict__ idx, uint64_t seed, int64_t rep0, uint4* __restrict__ Cd,
                                                           int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);      // KBw * 32 words: rows 2w, 2w+1 of the window in the halves of word w (zero beyond N)
    const int tid = threadIdx.x, nthr = blockDim.x;      // 256 ... 1,024 threads: the host fills the CU's wave slots whatever the histogram leaves room for
    const long b = blockIdx.x;
    constexpr int WKB = BYTES ? I8_HIST_KB_BYTES : I8_HIST_KB;
    const int kb0 = (int)blockIdx.y * WKB, KBw = min(WKB, KB - kb0);
    const unsigned r0 = (unsigned)kb0 * 64u, rspan = (unsigned)KBw * 64u;      // this window's rows [r0, r0 + rspan)
    const int nwords = KBw * (BYTES ? 16 : 32);
    auto count = [&](unsigned w) {                                               // row r0 + w of the window was drawn
        if (BYTES) atomicAdd(&hist[w >> 2], 1u << (8u * (w & 3u)));
        else atomicAdd(&hist[w >> 1], (w & 1u) ? 0x10000u : 1u);
    };
    for (int i = tid; i < nwords; i += nthr) hist[i] = 0u;
    __syncthreads();
    if (idx) {
        const int* my = idx + b * (long)N;
        for (int i = tid; i < N; i += nthr) {
            const int r = my[i];
            if ((unsigned)r < (unsigned)N) { const unsigned w = (unsigned)r - r0; if (w < rspan) count(w); }
            else if (blockIdx.y == 0) atomicOr(err, 1);
        }
    } else {
        const uint64_t rep = (uint64_t)(rep0 + b);
        const int nq = (N + 3) >> 2;
        for (int q = tid; q < nq; q += nthr) {
            const u32x4 u = resample_quad(seed, rep, (uint32_t)q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < N) { const unsigned w = to_index(u.v[j], (uint32_t)N) - r0; if (w < rspan) count(w); }
        }
    }
    __syncthreads();
    const int mt = (int)(b >> 4), r = (int)(b & 15);
    const uint4* h4 = reinterpret_cast<const uint4*>(hist);
    bool over = false;
    for (int c = tid; c < KBw * 4; c += nthr) {                     // piece c: rows 16c .. 16c+15 of the window = hist words 8c .. 8c+7 (bytes: 4c .. 4c+3)
        uint4 out;
        if (BYTES) {
            out = h4[c];
            over |= ((out.x | out.y | out.z | out.w) & 0x80808080u) != 0u;
        } else {
            const uint4 lo = h4[2 * c], hi = h4[2 * c + 1];
            const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned a = w[2 * k], bb = w[2 * k + 1];
                over |= ((a | bb) & 0xff80ff80u) != 0u;
                o[k] = (a & 0xffu) | ((a >> 8) & 0xff00u) | ((bb & 0xffu) << 16) | ((bb << 8) & 0xff000000u);
            }
            out.x = o[0]; out.y = o[1]; out.z = o[2]; out.w = o[3];
        }
        const int kb = kb0 + (c >> 2), g = c & 3;
        if (shape == 16) Cd[((long)kb * MT + mt) * 64 + g * 16 + r] = out;
        else Cd[((long)kb * MT + (b >> 5) * 2 + (g >> 1)) * 64 + (g & 1) * 32 + (int)(b & 31)] = out;       // block (tile of 32, half g / 2), piece (g % 2) 32 + row
    }
    if (over) atomicOr(err, 2);
}

// Round 5 -- FOUR-bit counters for Philox draws of data sets beyond one 16-bit window: N = 100,000 takes 50 KB of LDS instead of 100, so THREE workgroups
// share a CU and one replicate's clear / read-out / 6,250 scattered stores run under its neighbours' draws (the byte histogram's single workgroup per CU
// serialised them: 0.67-0.75 ms per 5,000 replicates, of which the generator is 0.11, tools/ubench/resample_rng.hip).  A count of 16 (P = 1.8e-14 per row
// and replicate: once in ~10^5 calls of 5,000 x 100,000) carries into the neighbouring nibble -- and is FOUND: every carry lowers the sum of the nibbles by
// 15 or 16, so "sum of the counts == draws that fell into the window" holds exactly when no counter overflowed.  The read-out accumulates that sum beside
// its stores; a workgroup whose sum is short draws the replicate again with 8-bit counters, two half-windows in the same LDS, and overwrites its pieces
// (`force_slow`: that path for every replicate -- tests).  Same draws, same layout, same bits as resample_i8_kernel.
#define I8_HIST_KB_NIB 4096
__device__ __forceinline__ unsigned nib_spread16(unsigned t) {      // four nibbles (bits 0 .. 15) -> four bytes
    t = (t | (t << 8)) & 0x00ff00ffu;
    return (t | (t << 4)) & 0x0f0f0f0fu;
}
__global__ void __launch_bounds__(1024) resample_i8_nib_kernel(int N, int KB, int MT, uint64_t seed, int64_t rep0, uint4* __restrict__ Cd, int* __restrict__ err, int force_slow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);      // KBw * 8 words: row 8 w + i of the window in nibble i of word w
    __shared__ unsigned tot[2];                                   // [sum of the counts read out, draws that fell into the window]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long b = blockIdx.x;
    const int kb0 = (int)blockIdx.y * I8_HIST_KB_NIB, KBw = min(I8_HIST_KB_NIB, KB - kb0);
    const unsigned r0 = (unsigned)kb0 * 64u, rspan = (unsigned)KBw * 64u;
    const int nwords = KBw * 8;
    const uint64_t rep = (uint64_t)(rep0 + b);
    const int nq = (N + 3) >> 2;
    const int mt = (int)(b >> 4), r = (int)(b & 15);
    for (int i = tid; i < nwords; i += nthr) hist[i] = 0u;
    if (tid < 2) tot[tid] = 0u;
    __syncthreads();
    unsigned inwin = 0u;
    for (int q = tid; q < nq; q += nthr) {
        const u32x4 u = resample_quad(seed, rep, (uint32_t)q);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < N) {
                const unsigned w = to_index(u.v[j], (uint32_t)N) - r0;
                if (w < rspan) { atomicAdd(&hist[w >> 3], 1u << (4u * (w & 7u))); ++inwin; }
            }
    }
    __syncthreads();
    unsigned seen = 0u;
    const uint2* h2 = reinterpret_cast<const uint2*>(hist);
    for (int c = tid; c < KBw * 4; c += nthr) {                     // piece c: rows 16 c .. 16 c + 15 of the window = words 2 c, 2 c + 1
        const uint2 w = h2[c];
        uint4 out;
        out.x = nib_spread16(w.x & 0xffffu); out.y = nib_spread16(w.x >> 16); out.z = nib_spread16(w.y & 0xffffu); out.w = nib_spread16(w.y >> 16);
        seen = __builtin_amdgcn_sad_u8(out.x, 0u, seen); seen = __builtin_amdgcn_sad_u8(out.y, 0u, seen);
        seen = __builtin_amdgcn_sad_u8(out.z, 0u, seen); seen = __builtin_amdgcn_sad_u8(out.w, 0u, seen);
        Cd[((long)(kb0 + (c >> 2)) * MT + mt) * 64 + (c & 3) * 16 + r] = out;
    }
    seen = wv::allsum(seen); inwin = wv::allsum(inwin);
    if ((tid & 63) == 0) { atomicAdd(&tot[0], seen); atomicAdd(&tot[1], inwin); }
    __syncthreads();
    if (tot[0] == tot[1] && !force_slow) return;                    // (uniform over the workgroup)
    // a counter overflowed: the replicate again, 8-bit counters, the window in two halves of KBw * 2 pieces (KBw * 32 rows) each
    bool over = false;
    for (int half = 0; half < 2; ++half) {
        const unsigned h0 = (unsigned)half * (unsigned)KBw * 32u, hspan = min((unsigned)KBw * 32u, rspan - h0);
        __syncthreads();
        for (int i = tid; i < nwords; i += nthr) hist[i] = 0u;
        __syncthreads();
        for (int q = tid; q < nq; q += nthr) {
            const u32x4 u = resample_quad(seed, rep, (uint32_t)q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < N) {
                    const unsigned w = to_index(u.v[j], (uint32_t)N) - r0 - h0;
                    if (w < hspan) atomicAdd(&hist[w >> 2], 1u << (8u * (w & 3u)));
                }
        }
        __syncthreads();
        const uint4* h4 = reinterpret_cast<const uint4*>(hist);
        for (int c = tid; c < KBw * 2; c += nthr) {                 // piece KBw * 2 * half + c of the window
            const uint4 out = h4[c];
            over |= ((out.x | out.y | out.z | out.w) & 0x80808080u) != 0u;
            const int cw = KBw * 2 * half + c;
            Cd[((long)(kb0 + (cw >> 2)) * MT + mt) * 64 + (cw & 3) * 16 + r] = out;
        }
    }
    if (over) atomicOr(err, 2);
}

// ---------------------------------------------------------------------------------------------- the GEMM
// Workgroup tile: 256 replicates x 2 pair groups (32 pairs x S digit planes); 4 waves as 2 (replicate halves) x 2 (pair groups);
// a wave keeps 8 x S accumulator tiles (S = 7: 224 AGPRs) -- one wave per SIMD, latency hidden inside the wave.
// k-step = one k-block of 64 rows = 16 + 2S contiguous 1 KB blocks, brought in by LDS-DMA (global_load_lds_dwordx4: one wave
// instruction = one block, no staging registers, no ds_write pass) into a ring of three LDS stages:
//     iteration kb:   issue DMA of k-step kb+2 -> stage (kb+2)%3        (its last readers finished before the previous barrier)
//                     s_waitcnt vmcnt(own DMAs of kb+2 may stay in flight) lgkmcnt(0);  s_barrier     -> k-step kb+1 is complete
//                     ds_read the fragments of kb+1 (second register set)  } overlap
//                     8 S MFMAs of k-step kb on the first register set     }
// so a DMA has two iterations (~1,800 cycles) to land and the matrix pipe only ever waits at the barrier.  The DMA statements are
// inline asm (hipcc would drain vmcnt(0) at every barrier for a builtin LDS-DMA); their completion is counted by hand: every wave
// issues exactly PER blocks per k-step (the 16 + 2S blocks dealt round-robin; the last wave repeats a few -- identical bytes to the
// same place), hence the constant vmcnt(PER).
// Tiles are enumerated XCD-aware: workgroup id mod 8 is the XCD, each XCD walks a contiguous range of an enumeration whose
// consecutive 32 tiles form a 4 (replicate tiles) x 8 (pair tiles) block, so an XCD's L2 serves 4 count chunks + 8 digit chunks
// per k-step to its 32 resident workgroups.
// Epilogue: lane (l & 15) owns pair 16 pg + (l & 15) in all S planes -> digits recombined in the lane (fp64 fma chain over exact terms, ~one ulp
// rounding chain), scaled by 2^-k_j and stored at the element's slot of the tile-packed moment matrix the solvers read.
// WM = wave rows of the workgroup (2: four waves, one per SIMD, 128 replicates x 16 pairs x S planes each; 4: eight waves, two per
// SIMD, 64 replicates each -- an LDS-DMA instruction holds its wave's issue port for ~60 cycles, which a single wave per SIMD cannot
// hide behind its own MFMAs: with two, the partner's MFMAs run meanwhile).
// VAR (experiments, tools/i8_bench.py): ring depth NS = 3 + VAR % 3, a fragment read after every (1 + VAR / 3 % 3)-th MFMA, DMA issue
// spread over the step (VAR / 9 == 0) or at its head (1).
// SH: MFMA shape.  16: v_mfma_i32_16x16x64_i8, blocks of 16 rows x 64 k, waves 2 (pair groups of 16) x WM.  32: v_mfma_i32_32x32x32_i8 (a 12 %
// higher measured ceiling, and 7 instead of 3 free issue slots behind every MFMA), blocks of 32 rows x 32 k (two per k-block: halves
// h = 0, 1), every wave spans the workgroup's 32 pairs x S planes and 256 / NW replicates.  The 1 KB blocks a workgroup needs per
// k-step are the same contiguous 16 + 2 S KB in both layouts; only the inside of a block differs (lane l's 16 bytes at 16 l in both).
// RT: count tiles (16 replicates each) per workgroup -- 16 = 256 replicates; 20 = 320 replicates with S = 6 (30 accumulator tiles per wave at
// eight waves: the six-plane default where its tile grid pays, with tile rows of 320 and 256 replicates in one launch -- MIX below);
// 12 / 8 = 192 / 128 replicates per workgroup, i.e. 6 / 4 accumulator rows per wave at four waves (168 / 112 accumulator registers instead
// of 224): the variants that leave half of a SIMD's register file to a co-resident wave of another kernel (measured: DESIGN.md 7b).
template <int S, int WM, int VAR = I8_DEFAULT_VAR, int SH = 16, int RT = 16>
struct GramI8 {
    static_assert(RT == 16 || SH == 16, "narrow workgroup tiles exist for the 16x16x64 layout only");
    static_assert(RT % WM == 0, "whole count tiles per wave");
    static constexpr int NW = 2 * WM;               // waves per workgroup
    static constexpr int MTW = RT / WM;             // SH 16: count tiles (16 replicates) per wave
    static constexpr int T32W = 8 / NW;             // SH 32: count tiles (32 replicates) per wave
    static constexpr int NA = SH == 16 ? MTW : 2 * T32W;      // operand fragments (16 B per lane) per wave and k-step: counts ...
    static constexpr int NB = SH == 16 ? S : 2 * S;           // ... and digit planes
    static constexpr int NMFMA = SH == 16 ? MTW * S : 2 * T32W * S;
    static constexpr int NBLK = RT + 2 * S;         // 1 KB blocks per k-step: RT count tiles + 2 pair groups x S planes
    static constexpr int PER = (NBLK + NW - 1) / NW;   // DMA instructions per wave and k-step
    static constexpr int STAGE_BYTES = NBLK * 1024;
    // VAR >= 18: ONE workgroup barrier per two k-steps on a ring of five stages (k-steps 2p+1 and 2p+2 have landed at the barrier of
    // pair p, 2p+3 is in flight, 2p+4 / 2p+5 are issued during the pair into the stages of 2p-1 / 2p, whose fragments every wave
    // read before that barrier) -- when five stages fit the 160 KB
    // VAR >= 100 (experiments build only): ABLATIONS of the default schedule V = VAR % 100 -- timing probes whose results are garbage:
    // bit 0 of VAR / 100: no LDS-DMA issue in the steady state, bit 1: no workgroup barrier, bit 2: no fragment reads
    static constexpr int ABL = (VAR / 100) & 7, V = VAR % 100;
    // VAR >= 800: the LDS-DMA as a MUBUF instruction, `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`, instead of the FLAT-global form
    // `global_load_lds_dwordx4 v_off, s[base]`: one descriptor per operand whose base is the workgroup's first block, 32-bit byte offsets
    // (the host takes this form whenever both operands' walks stay below 4 GiB).  Alternating A/B at working clocks: 0.4494 -> 0.4473 ms
    // with eight waves (all three rounds in favour), within the noise with four (profiles/r03_i8_dma_form.jsonl) -- the descriptor
    // form saves the 64-bit address VALU adds, not the issue slot.
    static constexpr int BUFM = VAR / 800;
    static_assert(BUFM == 0 || (SH == 16 && RT % NW == 0), "buffer form: 16x16x64 layout, count blocks dealt evenly to the waves");
    static constexpr bool PAIRB = V >= 18 && 5 * STAGE_BYTES <= 160 * 1024;
    static constexpr int NS_WANT = PAIRB ? 5 : 3 + V % 3, NS = NS_WANT * STAGE_BYTES <= 160 * 1024 ? NS_WANT : (160 * 1024) / STAGE_BYTES;      // LDS stages of the DMA ring
    static constexpr int RSTEP = 1 + (V / 3) % 3, DMA_HEAD = (V % 18) / 9;
    static constexpr int FLIGHT = PAIRB ? 1 : NS - 2;      // k-steps whose DMAs may still be in flight behind the barrier's wait
    static constexpr size_t LDS_BYTES = (size_t)NS * STAGE_BYTES;
    // MIX: the launch may hold tile rows of TWO heights -- RT count tiles ("tall") and RT - WM ("short": every wave drops its last count
    // tile, i.e. its last S MFMAs, one fragment read and the tile's share of the DMA blocks of a k-step).  A tile grid that fills the
    // last round of the machine only partly is then re-cut so that every CU carries about the same number of count-tile rows (host:
    // plspm_gram_i8.hip i8_mix_plan); sums are exact int32 either way, the results do not depend on the cut.  Eight-wave FLAT-global form of
    // the 320-replicate tile only (the buffer form fixes the operand of a DMA slot at compile time).
    static constexpr bool MIX = RT == 20 && WM == 4 && SH == 16 && BUFM == 0 && DMA_HEAD == 0 && ((PER - 1) * NMFMA) / PER + 1 < (MTW - 1) * S;
    static constexpr int RTS = RT - WM, NBLKS = RTS + 2 * S;      // count tiles / DMA blocks per k-step of a short tile
};

// one LDS-DMA block: 64 lanes x 16 B from `base + voff` to LDS byte address `lds_dst` (wave-uniform) + 16 lane
__device__ __forceinline__ void glds_block(const void* base, unsigned voff, unsigned lds_dst) {
    // (M0 is written in the statement that uses it; nothing else in these kernels reads M0, so it is not saved / restored: two scalar
    // moves less per DMA in a stream where every instruction between two MFMAs counts)
    const unsigned long long b = (unsigned long long)base;
    const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(ub), "s"(lds_dst) : "memory");
}

// IND ("independent planes"): the digit buffer holds ONE plane per pair group (0/1 indicator data, plspm_gram_i8.hip choose_slices) and the S
// "planes" a wave walks are S consecutive pair groups of it -- the same main loop at its full accumulator tile (a one-plane
// instantiation moves 16 KB of counts per 8 MFMAs of a wave and is bound by the loads); the epilogue stores S x 16 pairs per wave
// instead of recombining S planes into 16.
template <int S, int WM, int VAR = I8_DEFAULT_VAR, int SH = 16, int RT = 16, bool IND = false>
__global__ void __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(RT < 16 ? 2 : WM / 2, RT < 16 ? 2 : WM / 2)))
gram_i8_kernel(const uint4* __restrict__ Cd, const uint4* __restrict__ Zs, int KB, int MT, int NT, int ntx, int nty, const int* __restrict__ pair_dst,
               const int* __restrict__ pair_dst2, const double* __restrict__ pair_scale, int npair, long nrep, double* __restrict__ gram, long psize, int nty_short) {
    using G = GramI8<S, WM, VAR, SH, RT>;
    constexpr int MTW = G::MTW, NA = G::NA, NB = G::NB;
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    using AccT = typename std::conditional<SH == 16, i32x4, i32x16>::type;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = SH == 16 ? wave >> 1 : wave, wn = SH == 16 ? wave & 1 : 0;
    // XCD-aware tile enumeration.  MIX: nty - nty_short tall rows (count tiles 0 ..), then nty_short short rows; each list is cut into
    // eight contiguous ranges and an XCD walks its tall range first, then its short one (longest tiles first: the CUs that finish early
    // pick up the short ones).  ct0 = the tile's first count tile, shortT (wave-uniform) = its height.
    const int w = blockIdx.x;
    int slot = w >> 3, rows = nty, rbase = 0;
    bool shortT = false;
    if constexpr (G::MIX) {
        const int ntall = nty - nty_short, tper = (ntx * ntall + 7) >> 3;
        if (slot >= tper) { slot -= tper; rows = nty_short; rbase = ntall; shortT = true; }
        else rows = ntall;
    }
    const int total = ntx * rows, per = (total + 7) >> 3;
    const int gidx = (w & 7) * per + slot;
    if (slot >= per || gidx >= total) return;
    const int srow = 4 * ntx;
    const int sr = gidx / srow, rem = gidx - sr * srow;
    const int nr = min(4, rows - 4 * sr);
    const int tx = rem / nr, tyl = 4 * sr + (rem - tx * nr), ty = rbase + tyl;
    const int ct0 = G::MIX ? (shortT ? rbase * RT + tyl * G::RTS : tyl * RT) : ty * RT;
    const int RTr = (G::MIX && shortT) ? G::RTS : RT, NBLKr = (G::MIX && shortT) ? G::NBLKS : G::NBLK;

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_raw);
    const unsigned voff = (unsigned)lane * 16u;
    // this wave's blocks of a k-step, dealt round-robin (wave, wave + NW, ...): block b < 16 = count tile b of the workgroup's 16, else
    // digit block b - 16 of its 2 S.  NBLK is not always a multiple of NW: the last waves own one block less (`full` tells).
    const int myper = (NBLKr - wave + G::NW - 1) / G::NW;
    const bool full = myper == G::PER;
    const char* src[G::PER];
    long inc[G::PER];
    unsigned dst[G::PER];
#pragma unroll
    for (int i = 0; i < G::PER; ++i) {
        const int b = min(wave + G::NW * i, NBLKr - 1);
        const bool isA = b < RTr;
        src[i] = isA ? (const char*)(Cd + ((long)ct0 + b) * 64) : (const char*)(Zs + ((long)tx * 2 * S + (b - RTr)) * 64);
        inc[i] = (isA ? (long)MT : (long)NT) * 1024;
        // LDS slot: count tile a of wave row wm at wm MTW + a (a short tile leaves the last slot of every wave row unused), digits behind
        const int sl = isA ? ((G::MIX && shortT) ? (b / (MTW - 1)) * MTW + b % (MTW - 1) : b) : RT + (b - RTr);
        dst[i] = lds0 + (unsigned)sl * 1024u;
    }
    // the source pointers simply run on: the DMAs of the k-steps past the end (never consumed) read the I8_SLACK_KB k-blocks of slack
    // the host keeps behind both operands (a clamp cost a branch + 16 scalar selects per k-step)
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < G::PER; ++i) src[i] += inc[i];
    };
    // buffer form: rsA / rsB address the workgroup's count tiles / digit blocks of k-step 0; src[] then holds byte OFFSETS (< 4 GiB)
    i32x4 rsA = {0, 0, 0, 0}, rsB = {0, 0, 0, 0};
    if constexpr (G::BUFM != 0) {
        const unsigned long long bA = (unsigned long long)(Cd + (long)ct0 * 64), bB = (unsigned long long)(Zs + (long)tx * 2 * S * 64);
        rsA[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)bA); rsA[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(bA >> 32) & 0xffffu));
        rsB[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)bB); rsB[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(bB >> 32) & 0xffffu));
        rsA[2] = rsB[2] = (int)0xffffffffu;                      // num_records (bytes, stride 0): the 32-bit offsets are always in range
        rsA[3] = rsB[3] = 0x00027000;                            // dst_sel xyzw, 32-bit data format (raw buffer)
#pragma unroll
        for (int i = 0; i < G::PER; ++i) {
            const int b = min(wave + G::NW * i, G::NBLK - 1);
            src[i] = (const char*)(unsigned long long)((unsigned)(b < RT ? b : b - RT) * 1024u);
        }
    }
    auto issue_one = [&](int i, unsigned stage_off) {
        if (i < G::PER - 1 || full) {
            if constexpr (G::BUFM == 0) glds_block(src[i], voff, dst[i] + stage_off);
            else {
                // (block wave + NW i of the k-step: a count tile for i < RT / NW, else a digit block -- except that the LAST slot of a wave
                //  without a block of its own repeats block NBLK - 1, a digit block)
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)src[i]);
                const unsigned ld = dst[i] + stage_off;
                if (i < RT / G::NW) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsA), "s"(so), "s"(ld) : "memory");
                else asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsB), "s"(so), "s"(ld) : "memory");
            }
        }
    };
    // Fragment reads and MFMAs are asm statements (fixed order, nothing counted by the compiler):
    //  * accumulators constrained to AGPRs ("+a"): with the builtin hipcc kept a third of the 224 accumulator registers in VGPRs and
    //    copied them to and fro at every step (1,250 v_accvgpr moves in the loop);
    //  * ds_read_b128 of the NEXT k-step's fragments interleaved with the MFMAs of this one; their one wait is the lgkmcnt(0) in
    //    front of the next barrier (~900 cycles later), which names the registers "+v" so that nothing the compiler does with them
    //    can move above it -- compiler-issued LDS loads got an s_waitcnt lgkmcnt(0) in front of the first MFMA of every step;
    //  * an accumulator is touched once per k-step, so no MFMA depends on a neighbour; the epilogue reads them after the nops below.
    constexpr int NACC0 = SH == 16 ? MTW : G::T32W;
    AccT acc[NACC0][S];
#pragma unroll
    for (int mt = 0; mt < NACC0; ++mt)
#pragma unroll
        for (int s = 0; s < S; ++s) acc[mt][s] = AccT{};
    // fragment a of the counts: SH 16 tile wm MTW + a; SH 32 (tile wm T32W + a / 2, half a % 2) -- consecutive blocks either way;
    // fragment b of the digits: SH 16 plane b of pair group wn; SH 32 (plane b / 2, half b % 2)
    const unsigned fbaseA = voff + (unsigned)wm * (unsigned)(NA * 1024), fbaseB = voff + (unsigned)(RT + wn * S) * 1024u;
#define GI8_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "i"(off))
#define GI8_MFMA16(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define GI8_MFMA32(c, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
    // one k-step: MFMAs on (fc, fd); in their shadow the fragment reads of the following k-step from LDS stage R into (fna, fnb)
    // and this wave's DMA instructions of the k-step AHEAD into stage W (an LDS-DMA instruction occupies the wave's issue for
    // ~60 cycles: eight of them back to back in front of the barrier left the matrix pipe idle a third of every step)
    // (tools/i8_ablate.py: without the DMA issue the kernel takes 0.39 instead of 0.46 ms -- and the second wave of a SIMD in the
    // eight-wave form does not cover for it: with its DMA slots placed half an interval behind its partner's the kernel measured
    // 0.466-0.474 ms against 0.456-0.458 with both waves on the same schedule)
    auto step = [&](i32x4 (&fc)[NA], i32x4 (&fd)[NB], i32x4 (&fna)[NA], i32x4 (&fnb)[NB], unsigned Roff, unsigned Woff) {
        const unsigned ra = fbaseA + Roff, rb = fbaseB + Roff;
        constexpr int RSTEP = G::NMFMA / (NA + NB) >= G::RSTEP ? G::RSTEP : 1;          // a fragment read after every RSTEP-th MFMA
        int nread = 0, ndma = 0;
        auto fill = [&](int m) {                              // what rides behind MFMA number m of the step
            if (m % RSTEP == 0 && nread < NA + NB) {
                if (!(G::ABL & 4)) {
                    if (nread < NA) GI8_DSREAD(fna[nread], ra, nread * 1024);
                    else GI8_DSREAD(fnb[nread - NA], rb, (nread - NA) * 1024);
                }
                ++nread;
            }
            if (ndma < G::PER && m == (G::DMA_HEAD ? ndma : (ndma * G::NMFMA) / G::PER + 1)) { if (!(G::ABL & 1)) issue_one(ndma, Woff); ++ndma; }
        };
        if constexpr (G::MIX) {
            // the last count tile of the wave runs only in a tall workgroup tile (one uniform branch per k-step); every DMA and fragment
            // read of the step rides behind the MFMAs in front of it
            static_assert(!G::MIX || (G::DMA_HEAD == 0 && ((G::PER - 1) * G::NMFMA) / G::PER + 1 < (MTW - 1) * S), "DMA slots in front of the optional MFMAs");
#pragma unroll
            for (int mt = 0; mt < MTW - 1; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) { GI8_MFMA16(acc[mt][s], fc[mt], fd[s]); fill(mt * S + s); }
            if (!shortT) {
#pragma unroll
                for (int s = 0; s < S; ++s) GI8_MFMA16(acc[MTW - 1][s], fc[MTW - 1], fd[s]);
            }
        } else if constexpr (SH == 16) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) { GI8_MFMA16(acc[mt][s], fc[mt], fd[s]); fill(mt * S + s); }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < G::T32W; ++i)
#pragma unroll
                    for (int s = 0; s < S; ++s) { GI8_MFMA32(acc[i][s], fc[2 * i + h], fd[2 * s + h]); fill((h * G::T32W + i) * S + s); }
        }
        // stragglers (more fragments than RSTEP-spaced slots)
#pragma unroll
        for (int r = 0; r < NA + NB; ++r)
            if (r >= nread) { if (r < NA) GI8_DSREAD(fna[r], ra, r * 1024); else GI8_DSREAD(fnb[r - NA], rb, (r - NA) * 1024); }
        advance();
    };
    // wait until this wave's DMAs of the k-step after next are the only ones in flight and the fragment reads of the set consumed
    // next have returned (the registers are named so that nothing the compiler does with them moves above the wait); then the
    // workgroup barrier: every wave's share of the next k-step has landed
    auto wait_frags = [&](i32x4 (&fa)[NA], i32x4 (&fb)[NB]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(fb[i]));
    };
    auto wait_barrier = [&](i32x4 (&fa)[NA], i32x4 (&fb)[NB]) {
        // (the count differs between waves; the branch holds no register operands -- with the pins inside it hipcc merged the two
        // variants through copies of all the fragment registers every step)
        if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::FLIGHT * G::PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::FLIGHT * (G::PER - 1)) : "memory");
        wait_frags(fa, fb);
        if (!(G::ABL & 2)) asm volatile("s_barrier" ::: "memory");
    };

    i32x4 fa0[NA], fb0[NB], fa1[NA], fb1[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) { fa0[i] = (i32x4){0, 0, 0, 0}; fa1[i] = fa0[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) { fb0[i] = (i32x4){0, 0, 0, 0}; fb1[i] = fb0[i]; }
    const auto issue_all = [&](unsigned stage_off) {
#pragma unroll
        for (int i = 0; i < G::PER; ++i) issue_one(i, stage_off);
        advance();
    };
    issue_all(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");       // k-step 0 has landed
#pragma unroll
    for (int i = 0; i < NA; ++i) GI8_DSREAD(fa0[i], fbaseA, i * 1024);
#pragma unroll
    for (int i = 0; i < NB; ++i) GI8_DSREAD(fb0[i], fbaseB, i * 1024);
    constexpr int AHEAD = G::PAIRB ? G::NS - 1 : G::NS;        // the DMA issued during k-step kb carries k-step kb + AHEAD
#pragma unroll
    for (int st = 1; st < AHEAD; ++st) issue_all(st * G::STAGE_BYTES);
    // iteration kb: stage R = (kb+1) % NS is read now; stage W = (kb + AHEAD) % NS receives k-step kb + AHEAD (plain ring: the stage
    // of k-step kb itself, whose fragments are in registers; pair barrier: the stage of k-step kb-1)
    unsigned W = (AHEAD % G::NS) * G::STAGE_BYTES, R = G::STAGE_BYTES;
    auto next = [](unsigned st) { return (st == (G::NS - 1) * G::STAGE_BYTES) ? 0u : st + G::STAGE_BYTES; };
    // KB is even (the host pads the rows to whole pairs of k-blocks): two steps per trip, the fragment sets swap roles
    for (int kb = 0; kb < KB; kb += 2) {
        wait_barrier(fa0, fb0);
        step(fa0, fb0, fa1, fb1, R, W);
        W = next(W); R = next(R);
        if constexpr (G::PAIRB) wait_frags(fa1, fb1); else wait_barrier(fa1, fb1);
        step(fa1, fb1, fa0, fb0, R, W);
        W = next(W); R = next(R);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMA results -> readable
#undef GI8_DSREAD
#undef GI8_MFMA16
#undef GI8_MFMA32
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // no DMA may land after the workgroup has gone

    // Epilogue: the lane's pair and replicates from the MFMA's C/D map -- 16x16: column lane & 15, rows 4 (lane >> 4) + reg;
    // 32x32: column lane & 31, rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    if constexpr (IND) {
        static_assert(!IND || SH == 16, "independent planes: 16x16x64 layout");
        const long rep0 = (long)ty * (16 * RT) + wm * (MTW * 16) + (lane >> 4) * 4;
        if (nty_short < 0) {
            // round 5 (IND launches only; they have no short tile rows): the sums ARE co-occurrence counts -- they leave as uint16, upper triangle of the
            // replicate's [C x ld16] count matrix (pair_dst: p ld16 + q; `gram` / `psize` in uint16 units), the layout the categorical wave step streams
            // (kernels_nmw.h): a quarter of the bytes of the fp64 slots, 16 consecutive pairs of a replicate = one 32-byte run, and no scatter pass
            // behind the product (nmg_kernel<3>: 0.4-0.5 ms per 1,000 problems); the lower triangle is mirrored by nmg_kernel<4>
            unsigned short* g16 = reinterpret_cast<unsigned short*>(gram);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int js = ((tx * 2 + wn) * S + s) * 16 + (lane & 15);
                if (js >= npair) continue;
                const double scs = pair_scale[js];
                unsigned short* gp = g16 + rep0 * psize + pair_dst[js];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (rep0 + mt * 16 + reg < nrep) *gp = (unsigned short)__double2int_rn((double)acc[mt][s][reg] * scs);
                        gp += psize;
                        asm volatile("" : "+v"(gp)::"memory");
                    }
                    gp += 12 * psize;
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int js = ((tx * 2 + wn) * S + s) * 16 + (lane & 15);          // plane s of this wave = pair group (2 tx + wn) S + s of the buffer
            if (js >= npair) continue;
            const double scs = pair_scale[js];
            double* gp = gram + rep0 * psize + pair_dst[js];
            const long d2 = pair_dst2 ? (long)pair_dst2[js] - (long)pair_dst[js] : 0;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (rep0 + mt * 16 + reg < nrep) {
                        const double v = (double)acc[mt][s][reg] * scs;
                        *gp = v;
                        if (pair_dst2 && pair_dst2[js] >= 0) gp[d2] = v;
                    }
                    gp += psize;
                    asm volatile("" : "+v"(gp)::"memory");
                }
                gp += 12 * psize;
            }
        }
        return;
    }
    const int j = SH == 16 ? (tx * 2 + wn) * 16 + (lane & 15) : tx * 32 + (lane & 31);
    if (j >= npair) return;
    const long dstj = pair_dst[j];
    const long dstj2 = pair_dst2 ? (long)pair_dst2[j] : -1;       // dense layout: the mirrored slot (q, p); -1 on the diagonal / packed layout
    const double sc = pair_scale[j];
    auto emit = [&](double* gp, bool live, auto pick) {            // pick(s): the element's accumulator of plane s
        if (live) {
            // sum_s acc_s 256^s from the low planes up: every term is exact in fp64, the partial sums round only once they
            // pass 2^53 (relative 2^-53 each): the recombination costs about one ulp
            double v = (double)pick(0);
#pragma unroll
            for (int s = 1; s < S; ++s) v = fma((double)pick(s), (double)(1ll << (8 * s)), v);
            *gp = v * sc;
            if (dstj2 >= 0) gp[dstj2 - dstj] = v * sc;
        }
    };
    if constexpr (SH == 16) {
        const int mtw = (G::MIX && shortT) ? MTW - 1 : MTW;          // count tiles of this wave in this tile
        const long rep0 = (long)ct0 * 16 + wm * (mtw * 16) + (lane >> 4) * 4;
        double* gp = gram + rep0 * psize + dstj;       // walks the replicates of this lane; opaque to the compiler so that it does not
#pragma unroll                                         // precompute (and spill) 32 addresses
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                emit(gp, mt < mtw && rep0 + mt * 16 + reg < nrep, [&](int s) { return acc[mt][s][reg]; });
                gp += psize;
                asm volatile("" : "+v"(gp)::"memory");
            }
            gp += 12 * psize;
        }
    } else {
        const long rep0 = (long)ty * 256 + wm * (G::T32W * 32) + (lane >> 5) * 4;
        double* gp = gram + rep0 * psize + dstj;
#pragma unroll
        for (int i = 0; i < G::T32W; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {                   // row (reg & 3) + 8 (reg >> 2) of the tile (+ 4 for the upper lanes)
                emit(gp, rep0 + i * 32 + (reg & 3) + 8 * (reg >> 2) < nrep, [&](int s) { return acc[i][s][reg]; });
                gp += ((reg & 3) == 3 ? 5 : 1) * psize;            // rows 0..3, 8..11, 16..19, 24..27
                asm volatile("" : "+v"(gp)::"memory");
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- the GEMM, persistent ("stream-K") form
// The tiled launch above runs 1,200 tiles on 256 CUs at the headline size: 4.69 rounds of one workgroup per CU, i.e. a fifth round in
// which 80 CUs idle (6 % of the kernel).  Here ONE workgroup per CU stays for the whole product.  With W workgroups per XCD and n tiles
// in an XCD's range of the same XCD-aware enumeration, workgroup j takes the tiles j, j + W, ... of the first floor(n / W) rounds whole
// -- the same tiles at the same time as the tiled launch, so the L2 sharing is unchanged -- and an equal share of the k-steps of the
// n mod W tiles left over: its share is a contiguous range of (tile, pair of k-steps) units, at most the END of one tile and the START of
// the next.  int32 partial sums are exact in any order, so splitting k costs no determinism: a workgroup that does not hold a tile's
// last k-steps ("contributor") writes its 8 S accumulator tiles per wave to a scratch slot and raises a per-wave flag; the one that does
// ("owner") adds the contributors' slots to its own accumulators and runs the epilogue.  Every workgroup does its contributor piece
// BEFORE its owner piece, so a flag an owner waits for was raised long before unless the hardware has not started that workgroup yet
// (it will: nothing it needs depends on the waiting one) -- the wait is bounded all the same, an expired wait poisons the tile with NaN
// (status 3 on its replicates) instead of hanging the device.  Slots and flags are written with agent-scope stores (sc1: written through the XCD's L2) and
// read behind an agent-scope acquire, so nothing depends on which XCD a workgroup landed on; `epoch` distinguishes launches (flags are
// never reset).  Default schedule of GramI8 (three LDS stages), 16x16x64 instruction.
// (the s_nop: a store of more than 8 bytes reads its data registers late -- a VALU write of them in the next slot corrupts the first
// dword; hipcc's hazard recognizer inserts the wait state for its own stores, not behind an asm statement)
__device__ __forceinline__ void st_sys_b128(void* p, i32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sys_b32(void* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_sys_b32(const void* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int S, int WM>
__global__ void __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(WM / 2, WM / 2)))
gram_i8_sk_kernel(const uint4* __restrict__ Cd, const uint4* __restrict__ Zs, int KB, int MT, int NT, int ntx, int nty, const int* __restrict__ pair_dst,
                  const double* __restrict__ pair_scale, int npair, long nrep, double* __restrict__ gram, long psize, i32x4* partial, unsigned* flags, unsigned epoch, int* err) {
    using G = GramI8<S, WM, I8_DEFAULT_VAR, 16>;
    static_assert(!G::PAIRB && G::NS == 3, "default schedule");
    constexpr int MTW = G::MTW, NA = G::NA, NB = G::NB, NW = G::NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // this workgroup's share: XCD x = id mod 8 walks the tiles [x per, (x + 1) per) of the enumeration; W workgroups per XCD
    const int total = ntx * nty, per = (total + 7) >> 3;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, W = gridDim.x >> 3;
    const int nx = max(0, min(per, total - x * per));
    const int R = nx / W, mrem = nx - R * W, KB2 = KB >> 1;
    const long units = (long)mrem * KB2;
    const long ra = units * j / W, rb = units * (j + 1) / W;
    // pieces of the left-over tiles: A = [ra, end of its tile or rb), B = the start of the next tile (if the range crosses a tile end)
    const int i0 = (int)(ra / KB2), i1 = rb > ra ? (int)((rb - 1) / KB2) : i0;
    const int a0 = (int)(ra - (long)i0 * KB2), a1 = (int)(min(rb, (long)(i0 + 1) * KB2) - (long)i0 * KB2);
    const int b1 = (int)(rb - (long)i1 * KB2);
    const bool hasA = rb > ra, hasB = hasA && i1 > i0;
    const bool ownA = hasA && a1 == KB2, ownB = hasB && b1 == KB2;
    const int nitems = R + (hasA ? 1 : 0) + (hasB ? 1 : 0);

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_raw);
    const unsigned voff = (unsigned)lane * 16u;
    const int myper = (G::NBLK - wave + NW - 1) / NW;
    const bool full = myper == G::PER;
    const unsigned fbaseA = voff + (unsigned)wm * (unsigned)(NA * 1024), fbaseB = voff + (unsigned)(16 + wn * S) * 1024u;
    i32x4* const myslot = partial + ((size_t)blockIdx.x * NW + wave) * (size_t)(MTW * S) * 64 + lane;
#define GI8_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "i"(off))
#define GI8_MFMA16(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))

    for (int it = 0; it < nitems; ++it) {
        // item: (tile of the XCD's list, pairs of k-steps [p0, p1), owner?) -- whole tiles first, then the contributor piece, then the owner piece
        int tl, p0, p1; bool owner;
        if (it < R) { tl = it * W + j; p0 = 0; p1 = KB2; owner = true; }
        else {
            const bool second = it > R;
            // both pieces present: the contributor goes first (B is one unless it is a whole tile)
            const bool firstIsB = hasB && !ownB;
            const bool useB = hasB && (second ? !firstIsB : firstIsB);
            if (useB) { tl = R * W + i1; p0 = 0; p1 = b1; owner = ownB; }
            else { tl = R * W + i0; p0 = a0; p1 = a1; owner = ownA; }
        }
        const int gidx = x * per + tl;
        const int srow = 4 * ntx;
        const int sr = gidx / srow, rem = gidx - sr * srow;
        const int nr = min(4, nty - 4 * sr);
        const int tx = rem / nr, ty = 4 * sr + (rem - tx * nr);

        const char* src[G::PER];
        long inc[G::PER];
        unsigned dst[G::PER];
#pragma unroll
        for (int i = 0; i < G::PER; ++i) {
            const int b = min(wave + NW * i, G::NBLK - 1);
            const bool isA = b < 16;
            inc[i] = (isA ? (long)MT : (long)NT) * 1024;
            src[i] = (isA ? (const char*)(Cd + ((long)ty * 16 + b) * 64) : (const char*)(Zs + ((long)tx * 2 * S + (b - 16)) * 64)) + (long)(2 * p0) * inc[i];
            dst[i] = lds0 + (unsigned)b * 1024u;
        }
        auto advance = [&]() {
#pragma unroll
            for (int i = 0; i < G::PER; ++i) src[i] += inc[i];
        };
        auto issue_one = [&](int i, unsigned stage_off) {
            if (i < G::PER - 1 || full) glds_block(src[i], voff, dst[i] + stage_off);
        };
        i32x4 acc[MTW][S];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int s = 0; s < S; ++s) acc[mt][s] = i32x4{};
        auto step = [&](i32x4 (&fc)[NA], i32x4 (&fd)[NB], i32x4 (&fna)[NA], i32x4 (&fnb)[NB], unsigned Roff, unsigned Woff) {
            const unsigned ra_ = fbaseA + Roff, rb_ = fbaseB + Roff;
            constexpr int RSTEP = G::NMFMA / (NA + NB) >= G::RSTEP ? G::RSTEP : 1;
            int nread = 0, ndma = 0;
            auto fill = [&](int m) {
                if (m % RSTEP == 0 && nread < NA + NB) {
                    if (nread < NA) GI8_DSREAD(fna[nread], ra_, nread * 1024);
                    else GI8_DSREAD(fnb[nread - NA], rb_, (nread - NA) * 1024);
                    ++nread;
                }
                if (ndma < G::PER && m == (G::DMA_HEAD ? ndma : (ndma * G::NMFMA) / G::PER + 1)) { issue_one(ndma, Woff); ++ndma; }
            };
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) { GI8_MFMA16(acc[mt][s], fc[mt], fd[s]); fill(mt * S + s); }
#pragma unroll
            for (int r = 0; r < NA + NB; ++r)
                if (r >= nread) { if (r < NA) GI8_DSREAD(fna[r], ra_, r * 1024); else GI8_DSREAD(fnb[r - NA], rb_, (r - NA) * 1024); }
            advance();
        };
        auto wait_barrier = [&](i32x4 (&fa)[NA], i32x4 (&fb)[NB]) {
            if (full) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::FLIGHT * G::PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G::FLIGHT * (G::PER - 1)) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(fb[i]));
            asm volatile("s_barrier" ::: "memory");
        };
        i32x4 fa0[NA], fb0[NB], fa1[NA], fb1[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) { fa0[i] = (i32x4){0, 0, 0, 0}; fa1[i] = fa0[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { fb0[i] = (i32x4){0, 0, 0, 0}; fb1[i] = fb0[i]; }
        const auto issue_all = [&](unsigned stage_off) {
#pragma unroll
            for (int i = 0; i < G::PER; ++i) issue_one(i, stage_off);
            advance();
        };
        issue_all(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < NA; ++i) GI8_DSREAD(fa0[i], fbaseA, i * 1024);
#pragma unroll
        for (int i = 0; i < NB; ++i) GI8_DSREAD(fb0[i], fbaseB, i * 1024);
#pragma unroll
        for (int st = 1; st < G::NS; ++st) issue_all(st * G::STAGE_BYTES);
        unsigned Wst = 0, Rst = G::STAGE_BYTES;
        auto next = [](unsigned st) { return (st == (G::NS - 1) * G::STAGE_BYTES) ? 0u : st + G::STAGE_BYTES; };
        for (int pp = p0; pp < p1; ++pp) {
            wait_barrier(fa0, fb0);
            step(fa0, fb0, fa1, fb1, Rst, Wst);
            Wst = next(Wst); Rst = next(Rst);
            wait_barrier(fa1, fb1);
            step(fa1, fb1, fa0, fb0, Rst, Wst);
            Wst = next(Wst); Rst = next(Rst);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done with the LDS ring: the next item may overwrite it

        if (!owner) {
            // contributor: the accumulators to this workgroup's slot (system scope), then the wave's flag
            i32x4* sp = myslot;                               // a running pointer, opaque per store: hipcc otherwise keeps (and spills) 8 S addresses
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int s = 0; s < S; ++s) { st_sys_b128(sp, acc[mt][s]); sp += 64; asm volatile("" : "+v"(sp)); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_sys_b32(flags + (size_t)blockIdx.x * NW + wave, epoch);
            continue;
        }
        // owner: add the slots of the workgroups that hold the earlier k-steps of this tile (left-over tiles only: j' < j whose range
        // reaches into the tile)
        bool poisoned = false;
        if (it >= R) {
            const long tile_u0 = (long)(tl - R * W) * KB2;
            for (int jc = j - 1; jc >= 0 && units * (jc + 1) / W > tile_u0; --jc) {
                if (units * (jc + 1) / W == units * jc / W) continue;                   // (an empty share)
                const unsigned wc = (unsigned)(jc * 8 + x);
                const unsigned* fl = flags + (size_t)wc * NW + wave;
                int spins = 0;
                while (ld_sys_b32(fl) != epoch) {
                    if (++spins > (1 << 20)) { poisoned = true; if (lane == 0) atomicOr(err, 4); break; }      // (~1 s)
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const i32x4* slot = partial + ((size_t)wc * NW + wave) * (size_t)(MTW * S) * 64 + lane;
                // half of the slot in flight at a time (4 S loads = 16 S registers; the fragment registers are dead here): one memory
                // round trip per half -- S loads per batch made it 2 MTW / ... round trips of ~2 us each in front of every split tile's epilogue
                constexpr int HB = (MTW + 1) / 2;
#pragma unroll
                for (int h = 0; h < MTW; h += HB) {
                    i32x4 t[HB][S];
#pragma unroll
                    for (int mt = 0; mt < HB; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) if (h + mt < MTW) t[mt][s] = __builtin_nontemporal_load(slot + (size_t)(mt * S + s) * 64);
#pragma unroll
                    for (int mt = 0; mt < HB; ++mt)
#pragma unroll
                        for (int s = 0; s < S; ++s) if (h + mt < MTW) acc[h + mt][s] += t[mt][s];
                    slot += (size_t)HB * S * 64;
                    asm volatile("" : "+v"(slot) : : "memory");
                }
            }
        }
        const int jp = (tx * 2 + wn) * 16 + (lane & 15);
        if (jp < npair) {
            const long dstj = pair_dst[jp];
            const double sc = poisoned ? __builtin_nan("") : pair_scale[jp];
            const long rep0 = (long)ty * 256 + wm * (MTW * 16) + (lane >> 4) * 4;
            double* gp = gram + rep0 * psize + dstj;
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (rep0 + mt * 16 + reg < nrep) {
                        double v = (double)acc[mt][0][reg];
#pragma unroll
                        for (int s = 1; s < S; ++s) v = fma((double)acc[mt][s][reg], (double)(1ll << (8 * s)), v);
                        *gp = v * sc;
                    }
                    gp += psize;
                    asm volatile("" : "+v"(gp)::"memory");
                }
                gp += 12 * psize;
            }
        }
    }
#undef GI8_DSREAD
#undef GI8_MFMA16
}
