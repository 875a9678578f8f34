// kernels_gram.h -- Device kernels, part 2: the fp64 MFMA weighted Gram kernels (rows / wide / block variants) and the fixed-order reduce.
// Device code shared by the translation units of libplspm_hip.so (host_internal.h lists them); not a stand-alone header.
#pragma once
#include <utility>

// ------------------------------------------------------------------------------------------------ Gram (fp64 MFMA)
// v_mfma_f64_16x16x4_f64: D[16x16] += A[16x4] B[4x16].  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; for the
// Gram both are the SAME element xa[row_k][col_t(l&15)] (A additionally times the multiplicity), so one 16-byte
// load per 32 columns feeds two tiles: lane (k, i) reads columns 32*g + 2i, 2i+1 of row k -> tile 2g holds the even
// columns of group g, tile 2g+1 the odd ones (packed_tile_of / packed_pos_of in solver_core.h).  An odd tile count T (16-column
// granularity of the padded width: 201 columns -> 13 tiles = 91 MFMAs per k-group instead of 14 tiles = 105) leaves the last
// tile without a partner: lane (k, i) loads its single column 16(T-1) + i with an 8-byte load.
// Lane l ends up with D[(l>>4) + 4*reg][l&15] in acc[reg].
//
// Software pipeline.  hipcc de-pipelines a C++ prefetch here (it re-issues the loop-carried loads next to their
// use, exposing the full (list entry -> row address -> row data) latency every k-group -- measured 42 % MFMA
// utilisation), so the loads are issued with inline asm the compiler does not count, in a two-stage ping-pong:
//     stage s:  s_waitcnt vmcnt(0)                      rows(s) and entry(s+1) have landed
//               issue rows(s+1) <- Xa[entry(s+1).row]   } in flight under the MFMAs of stage s
//               issue entry(s+2)                        }
//               NT x v_mfma_f64_16x16x4_f64 on rows(s)
// Rules followed (cdna_hip_programming.md 5.7): destinations are tied "+v" operands (no compiler copy of an
// in-flight register), every destination is named by the wait statement before its first consumer, the
// accumulators are pinned "+a" at stage boundaries so no MFMA drifts across a wait, and a final vmcnt(0)
// precedes any other use of those registers.  Audit with tools/kernel_resources.py + -save-temps: the loop
// must show no v_accvgpr_* and no v_mov of a load destination.
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef int iv2 __attribute__((ext_vector_type(2)));

template <int T>
struct TileIdx {
    static constexpr int NTILE = T * (T + 1) / 2;
    __host__ __device__ static constexpr int of(int t, int u) { return t * T - t * (t - 1) / 2 + (u - t); }
};
// Tiles owned by wave W of NW when the upper tiles are dealt round-robin (NW = 1: one wave owns all).
template <int T, int NW, int W>
struct Own {
    static constexpr int COUNT = (TileIdx<T>::NTILE - W + NW - 1) / NW;
    __host__ __device__ static constexpr bool mine(int li) { return li % NW == W; }
    __host__ __device__ static constexpr int slot(int li) { return li / NW; }
};

template <int T, int NW, int W>
using AccArr = d4[Own<T, NW, W>::COUNT];
template <int T>
using RowArr = dv2[T / 2];

template <int Q, int G>
struct RowLoader {
    static __device__ __forceinline__ void issue(dv2 (&v)[G], const double* p) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(v[Q]) : "v"(p), "i"(Q * 256) : "memory");
        RowLoader<Q + 1, G>::issue(v, p);
    }
    static __device__ __forceinline__ void pin(dv2 (&v)[G]) { asm volatile("" : "+v"(v[Q])); RowLoader<Q + 1, G>::pin(v); }
};
template <int G>
struct RowLoader<G, G> {
    static __device__ __forceinline__ void issue(dv2 (&)[G], const double*) {}
    static __device__ __forceinline__ void pin(dv2 (&)[G]) {}
};
// Odd tile counts: the last tile has no partner; lane (k, i) loads its single column 16(T-1) + i of row k with an 8-byte load.
__device__ __forceinline__ void issue_tail(double& t, const double* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(t) : "v"(p) : "memory"); }
__device__ __forceinline__ void issue_entry(iv2& e, const int2* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(e) : "v"(p) : "memory"); }

// One pipeline stage (see above).  DENSE: rows are consecutive (single fit), no (row,count) list.
template <int T, int NW, int W, bool DENSE>
__device__ __forceinline__ void gram_stage(AccArr<T, NW, W>& acc, RowArr<T>& Vcur, RowArr<T>& Vnext, double& Tcur, double& Tnext, iv2& Enext, iv2& Eafter, double cnt,
                                           const double* xbase, const double* tbase, const int2* eptr_after, long dense_row_next) {
    constexpr int G = T / 2, PA = 16 * T;
    constexpr bool ODD = (T & 1) != 0;
#pragma unroll
    for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) asm volatile("" : "+a"(acc[t]));
    if (DENSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(Enext)::"memory");
    RowLoader<0, G>::pin(Vcur);
    if (ODD) asm volatile("" : "+v"(Tcur));
    double x[T];
#pragma unroll
    for (int q = 0; q < G; ++q) { x[2 * q] = Vcur[q].x; x[2 * q + 1] = Vcur[q].y; }
    if (ODD) x[T - 1] = Tcur;
    const long rnext = DENSE ? dense_row_next : (long)Enext.x;
    RowLoader<0, G>::issue(Vnext, xbase + rnext * PA);
    if (ODD) issue_tail(Tnext, tbase + rnext * PA);
    if (!DENSE) issue_entry(Eafter, eptr_after);
    // every MFMA operand below depends on a value pinned AFTER the loads were issued: none is scheduled above them.  DENSE rows
    // need no multiplicity (rows past the end are redirected to the all-zero pad row behind the matrix): A and B are the same register.
    if (DENSE) {
#pragma unroll
        for (int t = 0; t < T; ++t) asm volatile("" : "+v"(x[t]));
    } else asm volatile("" : "+v"(cnt));
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const double a = DENSE ? x[t] : cnt * x[t];
#pragma unroll
        for (int u = t; u < T; ++u) {
            const int li = TileIdx<T>::of(t, u);
            const int sl = Own<T, NW, W>::slot(li);
            if (Own<T, NW, W>::mine(li)) acc[sl] = MFMA_F64(a, x[u], acc[sl]);
        }
    }
}

// The k-group walk of one wave: groups g0, g0 + gs, ... < ng; accumulates its owned tiles.
template <int T, int NW, int W, bool DENSE>
__device__ __forceinline__ void gram_walk(AccArr<T, NW, W>& acc, const double* __restrict__ Xa, long N, const int2* __restrict__ e, int ng, int g0,
                                          int gs, int lane) {
    constexpr int G = T / 2, PA = 16 * T;
    const int k = lane >> 4, i = lane & 15;
    const double* xbase = Xa + 2 * i;
    const double* tbase = Xa + 32 * G + i;           // odd T: the partner-less last tile
    constexpr bool ODD = (T & 1) != 0;
    const int2* ek = e + k;
    const int niter = (g0 < ng) ? (ng - g0 + gs - 1) / gs : 0;
    const int last = ng - 1;
    auto gof = [&](int it) { const int g = g0 + it * gs; return g < last ? g : last; };
    auto eaddr = [&](int it) { return ek + 4 * (long)gof(it); };
    // dense walk: row N is the all-zero pad row plspm_upload keeps behind the matrix -- stages past the end and the rows >= N of the
    // last k-group read it instead of being masked by a multiplicity
    auto drow = [&](int it) { const long r = 4 * ((long)g0 + (long)it * gs) + k; return (it < niter && r < N) ? r : N; };
    auto dcnt = [&](int) { return 1; };
#pragma unroll
    for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) { acc[t] = (d4){0.0, 0.0, 0.0, 0.0}; asm volatile("" : "+a"(acc[t])); }
    iv2 EA = {0, 0}, EB = {0, 0};
    dv2 VA[G], VB[G];
    double TA = 0.0, TB = 0.0;
#pragma unroll
    for (int q = 0; q < G; ++q) { VA[q] = (dv2){0.0, 0.0}; VB[q] = (dv2){0.0, 0.0}; }
    int cA;
    if (DENSE) {
        RowLoader<0, G>::issue(VA, xbase + drow(0) * PA);
        if (ODD) issue_tail(TA, tbase + drow(0) * PA);
        cA = dcnt(0);
    } else {
        issue_entry(EA, eaddr(0));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(EA)::"memory");
        RowLoader<0, G>::issue(VA, xbase + (long)EA.x * PA);
        if (ODD) issue_tail(TA, tbase + (long)EA.x * PA);
        cA = (niter > 0) ? EA.y : 0;
        issue_entry(EB, eaddr(1));
    }
    for (int it = 0; it < niter; it += 2) {
        // stage A: consume VA; prefetch VB <- rows(it+1), EA <- entry(it+2)
        gram_stage<T, NW, W, DENSE>(acc, VA, VB, TA, TB, EB, EA, (double)cA, xbase, tbase, DENSE ? nullptr : eaddr(it + 2), DENSE ? drow(it + 1) : 0);
        const int cB = DENSE ? dcnt(it + 1) : ((it + 1 < niter) ? EB.y : 0);
        // stage B: consume VB; prefetch VA <- rows(it+2), EB <- entry(it+3)
        gram_stage<T, NW, W, DENSE>(acc, VB, VA, TB, TA, EA, EB, (double)cB, xbase, tbase, DENSE ? nullptr : eaddr(it + 3), DENSE ? drow(it + 2) : 0);
        cA = DENSE ? dcnt(it + 2) : EA.y;
    }
    if (DENSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(EA), "+v"(EB)::"memory");
    RowLoader<0, G>::pin(VA);
    RowLoader<0, G>::pin(VB);
    if (ODD) asm volatile("" : "+v"(TA), "+v"(TB));
#pragma unroll
    for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) asm volatile("" : "+a"(acc[t]));
}

// The dense walk on a ring of row buffers (round 6; single fits of wide models -- configs[4]: T = 13, four waves of 23 tiles, one wave per SIMD).  Counters of
// the two-stage ping-pong above on that launch (profiles/r05_pmc_rows.json): 3.9 % of the wave cycles wait for memory, 81 % wait to ISSUE, the matrix pipe is 82 %
// busy at 2.15 GHz -- what the pipe loses is not latency but the instructions between the MFMA blocks: 26 register copies per stage (x[] = the row buffer, so that
// the next stage's loads may overwrite the buffer while the MFMAs read the copies) + the loads' address arithmetic, on a SIMD with nobody else to issue.
// With D >= 4 buffers nothing needs a copy: stage s's MFMAs read buffer s % D in place, the loads issued in stage s go to the buffer of stage s + D - 2, last read
// by the MFMAs of stage s - 2 -- two whole stages (~3,000 clocks) before the data can land; s_waitcnt vmcnt((D - 3) x loads-per-stage) leaves the younger
// stages in flight.  Stages past the end read the all-zero pad row (plspm_upload keeps it behind the matrix).
template <class F, int... Bs>
__device__ __forceinline__ void ring_stages(F&& f, std::integer_sequence<int, Bs...>) { (f(std::integral_constant<int, Bs>{}), ...); }
template <int T, int NW, int W, int D>
__device__ __forceinline__ void gram_walk_dense_ring(AccArr<T, NW, W>& acc, const double* __restrict__ Xa, long N, int ng, int g0, int gs, int lane) {
    constexpr int G = T / 2, PA = 16 * T;
    constexpr bool ODD = (T & 1) != 0;
    constexpr int LPS = G + (ODD ? 1 : 0);                      // loads per stage and lane
    constexpr int PD = D - 2;                                   // prefetch distance in stages
    static_assert(D >= 4 && (PD - 1) * LPS <= 63, "ring depth: a buffer rests two stages before it is reloaded; the wait counter has six bits");
    const int k = lane >> 4, i = lane & 15;
    const double* xbase = Xa + 2 * i;
    const double* tbase = Xa + 32 * G + i;
    const int niter = (g0 < ng) ? (ng - g0 + gs - 1) / gs : 0;
    auto drow = [&](int it) { const long r = 4 * ((long)g0 + (long)it * gs) + k; return (it < niter && r < N) ? r : N; };
#pragma unroll
    for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) { acc[t] = (d4){0.0, 0.0, 0.0, 0.0}; asm volatile("" : "+a"(acc[t])); }
    dv2 V[D][G];
    double Tt[D];
#pragma unroll
    for (int b = 0; b < D; ++b) {
        Tt[b] = 0.0;
#pragma unroll
        for (int q = 0; q < G; ++q) V[b][q] = (dv2){0.0, 0.0};
    }
#pragma unroll
    for (int b = 0; b < PD; ++b) {                              // prologue: stages 0 .. PD - 1 in flight
        RowLoader<0, G>::issue(V[b], xbase + drow(b) * PA);
        if (ODD) issue_tail(Tt[b], tbase + drow(b) * PA);
    }
    for (int it = 0; it < niter; it += D) {
        ring_stages([&](auto bc) {                              // stage it + b on buffer b
            constexpr int b = decltype(bc)::value;
            constexpr int bn = (b + PD) % D;                    // the buffer of stage it + b + PD == that of stage it + b - 2
#pragma unroll
            for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) asm volatile("" : "+a"(acc[t]));
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * LPS) : "memory");
            RowLoader<0, G>::pin(V[b]);
            if (ODD) asm volatile("" : "+v"(Tt[b]));
            const long rnext = drow(it + b + PD);
            RowLoader<0, G>::issue(V[bn], xbase + rnext * PA);
            if (ODD) issue_tail(Tt[bn], tbase + rnext * PA);
            auto xv = [&](int t) -> double { return (ODD && t == T - 1) ? Tt[b] : ((t & 1) ? V[b][t >> 1].y : V[b][t >> 1].x); };
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int u = t; u < T; ++u) {
                    const int li = TileIdx<T>::of(t, u);
                    const int sl = Own<T, NW, W>::slot(li);
                    if (Own<T, NW, W>::mine(li)) acc[sl] = MFMA_F64(xv(t), xv(u), acc[sl]);
                }
            }
        }, std::make_integer_sequence<int, D>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < D; ++b) { RowLoader<0, G>::pin(V[b]); if (ODD) asm volatile("" : "+v"(Tt[b])); }
#pragma unroll
    for (int t = 0; t < Own<T, NW, W>::COUNT; ++t) asm volatile("" : "+a"(acc[t]));
}

// Rows-split variant (T <= 4): every wave of the 256-thread workgroup keeps ALL upper tiles and takes every 4th
// k-group of the (row,count) list; a two-level LDS tree adds the four partial accumulators at the end.
template <int T, bool DENSE>
__global__ void __launch_bounds__(256) gram_rows_kernel(const double* __restrict__ Xa, long N, const int2* __restrict__ ent,
                                                         const int* __restrict__ nent, long ent_stride, double* __restrict__ out) {
    constexpr int NT = TileIdx<T>::NTILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* red = reinterpret_cast<double*>(smem_raw);      // [2][NT*256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long problem = blockIdx.y;
    const int nchunks = gridDim.x, chunk = blockIdx.x;
    const int2* e = DENSE ? nullptr : ent + problem * ent_stride;
    const int ng = DENSE ? (int)((N + 3) >> 2) : ((nent[problem] + 3) >> 2);

    d4 acc[NT];
    gram_walk<T, 1, 0, DENSE>(acc, Xa, N, e, ng, chunk * 4 + wave, nchunks * 4, lane);

    // tree reduce: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores.  One tile at a time
    // (compiler fence per tile) so the epilogue does not inflate the kernel's VGPR budget past the main loop's.
#define TILE_FENCE() asm volatile("" ::: "memory")
    if (wave >= 2) {
        double* dst = red + (long)(wave - 2) * NT * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(t * 4 + r) * 64 + lane] = acc[t][r];
            TILE_FENCE();
        }
    }
    __syncthreads();
    if (wave < 2) {
        const double* src = red + (long)wave * NT * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += src[(t * 4 + r) * 64 + lane];
            TILE_FENCE();
        }
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] = acc[t][r];
            TILE_FENCE();
        }
    }
    __syncthreads();
    if (wave == 0) {
        double* dst = out + (problem * nchunks + chunk) * (long)(NT * 256);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(t * 4 + r) * 64 + lane] = acc[t][r] + red[(t * 4 + r) * 64 + lane];
            TILE_FENCE();
        }
    }
#undef TILE_FENCE
}

// Tile-split variant (6 <= T <= 16): the NW waves of a workgroup walk the SAME k-groups; wave W owns the upper tiles
// whose linear index == W (mod NW), so no reduction is needed and the accumulators stay within the register file.
template <int T, int NW, int W, bool DENSE, int RING = 0>
__device__ __forceinline__ void gram_wide_body(const double* __restrict__ Xa, long N, const int2* __restrict__ e, int ng, int chunk, int nchunks,
                                                double* __restrict__ dst, int lane) {
    d4 acc[Own<T, NW, W>::COUNT];
    if constexpr (DENSE && RING >= 4) gram_walk_dense_ring<T, NW, W, RING>(acc, Xa, N, ng, chunk, nchunks, lane);
    else gram_walk<T, NW, W, DENSE>(acc, Xa, N, e, ng, chunk, nchunks, lane);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int u = t; u < T; ++u) {
            const int li = TileIdx<T>::of(t, u);
            if (Own<T, NW, W>::mine(li)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(li * 4 + r) * 64 + lane] = acc[Own<T, NW, W>::slot(li)][r];
            }
        }
}
template <int T, int NW, int W, bool DENSE, int RING = 0>
struct WideDispatch {
    __device__ static __forceinline__ void run(int wave, const double* Xa, long N, const int2* e, int ng, int chunk, int nchunks, double* dst, int lane) {
        if (wave == W) gram_wide_body<T, NW, W, DENSE, RING>(Xa, N, e, ng, chunk, nchunks, dst, lane);
        else WideDispatch<T, NW, W + 1, DENSE, RING>::run(wave, Xa, N, e, ng, chunk, nchunks, dst, lane);
    }
};
template <int T, int NW, bool DENSE, int RING>
struct WideDispatch<T, NW, NW, DENSE, RING> {
    __device__ static __forceinline__ void run(int, const double*, long, const int2*, int, int, int, double*, int) {}
};
// NWV "virtual" waves share the tiles; a workgroup carries NWP of them and blockIdx.z selects which slice
// (NWV == NWP: one workgroup per k-group walk; NWV == 2*NWP: two workgroups walk the same rows, disjoint tiles).
template <int T, int NWV, int NWP, bool DENSE, int RING = 0>
__global__ void __launch_bounds__(NWP * 64) gram_wide_kernel(const double* __restrict__ Xa, long N, const int2* __restrict__ ent,
                                                              const int* __restrict__ nent, long ent_stride, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + NWP * (int)blockIdx.z;
    const long problem = blockIdx.y;
    const int2* e = DENSE ? nullptr : ent + problem * ent_stride;
    const int ng = DENSE ? (int)((N + 3) >> 2) : ((nent[problem] + 3) >> 2);
    double* dst = out + (problem * gridDim.x + blockIdx.x) * (long)(TileIdx<T>::NTILE * 256);
    WideDispatch<T, NWV, 0, DENSE, RING>::run(wave, Xa, N, e, ng, (int)blockIdx.x, (int)gridDim.x, dst, lane);
}


// Block variant (T > 16, i.e. 255 <= P <= 1022): the upper triangle of the T x T tile grid is cut into 4 x 4-tile super-blocks
// (64 x 64 columns); every wave owns ONE super-block (16 accumulator tiles, 10 on the diagonal), the four waves of a workgroup
// walk the same k-groups, and blockIdx.z enumerates groups of four super-blocks.  Tile coordinates are run-time values (wave-
// uniform), so one instantiation serves every T; out-of-range tiles of the last super-block row/column are skipped.
// two 16-byte loads: the first 32-column group of a super-block and (off doubles further) its second one
__device__ __forceinline__ void issue_pair(dv2 (&v)[2], const double* p, int off) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v[0]) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v[1]) : "v"(p + off) : "memory");
}
template <bool DENSE>
__device__ __forceinline__ void block_stage(d4 (&acc)[16], dv2 (&Rc)[2], dv2 (&Cc)[2], dv2 (&Rn)[2], dv2 (&Cn)[2], iv2& Enext, iv2& Eafter, double cnt,
                                            const double* rbase, const double* cbase, int r1off, int c1off, int PA, const int2* eptr_after,
                                            long dense_row_next, unsigned valid) {
#pragma unroll
    for (int t = 0; t < 16; ++t) asm volatile("" : "+a"(acc[t]));
    if (DENSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(Enext)::"memory");
    RowLoader<0, 2>::pin(Rc);
    RowLoader<0, 2>::pin(Cc);
    const double xr[4] = {Rc[0].x, Rc[0].y, Rc[1].x, Rc[1].y};
    const double xc[4] = {Cc[0].x, Cc[0].y, Cc[1].x, Cc[1].y};
    const long rnext = DENSE ? dense_row_next : (long)Enext.x;
    issue_pair(Rn, rbase + rnext * PA, r1off);
    issue_pair(Cn, cbase + rnext * PA, c1off);
    if (!DENSE) issue_entry(Eafter, eptr_after);
    double xa[4] = {xr[0], xr[1], xr[2], xr[3]};
    if (DENSE) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) asm volatile("" : "+v"(xa[ti]));
    } else asm volatile("" : "+v"(cnt));
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
        const double a = DENSE ? xa[ti] : cnt * xr[ti];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
            if (valid & (1u << (ti * 4 + tj))) acc[ti * 4 + tj] = MFMA_F64(a, xc[tj], acc[ti * 4 + tj]);      // wave-uniform mask: scalar branch
    }
}
template <bool DENSE>
__global__ void __launch_bounds__(256) gram_block_kernel(const double* __restrict__ Xa, long N, int T, const int2* __restrict__ ent,
                                                          const int* __restrict__ nent, long ent_stride, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, k = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long problem = blockIdx.y;
    const int PA = 16 * T, TB = (T + 3) >> 2, nsb = TB * (TB + 1) / 2;
    const int sb = (int)blockIdx.z * 4 + wave;
    if (sb >= nsb) return;
    int bi = 0, rem = sb;
    while (rem >= TB - bi) { rem -= TB - bi; ++bi; }
    const int bj = bi + rem;
    unsigned valid = 0;
    for (int ti = 0; ti < 4; ++ti)
        for (int tj = 0; tj < 4; ++tj) {
            const int t = 4 * bi + ti, u = 4 * bj + tj;
            if (t < T && u < T && t <= u) valid |= 1u << (ti * 4 + tj);
        }
    const int2* e = DENSE ? nullptr : ent + problem * ent_stride + k;
    const int ng = DENSE ? (int)((N + 3) >> 2) : ((nent[problem] + 3) >> 2);
    const int g0 = blockIdx.x, gs = gridDim.x;
    const int niter = (g0 < ng) ? (ng - g0 + gs - 1) / gs : 0;
    const int last = ng - 1;
    // row / column fragments: 32-column groups 2*bi, 2*bi+1 and 2*bj, 2*bj+1; a partial last super-block has no second group:
    // its load is pointed at the first group again (those tiles are masked out of `valid`)
    const double* rbase = Xa + 32 * (2 * bi) + 2 * i;
    const double* cbase = Xa + 32 * (2 * bj) + 2 * i;
    const int r1off = ((2 * bi + 1) < T / 2) ? 32 : 0, c1off = ((2 * bj + 1) < T / 2) ? 32 : 0;
    auto gof = [&](int it) { const int g = g0 + it * gs; return g < last ? g : last; };
    auto eaddr = [&](int it) { return e + 4 * (long)gof(it); };
    auto drow = [&](int it) { const long r = 4 * ((long)g0 + (long)it * gs) + k; return (it < niter && r < N) ? r : N; };      // N: the all-zero pad row
    auto dcnt = [&](int) { return 1; };
    d4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[t] = (d4){0.0, 0.0, 0.0, 0.0}; asm volatile("" : "+a"(acc[t])); }
    iv2 EA = {0, 0}, EB = {0, 0};
    dv2 RA[2], CA[2], RB[2], CB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { RA[q] = (dv2){0.0, 0.0}; CA[q] = RA[q]; RB[q] = RA[q]; CB[q] = RA[q]; }
    const double* rb = rbase;
    const double* cb = cbase;
    int cA;
    if (DENSE) {
        issue_pair(RA, rb + drow(0) * PA, r1off);
        issue_pair(CA, cb + drow(0) * PA, c1off);
        cA = dcnt(0);
    } else {
        issue_entry(EA, eaddr(0));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(EA)::"memory");
        issue_pair(RA, rb + (long)EA.x * PA, r1off);
        issue_pair(CA, cb + (long)EA.x * PA, c1off);
        cA = (niter > 0) ? EA.y : 0;
        issue_entry(EB, eaddr(1));
    }
    for (int it = 0; it < niter; it += 2) {
        block_stage<DENSE>(acc, RA, CA, RB, CB, EB, EA, (double)cA, rb, cb, r1off, c1off, PA, DENSE ? nullptr : eaddr(it + 2), DENSE ? drow(it + 1) : 0, valid);
        const int cB = DENSE ? dcnt(it + 1) : ((it + 1 < niter) ? EB.y : 0);
        block_stage<DENSE>(acc, RB, CB, RA, CA, EA, EB, (double)cB, rb, cb, r1off, c1off, PA, DENSE ? nullptr : eaddr(it + 3), DENSE ? drow(it + 2) : 0, valid);
        cA = DENSE ? dcnt(it + 2) : EA.y;
    }
    if (DENSE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(EA), "+v"(EB)::"memory");
    RowLoader<0, 2>::pin(RA); RowLoader<0, 2>::pin(CA); RowLoader<0, 2>::pin(RB); RowLoader<0, 2>::pin(CB);
#pragma unroll
    for (int t = 0; t < 16; ++t) asm volatile("" : "+a"(acc[t]));
    double* dst = out + (problem * gridDim.x + blockIdx.x) * (long)(T * (T + 1) / 2) * 256;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
            if (valid & (1u << (ti * 4 + tj))) {
                const int t = 4 * bi + ti, u = 4 * bj + tj;
                const long li = (long)t * T - (long)t * (t - 1) / 2 + (u - t);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(li * 4 + r) * 64 + lane] = acc[ti * 4 + tj][r];
            }
}

// out[e] = sum over chunks of partial[chunk][e]  (fixed order: deterministic)
__global__ void __launch_bounds__(256) gram_reduce_kernel(const double* __restrict__ partial, int nchunks, long size, double* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= size) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = 0;
    for (; c + 3 < nchunks; c += 4) {
        s0 += partial[(long)c * size + e];
        s1 += partial[(long)(c + 1) * size + e];
        s2 += partial[(long)(c + 2) * size + e];
        s3 += partial[(long)(c + 3) * size + e];
    }
    for (; c < nchunks; ++c) s0 += partial[(long)c * size + e];
    out[e] = (s0 + s1) + (s2 + s3);
}
