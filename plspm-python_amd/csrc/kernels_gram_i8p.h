// kernels_gram_i8p.h -- Device kernels, part 2c: the int8 digit-plane Gram with PRIVATE count fragments (round 4).
// Included by plspm_gram_i8.hip behind kernels_gram_i8.h (same product, same operand layouts, same epilogue); not a stand-alone header.
//
// gram_i8_kernel shares BOTH operands of a k-step through LDS: eight waves as 4 (replicate rows) x 2 (pair groups), 20 count blocks + 12
// digit blocks by LDS-DMA, every wave reading 5 + 6 fragments back -- 32 DMA instructions and 88 KB of fragment reads per CU and k-step
// for 240 MFMAs, which the round-3 ablation prices at 12 % + 10 % of the kernel.  The count operand does not need sharing at all: the
// fragment-major 1 KB block of (k-block, count tile) IS the A operand of one wave, lane l's 16 bytes at offset 16 l.  Here a workgroup is
// FOUR waves, one per SIMD, wave w = replicate row w of the tile: MTW count tiles x BOTH pair groups x S planes = MTW x 2S accumulator tiles
// (S = 6, MTW = 5: 60 tiles = 240 AGPRs; S = 7, MTW = 4: 56 = 224), its count fragments loaded straight from global memory into VGPRs
// (`global_load_dwordx4`, one coalesced KB per instruction, two k-steps ahead), and only the 2S digit blocks travel through the LDS ring:
//     per CU and k-step (S = 6):  12 LDS-DMA + 20 register loads (was 32 LDS-DMA),  48 KB of fragment reads (was 88),  12 KB of LDS writes (was 32).
// Schedule of k-step kb (fragment sets X = kb % 2 in use, Y = the other):
//     s_waitcnt vmcnt(ops of k-step kb-1 may fly) lgkmcnt(0);  s_barrier            -> digit blocks of kb+1 landed, X complete but its last fragment
//     MFMA stream mt-major:  acc[mt][j] += X.a[mt] x X.b[j]
//        behind it:  ds_read Y.b[*] of k-step kb+1 (stage (kb+1) % 3)
//                    global_load  Y.a[MTW-1] <- k-block kb+1   (first thing: that register was last read by k-step kb-1)
//                    global_load  X.a[mt]    <- k-block kb+2   right after tile mt's MFMAs have been issued (in-place double buffering: a
//                                 load needs hundreds of clocks to return, the MFMAs that read the register issued before it)
//                    LDS-DMA of this wave's digit blocks of k-step kb+3 -> stage kb % 3 (its fragments are in registers since k-step kb-1)
//        s_waitcnt vmcnt(..) in front of the last tile: X.a[MTW-1] (issued one k-step ago) has landed
// Everything VMEM is counted by hand (loads and LDS-DMA return in order): every wave issues the same sequence every k-step.
// Tile rows of two heights in one launch (MIX) as in gram_i8_kernel: a short row drops the last count tile of every wave.
#pragma once
#include <type_traits>

// Schedule of one k-step of a wave with MW count tiles: the MFMA (index within the k-step) each filler is issued behind.
// SCHED 0: the three kinds at their own strides (some MFMAs carry two fillers).  SCHED 1: ONE filler behind every gap-th MFMA -- a count
// load as soon as its register is free, else the next fragment read, the LDS-DMAs last (their data is three k-steps away) -- and the
// wave's digit blocks are CONSECUTIVE ones, so that M0 is written once per k-step and the blocks differ in the instruction offset only.
template <int S, int MW, int SCHED = 0>
struct GramI8PStep {
    static constexpr int NW = 4, NB = 2 * S, NMFMA = MW * NB;
    static constexpr int PERB = (NB + NW - 1) / NW;            // LDS-DMA instructions per wave and k-step
    struct Tab { int a[8], d[8], r[16]; bool ok; };
    static constexpr Tab build() {
        Tab t{};
        if (SCHED == 0) {
            for (int i = 0; i < MW; ++i) t.a[i] = i == 0 ? 1 : NB * i + 1;
            for (int i = 0; i < PERB; ++i) t.d[i] = (i * NMFMA) / PERB + 4;
            for (int r = 0; r < NB; ++r) t.r[r] = (r * (NMFMA - 16)) / NB + 2;      // the last read >= 16 MFMAs = 256 clocks before the k-step's end: its latency is not waited for
            t.ok = true;
        } else if (SCHED == 2) {
            // the VMEM operations (count loads and LDS-DMAs) evenly over the k-step, a count load wherever its register is free by then, else
            // a DMA; the fragment reads behind every second MFMA in between, from the start (all back long before the barrier)
            const int V = MW + PERB;
            bool used[128] = {};
            int na = 0, nd = 0;
            for (int k = 0; k < V; ++k) {
                int m = 1 + (k * NMFMA) / V;
                const bool a_ok = na < MW && m >= (na == 0 ? 1 : NB * na + 1);
                if (a_ok) t.a[na++] = m;
                else if (nd < PERB) t.d[nd++] = m;
                else { m = NB * na + 1; while (used[m]) ++m; t.a[na++] = m; }
                used[m] = true;
            }
            int nr = 0;
            for (int m = 3; m < NMFMA && nr < NB; m += 2) {
                while (m < NMFMA && used[m]) ++m;
                if (m < NMFMA) { t.r[nr++] = m; used[m] = true; }
            }
            t.ok = na == MW && nd == PERB && nr == NB;
            for (int i = 1; i < MW; ++i) t.ok = t.ok && t.a[i] >= NB * i + 1 && t.a[i] < NMFMA;
        } else {
            const int F = MW + NB + PERB, gap = NMFMA / F > 0 ? NMFMA / F : 1;
            int na = 0, nr = 0, nd = 0;
            for (int m = 1; m < NMFMA; ++m) {
                if ((m - 1) % gap) continue;
                if (na < MW && m >= (na == 0 ? 1 : NB * na + 1)) { t.a[na++] = m; continue; }
                if (nr < NB) { t.r[nr++] = m; continue; }
                if (nd < PERB) { t.d[nd++] = m; continue; }
            }
            t.ok = na == MW && nr == NB && nd == PERB;
        }
        return t;
    }
    static constexpr Tab tab = build();
    static constexpr int aslot(int i) { return tab.a[i]; }      // count load i: 0 = Y.a[MW-1], i >= 1 = X.a[i-1] (behind tile i-1's MFMAs)
    static constexpr int dslot(int i) { return tab.d[i]; }      // LDS-DMA i of the wave
    static constexpr int rslot(int r) { return tab.r[r]; }      // fragment read r
    static constexpr int vm_before(int m) {                     // VMEM operations of a k-step issued before MFMA m
        int n = 0;
        for (int i = 0; i < MW; ++i) n += aslot(i) < m ? 1 : 0;
        for (int i = 0; i < PERB; ++i) n += dslot(i) < m ? 1 : 0;
        return n;
    }
    static_assert(tab.ok, "every filler finds an MFMA of its own k-step to ride behind");
    static_assert(MW >= 2 && MW <= 5, "count fragments are addressed with 13-bit signed offsets around the wave's third tile");
    static_assert(rslot(NB - 1) < NMFMA && dslot(PERB - 1) < NMFMA && aslot(MW - 1) < NMFMA, "every filler rides behind an MFMA of its own k-step");
    static_assert(MW * NB * 4 <= 240, "accumulator tiles must fit the AGPR half");
};

template <int S, int MTW, int VAR = 0>
struct GramI8P {
    static constexpr int NW = 4, NB = 2 * S, RT = NW * MTW;
    static constexpr int PERB = (NB + NW - 1) / NW;
    // BREG (VAR bit 4): the digit blocks reach LDS through registers -- `global_load_dwordx4` into a staging quad, `ds_write_b128` two
    // k-steps later -- instead of by LDS-DMA: an LDS-DMA instruction costs a lone wave ~60 clocks of matrix pipe, a plain load ~22 and the
    // write a few (tools/ubench/mfma_i8_fillers.hip, profiles/r04_i8p_ablate.jsonl).  Two LDS stages then suffice.
    static constexpr bool BREG = ((VAR >> 4) & 1) != 0;
    static constexpr int SCHED = (VAR >> 5) & 3;               // GramI8PStep: 1 = one filler per MFMA gap, consecutive digit blocks per wave, M0 once per k-step
    // PAIRB (VAR bit 7): ONE workgroup barrier per two k-steps on a ring of five stages -- the DMA of k-step kb carries k-step kb + 4; at the
    // barrier in front of an even k-step the digit blocks of the next TWO k-steps have landed (a lone wave per SIMD pays every barrier in full)
    static constexpr bool PAIRB = ((VAR >> 7) & 1) != 0 && !BREG;
    static constexpr int STAGE_BYTES = NB * 1024, NS = BREG ? 2 : PAIRB ? 5 : 3, AHEAD = BREG ? 2 : PAIRB ? 4 : 3;
    static constexpr size_t LDS_BYTES = (size_t)NS * STAGE_BYTES;
    static constexpr int RTS = RT - NW;                        // count tiles of a short tile row
    // ablation probes (experiments build; results are garbage, only the time is read): bit 0 no LDS-DMA in the steady state, 1 no barrier,
    // 2 no fragment reads, 3 no count loads in the steady state
    static constexpr int ABL = VAR & 15;
};

// SHORTS: the launch holds short tile rows as well.  A workgroup runs the k-step loop of ITS height -- two instantiations of the whole body
// (loop + epilogue) behind one workgroup-uniform branch, so that neither loop contains control flow (hipcc answers a branch inside the
// k-step with copies of fragment registers whose loads are still in flight).
template <int S, int MTW, int VAR = 0, bool SHORTS = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gram_i8p_kernel(const uint4* __restrict__ Cd, const uint4* __restrict__ Zs, int KB, int MT, int NT, int ntx, int nty, const int* __restrict__ pair_dst,
                const double* __restrict__ pair_scale, int npair, long nrep, double* __restrict__ gram, long psize, int nty_short) {
    using G = GramI8P<S, MTW, VAR>;
    constexpr int NB = G::NB, RT = G::RT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile enumeration, tall rows first, then short ones (kernels_gram_i8.h gram_i8_kernel)
    const int w = blockIdx.x;
    int slot = w >> 3, rows = nty, rbase = 0;
    bool shortT = false;
    if constexpr (SHORTS) {
        const int ntall = nty - nty_short, tper = (ntx * ntall + 7) >> 3;
        if (slot >= tper) { slot -= tper; rows = nty_short; rbase = ntall; shortT = true; }
        else rows = ntall;
    }
    const int total = ntx * rows, per = (total + 7) >> 3;
    const int gidx = (w & 7) * per + slot;
    if (slot >= per || gidx >= total) return;
    const int srow = 4 * ntx;
    // (integer division runs on the vector unit even for uniform operands: the quotients go back to scalar registers here, so that every
    //  pointer below is scalar arithmetic instead of v_lshl_add_u64 + v_readfirstlane pairs inside the k-step)
    const int sr = __builtin_amdgcn_readfirstlane(gidx / srow), rem = gidx - sr * srow;
    const int nr = min(4, rows - 4 * sr);
    const int tx = __builtin_amdgcn_readfirstlane(rem / nr), tyl = 4 * sr + (rem - tx * nr);
    const int ct0 = shortT ? rbase * RT + tyl * G::RTS : tyl * RT;

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_raw);
    const unsigned voff = (unsigned)lane * 16u;
    const long binc = (long)NT * 1024, ainc = (long)MT * 1024;

    auto run = [&](auto mwc) {
        constexpr int MW = decltype(mwc)::value;               // count tiles of a wave in this tile
        using K = GramI8PStep<S, MW, G::SCHED>;
        const int ctw = ct0 + wave * MW;                       // ... the first of them
        // digit blocks of a k-step, dealt round-robin to the waves (wave, wave + 4, ...); when NB is not a multiple of 4 the last waves repeat
        // block NB - 1 in their last slot (identical bytes to the same place): every wave issues the same VMEM sequence
        const char* bsrc[K::PERB];
        unsigned bdst[K::PERB];
#pragma unroll
        for (int i = 0; i < K::PERB; ++i) {
            // SCHED 1: blocks b0 .. b0 + PERB - 1 with b0 = min(PERB wave, NB - PERB) (the last waves overlap when NB is not a multiple of 4)
            const int b = G::SCHED ? min(wave * K::PERB, NB - K::PERB) + i : min(wave + K::NW * i, NB - 1);
            bsrc[i] = (const char*)(Zs + ((long)tx * NB + b) * 64);
            bdst[i] = lds0 + (unsigned)b * 1024u;
        }
        // count fragments: the wave's tile 2 of the current k-block (fragment t at (t - 2) KB: signed 13-bit instruction offsets)
        const char* abase = (const char*)(Cd + ((long)ctw + 2) * 64);
        auto sgpr64 = [](const void* p) {
            const unsigned long long b = (unsigned long long)p;
            return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
        };
#define GI8P_ALOAD(dst, t) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sgpr64(abase)), "i"(((t) - 2) * 1024) : "memory")
#define GI8P_BLOAD(dst, i) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sgpr64(bsrc[i])) : "memory")
#define GI8P_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "i"(off))
#define GI8P_MFMA(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
        auto dma_one = [&](int i, unsigned stage_off) {
            if constexpr (G::SCHED == 0) glds_block(bsrc[i], voff, bdst[i] + stage_off);
            else {
                // (nothing else in this kernel touches M0: the first DMA of a k-step sets it, the others differ in the instruction offset, which
                //  moves the global and the LDS address alike)
                const unsigned long long ub = sgpr64(bsrc[0]);
                if (i == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(bdst[0] + stage_off) : "memory");
#pragma unroll
                for (int k = 0; k < K::PERB; ++k)
                    if (k == i) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(ub), "i"(k * 1024) : "memory");
            }
        };
        auto dma_advance = [&]() {
#pragma unroll
            for (int i = 0; i < K::PERB; ++i) bsrc[i] += binc;
        };

        i32x4 acc[MW][NB];
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[mt][j] = i32x4{};
        i32x4 fa0[MW], fa1[MW], fb0[NB], fb1[NB], gb0[K::PERB], gb1[K::PERB];
#pragma unroll
        for (int i = 0; i < K::PERB; ++i) { gb0[i] = (i32x4){0, 0, 0, 0}; gb1[i] = gb0[i]; }
#pragma unroll
        for (int i = 0; i < MW; ++i) { fa0[i] = (i32x4){0, 0, 0, 0}; fa1[i] = fa0[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { fb0[i] = (i32x4){0, 0, 0, 0}; fb1[i] = fb0[i]; }

        // prologue: counts of k-blocks 0 and 1 into the two sets, digit blocks of k-blocks 0 .. 2 into the three stages
#pragma unroll
        for (int t = 0; t < MW; ++t) GI8P_ALOAD(fa0[t], t);
        abase += ainc;
#pragma unroll
        for (int t = 0; t < MW; ++t) GI8P_ALOAD(fa1[t], t);
#pragma unroll
        for (int st = 0; st < G::AHEAD; ++st) {
#pragma unroll
            for (int i = 0; i < K::PERB; ++i) dma_one(i, st * G::STAGE_BYTES);
            dma_advance();
        }
        if constexpr (G::BREG) {                               // k-blocks 2 and 3 wait in the staging quads
#pragma unroll
            for (int i = 0; i < K::PERB; ++i) GI8P_BLOAD(gb0[i], i);
            dma_advance();
#pragma unroll
            for (int i = 0; i < K::PERB; ++i) GI8P_BLOAD(gb1[i], i);
            dma_advance();
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned fbase = lds0 + voff;
#pragma unroll
        for (int j = 0; j < NB; ++j) GI8P_DSREAD(fb0[j], fbase, j * 1024);
        // (abase: k-block 1 -- the first count load of k-step 0 re-reads Y.a[MW-1] of k-block 1, the same bytes)

        // one k-step: MFMAs on (xa, xb); behind them the reads of the next k-step's digit fragments into yb, the count loads into ya[MW-1]
        // (next k-step) and xa[0 .. MW-2] (the k-step after), this wave's LDS-DMA of the k-step three ahead into stage W
        auto step = [&](i32x4 (&xa)[MW], i32x4 (&xb)[NB], i32x4 (&ya)[MW], i32x4 (&yb)[NB], i32x4 (&gx)[K::PERB], unsigned Roff, unsigned Woff) {
            const unsigned rb = fbase + Roff;
            auto fill = [&](int m) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
                    if (m == K::rslot(r) && !(G::ABL & 4)) GI8P_DSREAD(yb[r], rb, r * 1024);
                if (m == K::aslot(0)) { if (!(G::ABL & 8)) GI8P_ALOAD(ya[MW - 1], MW - 1); abase += ainc; }
#pragma unroll
                for (int i = 1; i < MW; ++i)
                    if (m == K::aslot(i) && !(G::ABL & 8)) GI8P_ALOAD(xa[i - 1], i - 1);
#pragma unroll
                for (int i = 0; i < K::PERB; ++i)
                    if (m == K::dslot(i) && !(G::ABL & 1)) {
                        if constexpr (G::BREG) {
                            // gx[i] was loaded two k-steps ago at this very place of the sequence: 2 (MW + PERB) - 1 younger operations may fly.
                            // Into the stage of k-step kb + 2; the quad then takes the block of k-step kb + 4
                            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (MW + K::PERB) - 1) : "memory");
                            asm volatile("ds_write_b128 %1, %0" : "+v"(gx[i]) : "v"(voff + bdst[i] + Woff) : "memory");
                            GI8P_BLOAD(gx[i], i);
                        } else dma_one(i, Woff);
                    }
            };
            constexpr int M0 = NB * (MW - 1);
#pragma unroll
            for (int mt = 0; mt < MW - 1; ++mt)
#pragma unroll
                for (int j = 0; j < NB; ++j) { GI8P_MFMA(acc[mt][j], xa[mt], xb[j]); fill(mt * NB + j); }
            // xa[MW-1] was the first VMEM operation of the previous k-step: everything issued behind it may still fly
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MW + K::PERB - 1 + K::vm_before(M0)) : "memory");
            asm volatile("" : "+v"(xa[MW - 1]));
#pragma unroll
            for (int j = 0; j < NB; ++j) { GI8P_MFMA(acc[MW - 1][j], xa[MW - 1], xb[j]); fill(M0 + j); }
            dma_advance();
        };
        // the VMEM operations of the previous k-step may stay in flight (the digit blocks read next were issued the k-step before it); the
        // fragment reads of the set consumed now have returned; then the workgroup barrier: every wave's share of the next k-step has landed
        // and every wave is done reading the stage this k-step's DMA overwrites
        auto wait_barrier = [&](i32x4 (&xa)[MW], i32x4 (&xb)[NB], bool with_barrier) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MW + K::PERB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MW - 1; ++i) asm volatile("" : "+v"(xa[i]));
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(xb[i]));
            if (!(G::ABL & 2) && with_barrier) asm volatile("s_barrier" ::: "memory");
        };
        // k-step kb: stage R = (kb + 1) % NS is read; stage W = kb % NS receives k-step kb + 3 by LDS-DMA (three stages) or k-step kb + 2 out of
        // the staging quads (two stages)
        unsigned W = (G::AHEAD % G::NS) * G::STAGE_BYTES, R = G::STAGE_BYTES;
        auto next = [](unsigned st) { return (st == (G::NS - 1) * G::STAGE_BYTES) ? 0u : st + G::STAGE_BYTES; };
        for (int kb = 0; kb < KB; kb += 2) {          // KB is even: two k-steps per trip, the fragment sets swap roles
            wait_barrier(fa0, fb0, true);
            step(fa0, fb0, fa1, fb1, gb0, R, W);
            W = next(W); R = next(R);
            wait_barrier(fa1, fb1, !G::PAIRB);
            step(fa1, fb1, fa0, fb0, gb1, R, W);
            W = next(W); R = next(R);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // last MFMA results -> readable
#undef GI8P_ALOAD
#undef GI8P_BLOAD
#undef GI8P_DSREAD
#undef GI8P_MFMA
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // no load may land after its register / LDS stage has a new owner

        // Epilogue: lane (l & 15) owns pair 16 pg + (l & 15) of both pair groups in all S planes, replicates 4 (l >> 4) + reg of every count tile
        const long rep0 = (long)ctw * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int j = (tx * 2 + pg) * 16 + (lane & 15);
            if (j >= npair) continue;
            const long dstj = pair_dst[j];
            const double sc = pair_scale[j];
            double* gp = gram + rep0 * psize + dstj;       // walks the replicates of this lane; opaque to the compiler (no precomputed addresses)
#pragma unroll
            for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (rep0 + mt * 16 + reg < nrep) {
                        // sum_s acc_s 256^s from the low planes up: every term is exact in fp64 (kernels_gram_i8.h)
                        double v = (double)acc[mt][pg * S][reg];
#pragma unroll
                        for (int s = 1; s < S; ++s) v = fma((double)acc[mt][pg * S + s][reg], (double)(1ll << (8 * s)), v);
                        *gp = v * sc;
                    }
                    gp += psize;
                    asm volatile("" : "+v"(gp)::"memory");
                }
                gp += 12 * psize;
            }
        }
    };
    if constexpr (SHORTS) {
        if (shortT) run(std::integral_constant<int, MTW - 1>{});
        else run(std::integral_constant<int, MTW>{});
    } else run(std::integral_constant<int, MTW>{});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// gram_i8pp_kernel (round 5): the same product, the same tiles, the same k-step -- as ONE PERSISTENT workgroup per CU.
//
// The tiled launch pays a fixed cost per tile that the MFMA-only probe prices at ~8 us (profiles/r04_pmc.md: 0.2965 ms for 0.253-0.266 ms of
// pipe time, four rounds of tiles per CU): the dispatch of the next workgroup onto the CU, its cold first loads (two count fragments sets
// + three LDS stages before the first MFMA), the epilogue.  Here a workgroup keeps its CU for the whole launch:
//   * tiles come from the SAME per-XCD lists (tall rows first, then short ones; XCD = blockIdx & 7), handed out by one atomic counter per
//     XCD: workgroup j starts with slot j and grabs the next free slot while it works -- exactly the order in which the dispatcher hands
//     the tiled launch's workgroups to the CU that is free first (list scheduling), so a launch that shares the chip with another stream's
//     kernels still balances itself; the last workgroup of an XCD to leave resets the counters (no host bookkeeping between launches);
//   * the NEXT tile's prologue -- count fragments of k-blocks 0 / 1, digit blocks of k-blocks 0 .. 2 -- is issued IN FRONT of this tile's
//     epilogue (the accumulators are the only live state; the fragment registers and the LDS ring are free once the k-loop's last wait and
//     a workgroup barrier have passed), so its memory latency runs under the 40 stores + ~3,000 fp64 operations per lane of the epilogue;
//   * tall and short tiles are two PHASES of a workgroup (two instantiations of loop + epilogue, as in the tiled kernel): a workgroup walks
//     tall tiles until the counter hands it a short one, then short tiles; only the tile at the phase change starts cold.
// Exact integer sums, same epilogue arithmetic: matrices and records are bit-identical to the tiled launch ("i8_persist" 0).
struct GramI8PPCtl { unsigned next[8]; unsigned done[8]; };      // per XCD: slots handed out beyond the first of every workgroup / workgroups that left

template <int S, int MTW, int VAR = 0, bool SHORTS = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gram_i8pp_kernel(const uint4* __restrict__ Cd, const uint4* __restrict__ Zs, int KB, int MT, int NT, int ntx, int nty, const int* __restrict__ pair_dst,
                 const double* __restrict__ pair_scale, int npair, long nrep, double* __restrict__ gram, long psize, int nty_short, GramI8PPCtl* __restrict__ ctl) {
    using G = GramI8P<S, MTW, VAR>;
    static_assert(!G::BREG && !G::PAIRB && G::SCHED != 0 && G::ABL == 0, "the persistent form exists for the release schedule");
    constexpr int NB = G::NB, RT = G::RT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, wpx = gridDim.x >> 3;          // workgroups per XCD
    const int ntall = SHORTS ? nty - nty_short : nty;
    const int tper = (ntx * ntall + 7) >> 3, sper = SHORTS ? (ntx * nty_short + 7) >> 3 : 0, per_all = tper + sper;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem_raw);
    const unsigned lds_word = lds0 + (unsigned)G::LDS_BYTES;       // the next slot of this workgroup, written by wave 0
    const unsigned voff = (unsigned)lane * 16u;
    const long binc = (long)NT * 1024, ainc = (long)MT * 1024;
    auto sgpr64 = [](const void* p) {
        const unsigned long long b = (unsigned long long)p;
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    };
    // slot -> tile of its list: pair tile tx, first count tile ct0 (the enumeration of gram_i8p_kernel); false: the slot holds no tile (padding of
    // the eight ranges)
    auto decode = [&](int slot, bool shortT, int& tx, int& ct0) -> bool {
        const int rows = shortT ? nty_short : ntall, s = shortT ? slot - tper : slot;
        const int total = ntx * rows, per = (total + 7) >> 3;
        const int gidx = xcd * per + s;
        if (s >= per || gidx >= total) return false;
        const int srow = 4 * ntx;
        const int sr = __builtin_amdgcn_readfirstlane(gidx / srow), rem = gidx - sr * srow;
        const int nr = min(4, rows - 4 * sr);
        tx = __builtin_amdgcn_readfirstlane(rem / nr);
        const int tyl = 4 * sr + (rem - tx * nr);
        ct0 = shortT ? ntall * RT + tyl * G::RTS : tyl * RT;
        return true;
    };

    // One phase: tiles of one height, from `slot` on, while the counter hands out slots below `slot_end`.  Returns the first slot it was handed
    // that is not its kind (>= slot_end).
    auto phase = [&](auto mwc, int slot, const bool shortT, const int slot_end) -> int {
        constexpr int MW = decltype(mwc)::value;
        using K = GramI8PStep<S, MW, G::SCHED>;
        int tx = 0, ct0 = 0;
        // (slots of the padding hold no tile: skip them without touching the counter's order -- at most seven per list)
        const char* abase;
        const char* bsrc;
#define GI8P_ALOAD(dst, t) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sgpr64(abase)), "i"(((t) - 2) * 1024) : "memory")
#define GI8P_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "i"(off))
#define GI8P_MFMA(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define GI8P_MFMA0(c, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b))
        const int b0 = min(wave * K::PERB, NB - K::PERB);      // the wave's consecutive digit blocks
        const unsigned bdst0 = lds0 + (unsigned)b0 * 1024u;
        auto dma_one = [&](int i, unsigned stage_off) {
            const unsigned long long ub = sgpr64(bsrc);
            if (i == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(bdst0 + stage_off) : "memory");
#pragma unroll
            for (int k = 0; k < K::PERB; ++k)
                if (k == i) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(ub), "i"(k * 1024) : "memory");
        };
        i32x4 acc[MW][NB];
        i32x4 fa0[MW], fa1[MW], fb0[NB], fb1[NB];
#pragma unroll
        for (int i = 0; i < MW; ++i) { fa0[i] = (i32x4){0, 0, 0, 0}; fa1[i] = fa0[i]; }
#pragma unroll
        for (int i = 0; i < NB; ++i) { fb0[i] = (i32x4){0, 0, 0, 0}; fb1[i] = fb0[i]; }
        // prologue of tile (ptx, pct0): counts of k-blocks 0 and 1 into the two sets, digit blocks of k-blocks 0 .. 2 into the three stages;
        // leaves abase at k-block 1 and bsrc at k-block 3 (what the k-loop expects)
        auto issue_prologue = [&](int ptx, int pct0) {
            abase = (const char*)(Cd + ((long)(pct0 + wave * MW) + 2) * 64);
            bsrc = (const char*)(Zs + ((long)ptx * NB + b0) * 64);
#pragma unroll
            for (int t = 0; t < MW; ++t) GI8P_ALOAD(fa0[t], t);
            abase += ainc;
#pragma unroll
            for (int t = 0; t < MW; ++t) GI8P_ALOAD(fa1[t], t);
#pragma unroll
            for (int st = 0; st < G::AHEAD; ++st) {
#pragma unroll
                for (int i = 0; i < K::PERB; ++i) dma_one(i, st * G::STAGE_BYTES);
                bsrc += binc;
            }
        };
        const unsigned fbase = lds0 + voff;
        // FIRST: the k-step that opens a tile -- its MFMAs take the constant 0 as the accumulator input ("=a": the accumulators are DEFINED here, so
        // nothing of the previous tile's sums is carried round the tile loop and no zeroing pass is needed)
        auto step = [&](auto first, i32x4 (&xa)[MW], i32x4 (&xb)[NB], i32x4 (&ya)[MW], i32x4 (&yb)[NB], unsigned Roff, unsigned Woff) {
            constexpr bool FIRST = decltype(first)::value;
            const unsigned rb = fbase + Roff;
            auto fill = [&](int m) {
#pragma unroll
                for (int r = 0; r < NB; ++r)
                    if (m == K::rslot(r)) GI8P_DSREAD(yb[r], rb, r * 1024);
                if (m == K::aslot(0)) { GI8P_ALOAD(ya[MW - 1], MW - 1); abase += ainc; }
#pragma unroll
                for (int i = 1; i < MW; ++i)
                    if (m == K::aslot(i)) GI8P_ALOAD(xa[i - 1], i - 1);
#pragma unroll
                for (int i = 0; i < K::PERB; ++i)
                    if (m == K::dslot(i)) dma_one(i, Woff);
            };
            constexpr int M0 = NB * (MW - 1);
#pragma unroll
            for (int mt = 0; mt < MW - 1; ++mt)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if constexpr (FIRST) GI8P_MFMA0(acc[mt][j], xa[mt], xb[j]); else GI8P_MFMA(acc[mt][j], xa[mt], xb[j]);
                    fill(mt * NB + j);
                }
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MW + K::PERB - 1 + K::vm_before(M0)) : "memory");
            asm volatile("" : "+v"(xa[MW - 1]));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (FIRST) GI8P_MFMA0(acc[MW - 1][j], xa[MW - 1], xb[j]); else GI8P_MFMA(acc[MW - 1][j], xa[MW - 1], xb[j]);
                fill(M0 + j);
            }
            bsrc += binc;
        };
        auto wait_barrier = [&](i32x4 (&xa)[MW], i32x4 (&xb)[NB]) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MW + K::PERB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MW - 1; ++i) asm volatile("" : "+v"(xa[i]));
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(xb[i]));
            asm volatile("s_barrier" ::: "memory");
        };
        auto next_stage = [](unsigned st) { return (st == (G::NS - 1) * G::STAGE_BYTES) ? 0u : st + G::STAGE_BYTES; };
        // the epilogue of tile (etx, ect0): gram_i8p_kernel's, and the accumulators zeroed for the next tile on the way
        // (the tile's output slots and scales arrive as arguments: they are loaded with the tile's prologue -- a load of the compiler's own in here
        //  would make it wait for vmcnt(0), i.e. for the NEXT tile's prologue, at the head of the epilogue)
        auto epilogue = [&](int etx, int ect0, const long (&dst)[2], const double (&scl)[2]) {
            const long rep0 = (long)(ect0 + wave * MW) * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const int j = (etx * 2 + pg) * 16 + (lane & 15);
                if (j >= npair) continue;
                const long dstj = dst[pg];
                const double sc = scl[pg];
                double* gp = gram + rep0 * psize + dstj;
#pragma unroll
                for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (rep0 + mt * 16 + reg < nrep) {
                            double v = (double)acc[mt][pg * S][reg];
#pragma unroll
                            for (int s = 1; s < S; ++s) v = fma((double)acc[mt][pg * S + s][reg], (double)(1ll << (8 * s)), v);
                            *gp = v * sc;
                        }
                        gp += psize;
                        asm volatile("" : "+v"(gp)::"memory");
                    }
                    gp += 12 * psize;
                }
            }
        };

        // the first tile of the phase starts cold
        while (slot < slot_end && !decode(slot, shortT, tx, ct0)) slot = slot_end;      // (a padding slot can only be the list's tail: this workgroup has no tile of this kind)
        if (slot >= slot_end) return slot;
        issue_prologue(tx, ct0);
        for (;;) {
            // this tile's output slots / scales (clamped index instead of a branch: no control flow while the prologue's loads are in flight)
            long dst[2];
            double scl[2];
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const int jj = min((tx * 2 + pg) * 16 + (lane & 15), npair - 1);
                dst[pg] = pair_dst[jj];
                scl[pg] = pair_scale[jj];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(dst[0]), "+v"(dst[1]), "+v"(scl[0]), "+v"(scl[1]));      // (consumed here: the compiler's wait for them sits beside ours)
            // the slot after this one: one atomic per workgroup and tile, from asm (the compiler must not wait for it: its round trip -- a microsecond
            // or two at device scope -- runs under the first k-steps; wave 0's counted waits there only become conservative by one older operation).
            // The result is complete at the second k-step's wait and is published to the other waves behind the peeled pair.
            unsigned grabbed = 0u;
            if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(grabbed) : "v"(0u), "v"(1u), "s"(&ctl->next[xcd]) : "memory");
            asm volatile("s_barrier" ::: "memory");
            // (the digit fragment sets are DEFINED here ("=v"): their registers carry nothing from the previous tile, so they are free during its
            //  epilogue -- with "+v" the compiler keeps 96 dead registers alive across it and spills accumulators instead)
#pragma unroll
            for (int j = 0; j < NB; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb0[j]) : "v"(fbase), "i"(j * 1024));
#pragma unroll
            for (int j = 0; j < NB; ++j) asm volatile("" : "=v"(fb1[j]));
            unsigned W = (G::AHEAD % G::NS) * G::STAGE_BYTES, R = G::STAGE_BYTES;
            // (KB is even and >= 2: the first pair of k-steps is peeled, its first step defines the accumulators)
            wait_barrier(fa0, fb0);
            step(std::true_type{}, fa0, fb0, fa1, fb1, R, W);
            W = next_stage(W); R = next_stage(R);
            wait_barrier(fa1, fb1);
            step(std::false_type{}, fa1, fb1, fa0, fb0, R, W);
            W = next_stage(W); R = next_stage(R);
            if (tid == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(lds_word), "v"(grabbed + (unsigned)wpx) : "memory");
            for (int kb = 2; kb < KB; kb += 2) {
                wait_barrier(fa0, fb0);
                step(std::false_type{}, fa0, fb0, fa1, fb1, R, W);
                W = next_stage(W); R = next_stage(R);
                wait_barrier(fa1, fb1);
                step(std::false_type{}, fa1, fb1, fa0, fb0, R, W);
                W = next_stage(W); R = next_stage(R);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // (the accumulators are named here so that no read of them can be scheduled above the wait states of the last MFMAs)
#pragma unroll
            for (int mt = 0; mt < MW; ++mt)
#pragma unroll
                for (int j = 0; j < NB; ++j) asm volatile("" : "+a"(acc[mt][j]));
            asm volatile("s_barrier" ::: "memory");            // every wave is done with the LDS ring; wave 0 has published the next slot
            unsigned nxt_v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nxt_v) : "v"(lds_word) : "memory");
            const int nxt = __builtin_amdgcn_readfirstlane((int)nxt_v);
            int ntx2 = 0, nct0 = 0;
            const bool more = nxt < slot_end && decode(nxt, shortT, ntx2, nct0);
            if (more) {
                issue_prologue(ntx2, nct0);                    // under the epilogue: loads into the fragment sets, LDS-DMA into the ring
                epilogue(tx, ct0, dst, scl);
                tx = ntx2; ct0 = nct0;
            } else {
                epilogue(tx, ct0, dst, scl);
                return nxt;
            }
        }
#undef GI8P_ALOAD
#undef GI8P_DSREAD
#undef GI8P_MFMA
#undef GI8P_MFMA0
    };

    int slot = blockIdx.x >> 3;
    if (slot < per_all) {
        if (slot < tper) slot = phase(std::integral_constant<int, MTW>{}, slot, false, tper);
        if constexpr (SHORTS) {
            // (a slot of the tall list's padding sends the workgroup on to the short list: ask the counter for a fresh slot there)
            if (slot < per_all) slot = phase(std::integral_constant<int, MTW - 1>{}, slot, true, per_all);
        }
    }
    // the last workgroup of the XCD to leave resets its counters: the next launch on this handle starts from zero
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(&ctl->done[xcd], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)wpx - 1u) {
            __hip_atomic_store(&ctl->next[xcd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->done[xcd], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
