// device_exec.h -- the device executor the portable solver sources run on (DevExecT: thread roles, workgroup reductions, the segmented
// product of the rows solver) and the staging of the model descriptors in LDS.  Inline device code only: the solver units
// (plspm_fit.hip, plspm_nonmetric.hip) both include it.
#pragma once

// ------------------------------------------------------------------------------------------------ solver kernel
template <int KCAP>
struct DevExecT {
    static constexpr int kcap = KCAP;      // largest small regression solved in registers (solver_core.h spd_solve)
    int tid, nt;
    double* red;           // LDS scratch, one slot per wave
    long long* marks;      // debug: phase timestamps of problem 0 (PLSPM_DEBUG_MARKS)
#ifdef PLSPM_DEBUG_MARKS
    __device__ __forceinline__ void mark(int id) { if (marks && tid == 0) marks[id] = clock64(); }
#else
    __device__ __forceinline__ void mark(int) {}
#endif
    template <class F> __device__ __forceinline__ void par(int n, F f) { for (int i = tid; i < n; i += nt) f(i); __syncthreads(); }
    template <class F> __device__ __forceinline__ void one(F f) { if (tid == 0) f(); __syncthreads(); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    // out[j] = the `v` of thread first + j, j < 8 (one wave: v_readlane, the results are scalar operands of the consumers)
    template <int N> __device__ __forceinline__ void gather8(double v, int first, double (&out)[8]) {
        const int lo = __double2loint(v), hi = __double2hiint(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = __hiloint2double(__builtin_amdgcn_readlane(hi, first + j), __builtin_amdgcn_readlane(lo, first + j));
    }
    // Segmented products of the rows solver (one wave per problem, solver_core.h CovRows::block_products): thread p holds s[q] = S[q][p];
    //     vrow[m] = sum over the columns q of block m of s[q] w[q],     `ends` bit q = column q closes its block (wave-uniform).
    // w reaches the lanes as the DPP operand of the multiply-add itself: W[a] = w[16 a + lane % 16] (every row of 16 lanes holds a copy of
    // the vector in four register pairs), column q = `row_newbcast:q % 16` of W[q / 16] -- ONE v_fmac_f64_dpp per column, then a scalar bit
    // test whose taken side (add the two chains, store, advance, clear: six instructions) sits out of line behind the block.  A wave issues
    // one instruction per ~4 cycles whatever it is, so the count per column is the cost: 3 here; 7 + a TAKEN branch (an instruction-fetch
    // bubble of ~60 cycles) in the compiled form with v_readlane broadcasts, where hipcc had also expanded the loop-invariant mask into
    // 64 lane masks spilled to VGPR lanes (5.2k cycles per call at P = 60 against 2.6k for compiled DPP groups of four and this form's
    // ~1k).  Columns >= P hold s = 0 and W repeats w[P - 1] (finite).  Sixteen columns per asm statement (operand count); the leading
    // s_nop 4 covers the wait states a DPP read needs behind a VALU write of W (2: a copy the compiler may place in front) or of EXEC (5).
#define PLSPM_SEG_COL(J, ACC, SOP)                                                                                  \
    "v_fmac_f64_dpp " ACC ", %4, " SOP " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"                        \
    "s_bitcmp1_b32 %21, " #J "\n\t"                                                                                 \
    "s_cbranch_scc1 .Lc" #J "_%=\n"                                                                                 \
    ".Lb" #J "_%=:\n\t"
#define PLSPM_SEG_CLOSE(J)                                                                                          \
    ".Lc" #J "_%=:\n\t"                                                                                             \
    "v_add_f64 %3, %0, %1\n\t"                                                                                      \
    "ds_write_b64 %2, %3\n\t"                                                                                       \
    "v_add_u32 %2, 8, %2\n\t"                                                                                       \
    "v_mov_b64 %0, 0\n\t"                                                                                           \
    "v_mov_b64 %1, 0\n\t"                                                                                           \
    "s_branch .Lb" #J "_%=\n"
    template <int PMAX> __device__ __forceinline__ void seg_products(const double (&s)[PMAX], const double* w, int P, unsigned long long ends, double* vrow) {
        static_assert(PMAX % 16 == 0, "sixteen columns per statement");
        double W[PMAX / 16];
#pragma unroll
        for (int a = 0; a < PMAX / 16; ++a) W[a] = w[min(16 * a + (tid & 15), P - 1)];
        unsigned va = (unsigned)(size_t)vrow;                     // LDS byte address (generic -> local is a truncation)
        double r0 = 0.0, r1 = 0.0, t;                             // two chains per block (even / odd column): half the dependent latency
#pragma unroll
        for (int a = 0; a < PMAX / 16; ++a) {
            if (16 * a < P) {                                     // (uniform)
                const unsigned eh = (unsigned)(ends >> (16 * a)) & 0xffffu;
                const double* c = s + 16 * a;
                asm volatile("s_nop 4\n\t"
                             PLSPM_SEG_COL(0, "%0", "%5") PLSPM_SEG_COL(1, "%1", "%6") PLSPM_SEG_COL(2, "%0", "%7") PLSPM_SEG_COL(3, "%1", "%8")
                             PLSPM_SEG_COL(4, "%0", "%9") PLSPM_SEG_COL(5, "%1", "%10") PLSPM_SEG_COL(6, "%0", "%11") PLSPM_SEG_COL(7, "%1", "%12")
                             PLSPM_SEG_COL(8, "%0", "%13") PLSPM_SEG_COL(9, "%1", "%14") PLSPM_SEG_COL(10, "%0", "%15") PLSPM_SEG_COL(11, "%1", "%16")
                             PLSPM_SEG_COL(12, "%0", "%17") PLSPM_SEG_COL(13, "%1", "%18") PLSPM_SEG_COL(14, "%0", "%19") PLSPM_SEG_COL(15, "%1", "%20")
                             "s_branch .Lend_%=\n"
                             PLSPM_SEG_CLOSE(0) PLSPM_SEG_CLOSE(1) PLSPM_SEG_CLOSE(2) PLSPM_SEG_CLOSE(3) PLSPM_SEG_CLOSE(4) PLSPM_SEG_CLOSE(5)
                             PLSPM_SEG_CLOSE(6) PLSPM_SEG_CLOSE(7) PLSPM_SEG_CLOSE(8) PLSPM_SEG_CLOSE(9) PLSPM_SEG_CLOSE(10) PLSPM_SEG_CLOSE(11)
                             PLSPM_SEG_CLOSE(12) PLSPM_SEG_CLOSE(13) PLSPM_SEG_CLOSE(14) PLSPM_SEG_CLOSE(15)
                             ".Lend_%=:"
                             : "+v"(r0), "+v"(r1), "+v"(va), "=&v"(t)
                             : "v"(W[a]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]),
                               "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]), "s"(eh)
                             : "memory", "scc");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the stores above are not on the compiler's counters
    }
    // The same stream with a SECOND store per block: the block sum times `scale` (the thread's weight) at `OFF2` bytes behind the first -- solver_wave16.h
    // reads Q = W' S W from that copy (one load per term) instead of multiplying w_p V[p, m] term by term or building a transposed copy in a pass of its own.
#define PLSPM_SEG_COL2(J, ACC, SOP)                                                                                 \
    "v_fmac_f64_dpp " ACC ", %5, " SOP " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"                        \
    "s_bitcmp1_b32 %22, " #J "\n\t"                                                                                 \
    "s_cbranch_scc1 .Lc" #J "_%=\n"                                                                                 \
    ".Lb" #J "_%=:\n\t"
#define PLSPM_SEG_CLOSE2(J)                                                                                         \
    ".Lc" #J "_%=:\n\t"                                                                                             \
    "v_add_f64 %3, %0, %1\n\t"                                                                                      \
    "ds_write_b64 %2, %3\n\t"                                                                                       \
    "v_mul_f64 %4, %3, %23\n\t"                                                                                     \
    "ds_write_b64 %2, %4 offset:%24\n\t"                                                                            \
    "v_add_u32 %2, 8, %2\n\t"                                                                                       \
    "v_mov_b64 %0, 0\n\t"                                                                                           \
    "v_mov_b64 %1, 0\n\t"                                                                                           \
    "s_branch .Lb" #J "_%=\n"
    template <int PMAX, int OFF2> __device__ __forceinline__ void seg_products2(const double (&s)[PMAX], const double* w, int P, unsigned long long ends, double* vrow, double scale) {
        static_assert(PMAX % 16 == 0 && OFF2 > 0 && OFF2 < 65536 && OFF2 % 8 == 0, "sixteen columns per statement; the second copy within the instruction's offset field");
        double W[PMAX / 16];
#pragma unroll
        for (int a = 0; a < PMAX / 16; ++a) W[a] = w[min(16 * a + (tid & 15), P - 1)];
        unsigned va = (unsigned)(size_t)vrow;
        double r0 = 0.0, r1 = 0.0, t, t2;
#pragma unroll
        for (int a = 0; a < PMAX / 16; ++a) {
            if (16 * a < P) {                                     // (uniform)
                const unsigned eh = (unsigned)(ends >> (16 * a)) & 0xffffu;
                const double* c = s + 16 * a;
                asm volatile("s_nop 4\n\t"
                             PLSPM_SEG_COL2(0, "%0", "%6") PLSPM_SEG_COL2(1, "%1", "%7") PLSPM_SEG_COL2(2, "%0", "%8") PLSPM_SEG_COL2(3, "%1", "%9")
                             PLSPM_SEG_COL2(4, "%0", "%10") PLSPM_SEG_COL2(5, "%1", "%11") PLSPM_SEG_COL2(6, "%0", "%12") PLSPM_SEG_COL2(7, "%1", "%13")
                             PLSPM_SEG_COL2(8, "%0", "%14") PLSPM_SEG_COL2(9, "%1", "%15") PLSPM_SEG_COL2(10, "%0", "%16") PLSPM_SEG_COL2(11, "%1", "%17")
                             PLSPM_SEG_COL2(12, "%0", "%18") PLSPM_SEG_COL2(13, "%1", "%19") PLSPM_SEG_COL2(14, "%0", "%20") PLSPM_SEG_COL2(15, "%1", "%21")
                             "s_branch .Lend_%=\n"
                             PLSPM_SEG_CLOSE2(0) PLSPM_SEG_CLOSE2(1) PLSPM_SEG_CLOSE2(2) PLSPM_SEG_CLOSE2(3) PLSPM_SEG_CLOSE2(4) PLSPM_SEG_CLOSE2(5)
                             PLSPM_SEG_CLOSE2(6) PLSPM_SEG_CLOSE2(7) PLSPM_SEG_CLOSE2(8) PLSPM_SEG_CLOSE2(9) PLSPM_SEG_CLOSE2(10) PLSPM_SEG_CLOSE2(11)
                             PLSPM_SEG_CLOSE2(12) PLSPM_SEG_CLOSE2(13) PLSPM_SEG_CLOSE2(14) PLSPM_SEG_CLOSE2(15)
                             ".Lend_%=:"
                             : "+v"(r0), "+v"(r1), "+v"(va), "=&v"(t), "=&v"(t2)
                             : "v"(W[a]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]),
                               "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]), "s"(eh), "v"(scale), "i"(OFF2)
                             : "memory", "scc");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef PLSPM_SEG_COL2
#undef PLSPM_SEG_CLOSE2
#undef PLSPM_SEG_COL
#undef PLSPM_SEG_CLOSE
    // Block loader of the split rows solver (solver_core.h solve_problem_rows<64, true>): lane of a wave <-> MV pc (consecutive over the wave's
    // lanes: w0 .. w0 + 63, clamped to P - 1), registers <-> the nq <= 64 columns q0 .. of the thread's window; s[q] = S[q0 + q][pc] out of the
    // upper triangle M[r][c >= r] (row pitch PS).  Entries on or above the diagonal (q0 + q <= pc) are row q0 + q read along the lanes; the ones
    // below it are row pc of the matrix: read 16 rows at a time ALONG the lanes (lane <-> column q0 + lane) and turned through a 16 x 66 LDS
    // tile of the wave's own (`xstage`), as solver_wave's loader does for its one diagonal block.  Waves whose block lies wholly on one side of
    // the diagonal skip the other part (wave-uniform tests); every wave passes the same barriers.
    template <int PMAX> __device__ __forceinline__ void load_cov_block(const double* __restrict__ Md, int PS, int P, int pc, int q0, int nq, double (&s)[PMAX]) {
        static_assert(PMAX == 64, "one register per lane of the wave");
        const int lane = tid & 63;
        const int w0 = __builtin_amdgcn_readfirstlane((tid & ~63) & (2 * PMAX - 1));      // first MV of this wave
        double* stage = xstage + (tid >> 6) * (16 * 66);
        const bool direct = q0 <= min(w0 + 63, P - 1), turned = q0 + nq - 1 > w0;        // (uniform)
        if (direct) {
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                const int qc = q0 + min(q, nq - 1);
                const unsigned off = (unsigned)(qc * PS + max(pc, qc)) * 8u;
                s[q] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Md) + off);
            }
        }
        // The turned part (round 5, second form): a tile belongs to ONE wave, so the hand-over between its lanes needs the wave's own LDS order, not a
        // workgroup barrier (the four waves of a problem no longer wait for each other's memory round trips: eight barriers -> one); the rows of
        // round j + 1 are requested as soon as round j's registers have gone to the tile, in front of the reads of round j; the 64 reads of a lane are 32
        // 16-byte ones, taken unconditionally in groups of four and selected against the diagonal afterwards (the compiler had sunk every 8-byte
        // read into an exec-masked block of its own: 266 of them per problem); lanes beyond the window hold its last column already, so the
        // reads need no clamp; the rounds are a real loop (a quarter of the code).
        if (turned) {
            const int cq = q0 + min(lane, nq - 1);                                        // the column this lane reads in the turned part
            auto fetch = [&](double (&t)[16], int j) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rc = min(w0 + 16 * j + i, P - 1);
                    const unsigned off = (unsigned)(rc * PS + max(cq, rc)) * 8u;
                    t[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Md) + off);
                }
            };
            auto put = [&](const double (&t)[16]) {
#pragma unroll
                for (int i = 0; i < 16; ++i) stage[i * 66 + lane] = t[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            };
            const int below = pc - q0;                                                   // registers q > below hold entries under the diagonal
            auto take = [&](int j) {
                if ((lane >> 4) == j) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    const d2* row = reinterpret_cast<const d2*>(stage + (lane & 15) * 66);
#pragma unroll
                    for (int g = 0; g < PMAX; g += 8) {
                        d2 r[4];
#pragma unroll
                        for (int h = 0; h < 4; ++h) r[h] = row[(g >> 1) + h];
                        asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            s[g + 2 * h] = (g + 2 * h > below) ? r[h].x : s[g + 2 * h];
                            s[g + 2 * h + 1] = (g + 2 * h + 1 > below) ? r[h].y : s[g + 2 * h + 1];
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // (the tile is free for the next round's rows)
            };
            double ta[16];
            fetch(ta, 0);
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                put(ta);
                if (j < 3) fetch(ta, j + 1);
                take(j);
            }
        }
        __syncthreads();                                                                  // (callers re-use the tiles' memory)
    }
    double* xstage = nullptr;      // 16 x 66 doubles per wave (split rows solver only)
    // where threads without an item of their own may store (a shared dead array: every lane then runs the same store instruction)
    __device__ __forceinline__ double* sink(double* dead) { return dead; }
    // a value every thread of the group holds identically, made provably uniform (scalar registers, scalar branches)
    __device__ __forceinline__ unsigned long long uniform(unsigned long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    }
    // par over an n0 x n1 grid, first index fastest across threads; (i0, i1) advance incrementally (no integer division per item)
    template <class F> __device__ __forceinline__ void par2(int n0, int n1, F f) {
        int i0 = tid, i1 = 0;
        while (i0 >= n0) { i0 -= n0; ++i1; }
        while (i1 < n1) {
            f(i0, i1);
            i0 += nt;
            while (i0 >= n0) { i0 -= n0; ++i1; }
        }
        __syncthreads();
    }
    // src is a sequence of 64-double chunks (one 512-byte coalesced row each); wave w takes chunks w, w + nw, ... with
    // NB global loads issued before any is consumed.  The chunk index is wave-uniform (scalar decode).
    template <class F> __device__ __forceinline__ void par_chunks64(int nchunks, const double* __restrict__ src, F f) {
        const int lane = tid & 63, nw = nt >> 6;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        constexpr int NB = 20;                  // loads in flight per lane: the sweep is latency/queue bound (10 KB per wave outstanding)
        for (int base = wave; base < nchunks; base += NB * nw) {
            double v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) { const int c = base + j * nw; v[j] = (c < nchunks) ? src[c * 64 + lane] : 0.0; }
#pragma unroll
            for (int j = 0; j < NB; ++j) { const int c = base + j * nw; if (c < nchunks) f(c, lane, v[j]); }
        }
        __syncthreads();
    }
    // group-wide sum: per-thread strided partials -> 64-lane shuffle tree -> (several waves) LDS; fixed order, every thread gets it
    template <class F> __device__ __forceinline__ double sum(int n, F f) {
        double s = 0.0;
        for (int i = tid; i < n; i += nt) s += f(i);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        s = __shfl(s, 0, 64);
        if (nt > 64) {
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            s = 0.0;
            for (int w = 0; w < (nt >> 6); ++w) s += red[w];
        }
        __syncthreads();
        return s;
    }
    template <class F> __device__ __forceinline__ bool any(int n, F f) {
        int hit = 0;
        for (int i = tid; i < n; i += nt) hit |= f(i) ? 1 : 0;
        return __syncthreads_or(hit) != 0;
    }
};
using DevExec = DevExecT<8>;
// (SolverOut: host_internal.h -- the launch seams between the host units carry it)
// Stage the model descriptors in LDS (the solver consults them in every phase) and repoint md at the copies.
__device__ __forceinline__ void stage_descriptors(ModelDesc& md, double* lp) {
    const int P = md.P, L = md.L, ne = md.n_eff, tid = threadIdx.x, nt = blockDim.x;
    double* sh = lp; lp += P;
    int* ip = reinterpret_cast<int*>(lp);
    int* boff = ip; ip += L + 1;
    int* lvof = ip; ip += P;
    int* mode = ip; ip += L;
    int* choff = ip; ip += L;
    int* ef = ip; ip += ne;
    int* et = ip; ip += ne;
    const int nedge = md.n_edges;
    int* poff = ip; ip += L + 1;
    int* soff = ip; ip += L + 1;
    int* pidx = ip; ip += nedge;
    int* sidx = ip; ip += nedge;
    const int ntile = md.T * (md.T + 1) / 2;
    unsigned short* ttu = reinterpret_cast<unsigned short*>(ip); ip += (ntile + 1) / 2;
    unsigned char* Cb = reinterpret_cast<unsigned char*>(ip);
    for (int i = tid; i < ntile; i += nt) {
        int t = 0, rem = i;
        while (rem >= md.T - t) { rem -= md.T - t; ++t; }
        ttu[i] = (unsigned short)(t | ((t + rem) << 8));
    }
    for (int i = tid; i <= L; i += nt) { poff[i] = md.pred_off[i]; soff[i] = md.succ_off[i]; }
    for (int i = tid; i < nedge; i += nt) { pidx[i] = md.pred_idx[i]; sidx[i] = md.succ_idx[i]; }
    for (int i = tid; i < P; i += nt) { sh[i] = md.shift[i]; lvof[i] = md.lvof[i]; }
    for (int i = tid; i <= L; i += nt) boff[i] = md.boff[i];
    for (int i = tid; i < L; i += nt) { mode[i] = md.mode[i]; choff[i] = md.chol_off[i]; }
    for (int i = tid; i < ne; i += nt) { ef[i] = md.eff_from[i]; et[i] = md.eff_to[i]; }
    for (int i = tid; i < L * L; i += nt) Cb[i] = md.C[i];
    md.shift = sh; md.boff = boff; md.lvof = lvof; md.mode = mode; md.chol_off = choff; md.eff_from = ef; md.eff_to = et; md.C = Cb;
    md.pred_off = poff; md.succ_off = soff; md.pred_idx = pidx; md.succ_idx = sidx; md.tile_tu = ttu;
    __syncthreads();
}

#define SCORE_ROWS 16      // rows per tile of the gathering stop-rule pass (kernels_nonmetric.h nm_conv_kernel)
