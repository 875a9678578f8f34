// kernels_nonmetric.h -- Device kernels, part 4: non-metric solvers (NUM/RAW, categorical, missing data) and the stop-rule passes (gathering and dense).
// Device code shared by the translation units of libplspm_hip.so (host_internal.h lists them); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------ non-metric (NUM / RAW) kernels
// The correlation matrix R and the iteration state of every problem live in global memory between launches (gS / gstate); the
// small workspace and the descriptors are LDS-resident.  MODE 0 prepare, 1 step, 2 finish (solver_core.h nm_*).
template <int MODE>
__global__ void __launch_bounds__(256) nm_kernel(ModelDesc md, const double* __restrict__ Mp, long mp_stride, SolverOut so, double* gS, double* gstate,
                                                 const double* __restrict__ partial, int nparts, int* __restrict__ nactive, int fuse_finish, const int* __restrict__ live) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    // `live` (may be null): the problems still iterating after the previous step (active_list_kernel's list) -- the launch then has one workgroup per LIVE
    // problem instead of one per problem of the batch that finds itself stopped (the late iterations of a batch are a handful of problems among thousands)
    const long b = live ? live[blockIdx.x] : (long)blockIdx.x;
    Workspace ws;
    ws.PS = cov_ld(md.P);
    ws.S = gS + b * cov_doubles(md.P);
    const long small_doubles = workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol);
    lp += small_doubles;
    NmState st;
    nm_carve(st, gstate + b * nm_state_doubles(md.P, md.L, md.n_chol), md.P, md.L);
    if (MODE == 1 && st.scal[3] == 0.0) return;                 // finished problems cost nothing more
    stage_descriptors(md, lp);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, ws.red, nullptr};
    // Launch fusion: MODE 0 = prepare + the first step (nothing to decide before it); MODE 1 = decide + step, and -- with
    // fuse_finish -- the finish of a problem right where its stop is decided (every problem stops inside a MODE 1 launch, so the
    // separate finish launch and its reload of the correlation matrix disappear); MODE 2 = finish alone.
    bool finish_now = (MODE == 2);
    if (MODE == 0) {
        nm_prepare(ex, md, ws, st, Mp + b * mp_stride);
        const bool active = nm_step(ex, md, ws, st, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
    } else if (MODE == 1) {
        const bool active = nm_step(ex, md, ws, st, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
        finish_now = !active && fuse_finish;
    }
    if (finish_now) {
        FitOutputs out = so.fit;
        if (b != 0) out = FitOutputs{};
        out.row = so.row ? so.row + b * so.row_stride : nullptr;
        out.status = so.status ? so.status + b : nullptr;
        out.iters = so.iters ? so.iters + b : nullptr;
        nm_finish(ex, md, ws, st, out);
    }
}


// Categorical (Scale.ORD / NOM) non-metric problems (solver_nmg.h).  The aug-level moment matrix, the collapsed MV-level
// correlation matrix and the iteration state live in global memory; both small workspaces and the aug-level descriptors in LDS.
template <int MODE>
__global__ void __launch_bounds__(256) nmg_kernel(ModelDesc md, CatDesc cd, ModelDesc mdm, const double* __restrict__ Mp, long mp_stride, SolverOut so,
                                                  double* gS, double* gSm, double* gstate, long state_stride,
                                                  const double* __restrict__ partial, int nparts, int* __restrict__ nactive, int fuse_finish, int extra_in_lds,
                                                  unsigned short* gK16, int ld16, const int* __restrict__ live) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    const long b = live ? live[blockIdx.x] : (long)blockIdx.x;      // (nm_kernel: the live list of the previous step)
    const int Q = md.P, L = md.L;
    Workspace ws;
    ws.PS = cov_ld(Q);
    ws.S = gS + b * cov_doubles(Q);
    // (the aug-level solver indexes its workspace vectors by MV and by LV only -- wn, a, G, E, the regression scratch: Pm-sized suffices)
    carve_small(ws, lp, cd.Pm, L, md.kmax, 0);
    lp += workspace_small_doubles(cd.Pm, L, md.kmax, 0);
    Workspace wsm;
    wsm.PS = cov_ld(cd.Pm);
    wsm.S = gSm + b * cov_doubles(cd.Pm);
    carve_small(wsm, lp, cd.Pm, L, md.kmax, 0);
    lp += workspace_small_doubles(cd.Pm, L, md.kmax, 0);
    double* state = gstate + b * state_stride;
    NmState st;
    nm_carve(st, state, Q, L);
    if (MODE == 1 && st.scal[3] == 0.0) return;
    // The small arrays of the iteration (quantifications, block moment matrices, pooling scratch, per-MV scratch) are touched by hundreds of
    // short dependent phases: in global memory every one of them is a memory round trip (the per-LV loop and the quantification were 55 % of
    // a step).  When they fit they live in LDS for the launch; the two (Q+1) x L products stay global (streamed), the persistent head
    // (tq | tc | akk) is copied in and out.
    NmgExtra x, xg;
    if (extra_in_lds) {
        nmg_carve_fast(x, xg, state + nm_state_doubles(Q, L, 0), lp, Q, cd.Pm, L, cd.cmax, cd.kmv);
        lp += (nmg_fast_doubles(Q, cd.Pm, L, cd.cmax, cd.kmv) + 1) & ~1L;
        if (MODE != 0 && MODE != 3 && MODE != 4) {
            const long np = nmg_persistent_doubles(Q, cd.Pm, L);
            for (long i = threadIdx.x; i < np; i += blockDim.x) x.tq[i] = xg.tq[i];
            __syncthreads();
        }
    } else nmg_carve(x, state + nm_state_doubles(Q, L, 0), Q, cd.Pm, L, cd.cmax, cd.kmv);
    if (gK16) { x.k16 = gK16 + b * (long)(Q + 1) * ld16; x.ld16 = ld16; }       // all-indicator model: uint16 copy of the count matrix (solver_nmg.h)
    stage_descriptors(md, lp);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, ws.red, (b == 0) ? so.marks : nullptr};
    bool finish_now = (MODE == 2);
    if constexpr (MODE == 4) {
        // round 5: the int8 product wrote the UPPER triangle of the uint16 count matrix itself (gram_i8_kernel<.., IND>, nty_short < 0): mirror it -- tiles of
        // 64 x 64 counts through LDS, 16-byte loads and stores on both sides -- then the initial state; no packed fp64 matrix exists (Mp is null)
        __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
        const int C = Q + 1, nt = (C + 63) >> 6, tid = threadIdx.x;
        unsigned short* K = x.k16;
        for (int ti = 0; ti < nt; ++ti)
            for (int tj = ti; tj < nt; ++tj) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {                     // tile (ti, tj): 64 rows x 8 groups of 8 counts
                    const int e = tid + 256 * h, r = e >> 3, cg = e & 7;
                    const int row = ti * 64 + r, col = tj * 64 + cg * 8;
                    uint4 v = make_uint4(0u, 0u, 0u, 0u);
                    if (row < C && col < ld16) v = *reinterpret_cast<const uint4*>(K + (long)row * ld16 + col);
                    *reinterpret_cast<uint4*>(&tile[r][cg * 8]) = v;
                }
                __syncthreads();
#pragma unroll
                for (int h = 0; h < 2; ++h) {                     // its transpose into tile (tj, ti); on the diagonal: below the diagonal only
                    const int e = tid + 256 * h, r = e >> 3, cg = e & 7;
                    const int row = tj * 64 + r, col = ti * 64 + cg * 8;
                    if (row < C && col < ld16) {
                        unsigned short w[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int c = cg * 8 + u; w[u] = (ti != tj || c < r) ? tile[c][r] : tile[r][c]; }
                        uint4 v;
                        v.x = (unsigned)w[0] | ((unsigned)w[1] << 16); v.y = (unsigned)w[2] | ((unsigned)w[3] << 16);
                        v.z = (unsigned)w[4] | ((unsigned)w[5] << 16); v.w = (unsigned)w[6] | ((unsigned)w[7] << 16);
                        *reinterpret_cast<uint4*>(K + (long)row * ld16 + col) = v;
                    }
                }
                __syncthreads();
            }
        ws.S = nullptr;
        nmg_prepare(ex, md, cd, ws, st, x, (const double*)nullptr);
    } else if (MODE == 3) {
        // prepare alone, and only what the wave step reads (kernels_nmw.h): the uint16 counts + the initial state -- no fp64 copy of the matrix
        // (730 KB written per problem at 300 indicator columns, which the wave step and its fused finish never open)
        ws.S = nullptr;
        nmg_prepare(ex, md, cd, ws, st, x, Mp + b * mp_stride);
    } else if (MODE == 0) {
        nmg_prepare(ex, md, cd, ws, st, x, Mp + b * mp_stride);
        const bool active = nmg_step(ex, md, cd, ws, st, x, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
    } else if (MODE == 1) {
        const bool active = nmg_step(ex, md, cd, ws, st, x, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
        finish_now = !active && fuse_finish;
    }
    if (finish_now) {
        FitOutputs out = so.fit;
        if (b != 0) out = FitOutputs{};
        out.row = so.row ? so.row + b * so.row_stride : nullptr;
        out.status = so.status ? so.status + b : nullptr;
        out.iters = so.iters ? so.iters + b : nullptr;
        nmg_finish(ex, md, cd, mdm, ws, wsm, st, x, out);
    }
    if (extra_in_lds) {
        __syncthreads();
        const long np = nmg_persistent_doubles(Q, cd.Pm, L);
        for (long i = threadIdx.x; i < np; i += blockDim.x) xg.tq[i] = x.tq[i];
    }
}


// Non-metric data with missing values (solver_nmx.h).  MODE 0 also looks up the bootstrap weight of every incomplete row in the
// replicate's ordered (row, count) list (1 for a plain fit).
template <int MODE>
__global__ void __launch_bounds__(256) nmx_kernel(ModelDesc md, MissDesc xd, const int* __restrict__ rowid, const double* __restrict__ Mp, long mp_stride, SolverOut so,
                                                  double* gS, double* gstate, long state_stride, const double* __restrict__ partial, int nparts,
                                                  int* __restrict__ nactive, const int2* __restrict__ ent, const int* __restrict__ nent, long ent_stride, int fuse_finish,
                                                  const int* __restrict__ live) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    const long b = live ? live[blockIdx.x] : (long)blockIdx.x;      // (nm_kernel: the live list of the previous step)
    Workspace ws;
    ws.PS = cov_ld(md.P);
    ws.S = gS + b * cov_doubles(md.P);
    carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol);
    lp += workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    double* state = gstate + b * state_stride;
    NmState st;
    nm_carve(st, state, md.P, md.L);
    NmxExtra x;
    nmx_carve(x, state + nm_state_doubles(md.P, md.L, md.n_chol), md.P, md.L, xd.K);
    if (MODE == 1 && st.scal[3] == 0.0) return;
    stage_descriptors(md, lp);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, ws.red, nullptr};
    bool finish_now = (MODE == 2);
    if (MODE == 0) {
        const int2* e = ent ? ent + b * ent_stride : nullptr;
        const int ne = ent ? nent[b] : 0;
        for (int j = threadIdx.x; j < xd.K; j += blockDim.x) {
            double c = 1.0;
            if (e) {
                const int row = rowid[j];
                int lo = 0, hi = ne;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (e[mid].x < row) lo = mid + 1; else hi = mid; }
                c = (lo < ne && e[lo].x == row) ? (double)e[lo].y : 0.0;
            }
            x.ck[j] = c;
        }
        __syncthreads();
        nmx_prepare(ex, md, xd, ws, st, x, Mp + b * mp_stride);
        const bool active = nmx_step(ex, md, xd, ws, st, x, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
    } else if (MODE == 1) {
        const bool active = nmx_step(ex, md, xd, ws, st, x, partial + b * nparts, nparts);
        if (active && threadIdx.x == 0) atomicAdd(nactive, 1);
        finish_now = !active && fuse_finish;
    }
    if (finish_now) {
        FitOutputs out = so.fit;
        if (b != 0) out = FitOutputs{};
        out.row = so.row ? so.row + b * so.row_stride : nullptr;
        out.status = so.status ? so.status + b : nullptr;
        out.iters = so.iters ? so.iters + b : nullptr;
        nmx_finish(ex, md, xd, ws, st, x, out);
    }
}


// Streaming convergence pass (reference weights.py:120): for every still-active problem, sum over its observations (all rows,
// or the (row,count) list of a bootstrap replicate) of count * sum_l (|y_old| - |y_new|)^2, with y = xa . c + k for the two
// score maps in the state.  16-row tiles of Xa are staged in LDS like scores_kernel; blockIdx.x = part, blockIdx.y = problem.
__global__ void __launch_bounds__(256) nm_conv_kernel(const double* __restrict__ Xa, long N, int PA, int P, int L, int n_chol, const int* __restrict__ boff,
                                                       const int2* __restrict__ ent, const int* __restrict__ nent, long ent_stride,
                                                       const double* __restrict__ gstate, long state_stride, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);     // [16][PA+1]
    double* co = tile + SCORE_ROWS * (PA + 1);              // [P] c_old
    double* cn = co + P;                                    // [P] c_new
    double* ko = cn + P;                                    // [L]
    double* kn = ko + L;                                    // [L]
    double* cnt = kn + L;                                   // [16]
    double* red = cnt + SCORE_ROWS;                         // [256]
    int* bsh = reinterpret_cast<int*>(red + 256);           // [L+1]
    const long b = blockIdx.y;
    const int part = blockIdx.x, nparts = gridDim.x, tid = threadIdx.x;
    const double* st = gstate + b * state_stride;       // NmState-compatible head: scal[8] a_old a_new c_old c_new k_old k_new
    if (st[3] == 0.0) return;
    for (int p = tid; p < P; p += 256) { co[p] = st[8 + 2 * P + p]; cn[p] = st[8 + 3 * P + p]; }
    for (int l = tid; l < L; l += 256) { ko[l] = st[8 + 4 * P + l]; kn[l] = st[8 + 4 * P + L + l]; }
    for (int l = tid; l <= L; l += 256) bsh[l] = boff[l];
    const int2* e = ent ? ent + b * ent_stride : nullptr;
    const long nrows = ent ? (long)nent[b] : N;
    const long ntiles = (nrows + SCORE_ROWS - 1) / SCORE_ROWS;
    const int half = PA >> 1;
    const int r_c = tid & 15, lg = tid >> 4;
    double acc = 0.0;
    for (long tl = part; tl < ntiles; tl += nparts) {
        const long i0 = tl * SCORE_ROWS;
        const int rows = (int)lmin(SCORE_ROWS, nrows - i0);
        __syncthreads();
        if (tid < SCORE_ROWS) cnt[tid] = (tid < rows) ? (e ? (double)e[i0 + tid].y : 1.0) : 0.0;
        for (int el = tid; el < rows * half; el += 256) {
            const int r = el / half, c = 2 * (el - r * half);
            const long src_row = e ? (long)e[i0 + r].x : i0 + r;
            const double2 v = reinterpret_cast<const double2*>(Xa + src_row * PA)[c >> 1];
            tile[r * (PA + 1) + c] = v.x;
            tile[r * (PA + 1) + c + 1] = v.y;
        }
        __syncthreads();
        if (r_c < rows) {
            const double* row = tile + r_c * (PA + 1);
            double s = 0.0;
            for (int l = lg; l < L; l += 16) {
                double yo = ko[l], yn = kn[l];
                for (int p = bsh[l]; p < bsh[l + 1]; ++p) { const double x = row[p]; yo += x * co[p]; yn += x * cn[p]; }
                const double d = fabs(yo) - fabs(yn);
                s += d * d;
            }
            acc += cnt[r_c] * s;
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) red[tid] += red[tid + h]; __syncthreads(); }
    if (tid == 0) partial[b * nparts + part] = red[0];
}


// ------------------------------------------------------------------------------------------------ dense stop-rule pass (bootstrap)
// nm_conv_kernel gathers every replicate's surviving rows (3.2 MB of L2 reads per replicate and iteration at 10k x 60).  With
// thousands of replicates in flight it is cheaper to turn the loop inside out: a wave keeps a 16-ROW TILE of the data stationary
// -- in scalar registers, the tile is stored column-major (Xt[tile][p][16]) so one s_load fetches a column of it -- and walks
// over the replicates 64 at a time, one replicate per lane, their score-map coefficients staged through LDS from a table laid
// out [group][coefficient][lane].  Per (row, replicate) it forms the L old / new scores with block-sparse FMAs (scalar x,
// vector coefficient), accumulates (|y_old| - |y_new|)^2 and weights it with the row's count in that replicate (dense uint16
// histogram written by resample_kernel).  One partial per (replicate, tile); nm_step adds them in a fixed order.
__global__ void __launch_bounds__(256) tile_transpose_kernel(const double* __restrict__ Xa, long N, int PA, double* __restrict__ Xt) {
    const long tile = blockIdx.x;
    for (int e = threadIdx.x; e < 16 * PA; e += 256) {
        const int p = e >> 4, r = e & 15;
        const long i = tile * 16 + r;
        Xt[tile * 16 * PA + e] = (i < N) ? Xa[i * PA + p] : 0.0;
    }
}

// The problems still iterating, in problem order: list[0 .. count) = their ids, list[-1] ... the count itself goes to *count.  One
// workgroup; ballot prefix per wave, running offset across the trips (a few thousand problems: a few microseconds).  The stop-rule
// pass then walks ceil(count / 64) groups of LIVE problems -- in the late iterations of a batch most problems have stopped, and a
// group of 64 consecutive ids almost never stops as a whole.
// `host_count` (pinned host memory, may be null): the count once more, for the host's "anything left?" test -- written by the kernel, no copy
// operation (and no counter to clear) per iteration.
// `list2` / `count2` / `host_count2` (round 6, may be null): among them, the problems whose lower bound from the first row chunks did not decide and that ask for
// the pass over all rows (state word 5: kernels_nmw.h nmw_step_kernel) -- the same sweep files them a second time.
__global__ void __launch_bounds__(1024) active_list_kernel(const double* __restrict__ gstate, long state_stride, long nproblems, int* __restrict__ list, int* __restrict__ count,
                                                            int* __restrict__ host_count, int* __restrict__ list2 = nullptr, int* __restrict__ count2 = nullptr,
                                                            int* __restrict__ host_count2 = nullptr) {
    __shared__ int wcount[16], wcount2[16];
    __shared__ int base, base2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { base = 0; base2 = 0; }
    __syncthreads();
    for (long b0 = 0; b0 < nproblems; b0 += 1024) {
        const long b = b0 + tid;
        const bool on = b < nproblems && gstate[b * state_stride + 3] != 0.0;
        const bool on2 = list2 && on && gstate[b * state_stride + 5] == 1.0;
        const unsigned long long bal = __ballot(on), bal2 = __ballot(on2);
        if (lane == 0) { wcount[wave] = __popcll(bal); wcount2[wave] = __popcll(bal2); }
        __syncthreads();
        int off = base, off2 = base2;
        for (int w = 0; w < wave; ++w) { off += wcount[w]; off2 += wcount2[w]; }
        if (on) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)b;
        if (on2) list2[off2 + __popcll(bal2 & ((1ull << lane) - 1ull))] = (int)b;
        __syncthreads();
        if (tid == 0) { int t = 0, t2 = 0; for (int w = 0; w < 16; ++w) { t += wcount[w]; t2 += wcount2[w]; } base += t; base2 += t2; }
        __syncthreads();
    }
    if (tid == 0) { *count = base; if (host_count) *host_count = base; if (count2) *count2 = base2; if (host_count2) *host_count2 = base2; }
}

// table[g][q][lane] = state_b[8 + 2P + q], q < 2P + 2L (c_old | c_new | k_old | k_new), b = list[64 g + lane] (live problems only); row
// 2P + 2L: 1 for a live slot, 0 for the padding of the last group.  Grid (groups, chunks of 64 coefficients): the 64 coefficients of a
// problem are read as one 512-byte run and leave transposed through LDS (the first version read every coefficient as a cache line of
// its own: 71 us per call at 1,000 problems x 613 coefficients).
__global__ void __launch_bounds__(256) coef_table_kernel(const double* __restrict__ gstate, long state_stride, int P, int L, const int* __restrict__ list,
                                                          const int* __restrict__ count, double* __restrict__ table) {
    __shared__ double tile[64][65];
    const int n = *count;
    const long g = blockIdx.x;
    if (g * 64 >= n) return;
    const int rows = 2 * P + 2 * L + 1, q0 = (int)blockIdx.y * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = w; r < 64; r += 4) {                            // replicate slot r of the group: coefficients q0 .. q0 + 63
        const long slot = g * 64 + r;
        double v = 0.0;
        if (slot < n) {
            const double* st = gstate + (long)list[slot] * state_stride;
            const int q = q0 + lane;
            v = (q < rows - 1) ? st[8 + 2 * P + q] : (q == rows - 1 ? 1.0 : 0.0);
        }
        tile[r][lane] = v;
    }
    __syncthreads();
    double* out = table + g * (long)rows * 64;
    for (int qq = w; qq < 64; qq += 4) {
        const int q = q0 + qq;
        if (q < rows) out[(long)q * 64 + lane] = tile[lane][qq];
    }
}

typedef double d8 __attribute__((ext_vector_type(8)));
// acc += x * c with the wave-uniform x read straight from a scalar register (keeps the x columns out of the VGPR file)
#define FMAC_SV(acc, xs, cv) asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "s"(xs), "v"(cv))
// keeps an accumulator array in VGPRs across a scheduling point (see nm_conv_dense_kernel)
#define PIN_ACC(a)                                                                                                              \
    do {                                                                                                                        \
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));         \
        if (RW == 16) asm volatile("" : "+v"(a[RW - 8]), "+v"(a[RW - 7]), "+v"(a[RW - 6]), "+v"(a[RW - 5]), "+v"(a[RW - 4]), "+v"(a[RW - 3]), "+v"(a[RW - 2]), "+v"(a[RW - 1])); \
    } while (0)

// RW rows per wave (a 16-row tile or half of one), NW waves per workgroup, two workgroups per CU.  The x columns come through the
// scalar cache with L2-like latency; resident waves hide it better than a deeper per-wave pipeline can (the SGPR file holds two
// 16-double columns, not four): 16 rows x 8 waves (used) beat 32 rows x 4 waves by 30 % and 8 rows x 16 waves by 45 %.
// BLOCKED: the coefficient tile of a 64-replicate group does not fit LDS as a whole (wide models, e.g. 300 indicator columns):
// it is staged one LV block at a time ((2 kb + 2) x 64 doubles, kb = widest block), with two barriers per block.
// CNT8 (round 3): the row multiplicities come from the int8 counts the digit-plane Gram already consumed (`Cd`, fragment-major:
// k-block of 64 rows x replicate tile of 16 -> [g 0..3][r 0..15][16 B]; the 16 rows of Xt tile t are piece g = t % 4 of k-block t / 4) --
// 16 bytes per (replicate, tile), lanes of 16 consecutive replicates read 256 contiguous bytes -- instead of a second, uint16 histogram
// that a second resample kernel had to write (20 KB per replicate).  dcnt_stride then carries MT (replicate tiles of the counts).
template <int RW, int NW, bool BLOCKED, bool CNT8 = false>
__global__ void __launch_bounds__(64 * NW) nm_conv_dense_kernel(const double* __restrict__ Xt, long ntiles, int PA, int P, int L, const int* __restrict__ boff,
                                                                 const unsigned short* __restrict__ dcnt, long dcnt_stride, const double* __restrict__ table,
                                                                 const int* __restrict__ list, const int* __restrict__ count, double* __restrict__ partial, int nparts, int rbx,
                                                                 int gy, int kb, int by_slot = 0) {
    // by_slot (round 6: the exact pass of the one-launch NUM / RAW solver's verification, plspm_nonmetric.hip run_nonmetric_wave): the list holds VIRTUAL problems --
    // (replicate, step) pairs; list[slot] is the replicate whose counts weigh the rows, the result is filed under the slot.
    static_assert(!CNT8 || RW == 16, "the int8 counts come in pieces of 16 rows");
    // multiplicities of this wave's RW rows in replicate b: packed words (uint16 pairs, or bytes of the int8 counts)
    auto load_counts = [&](unsigned (&wq)[RW / 2], bool on, long b, long part) {
        if (CNT8) {
            const uint4* cd = reinterpret_cast<const uint4*>(dcnt);
            const uint4 c = on ? cd[((part >> 2) * dcnt_stride + (b >> 4)) * 64 + (part & 3) * 16 + (b & 15)] : make_uint4(0, 0, 0, 0);
            wq[0] = c.x; wq[1] = c.y; wq[2] = c.z; wq[3] = c.w;
        } else {
            const uint4* cp = reinterpret_cast<const uint4*>(dcnt + (on ? b : 0) * dcnt_stride + (on ? part : 0) * RW);
#pragma unroll
            for (int h = 0; h < RW / 8; ++h) {
                const uint4 c = on ? cp[h] : make_uint4(0, 0, 0, 0);
                wq[4 * h] = c.x; wq[4 * h + 1] = c.y; wq[4 * h + 2] = c.z; wq[4 * h + 3] = c.w;
            }
        }
    };
    auto count_of = [&](const unsigned (&wq)[RW / 2], int r) -> double {
        if (CNT8) return (double)((wq[r >> 2] >> (8 * (r & 3))) & 0xffu);
        return (double)((r & 1) ? (wq[r >> 1] >> 16) : (wq[r >> 1] & 0xffffu));
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* co = reinterpret_cast<double*>(smem_raw);           // [2P + 2L + 1][64]   (BLOCKED: [2 kb + 2][64])
    constexpr int PER_TILE = 16 / RW;                           // row parts per 16-row tile of Xt
    static_assert(RW == 8 || RW == 16, "a wave takes a 16-row tile or half of one");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // XCD-aware decomposition of the 1-D grid: workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so
    // XCD k takes the row blocks k, k + 8, ... for every replicate slice -- its share of Xt (1/8 of 5 MB at 10k x 60) stays in
    // its L2 while the coefficient table streams through.  rbx = row blocks per XCD, gy = replicate slices.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int rb = (jj % rbx) * 8 + xcd, gy0 = jj / rbx;
    const long nparts_all = ntiles * PER_TILE;
    const long part = (long)rb * NW + wave;                     // wave-uniform: rows [RW part, RW part + RW)
    if ((long)rb * NW >= nparts_all) return;                    // whole workgroup beyond the data (uniform: no barrier is skipped)
    const bool have = part < nparts_all;
    const double* __restrict__ xt = Xt + (have ? part / PER_TILE : 0) * 16 * PA + (part % PER_TILE) * RW;
    const int rows = 2 * P + 2 * L + 1;
    const double* cn = co + (long)P * 64;
    const double* ko = co + 2L * P * 64;
    const double* kn = ko + (long)L * 64;
    // groups of 64 LIVE problems (active_list_kernel / coef_table_kernel): slot 64 g + lane is problem list[slot]; nothing live (the
    // speculative pass after the last iteration): no trip at all
    const int nlive = *count, ngroups = (nlive + 63) / 64;
    for (int g = gy0; g < ngroups; g += gy) {
        if (BLOCKED) {
            const bool live = (long)g * 64 + lane < nlive;
            const long b = live ? (long)list[(long)g * 64 + lane] : 0;
            unsigned wq[RW / 2];
            load_counts(wq, live && have, b, part);
            const double* tg = table + (long)g * rows * 64;
            double* bcn = co + (long)kb * 64;                   // [kb][64] new coefficients, then k_old[64], k_new[64]
            double* bk = bcn + (long)kb * 64;
            const unsigned lds_co = (unsigned)(size_t)(co + lane) & 0xffffffffu, lds_cn = (unsigned)(size_t)(bcn + lane) & 0xffffffffu;
            double acc = 0.0;
            for (int l = 0; l < L; ++l) {
                const int p0 = boff[l], nb = boff[l + 1] - p0;
                __syncthreads();                                // everybody is done with the previous block's coefficients
                for (int e = threadIdx.x; e < nb * 64; e += 64 * NW) { co[e] = tg[(long)p0 * 64 + e]; bcn[e] = tg[((long)P + p0) * 64 + e]; }
                if (threadIdx.x < 64) { bk[lane] = tg[(2L * P + l) * 64 + lane]; bk[64 + lane] = tg[(2L * P + L + l) * 64 + lane]; }
                __syncthreads();
                if (!have) continue;                            // (uniform per wave; the barriers above are reached by every wave)
                double ao[RW], an[RW];
                const double k0 = bk[lane], k1 = bk[64 + lane];
#pragma unroll
                for (int r = 0; r < RW; ++r) { ao[r] = k0; an[r] = k1; }
                double c0 = co[lane], c1 = bcn[lane];
                d8 xa0, xa1 = {};
                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(xa0) : "s"(xt + p0 * 16));
                if (RW == 16) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(xa1) : "s"(xt + p0 * 16));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xa0), "+s"(xa1));
                for (int q = 0; q < nb; ++q) {
                    const int qn = (q + 1 < nb) ? q + 1 : q;
                    const double* xn = xt + (p0 + qn) * 16;
                    d8 xn0, xn1 = {};
                    double c0n, c1n;
                    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(xn0) : "s"(xn));
                    if (RW == 16) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(xn1) : "s"(xn));
                    asm volatile("ds_read_b64 %0, %1" : "=v"(c0n) : "v"(lds_co + (unsigned)qn * 512u));
                    asm volatile("ds_read_b64 %0, %1" : "=v"(c1n) : "v"(lds_cn + (unsigned)qn * 512u));
                    PIN_ACC(ao); PIN_ACC(an);
#pragma unroll
                    for (int r = 0; r < 8; ++r) { FMAC_SV(ao[r], xa0[r], c0); FMAC_SV(an[r], xa0[r], c1); }
                    if (RW == 16) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { FMAC_SV(ao[8 + r], xa1[r], c0); FMAC_SV(an[8 + r], xa1[r], c1); }
                    }
                    PIN_ACC(ao); PIN_ACC(an);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xn0), "+s"(xn1), "+v"(c0n), "+v"(c1n));
                    xa0 = xn0; xa1 = xn1; c0 = c0n; c1 = c1n;
                }
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const double d = fabs(ao[r]) - fabs(an[r]);
                    const double w = count_of(wq, r);
                    acc = fma(w * d, d, acc);
                }
            }
            if (live && have) partial[(by_slot ? (long)g * 64 + lane : b) * nparts + part] = acc;
            continue;
        }
        __syncthreads();
        const double2* src = reinterpret_cast<const double2*>(table + (long)g * rows * 64);
        double2* dst = reinterpret_cast<double2*>(co);
        for (int e = threadIdx.x; e < rows * 32; e += 64 * NW) dst[e] = src[e];
        __syncthreads();
        if (!have) continue;
        const bool live = (long)g * 64 + lane < nlive;
        const long b = live ? (long)list[(long)g * 64 + lane] : 0;
        unsigned wq[RW / 2];                                    // two uint16 counts (or four int8 counts) per word
        load_counts(wq, live, b, part);
        double acc = 0.0;
        // Column loop, software-pipelined by hand (hipcc sinks a C++ prefetch below the FMAs and waits right after issuing it):
        // the loads of column p + 1 -- one or two s_load_dwordx16 for the x values, two ds_read_b64 for the lane's coefficients --
        // are ISSUED before the FMAs of column p and WAITED for after them.  The empty asm statements pin the accumulators on
        // both sides of the FMA block so that the compiler cannot move the FMAs across the issue / wait points.
        const unsigned lds_co = (unsigned)(size_t)(co + lane) & 0xffffffffu, lds_cn = (unsigned)(size_t)(cn + lane) & 0xffffffffu;
        double c0 = co[lane], c1 = cn[lane];
        d8 xa0, xa1 = {};                                       // column 0 (through the same scalar path: the loop-carried x stays in SGPRs)
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(xa0) : "s"(xt));
        if (RW == 16) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(xa1) : "s"(xt));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xa0), "+s"(xa1));
        int p = 0;
        for (int l = 0; l < L; ++l) {
            double ao[RW], an[RW];
            const double k0 = ko[l * 64 + lane], k1 = kn[l * 64 + lane];
#pragma unroll
            for (int r = 0; r < RW; ++r) { ao[r] = k0; an[r] = k1; }
            const int pend = boff[l + 1];
            for (; p < pend; ++p) {
                const int pn = (p + 1 < P) ? p + 1 : p;
                const double* xn = xt + pn * 16;
                d8 xn0, xn1 = {};
                double c0n, c1n;
                asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(xn0) : "s"(xn));
                if (RW == 16) asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(xn1) : "s"(xn));
                asm volatile("ds_read_b64 %0, %1" : "=v"(c0n) : "v"(lds_co + (unsigned)pn * 512u));
                asm volatile("ds_read_b64 %0, %1" : "=v"(c1n) : "v"(lds_cn + (unsigned)pn * 512u));
                PIN_ACC(ao); PIN_ACC(an);
#pragma unroll
                for (int r = 0; r < 8; ++r) { FMAC_SV(ao[r], xa0[r], c0); FMAC_SV(an[r], xa0[r], c1); }
                if (RW == 16) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { FMAC_SV(ao[8 + r], xa1[r], c0); FMAC_SV(an[8 + r], xa1[r], c1); }
                }
                PIN_ACC(ao); PIN_ACC(an);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xn0), "+s"(xn1), "+v"(c0n), "+v"(c1n));
                xa0 = xn0; xa1 = xn1; c0 = c0n; c1 = c1n;
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const double d = fabs(ao[r]) - fabs(an[r]);
                const double w = count_of(wq, r);
                acc = fma(w * d, d, acc);
            }
        }
        if (live) partial[(by_slot ? (long)g * 64 + lane : b) * nparts + part] = acc;
    }
}

// ---------------------------------------------------------------------------------------------- verification of the one-launch NUM / RAW solver (round 6)
// solver_nmwave_kernel (kernels_solver.h; solver_wave16.h NM) runs prepare + all steps + finish of a replicate in ONE launch: a step stops on the quadratic
// upper bound of the reference's score criterion and continues otherwise -- speculatively, since "the bound is not below the tolerance" does not say that the
// criterion is not.  What was speculated is checked here, on the observations, for every step a problem continued behind: VIRTUAL problems (b, j), j = 1 ..
// steps_b - 1, "step j of replicate b: sum_il c_i (|y_{j-1}| - |y_j|)^2 >= tol?".  All terms are non-negative, so any subset of the rows gives a lower bound:
// pass A evaluates the first rows only (an eighth; step 2 of the headline's replicates sits at 1e-4 against 1e-6, step 1 at 1e5) and confirms nearly
// everything; what it cannot confirm goes through pass B over all rows with fixed-order sums, and a (b, j) whose exact value IS below the tolerance moves
// replicate b's stop to step j (force[b] = min j; the solver replays those replicates).  Maps: maps[b][j] = [c_p (P) | k_l (L)] of step j's scores.
//
// vlist: the virtual problems of steps j0 .. j0 + JR - 1, step-major (consecutive slots = consecutive replicates: coalesced count loads), vb / vj, *count;
// also max steps -> *host_max (pinned), vsum cleared, force[] = INT_MAX on the first round.
// One workgroup, two sweeps: wave w owns a contiguous range of replicates, counts its virtual problems per step, then -- behind one barrier -- files them.
template <int JR>
__global__ void __launch_bounds__(1024) nm_vlist_kernel(const int* __restrict__ steps, long nproblems, int j0, int* __restrict__ vb, int* __restrict__ vj,
                                                         int* __restrict__ count, double* __restrict__ vsum, int* __restrict__ force, int* __restrict__ host_max, int cap) {
    // `cap` > 0 (a long round behind the batch's main body: the few replicates that are still iterating, many steps each): more than `cap` slots would not fit the
    // round's buffers -- nothing is filed then, host_max[3] = 1, and the host walks the same steps in short rounds
    __shared__ int wtot[16][JR];
    __shared__ int smax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) smax = 0;
    const long per = (((nproblems + 15) / 16) + 63) & ~63L;      // replicates per wave, whole trips of 64
    const long w0 = (long)wave * per, w1 = lmin(nproblems, w0 + per);
    int tot[JR], mx = 0;
#pragma unroll
    for (int jj = 0; jj < JR; ++jj) tot[jj] = 0;
    for (long b0 = w0; b0 < w1; b0 += 64) {
        const long b = b0 + lane;
        const int st = b < w1 ? steps[b] : 0;
        mx = max(mx, st);
        if (j0 == 1 && b < w1) force[b] = 0x7fffffff;
#pragma unroll
        for (int jj = 0; jj < JR; ++jj) tot[jj] += __popcll(__ballot(st - 1 >= j0 + jj));
    }
    if (lane == 0) {
#pragma unroll
        for (int jj = 0; jj < JR; ++jj) wtot[wave][jj] = tot[jj];
    }
    __syncthreads();
    atomicMax(&smax, mx);
    int base[JR], total = 0;                                     // first slot of (step j0 + jj, this wave)
#pragma unroll
    for (int jj = 0; jj < JR; ++jj) {
        int before = 0, all = 0;
        for (int w = 0; w < 16; ++w) { const int c = wtot[w][jj]; all += c; if (w < wave) before += c; }
        base[jj] = total + before;
        total += all;
    }
    const bool over = cap > 0 && total > cap;
    for (long b0 = w0; b0 < (over ? w0 : w1); b0 += 64) {
        const long b = b0 + lane;
        const int st = b < w1 ? steps[b] : 0;
#pragma unroll
        for (int jj = 0; jj < JR; ++jj) {
            const bool on = st - 1 >= j0 + jj;
            const unsigned long long bal = __ballot(on);
            if (on) {
                const int v = base[jj] + __popcll(bal & ((1ull << lane) - 1ull));
                vb[v] = (int)b; vj[v] = j0 + jj; vsum[v] = 0.0;
            }
            base[jj] += __popcll(bal);
        }
    }
    __syncthreads();
    if (tid == 0) { *count = over ? 0 : total; if (host_max) { host_max[0] = smax; if (cap > 0) host_max[3] = over ? 1 : 0; } }
}

// table[g][q][lane] of the virtual problems: q < P old coefficients (step j - 1), q < 2P new (step j), then k_old[L], k_new[L], live flag -- the layout
// coef_table_kernel writes for the pass.
// vneed[v] (may be null): the row blocks (128 rows) pass A reads for slot v -- the solver left the quadratic bound ub_j of every step beside its map, and the
// criterion it bounds is nearly always a few sign flips below it, so a fraction ~ 8 tol / ub_j of the rows carries the lower bound over the tolerance: one
// block for a first step (ub ~ 1e5), a handful for a second (ub ~ 1e-4 against 1e-6), at most `cap_blocks`; `fixed_blocks` > 0 (option nm_verify_rows): that
// many for every slot.
__global__ void __launch_bounds__(256) nm_vtable_kernel(const double* __restrict__ maps, long maps_stride, int P, int L, const int* __restrict__ vb, const int* __restrict__ vj,
                                                         const int* __restrict__ count, double* __restrict__ table, int* __restrict__ vneed, double tol, int nblocks_all,
                                                         int cap_blocks, int fixed_blocks) {
    __shared__ double tile[64][65];
    const int n = *count;
    const long g = blockIdx.x;
    if (g * 64 >= n) return;
    if (vneed && blockIdx.y == 0 && threadIdx.x < 64 && g * 64 + threadIdx.x < n) {
        const long v = g * 64 + threadIdx.x;
        int need = fixed_blocks > 0 ? fixed_blocks : cap_blocks;
        if (fixed_blocks <= 0) {
            const double ub = maps[(long)vb[v] * maps_stride + (long)vj[v] * (P + L + 1) + P + L];
            const double want = 8.0 * tol / ub * (double)nblocks_all;       // (ub <= 0 or NaN: the cap)
            if (want >= 0.0 && want < (double)cap_blocks) need = max(1, (int)want + 1);
        }
        vneed[v] = need;
    }
    const int rows = 2 * P + 2 * L + 1, q0 = (int)blockIdx.y * 64, W = P + L + 1;      // (a map: c_p | k_l | the bound of its step)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = w; r < 64; r += 4) {
        const long slot = g * 64 + r;
        double v = 0.0;
        if (slot < n) {
            const double* mo = maps + (long)vb[slot] * maps_stride + (long)(vj[slot] - 1) * W;
            const double* mn = mo + W;
            const int q = q0 + lane;
            if (q < P) v = mo[q];
            else if (q < 2 * P) v = mn[q - P];
            else if (q < 2 * P + L) v = mo[P + q - 2 * P];
            else if (q < rows - 1) v = mn[P + q - 2 * P - L];
            else if (q == rows - 1) v = 1.0;
        }
        tile[r][lane] = v;
    }
    __syncthreads();
    double* out = table + g * (long)rows * 64;
    for (int qq = w; qq < 64; qq += 4) {
        const int q = q0 + qq;
        if (q < rows) out[(long)q * 64 + lane] = tile[lane][qq];
    }
}

// Pass A: the lower bound of the criterion of every virtual problem from its first `need` row blocks.  The work is small (a few hundred workgroup-sized units
// per 5,000 replicates) and nm_conv_dense_kernel -- built for throughput: its x columns come one scalar load per column through the scalar cache, whose latency
// the many resident waves of a full pass hide -- runs it as a chain of ~60 dependent memory round trips per workgroup (85 us whatever the row count,
// tools/experiments/nm_verify_prof.sh).  Here a wave copies its 16-row tile of Xt into LDS once (coalesced) and the column loop reads x as LDS broadcasts
// beside the lane's two coefficients: the chain is the FMAs'.  Workgroup = (row block of 8 tiles, group of 64 slots); lane = slot; one atomic add per lane.
__global__ void __launch_bounds__(512) nm_verify_kernel(const double* __restrict__ Xt, long ntiles, int PA, int P, int L, const int* __restrict__ boff, const uint4* __restrict__ cd,
                                                         long MT, const double* __restrict__ table, const int* __restrict__ vb, const int* __restrict__ count,
                                                         const int* __restrict__ vneed, double* __restrict__ vsum, int nrb, int gy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int rows = 2 * P + 2 * L + 1;
    double* co = reinterpret_cast<double*>(smem_raw);           // [rows][64]
    double* xs = co + (long)rows * 64;                           // [8 waves][P][16]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rb = blockIdx.x % nrb, gy0 = blockIdx.x / nrb;
    const long tile = (long)rb * 8 + wave;
    const bool have = tile < ntiles;
    double* xw = xs + (long)wave * P * 16;
    const int nlive = *count, ngroups = (nlive + 63) / 64;
    bool tile_loaded = false;
    for (int g = gy0; g < ngroups; g += gy) {
        const long slot = (long)g * 64 + lane;
        const int mine = slot < nlive ? vneed[slot] : 0;
        if (rb >= wv::allreduce(mine, [](int a, int b) { return a > b ? a : b; })) continue;      // (uniform over the workgroup)
        __syncthreads();
        {
            const double2* src = reinterpret_cast<const double2*>(table + (long)g * rows * 64);
            double2* dst = reinterpret_cast<double2*>(co);
            for (int e = threadIdx.x; e < rows * 32; e += 512) dst[e] = src[e];
        }
        if (have && !tile_loaded) {
            const double2* src = reinterpret_cast<const double2*>(Xt + tile * 16 * PA);
            double2* dst = reinterpret_cast<double2*>(xw);
            for (int e = lane; e < P * 8; e += 64) dst[e] = src[e];
            tile_loaded = true;
        }
        __syncthreads();
        if (!have) continue;                                     // (the barriers of the next trip are reached by every wave: `continue` above is workgroup-uniform)
        const bool live = slot < nlive;
        const long b = live ? (long)vb[slot] : 0;
        const uint4 cw = live ? cd[((tile >> 2) * MT + (b >> 4)) * 64 + (tile & 3) * 16 + (b & 15)] : make_uint4(0, 0, 0, 0);
        const unsigned wq[4] = {cw.x, cw.y, cw.z, cw.w};
        const double* cn = co + (long)P * 64;
        const double* ko = co + 2L * P * 64;
        const double* kn = ko + (long)L * 64;
        double acc = 0.0;
        int p = 0;
        for (int l = 0; l < L; ++l) {
            double ao[16], an[16];
            const double k0 = ko[l * 64 + lane], k1 = kn[l * 64 + lane];
#pragma unroll
            for (int r = 0; r < 16; ++r) { ao[r] = k0; an[r] = k1; }
            const int pend = boff[l + 1];
            for (; p < pend; ++p) {
                const double c0 = co[p * 64 + lane], c1 = cn[p * 64 + lane];
                const double2* xr = reinterpret_cast<const double2*>(xw + p * 16);
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const double2 x = xr[h];
                    ao[2 * h] = fma(x.x, c0, ao[2 * h]); an[2 * h] = fma(x.x, c1, an[2 * h]);
                    ao[2 * h + 1] = fma(x.y, c0, ao[2 * h + 1]); an[2 * h + 1] = fma(x.y, c1, an[2 * h + 1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double d = fabs(ao[r]) - fabs(an[r]);
                const double w = (double)((wq[r >> 2] >> (8 * (r & 3))) & 0xffu);
                acc = fma(w * d, d, acc);
            }
        }
        if (live) unsafeAtomicAdd(&vsum[slot], acc);
    }
}

// What pass A could not confirm: slots whose partial sum is not safely at or above the tolerance -> (fb, fj), *fcount (+ pinned copy).  One workgroup.
__global__ void __launch_bounds__(1024) nm_vflag_kernel(const double* __restrict__ vsum, const int* __restrict__ vb, const int* __restrict__ vj, const int* __restrict__ count,
                                                         double tol, int* __restrict__ fb, int* __restrict__ fj, int* __restrict__ fcount, int* __restrict__ host_count) {
    __shared__ int wcount[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    const int n = *count;
    for (int v0 = 0; v0 < n; v0 += 1024) {
        const int v = v0 + tid;
        const bool on = v < n && !(vsum[v] >= tol * (1.0 + 1e-6));      // (NaN: flagged; the exact pass leaves it alone)
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wcount[w];
        if (on) { const int o = off + __popcll(bal & ((1ull << lane) - 1ull)); fb[o] = vb[v]; fj[o] = vj[v]; }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wcount[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) { *fcount = base; if (host_count) *host_count = base; }
}

// Pass B's verdict: the exact criterion of flagged slot v (fixed-order sum of its row parts) below the tolerance -> the reference stops replicate fb[v] at step fj[v].
__global__ void __launch_bounds__(64) nm_vcheck_kernel(const double* __restrict__ partial, int nparts, const int* __restrict__ fb, const int* __restrict__ fj,
                                                        const int* __restrict__ fcount, double tol, int* __restrict__ force) {
    const int v = blockIdx.x;
    if (v >= *fcount) return;
    double s = 0.0;
    for (int c = threadIdx.x; c < nparts; c += 64) s += partial[(long)v * nparts + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0 && s < tol) atomicMin(&force[fb[v]], fj[v]);
}

// Replicates whose stop moved: force[b] < steps[b] -> list (+ count, pinned copy).  One workgroup.
__global__ void __launch_bounds__(1024) nm_vfix_kernel(const int* __restrict__ steps, const int* __restrict__ force, long nproblems, int* __restrict__ list, int* __restrict__ count,
                                                        int* __restrict__ host_count) {
    __shared__ int wcount[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (long b0 = 0; b0 < nproblems; b0 += 1024) {
        const long b = b0 + tid;
        const bool on = b < nproblems && force[b] < steps[b];
        const unsigned long long bal = __ballot(on);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wcount[w];
        if (on) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)b;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wcount[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) { *count = base; if (host_count) *host_count = base; }
}

// ---------------------------------------------------------------------------------------------- dense stop-rule pass on category codes
// All-indicator categorical models (every device column a 0/1 indicator of one category of one MV: plspm_model::cat_pure): a row holds
// exactly one 1 among the columns of an MV, and the rows of a tile are the same for every lane (= replicate).  So instead of 2 x 16
// multiply-adds per COLUMN of the block (nm_conv_dense_kernel: ten per row and five-category MV, nine of them with x = 0), the wave reads
// the 16 category codes of (tile, MV) through the scalar cache and ADDS the old and the new coefficient of that one column -- the same
// additions in the same order (fma(1, c, a) = a + c; fma(0, c, a) = a), so the partial sums are bit for bit those of the dense pass.
// Coefficients of a block in LDS interleaved [column][old | new][64 lanes] (one address per row and MV); slot kb stays zero: the code of
// a row without a 1 in that MV (the pad rows of the last tile).
// codes[(tile * Pm + mv) * 16 + r]: the block-relative column of MV mv's category in row 16 tile + r.
// mv_base[mv]: first column of the block the pass files MV mv under (its LV's block; a second HOC stage: the block of the second-stage LV
// that stands for it).
__global__ void __launch_bounds__(256) cat_codes_kernel(const double* __restrict__ Xa, long N, int PA, int Pm, const int* __restrict__ mv_off, const int* __restrict__ mv_base,
                                                         int none, long ntiles, unsigned short* __restrict__ codes) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= ntiles * Pm * 16) return;
    const int r = (int)(e & 15);
    const long tm = e >> 4;
    const int mv = (int)(tm % Pm);
    const long i = (tm / Pm) * 16 + r;
    int code = none;
    if (i < N) {
        const int c0 = mv_off[mv], c1 = mv_off[mv + 1], b0 = mv_base[mv];
        for (int c = c0; c < c1; ++c)
            if (Xa[i * PA + c] != 0.0) { code = c - b0; break; }
    }
    codes[e] = (unsigned short)code;
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) nm_conv_codes_kernel(const unsigned short* __restrict__ codes, long ntiles, int Pm, int P, int L, const int* __restrict__ boff,
                                                                 const int* __restrict__ lmv_off, const uint4* __restrict__ cd, long MT, const double* __restrict__ table,
                                                                 const int* __restrict__ list, const int* __restrict__ count, double* __restrict__ partial, int nparts, int rbx,
                                                                 int gy, int kb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* co = reinterpret_cast<double*>(smem_raw);           // [kb + 1][2][64], then k_old[64], k_new[64]
    double* bk = co + (long)(kb + 1) * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;       // the dense pass's XCD-aware decomposition: row blocks k, k + 8, ... on XCD k
    const int rb = (jj % rbx) * 8 + xcd, gy0 = jj / rbx;
    const long part = (long)rb * NW + wave;                     // this wave's 16-row tile
    if ((long)rb * NW >= ntiles) return;
    const bool have = part < ntiles;
    const uint4* __restrict__ cq = reinterpret_cast<const uint4*>(codes + (have ? part : 0) * (long)Pm * 16);      // two uint4 per MV
    const int rows = 2 * P + 2 * L + 1;
    for (int e = threadIdx.x; e < 128; e += 64 * NW) co[(long)kb * 128 + e] = 0.0;
    const int nlive = *count, ngroups = (nlive + 63) / 64;
    for (int g = gy0; g < ngroups; g += gy) {
        const bool live = (long)g * 64 + lane < nlive;
        const long b = live ? (long)list[(long)g * 64 + lane] : 0;
        const uint4 cw = (live && have) ? cd[((part >> 2) * MT + (b >> 4)) * 64 + (part & 3) * 16 + (b & 15)] : make_uint4(0, 0, 0, 0);
        const unsigned wq[4] = {cw.x, cw.y, cw.z, cw.w};
        const double* tg = table + (long)g * rows * 64;
        double acc = 0.0;
        for (int l = 0; l < L; ++l) {
            const int p0 = boff[l], nb = boff[l + 1] - p0;
            __syncthreads();                                    // everybody is done with the previous block's coefficients
            for (int e = threadIdx.x; e < nb * 64; e += 64 * NW) {
                const int q = e >> 6, ln = e & 63;
                co[q * 128 + ln] = tg[(long)p0 * 64 + e];
                co[q * 128 + 64 + ln] = tg[((long)P + p0) * 64 + e];
            }
            if (threadIdx.x < 64) { bk[lane] = tg[(2L * P + l) * 64 + lane]; bk[64 + lane] = tg[(2L * P + L + l) * 64 + lane]; }
            __syncthreads();
            if (!have) continue;                                // (uniform per wave; the barriers above are reached by every wave)
            double ao[16], an[16];
            const double k0 = bk[lane], k1 = bk[64 + lane];
#pragma unroll
            for (int r = 0; r < 16; ++r) { ao[r] = k0; an[r] = k1; }
            const double* cl = co + lane;
            const int m1 = lmv_off[l + 1];
            for (int mv = lmv_off[l]; mv < m1; ++mv) {
                const uint4 ca = cq[2 * mv], cb = cq[2 * mv + 1];      // (wave-uniform address: scalar loads)
                const unsigned c16[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned code = (c16[r >> 1] >> (16 * (r & 1))) & 0xffffu;
                    const double* pc = cl + code * 128u;
                    ao[r] += pc[0];
                    an[r] += pc[64];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const double d = fabs(ao[r]) - fabs(an[r]);
                const double w = (double)((wq[r >> 2] >> (8 * (r & 3))) & 0xffu);
                acc = fma(w * d, d, acc);
            }
        }
        if (live && have) partial[b * nparts + part] = acc;
    }
}

// second stage of a HOC pair: its score maps composed with the first stage's, in the layout the stop-rule passes read (solver_hoc.h)
__global__ void __launch_bounds__(64) hoc_compose_kernel(HocDesc hd, const double* __restrict__ state1, long st1_stride, double* state2, long st2_stride, int n_chol2,
                                                         double* pseudo, long ps_stride, const int* __restrict__ live) {
    const long b = live ? live[blockIdx.x] : (long)blockIdx.x;      // (nm_kernel: the live list of the previous step -- a problem that stopped in this step still
                                                                    //  gets its flag copied; one that stopped earlier keeps the cleared flag it got then)
    const double* st = state1 + b * st1_stride;
    NmState st2;
    nm_carve(st2, state2 + b * st2_stride, hd.P2, hd.L2);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, nullptr, nullptr};
    hoc_compose_score_maps(ex, hd, st + 8 + 3 * hd.P1, st + 8 + 4 * hd.P1 + hd.L1, st2, pseudo + b * ps_stride);
}
