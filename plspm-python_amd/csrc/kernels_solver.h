// kernels_solver.h -- Device kernels, part 3: the metric solver kernels (LDS / rows / wave), imputation and HOC moment transforms, operator seam.
// Included by plspm_fit.hip only (device_exec.h in front); not a stand-alone header.
#pragma once

// LDS: [S (if s_in_lds)] [small workspace (if small_in_lds)]; otherwise the global scratch areas are used.
// The placement is a template parameter so that every workspace pointer has ONE provenance: the compiler then proves the
// LDS ones to be address-space-3 (ds_read / ds_write) instead of falling back to flat_load / flat_store.
template <bool S_IN_LDS, bool SMALL_IN_LDS>
__global__ void __launch_bounds__(256) solver_kernel(ModelDesc md, const double* __restrict__ Mp, long mp_stride, SolverOut so, double* gS, double* gsmall) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lds = reinterpret_cast<double*>(smem_raw);
    const long b = blockIdx.x;
    const int PS = cov_ld(md.P);
    const long s_doubles = cov_doubles(md.P), small_doubles = workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    Workspace ws;
    ws.PS = PS;
    double* lp = lds;
    if (S_IN_LDS) { ws.S = lp; lp += s_doubles; } else ws.S = gS + b * s_doubles;
    if (SMALL_IN_LDS) { carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol); lp += small_doubles; }
    else carve_small(ws, gsmall + b * small_doubles, md.P, md.L, md.kmax, md.n_chol);
    stage_descriptors(md, lp);
    FitOutputs out = so.fit;
    if (b != 0) out = FitOutputs{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, ws.red, (b == 0) ? so.marks : nullptr};
    solve_problem(ex, md, ws, Mp + b * mp_stride, out);
}


#ifndef PLSPM_ROWS_WAVES
#define PLSPM_ROWS_WAVES 2
#endif
// Rows variant (solver_core.h solve_problem_rows): ONE wave per problem, column p of the covariance in the registers of lane p,
// the small workspace + descriptors in LDS (~11 KB at P = 60, L = 6).  Bootstrap batches of metric models with P <= 64 whose
// moment matrices arrive dense from the int8 digit-plane Gram.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PLSPM_ROWS_WAVES, PLSPM_ROWS_WAVES))) solver_rows_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    const long b = blockIdx.x;
    Workspace ws;
    ws.PS = cov_ld(md.P);
    ws.S = nullptr;
    carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol);
    lp += workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    stage_descriptors(md, lp);
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevExecT<4> ex{(int)threadIdx.x, 64, ws.red, (b == 0) ? so.marks : nullptr};
    solve_problem_rows<64>(ex, md, ws, Md + b * md_stride, out);
}


#define PLSPM_ROWS_SPLIT_STAGE_DOUBLES (4 * 16 * 66)
// Split rows variant (round 4; solve_problem_rows<64, true>): 64 < P <= 128 MVs, FOUR waves per problem -- thread t serves MV t mod 128 with the
// columns on side t / 128 of a block boundary, so the covariance of a 120-MV model lives in the registers of four waves (two problems per
// CU, small workspace ~30 KB of LDS each) where solver_kernel keeps 115 KB of LDS per problem (one per CU).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) solver_rows_split_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    double* turn = lp;                              // four 16 x 66 tiles: the block loader's transposition staging (device_exec.h load_cov_block)
    lp += PLSPM_ROWS_SPLIT_STAGE_DOUBLES;
    const long b = blockIdx.x;
    Workspace ws;
    ws.PS = cov_ld(md.P);
    ws.S = nullptr;
    carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol);
    lp += workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    stage_descriptors(md, lp);
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevExecT<4> ex{(int)threadIdx.x, 256, ws.red, (b == 0) ? so.marks : nullptr};
    ex.xstage = turn;
    solve_problem_rows<64, true>(ex, md, ws, Md + b * md_stride, out);
}


// Wave variant (solver_wave.h solve_problem_wave): ONE wave per problem with fixed lane roles -- the bootstrap solver of metric Mode-A
// models with at most 64 MVs and 8 LVs.  Executor = the rows executor + the wave primitives.
struct DevWaveExec : DevExecT<4> {
    // butterfly sum (lane ^ 1, 2, ... 32; wave_ops.h: DPP + permlane swaps, no LDS crossbar): every lane ends with bitwise the same value
    __device__ __forceinline__ double allsum(double v) { return wv::allsum(v); }
    // value of `v` on lane q (q wave-uniform): two v_readlane into scalar registers -- no LDS round trip; `published` serves the CPU emulation
    __device__ __forceinline__ double bcast(double v, int q, const double*) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), q), __builtin_amdgcn_readlane(__double2loint(v), q));
    }
    __device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ double uniform_d(double v) {                           // a value every lane holds identically -> scalar registers
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    __device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }       // nothing that touches memory moves across
    __device__ __forceinline__ void opaque(int& x) { asm volatile("" : "+v"(x)); }
    __device__ __forceinline__ void opaque(unsigned& x) { asm volatile("" : "+v"(x)); }      // the optimiser may not look through x (no hoisting of what derives from it)
    // ... and these eight values are complete at this point (a fence the arithmetic cannot sink below)
    __device__ __forceinline__ void pin8(double& a, double& b, double& c, double& d, double& e, double& f, double& g, double& h) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
    }
    __device__ __forceinline__ int vote_count(bool b) { return __popcll(__ballot(b)); }
    __device__ __forceinline__ int wave_vote_count(bool b) { return __popcll(__ballot(b)); }      // (solver_quad.h: four waves per problem, votes per wave)
    __device__ __forceinline__ bool vote_any(bool b) { return __ballot(b) != 0ull; }
    // Column `lane` of the symmetric moment matrix out of its upper triangle (entry (r, c >= r) at r * PS + c), into s[0 .. 63].
    //   A. row r across the lanes, r = 0 .. 63: lane c >= r reads M[r][c] -- coalesced 512-byte rows; lanes c < r re-read the diagonal
    //      element (same line): register r of lane c is column c's entry r wherever r <= c.  (The rows solver read the other half as
    //      a strided run per lane: 134 MB fetched for 75 MB of triangles, the load phase of 2,048 resident waves bound by the address
    //      units -- 29 k clocks in the first round, profiles/r02c_pmc.md.)
    //   B. the other half is the transpose, s_p[q] = s_q[p] for q > p: four passes of 16 registers through a [16][66]-double LDS tile
    //      (written row-wise, one row per register; lane p of the pass's 16 lanes reads row p - 16 j: pitch 66 = conflict-free 8-byte
    //      reads, 16-byte aligned rows).
    // Rows / lanes >= P repeat valid entries (finite; the caller zeroes them).
    template <int PMAX> __device__ __forceinline__ void load_cov(const double* __restrict__ Md, int PS, int P, double (&s)[PMAX], double* stage, double& mu, double& diag) {
        static_assert(PMAX == 64, "one register per lane of the wave");
        // (lane P -- when there is one -- takes the ones column: row r of the staging tile then carries the column sum M[r][P] as well, and
        //  lane p picks its column sum and its diagonal entry out of its own row: no strided loads of 2 x P cache lines per problem)
        const int lane = tid, cl = min(lane, P);
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const int rc = min(r, P - 1);
            const unsigned off = (unsigned)(rc * PS + max(cl, rc)) * 8u;
            s[r] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Md) + off);
        }
        const bool ones_lane = P < 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[r * 66 + lane] = s[16 * j + r];
            __syncthreads();
            if ((lane >> 4) == j) {
                const double* row = stage + (lane - 16 * j) * 66;
                diag = row[lane];
                mu = ones_lane ? row[min(P, 63)] : Md[(long)min(lane, P - 1) * PS + P];
#pragma unroll
                for (int q = 16 * j + 1; q < 16 * (j + 1); ++q) s[q] = (q > lane) ? row[q] : s[q];
#pragma unroll
                for (int q = 16 * (j + 1); q < 64; ++q) s[q] = row[q];
            }
            __syncthreads();
        }
    }
};

template <int LMAX, bool MODEB = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) solver_wave_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long b = blockIdx.x;
    WaveWs<LMAX> ws;
    wave_carve(ws, reinterpret_cast<double*>(smem_raw));
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevWaveExec ex;
    ex.tid = (int)threadIdx.x; ex.nt = 64; ex.red = nullptr; ex.marks = (b == 0) ? so.marks : nullptr;
    solve_problem_wave<LMAX, MODEB>(ex, md, ws, Md + b * md_stride, out);
}


// Wave variant for 9 .. 16 LVs (solver_wave16.h solve_problem_wave16; round 5): one wave per problem, four matrix entries per pair lane, V in LDS.
template <int LMAX, bool MODEB = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LMAX > 16 ? 1 : 2, LMAX > 16 ? 1 : 2))) solver_wave16_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long b = blockIdx.x;
    Wave16Ws<LMAX> ws;
    wave16_carve(ws, reinterpret_cast<double*>(smem_raw), md.L, md.kmax);
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevWaveExec ex;
    ex.tid = (int)threadIdx.x; ex.nt = 64; ex.red = nullptr; ex.marks = (b == 0) ? so.marks : nullptr;
    solve_problem_wave16<LMAX, MODEB>(ex, md, ws, Md + b * md_stride, out);
}


// The same wave on Scale.NUM / RAW non-metric data (solver_wave16.h NM; round 6): prepare + every step + finish of a replicate in one launch, the steps it
// continued behind left as score maps for the verification pass (kernels_nonmetric.h nm_vlist_kernel ...).  `live` / `force`: the replay of the replicates
// whose stop the verification moved (no maps stored).  The per-block sums of the map constants take 64 doubles behind the metric workspace.
template <int LMAX, bool MODEB = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LMAX > 16 ? 1 : 2, LMAX > 16 ? 1 : 2))) solver_nmwave_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so,
                                                                                                        double* __restrict__ maps, long maps_stride, int* __restrict__ steps,
                                                                                                        const int* __restrict__ force, const int* __restrict__ live, double bound_scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long b = live ? (long)live[blockIdx.x] : (long)blockIdx.x;
    Wave16Ws<LMAX> ws;
    wave16_carve(ws, reinterpret_cast<double*>(smem_raw), md.L, md.kmax);
    double* nmk = reinterpret_cast<double*>(smem_raw) + wave16_ws_doubles<LMAX>(md.L, md.kmax, md.n_chol);
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    NmWaveIo io;
    io.maps = force ? nullptr : maps + b * maps_stride;
    io.force_T = force ? force[b] : 0;
    io.steps = force ? nullptr : steps + b;
    io.bound_scale = bound_scale;
    DevWaveExec ex;
    ex.tid = (int)threadIdx.x; ex.nt = 64; ex.red = nullptr; ex.marks = (b == 0) ? so.marks : nullptr;
    solve_problem_wave16<LMAX, MODEB, true>(ex, md, ws, Md + b * md_stride, out, &io, nmk);
}


// Quad variant (solver_quad.h solve_problem_quad; round 5): the wave solver's lane roles on FOUR waves per problem -- metric Mode-A models of
// 65 .. 128 MVs and at most 16 LVs; two problems per CU (one wave of each per SIMD), ~52 KB of LDS per problem.
template <int LMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) solver_quad_kernel(ModelDesc md, const double* __restrict__ Md, long md_stride, SolverOut so) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const long b = blockIdx.x;
    QuadWs<LMAX> ws;
    quad_carve(ws, reinterpret_cast<double*>(smem_raw));
    FitOutputs out{};
    out.row = so.row ? so.row + b * so.row_stride : nullptr;
    out.status = so.status ? so.status + b : nullptr;
    out.iters = so.iters ? so.iters + b : nullptr;
    DevWaveExec ex;
    ex.tid = (int)threadIdx.x; ex.nt = 256; ex.red = nullptr; ex.marks = (b == 0) ? so.marks : nullptr;
    ex.xstage = ws.stage;
    solve_problem_quad<LMAX>(ex, md, ws, Md + b * md_stride, out);
}


// Metric data with missing values: per problem, the Gram of [data | missing indicators | 1] -> mean-imputed moments of the P
// data columns (solver_core.h impute_collapse).  One workgroup per problem.
__global__ void __launch_bounds__(256) impute_kernel(int P, int Qa, int Ta, int Ts, const int* __restrict__ ind_of, const double* __restrict__ Min, long in_stride,
                                                     double* __restrict__ Mout, long out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* gam = reinterpret_cast<double*>(smem_raw);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, nullptr, nullptr};
    impute_collapse(ex, P, Qa, Ta, Ts, ind_of, Min + blockIdx.x * in_stride, Mout + blockIdx.x * out_stride, gam);
}


// Two-stage HOC bootstrap (solver_hoc.h): stage-1 Gram + final stage-1 score maps -> stage-2 moment matrix, one workgroup per replicate.
__global__ void __launch_bounds__(256) hoc_moments_kernel(HocDesc hd, const double* __restrict__ M1, long m1_stride, const double* __restrict__ state1, long st1_stride,
                                                          double* __restrict__ M2, long m2_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* V = reinterpret_cast<double*>(smem_raw);
    const long b = blockIdx.x;
    const double* st = state1 + b * st1_stride;
    const double* c1 = st + 8 + 3 * hd.P1;                  // NmState: scal[8] a_old a_new c_old c_new k_old k_new
    const double* k1 = st + 8 + 4 * hd.P1 + hd.L1;
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, nullptr, nullptr};
    hoc_second_stage_moments(ex, hd, M1 + b * m1_stride, c1, k1, st[1] == (double)ST_OK, M2 + b * m2_stride, V);
}



// ------------------------------------------------------------------------------------------------ operator seam (solver_ops.h)
// One workgroup: descriptors + the small workspace in LDS (the handle's "MVs" are the L score columns).
__global__ void __launch_bounds__(256) op_inner_kernel(ModelDesc md, const double* __restrict__ Mp, double* __restrict__ E_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* lp = reinterpret_cast<double*>(smem_raw);
    Workspace ws;
    ws.PS = cov_ld(md.P);
    ws.S = nullptr;
    carve_small(ws, lp, md.P, md.L, md.kmax, md.n_chol);
    lp += workspace_small_doubles(md.P, md.L, md.kmax, md.n_chol);
    stage_descriptors(md, lp);
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, ws.red, nullptr};
    op_inner_weights(ex, md, ws, Mp, E_out);
}
// scratch: [A k*k | F k*k | V k*k | flag] in global memory (a block may have up to 1020 MVs)
__global__ void __launch_bounds__(256) op_outer_kernel(int mode, int k, int T, const double* __restrict__ Mp, const double* __restrict__ shift, double* scratch,
                                                        double* __restrict__ w_out) {
    __shared__ double red[16];
    DevExec ex{(int)threadIdx.x, (int)blockDim.x, red, nullptr};
    const long kk = (long)k * k;
    op_outer_weights(ex, mode, k, T, Mp, shift, scratch, scratch + kk, scratch + 2 * kk, w_out, scratch + 3 * kk);
}

// Mode.X.value.outer_weights_nonmetric (reference plspm/mode.py:31-42, 54-61) on the block's quantified MVs: one workgroup of 1,024 threads,
// three passes over the N x k block.  Every sum runs over a fixed thread-to-row map and a fixed LDS tree: bit-reproducible.
//   Mode A, complete block   w = X' z / sum z^2 (mode.py:38-39);      Y = X w
//   Mode A, NaN cells        w_p = nansum_i(x_ip z_i) / sum_i (m_ip z_i)^2,  Y_i = nansum_p(x_ip w_p) / sum_p (m_ip w_p)^2 (mode.py:33-37; m = presence mask)
//   Mode B                   w arrives (least squares of z on X, plspm_op_outer_weights);  Y = X w (mode.py:58-59)
//   then Y <- treat_numpy(Y) * correction = (Y - nanmean) / nanstd1 * correction (util.py:43-53, mode.py:41,60)
// red: LDS [1024] doubles.
__device__ __forceinline__ double op_block_sum(double v, double* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    const double s = red[0];
    __syncthreads();
    return s;
}
__global__ void __launch_bounds__(1024) op_nm_outer_kernel(int have_w, long N, int k, const double* __restrict__ X, const unsigned char* __restrict__ present,
                                                           const double* __restrict__ z, double correction, double* __restrict__ w, double* __restrict__ Y) {
    __shared__ double red[1024];
    const int t = threadIdx.x;
    if (!have_w) {
        double zz = 0.0;
        if (!present) { for (long i = t; i < N; i += 1024) zz += z[i] * z[i]; zz = op_block_sum(zz, red); }
        for (int p = 0; p < k; ++p) {
            double num = 0.0, den = 0.0;
            for (long i = t; i < N; i += 1024) {
                const double x = X[i * k + p], zi = z[i];
                if (x == x) num += x * zi;
                if (present) { const double mz = present[i * k + p] ? zi : 0.0; den += mz * mz; }
            }
            num = op_block_sum(num, red);
            if (present) den = op_block_sum(den, red); else den = zz;
            if (t == 0) w[p] = num / den;
            __syncthreads();
        }
    }
    __threadfence_block();
    __syncthreads();
    double sum = 0.0, cnt = 0.0;
    for (long i = t; i < N; i += 1024) {
        double y = 0.0, den = 0.0;
        for (int p = 0; p < k; ++p) {
            const double x = X[i * k + p], wp = w[p];
            if (x == x) y += x * wp;
            if (present) { const double mw = present[i * k + p] ? wp : 0.0; den += mw * mw; }
        }
        if (present) y = y / den;
        Y[i] = y;
        if (y == y) { sum += y; cnt += 1.0; }
    }
    sum = op_block_sum(sum, red);
    cnt = op_block_sum(cnt, red);
    const double mean = sum / cnt;
    double ss = 0.0;
    for (long i = t; i < N; i += 1024) { const double d = Y[i] - mean; if (d == d) ss += d * d; }
    ss = op_block_sum(ss, red);
    const double sd = sqrt(ss / (cnt - 1.0));
    for (long i = t; i < N; i += 1024) Y[i] = (Y[i] - mean) / sd * correction;
}


