// plspm_fit.hip -- host side, part 2: the fp64 MFMA Gram launches, the metric solvers, the single fit (plspm_fit) and the operator seam
// (plspm_op_*).  Kernels: kernels_gram.h, kernels_solver.h, kernels_scores.h.  (The non-metric iteration: plspm_nonmetric.hip.)
#include "host_internal.h"
#include <hip/hip_ext.h>

#include "wave_ops.h"
#include "device_exec.h"
#include "kernels_gram.h"
#include "kernels_solver.h"
#include "kernels_scores.h"

// ------------------------------------------------------------------------------------------------ launch helpers
template <bool DENSE>
static int launch_gram(plspm_model* m, long nproblems, int nchunks, const int2* ent, const int* nent, long ent_stride, double* out) {
    const dim3 grid(nchunks, (unsigned)nproblems);
    const long N = m->N;
    hipStream_t s = m->stream;
#define ROWS(TT)                                                                                                         \
    {                                                                                                                    \
        const size_t lds = std::max<size_t>(2 * (size_t)TileIdx<TT>::NTILE * 256 * sizeof(double), (size_t)m->tune.gram_lds_kb * 1024); \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)gram_rows_kernel<TT, DENSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gram_rows_kernel<TT, DENSE>), grid, dim3(256), lds, s, m->d_Xa, N, ent, nent, ent_stride, out); \
    }
#define WIDE(TT, NWV, NWP)                                                                                                     \
    hipLaunchKernelGGL((gram_wide_kernel<TT, NWV, NWP, DENSE>), dim3(nchunks, (unsigned)nproblems, NWV / NWP), dim3(NWP * 64), 0, s, m->d_Xa, N, ent, \
                       nent, ent_stride, out);
    switch (m->T) {
        case 2: ROWS(2) break;
        case 4: ROWS(4) break;
        case 5: WIDE(5, 4, 4) break;
        case 6: WIDE(6, 4, 4) break;
        case 7: WIDE(7, 4, 4) break;
        case 9: WIDE(9, 4, 4) break;
        case 11: WIDE(11, 4, 4) break;
        case 13:
            // (configs[4]: the dense walk on a ring of row buffers -- kernels_gram.h gram_walk_dense_ring; option wide_ring 0: the two-stage ping-pong)
            if (DENSE && m->tune.wide_ring == 4) { hipLaunchKernelGGL((gram_wide_kernel<13, 4, 4, DENSE, DENSE ? 4 : 0>), dim3(nchunks, (unsigned)nproblems, 1), dim3(256), 0, s, m->d_Xa, N, ent, nent, ent_stride, out); }
            else if (DENSE && m->tune.wide_ring >= 5) { hipLaunchKernelGGL((gram_wide_kernel<13, 4, 4, DENSE, DENSE ? 6 : 0>), dim3(nchunks, (unsigned)nproblems, 1), dim3(256), 0, s, m->d_Xa, N, ent, nent, ent_stride, out); }
            else WIDE(13, 4, 4)
            break;
        case 15: WIDE(15, 4, 4) break;
        case 8: WIDE(8, 4, 4) break;
        case 10: WIDE(10, 4, 4) break;
        case 12: WIDE(12, 4, 4) break;
        case 14: {
            const int sel = m->tune.wide_nw;        // measured on 1M x 200: NW=4 0.99 ms, 8: 1.12 ms, 16 (two workgroups per walk): 1.37 ms
            if (sel == 4) WIDE(14, 4, 4) else if (sel == 16) WIDE(14, 16, 8) else WIDE(14, 8, 8)
        } break;
        case 16: WIDE(16, 16, 8) break;
        default: {
            if (m->T < 18 || (m->T & 1)) return fail(m, PLSPM_E_LIMIT, "unsupported tile count");
            const int TB = (m->T + 3) / 4, nsb = TB * (TB + 1) / 2;
            hipLaunchKernelGGL((gram_block_kernel<DENSE>), dim3(nchunks, (unsigned)nproblems, (nsb + 3) / 4), dim3(256), 0, s, m->d_Xa, N, m->T, ent, nent, ent_stride, out);
        } break;
    }
#undef ROWS
#undef WIDE
    HIPCHK(m, hipGetLastError());
    return 0;
}

size_t desc_lds_bytes(int P, int L, int ne, int nedge) {
    const size_t T = ((size_t)P + 1 + 31) / 32 * 2, ntile = T * (T + 1) / 2;
    return (size_t)P * 8 + (3 * (size_t)(L + 1) + P + 2 * (size_t)L + 2 * (size_t)ne + 2 * (size_t)nedge + (ntile + 1) / 2 + 4) * 4 + (((size_t)L * L + 15) & ~(size_t)15) + 16;
}

// Missing-data models: collapse the aug Gram(s) at `Min` into mean-imputed P-column moments; returns the matrix the solver reads.
int run_impute(plspm_model* m, long nproblems, const double* Min, const double** Mp, long* mp_stride) {
    *Mp = Min; *mp_stride = packed_size(m->T);
    if (!m->n_ind) return 0;
    const long out_stride = packed_size(m->Ts);
    int rc = ensure(m, m->gram2, (size_t)nproblems * out_stride * sizeof(double));
    if (rc) return rc;
    ProfScope ps(m, PLSPM_K_REDUCE);
    hipLaunchKernelGGL(impute_kernel, dim3((unsigned)nproblems), dim3(256), (size_t)m->P * sizeof(double), m->stream, m->P, m->Pg, m->T, m->Ts, m->d_ind_of, Min,
                       packed_size(m->T), (double*)m->gram2.p, out_stride);
    *Mp = (const double*)m->gram2.p; *mp_stride = out_stride;
    return 0;
}

int launch_solver(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so, int threads) {
    const int P = m->P, L = m->L;
    const size_t s_bytes = (size_t)cov_doubles(P) * sizeof(double);
    const size_t small_bytes = (size_t)workspace_small_doubles(P, L, m->kmax, m->n_chol) * sizeof(double);
    const size_t lds_budget = (nproblems == 1) ? kMaxLds : 64 * 1024;   // batched: keep >= 2 workgroups per CU
    int s_in_lds = 0, small_in_lds = 0;
    size_t lds = desc_lds_bytes(P, L, m->n_eff, (int)m->pred_idx.size());
    if (lds + small_bytes <= lds_budget) { small_in_lds = 1; lds += small_bytes; }
    if (small_in_lds && lds + s_bytes <= lds_budget) { s_in_lds = 1; lds += s_bytes; }
    if (!s_in_lds) { int rc = ensure(m, m->gS, (size_t)nproblems * s_bytes); if (rc) return rc; }
    if (!small_in_lds) { int rc = ensure(m, m->gsmall, (size_t)nproblems * small_bytes); if (rc) return rc; }
#define SOLVER_LAUNCH(A, B)                                                                                                        \
    {                                                                                                                              \
        int rc = allow_lds(m, (const void*)solver_kernel<A, B>, lds);                                                               \
        if (rc) return rc;                                                                                                         \
        hipLaunchKernelGGL((solver_kernel<A, B>), dim3((unsigned)nproblems), dim3(threads), lds, m->stream, make_desc(m), Mp, mp_stride, so, \
                           (double*)m->gS.p, (double*)m->gsmall.p);                                                                \
    }
    if (s_in_lds && small_in_lds) SOLVER_LAUNCH(true, true)
    else if (small_in_lds) SOLVER_LAUNCH(false, true)
    else SOLVER_LAUNCH(false, false)
#undef SOLVER_LAUNCH
    HIPCHK(m, hipGetLastError());
    return 0;
}


// Moment matrix of ALL uploaded rows (single fit, operator seam): the dense MFMA Gram split over row chunks + the fixed-order
// reduce -> m->gram (tile-packed, of the shifted columns + ones).
int dense_moments(plspm_model* m) {
    const long N = m->N;
    const long psize = packed_size(m->T);
    const long ng = (N + 3) / 4;
    // row chunks (workgroups) of the dense Gram: enough k-groups per wave to amortise the pipeline prologue; at most two workgroups
    // per CU for the rows kernel (every wave holds all tiles), ONE for the tile-split kernels (a wave per SIMD already fills the
    // register file; measured on 1M x 200: 256 chunks 0.941 + reduce 0.027 ms, 512: 0.939 + 0.063, 1024: 0.954 + 0.132)
    const int waves_per_wg = (m->T <= 4) ? 4 : 1;
    const long per_wave = 16;
    const long max_chunks = (m->T <= 4) ? 512 : 256;
    const int nchunks = m->tune.fit_chunks > 0 ? m->tune.fit_chunks : (int)std::max<long>(1, std::min<long>(max_chunks, (ng + waves_per_wg * per_wave - 1) / (waves_per_wg * per_wave)));
    int rc;
    if ((rc = ensure(m, m->gram_partial, (size_t)nchunks * psize * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->gram, (size_t)psize * sizeof(double)))) return rc;
    {
        ProfScope ps(m, PLSPM_K_GRAM);
        if ((rc = launch_gram<true>(m, 1, nchunks, nullptr, nullptr, 0, (double*)m->gram_partial.p))) return rc;
    }
    {
        ProfScope ps(m, PLSPM_K_REDUCE);
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((psize + 255) / 256)), dim3(256), 0, m->stream, (const double*)m->gram_partial.p, nchunks, psize,
                           (double*)m->gram.p);
    }
    HIPCHK(m, hipGetLastError());
    return 0;
}

int launch_gram_lists(plspm_model* m, long nproblems, const int2* ent, const int* nent, long ent_stride, double* out) {
    return launch_gram<false>(m, nproblems, 1, ent, nent, ent_stride, out);
}

int run_hoc_moments(plspm_model* m, plspm_model* m2, long nb) {
    const long psize = packed_size(m->T), psize2 = packed_size(m2->Ts);
    int rc;
    if ((rc = ensure(m, m2->gram, (size_t)nb * psize2 * sizeof(double)))) return rc;
    const HocDesc hd = make_hoc_desc(m2);
    const size_t vlds = std::max<size_t>(1, (size_t)hd.nh * (hd.P1 + 1)) * sizeof(double);
    if ((rc = allow_lds(m, (const void*)hoc_moments_kernel, vlds))) return rc;
    ProfScope ps(m, PLSPM_K_REDUCE);
    hipLaunchKernelGGL(hoc_moments_kernel, dim3((unsigned)nb), dim3(256), vlds, m->stream, hd, (const double*)m->gram.p, psize, (const double*)m->nmstate.p,
                       (long)nm_state_doubles_of(m), (double*)m2->gram.p, psize2);
    HIPCHK(m, hipGetLastError());
    return 0;
}

bool nm_wave_solver_covers(const plspm_model* m) {
    if (!m->nonmetric || m->categorical || m->n_ind || m->nmx_K > 0 || m->stage1 || m->stage2 || m->P > 64) return false;
    if (m->L <= 8) return m->P >= 1 && (wave16_ws_doubles<8>(m->L, m->kmax, m->n_chol) + 64) * sizeof(double) <= 20 * 1024 && m->n_chol / 2 <= 16 * 66;
    if (m->L <= 16) return wave16_solver_covers<16>(m->P, m->L, m->n_chol, m->kmax);
    return m->n_chol == 0 && wave16_solver_covers<32>(m->P, m->L, 0, m->kmax);      // 17 .. 32 LVs: all Mode A, as the metric <32> form
}

int launch_nm_wave_solver(plspm_model* m, long nb, const SolverOut& so, double* maps, long maps_stride, int* steps, const int* force, const int* live) {
    int rc;
    const double* gram_buf = (const double*)m->gram.p;
    ProfScope ps(m, PLSPM_K_SOLVER);
    if (m->L <= 8) {
        const size_t lds = (size_t)(wave16_ws_doubles<8>(m->L, m->kmax, m->n_chol) + 64) * sizeof(double);
        auto k = m->n_chol > 0 ? solver_nmwave_kernel<8, true> : solver_nmwave_kernel<8, false>;
        if ((rc = allow_lds(m, (const void*)k, lds))) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so, maps, maps_stride, steps, force, live, std::ldexp(1.0, m->tune.nm_bound_shift));
        m->last_solver = 9;
    } else if (m->L <= 16) {
        const size_t lds = (size_t)(wave16_ws_doubles<16>(m->L, m->kmax, m->n_chol) + 64) * sizeof(double);
        auto k = m->n_chol > 0 ? solver_nmwave_kernel<16, true> : solver_nmwave_kernel<16, false>;
        if ((rc = allow_lds(m, (const void*)k, lds))) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so, maps, maps_stride, steps, force, live, std::ldexp(1.0, m->tune.nm_bound_shift));
        m->last_solver = 10;
    } else {
        const size_t lds = (size_t)(wave16_ws_doubles<32>(m->L, m->kmax, 0) + 64) * sizeof(double);
        auto k = solver_nmwave_kernel<32, false>;
        if ((rc = allow_lds(m, (const void*)k, lds))) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so, maps, maps_stride, steps, force, live, std::ldexp(1.0, m->tune.nm_bound_shift));
        m->last_solver = 11;
    }
    return 0;
}

int launch_batch_solver(plspm_model* m, long nb, bool dense, const SolverOut& so_in) {
    SolverOut so = so_in;
    int rc;
    const double* gram_buf = (const double*)m->gram.p;
#ifdef PLSPM_DEBUG_MARKS      // phase clocks of one solver problem (make marks); never in the release library
    long long* d_marks = nullptr;
    HIPCHK(m, plspm_dmalloc((void**)&d_marks, 32 * sizeof(long long))); so.marks = d_marks;
#endif
    if (dense && (m->tune.solver_wave == 1 || m->tune.solver_wave == 2) && m->L <= 8 && wave_solver_covers<8>(m->P, m->L, m->n_chol) &&
        wave16_ws_doubles<8>(m->L, m->kmax, m->n_chol) * sizeof(double) <= 20 * 1024) {
        // round 5 (second half): models of at most 8 LVs -- the headline's class -- in the arrangement of solver_wave16.h at LMAX = 8: V in LDS, the product
        // stream's second copy w V for the Q sums, a folded into E (0.095 -> 0.084 ms per 5,000 replicates; set_option("solver_wave", 3): the round-3 kernel);
        // Mode-B blocks: the round-4 block inverses on this workspace, a second instantiation
        const size_t lds = (size_t)wave16_ws_doubles<8>(m->L, m->kmax, m->n_chol) * sizeof(double);
        auto k8 = m->n_chol > 0 ? solver_wave16_kernel<8, true> : solver_wave16_kernel<8, false>;
        if ((rc = allow_lds(m, (const void*)k8, lds))) return rc;
        ProfScope ps(m, PLSPM_K_SOLVER);
        hipEvent_t stop = m->stop_event;
        m->stop_event = nullptr;
        hipExtLaunchKernelGGL(k8, dim3((unsigned)nb), dim3(64), lds, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        m->last_solver = 7;
    } else if (dense && m->tune.solver_wave != 0 && wave_solver_covers<8>(m->P, m->L, m->n_chol)) {
        // one wave per problem with fixed lane roles (solver_wave.h): at most 64 MVs and 8 LVs; Mode-B blocks keep their inverses behind the workspace
        const size_t lds = (size_t)wave_ws_doubles<8>(m->n_chol) * sizeof(double);
        ProfScope ps(m, PLSPM_K_SOLVER);
        // (a caller that wants an event behind this batch -- plspm_group.cpp: the `computed` event its collective waits for -- hands it over as the
        //  launch's own completion signal: a separate hipEventRecord is one more packet the queue drains the device for, ~5 us of every step)
        hipEvent_t stop = m->stop_event;
        m->stop_event = nullptr;
        if (m->n_chol > 0) hipExtLaunchKernelGGL((solver_wave_kernel<8, true>), dim3((unsigned)nb), dim3(64), lds, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        else hipExtLaunchKernelGGL((solver_wave_kernel<8, false>), dim3((unsigned)nb), dim3(64), lds, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        m->last_solver = 3;
    } else if (dense && m->tune.solver_wave != 0 && wave16_solver_covers<16>(m->P, m->L, m->n_chol, m->kmax)) {
        // the same for 9 .. 16 LVs (solver_wave16.h; round 5): four matrix entries per pair lane, V in LDS
        const size_t lds = (size_t)wave16_ws_doubles<16>(m->L, m->kmax, m->n_chol) * sizeof(double);
        auto k16 = m->n_chol > 0 ? solver_wave16_kernel<16, true> : solver_wave16_kernel<16, false>;
        if ((rc = allow_lds(m, (const void*)k16, lds))) return rc;
        ProfScope ps(m, PLSPM_K_SOLVER);
        hipEvent_t stop = m->stop_event;
        m->stop_event = nullptr;
        hipExtLaunchKernelGGL(k16, dim3((unsigned)nb), dim3(64), lds, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        m->last_solver = 6;
    } else if (dense && m->tune.solver_wave != 0 && m->n_chol == 0 && wave16_solver_covers<32>(m->P, m->L, m->n_chol, m->kmax)) {
        // ... and for 17 .. 32 LVs: sixteen matrix entries per pair lane, three problems per CU (all Mode A)
        const size_t lds = (size_t)wave16_ws_doubles<32>(m->L, m->kmax, 0) * sizeof(double);
        if ((rc = allow_lds(m, (const void*)solver_wave16_kernel<32, false>, lds))) return rc;
        ProfScope ps(m, PLSPM_K_SOLVER);
        hipEvent_t stop = m->stop_event;
        m->stop_event = nullptr;
        hipExtLaunchKernelGGL((solver_wave16_kernel<32, false>), dim3((unsigned)nb), dim3(64), lds, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        m->last_solver = 8;
    } else if (dense) {
        const size_t lds = desc_lds_bytes(m->P, m->L, m->n_eff, (int)m->pred_idx.size()) + (size_t)workspace_small_doubles(m->P, m->L, m->kmax, m->n_chol) * sizeof(double);
        if (m->P > 64 && m->tune.solver_quad != 0 && quad_solver_covers<16>(m->P, m->L, m->n_chol, m->kmax, m->boff.data())) {
            // four waves per problem with fixed lane roles (solver_quad.h; round 5): Mode-A models of 65 .. 128 MVs and at most 16 LVs
            m->last_solver = 5;
            const size_t ldsq = (size_t)quad_ws_doubles<16>(m->L, m->kmax) * sizeof(double);
            if ((rc = allow_lds(m, (const void*)solver_quad_kernel<16>, ldsq))) return rc;
            ProfScope ps(m, PLSPM_K_SOLVER);
            hipEvent_t stop = m->stop_event;
            m->stop_event = nullptr;
            hipExtLaunchKernelGGL((solver_quad_kernel<16>), dim3((unsigned)nb), dim3(256), ldsq, m->stream, nullptr, stop, 0, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        } else if (m->P > 64) {                    // split form: two threads per MV (plspm_detail_bootstrap asked rows_split_block)
            m->last_solver = 4;
            const size_t lds4 = lds + PLSPM_ROWS_SPLIT_STAGE_DOUBLES * sizeof(double);
            if ((rc = allow_lds(m, (const void*)solver_rows_split_kernel, lds4))) return rc;
            ProfScope ps(m, PLSPM_K_SOLVER);
            hipLaunchKernelGGL(solver_rows_split_kernel, dim3((unsigned)nb), dim3(256), lds4, m->stream, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        } else {
            m->last_solver = 2;
            if ((rc = allow_lds(m, (const void*)solver_rows_kernel, lds))) return rc;
            ProfScope ps(m, PLSPM_K_SOLVER);
            hipLaunchKernelGGL(solver_rows_kernel, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), gram_buf, (long)cov_doubles(m->Pg), so);
        }
    } else {
        const double* Mp; long mp_stride;
        if ((rc = run_impute(m, nb, gram_buf, &Mp, &mp_stride))) return rc;
        ProfScope ps(m, PLSPM_K_SOLVER);
        m->last_solver = 1;
        if ((rc = launch_solver(m, nb, Mp, mp_stride, so, m->tune.solver_threads))) return rc;
    }
    HIPCHK(m, hipGetLastError());
#ifdef PLSPM_DEBUG_MARKS
    {
        long long h[32];
        HIPCHK(m, hipStreamSynchronize(m->stream));
        HIPCHK(m, hipMemcpy(h, d_marks, sizeof(h), hipMemcpyDeviceToHost));
        if (m->last_solver == 3 || m->last_solver >= 5) {
            fprintf(stderr, "[plspm wave clocks] load %lld  treat %lld  init %lld  iterations %lld  finalize %lld  inner %lld  effects %lld  outputs %lld  total %lld\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[13] - h[7], h[13] - h[0]);
            fprintf(stderr, "[plspm wave last iterate] apply_cov %lld  a/G/E %lld  regress %lld  outer+conv %lld\n", h[9] - h[8], h[10] - h[9], h[11] - h[10], h[12] - h[11]);
            fprintf(stderr, "[plspm wave last apply_cov] seg_products+T %lld  sync %lld  Q %lld\n", h[17] - h[16], h[18] - h[17], h[19] - h[18]);
            if (m->n_chol > 0) fprintf(stderr, "[plspm wave Mode-B inverses] setup %lld  fill %lld  rows %lld  sweep %lld  store %lld  fallback test %lld  rest of init %lld\n", h[20] - h[2], h[21] - h[20], h[22] - h[21],
                                       h[23] - h[22], h[24] - h[23], h[25] - h[24], h[3] - h[25]);
        } else {
        fprintf(stderr, "[plspm solver clocks] cov %lld  chol+init %lld  iterate %lld  finalize %lld  inner %lld  effects %lld  outputs %lld  total %lld\n",
                h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
        fprintf(stderr, "[plspm cov] sweep %lld  scale-factor %lld  centre+sd %lld\n", h[14] - h[0], h[15] - h[14], h[1] - h[15]);
        fprintf(stderr, "[plspm last iterate] apply_cov %lld  a+G %lld  inner_weights %lld  outer %lld  conv+copy %lld\n", h[9] - h[8], h[10] - h[9],
                h[11] - h[10], h[12] - h[11], h[13] - h[12]);
        fprintf(stderr, "[plspm last apply_cov] block products %lld  Q %lld\n", h[17] - h[16], h[18] - h[17]);
        }
        plspm_dfree(d_marks);
    }
#endif
    return 0;
}

extern "C" {

int plspm_fit(plspm_model_t* m, const plspm_fit_result_t* out) {
    if (!m || !out) return PLSPM_E_ARG;
    if (!m->d_Xa || m->N < 2) return fail(m, PLSPM_E_STATE, "plspm_fit: no data uploaded");
    HIPCHK(m, hipSetDevice(m->device));
    const int P = m->P, L = m->L, ne = m->n_eff;
    const long N = m->N;
    const long psize = packed_size(m->T);
    int rc;
    // device-side result block
    const long o_w = 0, o_ld = o_w + P, o_cl = o_ld + P, o_pc = o_cl + (long)P * L, o_r2 = o_pc + (long)L * L, o_lc = o_r2 + L,
               o_row = o_lc + (long)L * L, o_ind = o_row + (2L * P + L + 2L * ne + 2), o_sw = o_ind + std::max(ne, 1), o_sc = o_sw + P, o_mean = o_sc + L,
               o_cov = o_mean + P, o_end = o_cov + (long)P * P;
    const size_t fit_bytes = (size_t)o_end * sizeof(double) + 64 + (size_t)L + 16;
    if ((rc = ensure(m, m->fitout, fit_bytes))) return rc;
    double* d = (double*)m->fitout.p;
    int* d_int = (int*)(d + o_end);           // [0] iters, [1] status
    int8_t* d_sign = (int8_t*)(d_int + 4);
    if ((rc = dense_moments(m))) return rc;
    SolverOut so{};
    so.row = d + o_row; so.row_stride = 0; so.status = d_int + 1; so.iters = d_int;
    so.fit.weights = d + o_w; so.fit.loadings = d + o_ld; so.fit.crossloadings = d + o_cl; so.fit.path_coef = d + o_pc; so.fit.r2 = d + o_r2;
    so.fit.lv_cov = d + o_lc; so.fit.indirect = d + o_ind; so.fit.score_w = d + o_sw; so.fit.score_c = d + o_sc;
    so.fit.cov = out->cov ? d + o_cov : nullptr; so.fit.mean = d + o_mean; so.fit.sign = d_sign;
    if (m->nonmetric) {
        if ((rc = run_nonmetric(m, 1, (const double*)m->gram.p, psize, so, nullptr, nullptr, 0, 256))) return rc;
    } else {
        const double* Mp; long mp_stride;
        if ((rc = run_impute(m, 1, (const double*)m->gram.p, &Mp, &mp_stride))) return rc;
        ProfScope ps(m, PLSPM_K_SOLVER);
        if ((rc = launch_solver(m, 1, Mp, mp_stride, so, 256))) return rc;
    }
    if (out->scores) {
        if ((rc = ensure(m, m->scores, (size_t)N * L * sizeof(double)))) return rc;
        // tile rows: 32 while two workgroups of them fit one CU's LDS, else 16 (wide models; option "scores_tile" overrides);
        // chunk count per thread selects the prefetching instantiation (PA <= 256), wider matrices take the plain one
        auto lds_of = [&](int tr) { return ((size_t)tr * (m->PA + 1) + P + L + (size_t)tr * L) * sizeof(double) + (size_t)(L + 2) * sizeof(int); };
        const bool tr32 = m->tune.scores_tile ? (m->tune.scores_tile == 32) : (lds_of(32) <= 72 * 1024);
        const int TRows = tr32 ? 32 : 16;
        const size_t lds = lds_of(TRows);
        const int nch = (m->PA / 2 + 15) / 16;
        typedef void (*ScoresFn)(const double*, long, int, int, int, const int*, const double*, const double*, double*);
        ScoresFn fn;
        if (tr32) fn = nch <= 2 ? scores_kernel<32, 2> : nch <= 4 ? scores_kernel<32, 4> : nch <= 6 ? scores_kernel<32, 6> : nch <= 8 ? scores_kernel<32, 8> : scores_kernel<32, 0>;
        else fn = nch <= 2 ? scores_kernel<16, 2> : nch <= 4 ? scores_kernel<16, 4> : nch <= 6 ? scores_kernel<16, 6> : nch <= 8 ? scores_kernel<16, 8> : scores_kernel<16, 0>;
        if ((rc = allow_lds(m, (const void*)fn, lds))) return rc;
        const long ntl = (N + TRows - 1) / TRows;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, kMaxLds / lds));
        const int grid = (int)std::min<long>(256L * per_cu, ntl);         // resident workgroups only: each walks its tiles with the prefetch running
        ProfScope ps(m, PLSPM_K_SCORES);
        hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, m->stream, (const double*)m->d_Xa, N, m->PA, P, L, (const int*)m->d_boff, (const double*)(d + o_sw), (const double*)(d + o_sc), (double*)m->scores.p);
        if (m->nmx_K) {                                     // the incomplete rows' scores are not affine in the columns: take them from the state
            const double* Yn = (const double*)m->nmstate.p + nm_state_doubles(P, L, m->n_chol) + m->nmx_K + 2L * P + (long)m->nmx_K * P + (long)m->nmx_K * L;
            hipLaunchKernelGGL(patch_scores_kernel, dim3((unsigned)m->nmx_K), dim3(64), 0, m->stream, (double*)m->scores.p, L, (const int*)m->d_rowid, Yn);
        }
    }
    HIPCHK(m, hipGetLastError());
    // ONE device->host copy of the whole result block into a pinned staging buffer, then scatter on the host
    // (15 separate small copies cost more than the four kernels of a 10k x 60 fit).
    const size_t block_bytes = (size_t)(out->cov ? o_end : o_cov) * sizeof(double);
    const size_t tail_bytes = 64 + (size_t)L + 16;
    const size_t score_bytes = out->scores ? sizeof(double) * (size_t)N * L : 0;
    const bool stage_scores = score_bytes > 0 && score_bytes <= ((size_t)8 << 20);      // small score matrices ride the pinned buffer too
    const size_t stage_need = fit_bytes + (stage_scores ? score_bytes : 0);
    if (m->h_stage_cap < stage_need) {
        if (m->h_stage) plspm_hfree(m->h_stage);
        m->h_stage = nullptr; m->h_stage_cap = 0;
        HIPCHK(m, plspm_hmalloc(&m->h_stage, stage_need));
        m->h_stage_cap = stage_need;
    }
    char* hs = (char*)m->h_stage;
    HIPCHK(m, hipMemcpyAsync(hs, d, block_bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipMemcpyAsync(hs + (size_t)o_end * sizeof(double), d + o_end, tail_bytes, hipMemcpyDeviceToHost, m->stream));
    if (stage_scores) HIPCHK(m, hipMemcpyAsync(hs + fit_bytes, m->scores.p, score_bytes, hipMemcpyDeviceToHost, m->stream));
    else if (out->scores) HIPCHK(m, hipMemcpyAsync(out->scores, m->scores.p, score_bytes, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    // plspm_bootstrap_prepare ran before this fit: its column statistics are on the host now, so the digit planes are cut (enqueue only)
    // while the caller unpacks the fit -- the first bootstrap call finds them ready.  (A failure here is reported by that call, which retries.)
    if (m->zs_stats_ready && !m->zs_valid) { const std::string keep = m->error; if (prepare_zs(m)) { (void)hipGetLastError(); m->error = keep; } }
    if (stage_scores) memcpy(out->scores, hs + fit_bytes, score_bytes);
    const double* h = (const double*)hs;
    const int* h_int = (const int*)(h + o_end);
    auto put = [&](void* dst, const void* src, size_t bytes) { if (dst) memcpy(dst, src, bytes); };
    const int Po = m->categorical ? m->Pm : P;             // categorical handles report per logical MV, not per aug column
    put(out->weights, h + o_w, sizeof(double) * Po);
    put(out->loadings, h + o_ld, sizeof(double) * Po);
    put(out->crossloadings, h + o_cl, sizeof(double) * Po * L);
    put(out->path_coef, h + o_pc, sizeof(double) * L * L);
    put(out->r2, h + o_r2, sizeof(double) * L);
    put(out->lv_cov, h + o_lc, sizeof(double) * L * L);
    put(out->total, h + o_row + Po + L, sizeof(double) * ne);
    put(out->direct, h + o_row + Po + L + ne, sizeof(double) * ne);
    put(out->indirect, h + o_ind, sizeof(double) * ne);
    put(out->cov, h + o_cov, sizeof(double) * Po * Po);
    if (m->categorical) { if (out->mean) memset(out->mean, 0, sizeof(double) * Po); }
    else put(out->mean, h + o_mean, sizeof(double) * P);
    put(out->sign, h_int + 4, (size_t)L);
    put(out->iterations, h_int, sizeof(int));
    put(out->status, h_int + 1, sizeof(int));
    return 0;
}


// ---- operator seam (solver_ops.h): the reference's Scheme / Mode plug-ins, one call = upload + MFMA Gram + one small kernel ----
int plspm_op_inner_weights(int32_t device_id, int32_t scheme, int32_t L, const uint8_t* path, const double* y, int64_t N, double* E) {
    g_create_error.clear();
    if (!path || !y || !E || L < 1 || L > 64 || N < 2) return fail(nullptr, PLSPM_E_ARG, "plspm_op_inner_weights: bad arguments (1 <= L <= 64, N >= 2)");
    std::vector<int32_t> boff(L + 1), mode(L, PLSPM_MODE_A);
    for (int l = 0; l <= L; ++l) boff[l] = l;                       // every LV's "block" is its own score column
    plspm_model* m = plspm_model_create(L, L, boff.data(), path, mode.data(), scheme, 0, 1, 1.0, device_id);
    if (!m) return PLSPM_E_ARG;                                     // text in plspm_last_error(NULL)
    auto done = [&](int rc) { if (rc) g_create_error = m->error; plspm_model_destroy(m); return rc; };
    int rc;
    if ((rc = plspm_upload(m, y, N, L, 0, nullptr)) || (rc = dense_moments(m))) return done(rc);
    if ((rc = ensure(m, m->fitout, sizeof(double) * (size_t)L * L))) return done(rc);
    const size_t lds = (size_t)workspace_small_doubles(L, L, m->kmax, m->n_chol) * sizeof(double) + desc_lds_bytes(L, L, m->n_eff, (int)m->pred_idx.size());
    if ((rc = allow_lds(m, (const void*)op_inner_kernel, lds))) return done(rc);
    hipLaunchKernelGGL(op_inner_kernel, dim3(1), dim3(256), lds, m->stream, make_desc(m), (const double*)m->gram.p, (double*)m->fitout.p);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(E, m->fitout.p, sizeof(double) * (size_t)L * L, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipStreamSynchronize(m->stream) != hipSuccess)
        return done(fail(m, PLSPM_E_STATE, "plspm_op_inner_weights: launch / copy failed"));
    return done(0);
}

int plspm_op_outer_weights(int32_t device_id, int32_t mode, const double* Xk, const double* z, int64_t N, int32_t k, double* w) {
    g_create_error.clear();
    if (!Xk || !z || !w || k < 1 || k > 1020 || N < 2 || (mode != PLSPM_MODE_A && mode != PLSPM_MODE_B))
        return fail(nullptr, PLSPM_E_ARG, "plspm_op_outer_weights: bad arguments (1 <= k <= 1020, N >= 2)");
    const int P = k + 1;                                            // [X_k | z]
    const int32_t boff[2] = {0, P}, modes[1] = {PLSPM_MODE_A};
    const uint8_t path[1] = {0};
    plspm_model* m = plspm_model_create(P, 1, boff, path, modes, PLSPM_SCHEME_CENTROID, 0, 1, 1.0, device_id);
    if (!m) return PLSPM_E_ARG;
    auto done = [&](int rc) { if (rc) g_create_error = m->error; plspm_model_destroy(m); return rc; };
    std::vector<double> both;                                       // the two host arrays side by side (one upload, one Gram)
    try { both.resize((size_t)N * P); } catch (...) { return done(fail(m, PLSPM_E_STATE, "out of host memory")); }
    for (int64_t i = 0; i < N; ++i) { memcpy(&both[(size_t)i * P], Xk + (size_t)i * k, sizeof(double) * k); both[(size_t)i * P + k] = z[i]; }
    int rc;
    if ((rc = plspm_upload(m, both.data(), N, P, 0, nullptr)) || (rc = dense_moments(m))) return done(rc);
    const size_t kk = (size_t)k * k;
    if ((rc = ensure(m, m->fitout, sizeof(double) * (3 * kk + 1 + k)))) return done(rc);
    double* scratch = (double*)m->fitout.p;
    double* d_w = scratch + 3 * kk + 1;
    hipLaunchKernelGGL(op_outer_kernel, dim3(1), dim3(256), 0, m->stream, (int)mode, (int)k, m->T, (const double*)m->gram.p, (const double*)m->d_shift, scratch, d_w);
    double flag = 0.0;
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(w, d_w, sizeof(double) * k, hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
        hipMemcpyAsync(&flag, scratch + 3 * kk, sizeof(double), hipMemcpyDeviceToHost, m->stream) != hipSuccess || hipStreamSynchronize(m->stream) != hipSuccess)
        return done(fail(m, PLSPM_E_STATE, "plspm_op_outer_weights: launch / copy failed"));
    if (flag == 0.0) return done(fail(m, PLSPM_SINGULAR, "plspm_op_outer_weights: the Mode-B least squares did not converge"));
    return done(0);
}

int plspm_op_outer_weights_nonmetric(int32_t device_id, int32_t mode, const double* Xk, const uint8_t* present, const double* z, int64_t N, int32_t k,
                                     double correction, double* w, double* Y) {
    g_create_error.clear();
    if (!Xk || !z || !w || !Y || k < 1 || k > 1020 || N < 2 || (mode != PLSPM_MODE_A && mode != PLSPM_MODE_B))
        return fail(nullptr, PLSPM_E_ARG, "plspm_op_outer_weights_nonmetric: bad arguments (1 <= k <= 1020, N >= 2)");
    if (mode == PLSPM_MODE_B && present) return fail(nullptr, PLSPM_E_ARG, "plspm_op_outer_weights_nonmetric: Mode B takes no missing values (mode.py:55-56)");
    int rc;
    // Mode B: the least-squares weights of z on the block (minimum norm when rank deficient), as the metric operator computes them
    if (mode == PLSPM_MODE_B && (rc = plspm_op_outer_weights(device_id, PLSPM_MODE_B, Xk, z, N, k, w))) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return fail(nullptr, PLSPM_E_STATE, "plspm_op_outer_weights_nonmetric: no such HIP device");
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, PLSPM_E_STATE, "hipSetDevice failed");
    const size_t nx = (size_t)N * k;
    const size_t bytes = sizeof(double) * (nx + 2 * (size_t)N + k) + (present ? nx : 0);
    void* base = nullptr;
    hipStream_t st = nullptr;
    if (plspm_dmalloc(&base, bytes) != hipSuccess) return fail(nullptr, PLSPM_E_STATE, "plspm_op_outer_weights_nonmetric: out of device memory");
    auto done = [&](int code, const char* why) { if (st) { hipStreamSynchronize(st); plspm_stream_release(st); } plspm_dfree(base); return code ? fail(nullptr, code, why) : 0; };
    if (plspm_stream_acquire(&st) != hipSuccess) return done(PLSPM_E_STATE, "plspm_op_outer_weights_nonmetric: no stream");
    double* d_X = (double*)base; double* d_z = d_X + nx; double* d_Y = d_z + N; double* d_w = d_Y + N;
    unsigned char* d_m = present ? (unsigned char*)(d_w + k) : nullptr;
    bool ok = hipMemcpyAsync(d_X, Xk, sizeof(double) * nx, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(d_z, z, sizeof(double) * N, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok && present) ok = hipMemcpyAsync(d_m, present, nx, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok && mode == PLSPM_MODE_B) ok = hipMemcpyAsync(d_w, w, sizeof(double) * k, hipMemcpyHostToDevice, st) == hipSuccess;
    if (!ok) return done(PLSPM_E_STATE, "plspm_op_outer_weights_nonmetric: upload failed");
    hipLaunchKernelGGL(op_nm_outer_kernel, dim3(1), dim3(1024), 0, st, mode == PLSPM_MODE_B ? 1 : 0, (long)N, (int)k, (const double*)d_X, (const unsigned char*)d_m, (const double*)d_z,
                       correction, d_w, d_Y);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(w, d_w, sizeof(double) * k, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(Y, d_Y, sizeof(double) * N, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return done(PLSPM_E_STATE, "plspm_op_outer_weights_nonmetric: launch / copy failed");
    return done(0, "");
}

}  // extern "C"
