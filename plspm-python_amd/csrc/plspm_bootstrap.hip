// plspm_bootstrap.hip -- host side, part 4: the bootstrap driver (resample -> Gram -> solver per chunk of replicates), its C-ABI entry points,
// the record download and the device summaries.  Kernels: kernels_resample.h, kernels_summary.h.
#include "host_internal.h"

#include "philox.h"
#include "wave_ops.h"
#include "kernels_resample.h"
#include "kernels_summary.h"

// dense [C x C] symmetric moment matrix of every replicate out of the tile-packed one (plspm_bootstrap_moments)
__global__ void __launch_bounds__(256) moments_unpack_kernel(const double* __restrict__ gram, long psize, int T, int C, double* __restrict__ out) {
    const double* g = gram + (long)blockIdx.x * psize;
    double* o = out + (long)blockIdx.x * C * C;
    for (int e = threadIdx.x; e < C * C; e += 256) { const int p = e / C, q = e - p * C; o[e] = g[packed_index(T, p, q)]; }
}

int plspm_detail_bootstrap(plspm_model* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* d_idx, double* rows_out) {
    if (!m || B < 1 || rep_offset < 0 || B > ((int64_t)1 << 30)) return fail(m, PLSPM_E_ARG, "plspm_bootstrap: bad arguments (1 <= B <= 2^30, rep_offset >= 0)");
    if (!m->d_Xa || m->N < 2) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: no data uploaded");
    const long N = m->N;
    const bool lds_hist = (N <= 65535);        // N * 2 bytes of LDS histogram (16-bit counters, <= 128 KB); beyond that a global scratch slice per replicate
    HIPCHK(m, hipSetDevice(m->device));
    const int R = plspm_row_stride(m);
    const long psize = packed_size(m->T);
    const long ent_stride = ((N + 3) & ~3L) + 4;
    // replicates per pass: bound the (row,count) + Gram scratch to ~2 GiB
    // non-metric solvers: dense uint16 histograms for the dense stop-rule pass (LDS-histogram path only)
    // non-metric models on the int8 route with Philox draws (round 3): the dense stop-rule pass reads its row multiplicities from the int8
    // counts the Gram consumed -- no second resample kernel, no (row,count) lists, no uint16 histograms (set_option "nm_counts8" 0: the
    // round-2 path, kept for A/B and for the cases below)
    const int gpath_plan = choose_gram_path(m, B);
    // (explicit index lists of at most 65,535 rows keep the round-3 arrangement -- uint16 histograms beside the lists they need anyway; beyond
    //  one window the int8 counts are the only dense multiplicities there are)
    const bool counts8_plan = gpath_plan == 2 && (!d_idx || !lds_hist) && nm_counts8_possible(m);
    const bool want_dcnt = m->nonmetric && lds_hist && !counts8_plan;
    const long dcnt_stride = ((N + 15) & ~15L);
    const int gpath = gpath_plan;
    m->last_gram_path = gpath;
    // one wave per problem on dense moment matrices (solver_rows_kernel): metric models of at most 64 MVs behind the int8 Gram
    // rows solver: one wave per problem, its small workspace + descriptors in LDS -- eight problems per CU at the headline size
    // (at least four problems per CU; wide inner models, L >~ 20, take the LDS solver, which can move its workspace to global scratch)
    const size_t rows_lds = desc_lds_bytes(m->P, m->L, m->n_eff, (int)m->pred_idx.size()) + (size_t)workspace_small_doubles(m->P, m->L, m->kmax, m->n_chol) * sizeof(double);
    // (round 4: 64 < P <= 128 in the split form -- two threads per MV on either side of a block boundary, four waves and ~30 KB of LDS per problem)
    // (round 5: the quad solver -- solver_quad.h, Mode-A models of 65 .. 128 MVs and at most 16 LVs -- has a workspace of its own, ~52 KB whatever the inner model)
    const bool quad_width = m->tune.solver_quad != 0 && quad_solver_covers<16>(m->P, m->L, m->n_chol, m->kmax, m->boff.data());
    const bool w16_width = m->tune.solver_wave != 0 && (wave16_solver_covers<16>(m->P, m->L, m->n_chol, m->kmax) || (m->n_chol == 0 && wave16_solver_covers<32>(m->P, m->L, 0, m->kmax)));      // (solver_wave16.h: 9 .. 16 LVs, its own ~16 KB workspace)
    const bool rows_width = m->P <= 64 ? (rows_lds <= kMaxLds / 4 || w16_width) : (quad_width || (m->P <= 128 && rows_split_block(m->boff.data(), m->L, 64) > 0 && rows_lds + 4 * 16 * 66 * sizeof(double) <= kMaxLds / 2));
    const bool rows_solver = gpath == 2 && m->tune.solver_rows != 0 && rows_width && !m->n_ind && !m->nonmetric && !m->moments_out;
    // round 6: Scale.NUM / RAW batches on the int8 route as one solver launch on dense moment matrices + a verification pass (plspm_nonmetric.hip
    // run_nonmetric_wave) -- needs the int8 counts the Gram consumed (explicit index lists of at most 65,535 rows keep the per-iteration launches and their
    // uint16 histograms; so does a chunk whose explicit indices made the int8 Gram fall back to the fp64 one)
    const bool nm_wave = gpath == 2 && counts8_plan && !m->moments_out && nm_wave_route_planned(m);
    // the fp64 Gram walks (row,count) lists (explicit indices may fall back to it); so do the stop-rule passes of the non-metric solvers
    const bool need_lists = gpath == 1 || d_idx != nullptr || (m->nonmetric && !counts8_plan);
    const size_t kpad = (size_t)i8_kblocks(N) * 64;
    // (the global-scratch histogram serves the (row,count) lists only: the int8 route on Philox draws never builds them)
    const bool need_ghist = !lds_hist && need_lists;
    const bool cat_one = gpath_plan == 2 && counts8_plan && !m->stage1 && m->tune.nm_cat_one != 0 && m->tune.nm_subset != 0 && m->tune.nm_mfma != 0 && nm_wave_step_planned(m);
    const size_t per_rep = (need_lists ? (size_t)ent_stride * sizeof(int2) : 0) + (size_t)std::max<long>(psize, cov_doubles(m->Pg)) * sizeof(double) + (need_ghist ? (size_t)N * sizeof(unsigned) : 0) +
                           (want_dcnt ? (size_t)dcnt_stride * sizeof(unsigned short) : 0) + (gpath == 2 ? kpad : 0) +
                           (nm_wave ? (size_t)(m->max_iter + 2) * (m->P + m->L + 1) * sizeof(double) + 4 * ((size_t)(2 * m->P + 2 * m->L + 1) + 8) * sizeof(double) : 0) +      // (score maps + verification tables)
                           // (the categorical one-launch form, plspm_nonmetric.hip: a map per step and replicate, digit planes of eight (replicate, step) slots per replicate)
                           (cat_one ? (size_t)(m->max_iter + 2) * (m->P + m->L + 1) * sizeof(double) + 8 * ((size_t)m->L * 2 * 7 * 2 * 64 + 64) : 0);
    // (the one-launch categorical batch lasts as long as its slowest replicate -- the ones that never converge run max_iter + 1 steps in one wave --, so cutting a call
    //  into passes multiplies that tail: 16 GiB of the 288 for it instead of 2)
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(B, (int64_t)(((cat_one ? 16ull : 2ull) << 30) / per_rep)));
    if (gpath == 2 && chunk < B) chunk = std::max<int64_t>(256, chunk & ~(int64_t)255);      // whole 256-replicate tiles per pass
    int rc;
    if (gpath == 2) {
        if ((rc = prepare_zs(m))) return rc;
    }
    if (need_lists) {
        if ((rc = ensure(m, m->ent, (size_t)chunk * ent_stride * sizeof(int2)))) return rc;
        if ((rc = ensure(m, m->nent, (size_t)chunk * sizeof(int)))) return rc;
    }
    if ((rc = ensure(m, m->gram, (size_t)chunk * std::max<long>(psize, (rows_solver || nm_wave) ? cov_doubles(m->Pg) : 0) * sizeof(double)))) return rc;
    if (!rows_out) {
        m->rows_B = 0;
        if ((rc = ensure(m, m->rows, (size_t)B * R * sizeof(double)))) return rc;
        rows_out = (double*)m->rows.p;
    }
    if ((rc = ensure(m, m->status, (size_t)B * sizeof(int)))) return rc;
    if ((rc = ensure(m, m->iters, (size_t)B * sizeof(int)))) return rc;
    const void* err_before = m->err.p;
    if ((rc = ensure(m, m->err, sizeof(int)))) return rc;
    if (m->err.p != err_before) m->err_clean = false;
    if (need_ghist && (rc = ensure(m, m->ghist, (size_t)chunk * N * sizeof(unsigned)))) return rc;
    if (want_dcnt && (rc = ensure(m, m->dcnt, (size_t)chunk * dcnt_stride * sizeof(unsigned short)))) return rc;
    m->dcnt_stride = dcnt_stride; m->dcnt_ready = want_dcnt;
    // The error word (index out of range / multiplicity above 127 / stream-K wait expired) can only be raised by a call that brings explicit
    // indices or runs the persistent Gram (data sets of fewer than 128 rows cannot exceed a multiplicity of 127; they are kept on the clearing
    // side because their launches take the short-N forms that share the word with the index check): a Philox call on the tiled launch neither raises nor needs to clear it (the 4-byte memset is a
    // kernel of its own on the stream: ~8 us with its gaps, 1.5 % of a 5,000-replicate step)
    const bool may_raise = d_idx != nullptr || m->tune.i8_sched != 0 || N < 128;
    if (!m->err_clean || may_raise) HIPCHK(m, hipMemsetAsync(m->err.p, 0, sizeof(int), m->stream));
    m->err_clean = !may_raise;
    double* const gram_buf = (double*)m->gram.p;
    // round 5: all-indicator categorical models on the wave step -- the int8 product writes the uint16 count matrices the step streams itself (no fp64 slots, no
    // scatter pass: kernels_gram_i8.h IND epilogue, kernels_nonmetric.h nmg_kernel<4>); option "nm_direct16" 0: the packed fp64 matrices + nmg_kernel<3>
    const bool want16 = gpath == 2 && m->nonmetric && !m->stage2 && !m->stage1 && !m->moments_out && m->tune.nm_direct16 != 0 && nm_wave_step_planned(m);
    if (want16 && (rc = ensure(m, m->gK16, (size_t)chunk * (m->P + 1) * ((m->P + 1 + 7) & ~7) * sizeof(unsigned short) + 64))) return rc;
    hipEvent_t parked_stop = m->stop_event;
    m->stop_event = nullptr;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = std::min<int64_t>(chunk, B - b0);
        if (b0 + nb >= B) m->stop_event = parked_stop;       // the last chunk's solver launch may signal the caller's event
        bool f64_gram = gpath == 1;
        const void* cd8 = nullptr;
        int cd8_MT = 0;
        bool wrote16 = false;
        if (gpath == 2) {
            bool fallback = false;
            if ((rc = run_gram_i8(m, nb, seed, rep_offset + b0, d_idx ? d_idx + b0 * N : nullptr, gram_buf, rows_solver || nm_wave, &fallback, &cd8, &cd8_MT,
                                  want16 ? (unsigned short*)m->gK16.p : nullptr, &wrote16))) return rc;
            f64_gram = fallback;
        }
        if (!counts8_plan || f64_gram) cd8 = nullptr;                  // (a chunk that fell back has no usable int8 counts)
        if (f64_gram || (m->nonmetric && !cd8)) {                      // (row,count) lists (+ dense uint16 histograms): the same draws as the int8 counts
            if (lds_hist) {
                const size_t hist_bytes = (size_t)((N + 1) / 2) * sizeof(unsigned);
                if ((rc = allow_lds(m, (const void*)resample_kernel, hist_bytes))) return rc;
                ProfScope ps(m, PLSPM_K_RESAMPLE);
                hipLaunchKernelGGL(resample_kernel, dim3((unsigned)nb), dim3(256), hist_bytes, m->stream, (int)N,
                                   d_idx ? d_idx + b0 * N : nullptr, seed, rep_offset + b0, (int2*)m->ent.p, (int*)m->nent.p, ent_stride, (int*)m->err.p,
                                   want_dcnt ? (unsigned short*)m->dcnt.p : (unsigned short*)nullptr, dcnt_stride);
            } else {
                ProfScope ps(m, PLSPM_K_RESAMPLE);
                hipLaunchKernelGGL(resample_global_kernel, dim3((unsigned)nb), dim3(256), 0, m->stream, (int)N, d_idx ? d_idx + b0 * N : nullptr, seed,
                                   rep_offset + b0, (unsigned*)m->ghist.p, (int2*)m->ent.p, (int*)m->nent.p, ent_stride, (int*)m->err.p);
            }
        }
        if (f64_gram) {
            ProfScope ps(m, PLSPM_K_GRAM);
            if ((rc = launch_gram_lists(m, nb, (const int2*)m->ent.p, (const int*)m->nent.p, ent_stride, (double*)m->gram.p))) return rc;
        }
        if (m->moments_out) {                      // plspm_bootstrap_moments (test seam): the replicates' moment matrices, dense, no solver
            const int C = m->Pg + 1;
            hipLaunchKernelGGL(moments_unpack_kernel, dim3((unsigned)nb), dim3(256), 0, m->stream, (const double*)m->gram.p, psize, m->T, C, m->moments_out + b0 * C * C);
            continue;
        }
        SolverOut so{};
        so.row = rows_out + b0 * R; so.row_stride = R; so.status = (int*)m->status.p + b0; so.iters = (int*)m->iters.p + b0;
        if (m->nonmetric && m->stage2) {
            // two-stage HOC estimation per replicate (solver_hoc.h): stage 1 to convergence (no report), stage-2 moments by congruence,
            // stage 2 on the second handle's descriptors with the convergence pass streaming THIS handle's data
            plspm_model* m2 = m->stage2;
            const long psize2 = packed_size(m2->Ts);
            const int2* ent_l = (need_lists && !cd8) ? (const int2*)m->ent.p : nullptr;      // (built above only when the int8 counts are not used)
            const int* nent_l = (need_lists && !cd8) ? (const int*)m->nent.p : nullptr;
            // (threads per problem: 128 unless the handle's "nm_threads" option says otherwise -- each stage reads its own handle's)
            const int t1 = m->tune.nm_threads > 0 ? m->tune.nm_threads : 128, t2 = m2->tune.nm_threads > 0 ? m2->tune.nm_threads : 128;
            if ((rc = run_nonmetric(m, nb, (const double*)m->gram.p, psize, SolverOut{}, ent_l, nent_l, ent_stride, t1, false, cd8, cd8_MT))) return rc;
            if ((rc = run_hoc_moments(m, m2, nb))) return rc;
            rc = run_nonmetric(m2, nb, (const double*)m2->gram.p, psize2, so, ent_l, nent_l, ent_stride, t2, true, cd8, cd8_MT);
            if (rc) return fail(m, rc, "second stage: " + m2->error);
            continue;
        }
        if (m->nonmetric && nm_wave && cd8 && !f64_gram) {
            if ((rc = run_nonmetric_wave(m, nb, so, cd8, cd8_MT))) return rc;
            continue;
        }
        if (m->nonmetric) {
            m->last_nm_wave16 = 0;
            // threads per problem by model width (measured: 60 columns 0.60 / 0.64 / 0.81 ms with 64 / 128 / 256 threads; 300 indicator
            // columns 21.0 / 13.5 / 10.0 ms)
            const int nm_threads = m->tune.nm_threads > 0 ? m->tune.nm_threads : (m->P > 128 ? 256 : (m->P > 64 ? 128 : 64));
            if ((rc = run_nonmetric(m, nb, (const double*)m->gram.p, psize, so, (need_lists && !cd8) ? (const int2*)m->ent.p : nullptr, (need_lists && !cd8) ? (const int*)m->nent.p : nullptr, ent_stride,
                                    nm_threads, true, cd8, cd8_MT, wrote16))) return rc;
            continue;
        }
        if ((rc = launch_batch_solver(m, nb, rows_solver && !f64_gram, so))) return rc;
    }
    HIPCHK(m, hipGetLastError());
    if (rows_out == (double*)m->rows.p) m->rows_B = B;
    return 0;
}
// (m->stop_event: only the last chunk's solver launch may take it -- plspm_detail_bootstrap parks it while earlier chunks run)

static int fetch_segments(plspm_model* m, hipStream_t cs, const double* d_records, int32_t stride, const FetchSeg* segs, int nsegs, int64_t B_total, double* out, int32_t* status,
                          int32_t* iters);

extern "C" {

int plspm_bootstrap_device(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* d_idx, void** d_out, void** d_status,
                           void** d_iters) {
    int rc = plspm_detail_bootstrap(m, B, seed, rep_offset, d_idx, nullptr);
    if (rc) return rc;
    if (d_out) *d_out = m->rows.p;
    if (d_status) *d_status = m->status.p;
    if (d_iters) *d_iters = m->iters.p;
    return 0;
}

int plspm_bootstrap(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* idx, double* out, int32_t* status, int32_t* iters) {
    if (!m || !out || B < 1 || B > ((int64_t)1 << 30)) return fail(m, PLSPM_E_ARG, "plspm_bootstrap: bad arguments");
    if (!m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: no data uploaded");
    HIPCHK(m, hipSetDevice(m->device));
    const int32_t* d_idx = nullptr;
    int rc;
    if (idx) {
        const size_t bytes = (size_t)B * m->N * sizeof(int32_t);
        if ((rc = ensure(m, m->idx, bytes))) return rc;
        if ((rc = plspm_detail_h2d(m, m->idx.p, idx, bytes))) return rc;
        d_idx = (const int32_t*)m->idx.p;
    }
    // The call as SUB-BATCHES (round 5): the records of sub-batch k cross PCIe on a copy stream of the handle's own
    // -- and are unpacked into the caller's buffers -- while the kernels of sub-batch k + 1 run; only the last, smallest sub-batch's download is
    // exposed.  Metric models on Philox draws (explicit index lists and the non-metric iteration make host round trips inside the call as
    // it is).  The records of a replicate do not depend on the sub-batch it travels in (exact integer Gram, per-replicate solver).
    const int RS = plspm_row_stride(m);
    int64_t parts[kBootChunksMax];
    int nparts = 1;
    parts[0] = B;
    // (a call whose launches may raise the device error word -- plspm_detail_bootstrap's may_raise -- clears that word at the start of every
    //  sub-batch, which would wipe the bits of the sub-batch before: such calls run as one batch, one memset, one read)
    const bool may_raise = m->tune.i8_sched != 0 || m->N < 128;
    if (!d_idx && !m->nonmetric && !m->moments_out && !may_raise)
        nparts = plspm_detail_chunk_plan(B, (int64_t)RS * (int64_t)sizeof(double), m->tune.boot_chunks, m->tune.boot_ratio, parts, m->tune.boot_align > 0 ? m->tune.boot_align : plspm_detail_round_units(m));
    if (nparts > 1) {
        if (!m->dl) HIPCHK(m, plspm_stream_acquire(&m->dl));
        for (int k = 0; k < nparts; ++k)
            if (!m->ev_part[k]) HIPCHK(m, hipEventCreateWithFlags(&m->ev_part[k], hipEventDisableTiming));
    }
    m->rows_B = 0;
    if ((rc = ensure(m, m->rows, (size_t)B * RS * sizeof(double)))) return rc;
    double* const rows = (double*)m->rows.p;
    int* h_err = (int*)m->h_flag + 10;                   // (+8: plspm_bootstrap_fetch, +9: run_gram_i8's look at the multiplicity flag of an explicit index list)
    h_err[0] = h_err[1] = 0;
    FetchSeg segs[kBootChunksMax];
    int64_t b0 = 0;
    for (int k = 0; k < nparts; ++k) {
        if ((rc = plspm_detail_bootstrap(m, parts[k], seed, rep_offset + b0, d_idx ? d_idx + b0 * m->N : nullptr, rows + b0 * RS))) return rc;
        if (k == nparts - 1) {
            // ONE error word, read with the last sub-batch: copied behind the last kernel, in front of the event its download waits for
            HIPCHK(m, hipMemcpyAsync(h_err, m->err.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
            if (m->err2.p) HIPCHK(m, hipMemcpyAsync(h_err + 1, m->err2.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));      // (experiments build: counts drawn on a second stream)
        }
        segs[k].b0 = b0; segs[k].nb = parts[k]; segs[k].ready = nullptr;
        if (nparts > 1) { HIPCHK(m, hipEventRecord(m->ev_part[k], m->stream)); segs[k].ready = m->ev_part[k]; }
        b0 += parts[k];
    }
    rc = fetch_segments(m, nparts > 1 ? m->dl : m->stream, rows, RS, segs, nparts, B, out, status, iters);
    if (nparts > 1) HIPCHK(m, hipStreamSynchronize(m->stream));      // (done already: the last download waited for the last kernel; keeps the handle's invariants simple)
    if (rc) return rc;
    m->rows_B = B;
    if (h_err[0] & 4) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: the persistent Gram gave up waiting for a partial tile (device shared with a long-running kernel?); set_option i8_sched 0");
    if (h_err[0] & 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap: resample index outside [0, N)");
    if ((h_err[0] & 2) || h_err[1]) return fail(m, PLSPM_E_LIMIT, "plspm_bootstrap: a resample multiplicity exceeded 127 on the int8 Gram path (set_option gram_path 1)");
    if (h_err[0]) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: the device reported error bits " + std::to_string(h_err[0]));
    return 0;
}

int plspm_bootstrap_fetch(plspm_model_t* m, int64_t first, int64_t count, double* out, int32_t* status, int32_t* iters) {
    if (!m || first < 0 || count < 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_fetch: bad arguments");
    if (!m->rows_B || !m->rows.p) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_fetch: no bootstrap records on this handle (a later call replaced them)");
    if (first + count > m->rows_B) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_fetch: range exceeds the last bootstrap's replicates");
    HIPCHK(m, hipSetDevice(m->device));
    const int RS = plspm_row_stride(m);
    int rc = plspm_detail_fetch_records(m, (const double*)m->rows.p + first * RS, count, RS, out, status, iters);
    if (rc) return rc;
    if (m->sk_epoch && m->err.p) {          // the persistent Gram's bounded wait (its tiles are NaN / status 3 then; say why)
        int* h_err = (int*)m->h_flag + 8;
        HIPCHK(m, hipMemcpyAsync(h_err, m->err.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (*h_err & 4) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_fetch: the persistent Gram gave up waiting for a partial tile (device shared with a long-running kernel?); set_option i8_sched 0");
    }
    return 0;
}

int plspm_bootstrap_store(plspm_model_t* m, const double* records, int64_t B) {
    if (!m || !records || B < 1 || B > ((int64_t)1 << 30)) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_store: bad arguments");
    HIPCHK(m, hipSetDevice(m->device));
    const size_t bytes = (size_t)B * plspm_row_stride(m) * sizeof(double);
    m->rows_B = 0;
    int rc;
    if ((rc = ensure(m, m->rows, bytes))) return rc;
    if ((rc = plspm_detail_h2d(m, m->rows.p, records, bytes))) return rc;
    m->rows_B = B;
    return 0;
}

int plspm_bootstrap_summary(plspm_model_t* m, const void* d_rows, int64_t B, int32_t stride, const double* original, double* summary, int64_t* n_used) {
    if (!m || !original || !summary || B < 1 || B > ((int64_t)1 << 30)) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_summary: bad arguments (1 <= B <= 2^30)");
    const double* rows = (const double*)d_rows;
    if (!rows) {
        // the handle's own records: exactly the replicates of the last plspm_bootstrap(_device) call (the buffer is never shared
        // with another result, and a different B would read stale or foreign memory as records)
        if (!m->rows_B || !m->rows.p) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_summary: no bootstrap result on this handle");
        if (B != m->rows_B) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_summary: B differs from the last bootstrap on this handle");
        rows = (const double*)m->rows.p;
        stride = plspm_row_stride(m);
    }
    return plspm_detail_summary(m, rows, B, stride, original, summary, n_used);
}

}  // extern "C"

// Unpacking of downloaded records (strided pinned staging -> the caller's pageable rows / status / iterations) by a small resident crew:
// one thread needs ~0.3 ms for the 6.3 MB of 5,000 records, more than their four DMA chunks take (plspm_bootstrap: 0.84 ms per call against
// 0.53 on the device; 0.76 with the crew -- what is left is the device -> host copy itself, ~30 GB/s at this size on a copy-only stream or
// behind the kernels, into coherent or non-coherent pinned memory alike).  Three helper threads, started on first use and leaked with the
// process (like the memory cache): asleep on a condition variable between downloads, woken when a download starts -- the first chunk's
// DMA covers the wake-up -- and spinning on a sequence number only while that download lasts, so that handing them a chunk costs no
// system call.
namespace {
struct UnpackCrew {
    static constexpr int kHelpers = 3;
    std::mutex session;                                   // one download at a time uses the crew (handles may live on different threads)
    std::mutex mu;
    std::condition_variable cv;
    bool started = false, broken = false;
    std::atomic<int> active{0};                           // a download is running: helpers spin instead of sleeping
    std::atomic<uint64_t> seq{0};
    std::atomic<int> done{0};
    void (*fn)(void*, int, int) = nullptr;
    void* arg = nullptr;
    void helper(int t) {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return active.load(std::memory_order_acquire) > 0; }); }
            int idle = 0;
            while (active.load(std::memory_order_acquire) > 0) {
                const uint64_t s = seq.load(std::memory_order_acquire);
                if (s != seen) { seen = s; fn(arg, t, kHelpers + 1); done.fetch_add(1, std::memory_order_release); idle = 0; }
                else if (++idle > 4000) std::this_thread::yield();
            }
        }
    }
    bool begin() {
        session.lock();
        if (!started) {
            started = true;
            broken = std::thread::hardware_concurrency() < 8;
            if (!broken) {
                try { for (int t = 1; t <= kHelpers; ++t) std::thread([this, t]() { helper(t); }).detach(); }
                catch (...) { broken = true; }            // (helpers that did start sleep for ever: the crew is never activated)
            }
        }
        if (broken) { session.unlock(); return false; }
        { std::lock_guard<std::mutex> lk(mu); active.store(1, std::memory_order_release); }
        cv.notify_all();
        return true;
    }
    void run(void (*f)(void*, int, int), void* a) {       // f(a, t, T) on the caller (t = 0) and the helpers (t = 1 .. kHelpers); returns when all are done
        fn = f; arg = a;
        done.store(0, std::memory_order_relaxed);
        seq.fetch_add(1, std::memory_order_release);
        f(a, 0, kHelpers + 1);
        while (done.load(std::memory_order_acquire) < kHelpers) std::this_thread::yield();
    }
    void end() { active.store(0, std::memory_order_release); session.unlock(); }
};
UnpackCrew& unpack_crew() { static UnpackCrew* c = new UnpackCrew(); return *c; }      // leaked on purpose (must outlive every static destructor)
}  // namespace

// Record exchange of a group whose handles all live on ONE device (tests and single-GPU runs of the multi-GPU path; real multi-GPU
// groups use ncclAllGather): one launch copies every send buffer into its slot of every receive buffer -- the same G x G block copies the
// peer-to-peer route issues one by one (64 hipMemcpyAsync + 64 event waits at eight handles: 0.35 ms of host time).
struct GatherLocalArgs { const double* send[PLSPM_GATHER_LOCAL_MAX]; double* recv[PLSPM_GATHER_LOCAL_MAX]; };
template <typename T>      // double2 when a record block is a whole number of 16-byte pieces
__global__ void __launch_bounds__(256) gather_local_kernel(GatherLocalArgs a, long n) {
    const T* __restrict__ src = reinterpret_cast<const T*>(a.send[blockIdx.y]);
    T* __restrict__ dst = reinterpret_cast<T*>(a.recv[blockIdx.z]) + (long)blockIdx.y * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
int plspm_detail_gather_local(hipStream_t stream, int n, const double* const* send, double* const* recv, size_t doubles, int ndst) {
    if (n < 1 || n > PLSPM_GATHER_LOCAL_MAX || ndst < 1 || ndst > n) return PLSPM_E_ARG;
    GatherLocalArgs a;
    for (int i = 0; i < n; ++i) { a.send[i] = send[i]; a.recv[i] = recv[i]; }
    for (int i = n; i < PLSPM_GATHER_LOCAL_MAX; ++i) { a.send[i] = nullptr; a.recv[i] = nullptr; }
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((doubles + 2047) / 2048, 512));
    if (doubles % 2 == 0) hipLaunchKernelGGL(gather_local_kernel<double2>, dim3(gx, (unsigned)n, (unsigned)ndst), dim3(256), 0, stream, a, (long)(doubles / 2));
    else hipLaunchKernelGGL(gather_local_kernel<double>, dim3(gx, (unsigned)n, (unsigned)ndst), dim3(256), 0, stream, a, (long)doubles);
    return hipGetLastError() == hipSuccess ? 0 : PLSPM_E_STATE;
}

// Host copy of device records, segment by segment: segment k = records [b0, b0 + nb) of `d_records`, downloaded on `cs` once its `ready` event
// (may be null) has fired on the device side -- the copy stream waits, not the host.  Pieces of at most half the staging area -- and of at
// most a quarter of all records, so that the host's unpacking of one piece runs beside the DMA of the next even when everything would fit a
// single piece (5,000 x 158 records = 6.3 MB).
static int fetch_segments(plspm_model* m, hipStream_t cs, const double* d_records, int32_t stride, const FetchSeg* segs, int nsegs, int64_t B_total, double* out, int32_t* status,
                          int32_t* iters) {
    const int R = stride - 2;
    int rc = pin_ready(m);
    if (rc) return rc;
    int64_t per = std::max<int64_t>(1, (int64_t)(kPinHalf / ((size_t)stride * sizeof(double))));
    per = std::min<int64_t>(per, std::max<int64_t>(256, (B_total + 3) / 4));
    struct Job { const double* rec; int64_t b0, nb; int32_t stride, R; double* out; int32_t* status; int32_t* iters; };
    auto unpack_part = [](void* a, int t, int T) {
        const Job& j = *(const Job*)a;
        const int64_t lo = j.nb * t / T, hi = j.nb * (t + 1) / T;
        const double* rec = j.rec + lo * j.stride;
        for (int64_t b = lo; b < hi; ++b, rec += j.stride) {
            if (j.out) memcpy(j.out + (j.b0 + b) * j.R, rec, (size_t)j.R * sizeof(double));
            if (j.status) j.status[j.b0 + b] = (rec[j.R] == rec[j.R]) ? (int32_t)rec[j.R] : -1;      // NaN marks the padding records of a ragged shard
            if (j.iters) j.iters[j.b0 + b] = (rec[j.R + 1] == rec[j.R + 1]) ? (int32_t)rec[j.R + 1] : 0;
        }
    };
    // the crew from a megabyte of records on (smaller downloads are done before a helper has woken up)
    struct Session { bool on = false; ~Session() { if (on) unpack_crew().end(); } } crew;
    if ((size_t)B_total * stride * sizeof(double) >= ((size_t)1 << 20)) crew.on = unpack_crew().begin();
    auto unpack = [&](int h, int64_t b0, int64_t nb) {
        Job j{(const double*)((const char*)m->h_pin + h * kPinHalf), b0, nb, stride, R, out, status, iters};
        if (crew.on) unpack_crew().run(unpack_part, &j); else unpack_part(&j, 0, 1);
    };
    int64_t prev_b0 = 0, prev_nb = 0;
    int k = 0;
    for (int sg = 0; sg < nsegs; ++sg) {
        if (segs[sg].ready) HIPCHK(m, hipStreamWaitEvent(cs, segs[sg].ready, 0));
        for (int64_t b0 = segs[sg].b0; b0 < segs[sg].b0 + segs[sg].nb; b0 += per, ++k) {
            const int h = k & 1;
            const int64_t nb = std::min<int64_t>(per, segs[sg].b0 + segs[sg].nb - b0);
            // (half h was unpacked before the piece before this one was waited for: free)
            HIPCHK(m, hipMemcpyAsync((char*)m->h_pin + h * kPinHalf, d_records + b0 * stride, (size_t)nb * stride * sizeof(double), hipMemcpyDeviceToHost, cs));
            HIPCHK(m, hipEventRecord(m->ev_pin[h], cs));
            if (prev_nb) { HIPCHK(m, hipEventSynchronize(m->ev_pin[h ^ 1])); unpack(h ^ 1, prev_b0, prev_nb); }
            prev_b0 = b0; prev_nb = nb;
        }
    }
    if (prev_nb) { HIPCHK(m, hipEventSynchronize(m->ev_pin[(k - 1) & 1])); unpack((k - 1) & 1, prev_b0, prev_nb); }
    return 0;
}

int plspm_detail_fetch_records(plspm_model* m, const double* d_records, int64_t B, int32_t stride, double* out, int32_t* status, int32_t* iters) {
    const FetchSeg seg{0, B, nullptr};
    return fetch_segments(m, m->stream, d_records, stride, &seg, 1, B, out, status, iters);
}

int plspm_detail_summary(plspm_model* m, const double* rows, int64_t B, int32_t stride, const double* original, double* summary, int64_t* n_used) {
    HIPCHK(m, hipSetDevice(m->device));
    const int R = plspm_row_width(m);
    if (stride < R + 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_summary: stride must cover the status column");
    const int npad = (int)((B + 1) & ~(int64_t)1);                   // values per column (no padding needed: nothing is sorted)
    const bool in_lds = (size_t)npad * sizeof(double) <= (size_t)128 * 1024;
    int rc;
    if ((rc = pin_ready(m))) return rc;
    if ((size_t)R * 7 * sizeof(double) + 64 > m->h_pin_cap) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_summary: record too wide for the staging area");
    if (!in_lds && (rc = ensure(m, m->sum_buf, (size_t)R * npad * sizeof(double)))) return rc;
    // [original R | summary 6R | n_used] in the handle's pinned staging area, read and written by the kernel itself (R + 6R + 1 words
    // across the host link): no copy engine operation in front of or behind the kernel, whose scheduling gaps cost more than the bytes
    double* h_io = (double*)m->h_pin;
    memcpy(h_io, original, sizeof(double) * R);
    // the records column-major first (values + status: R + 1 columns of B, one small tiled transpose): read in place, every value of a
    // column costs the summary workgroup a 128-byte line of its own (2 x B L1 fills per column were 40 % of the kernel)
    const long cols_ld = (long)((B + 63) & ~(int64_t)63);
    if ((rc = ensure(m, m->cols, (size_t)(R + 1) * cols_ld * sizeof(double)))) return rc;
    const double* cols = (const double*)m->cols.p;
    hipLaunchKernelGGL(records_transpose_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)((R + 1 + 63) / 64)), dim3(256), 0, m->stream, rows, (long)B, (int)stride, R + 1, (double*)m->cols.p,
                       cols_ld);
    double* h_out = h_io + R;
    int* h_used = (int*)(h_out + (size_t)R * 6);
    if (in_lds) {
        const size_t lds = (size_t)npad * sizeof(double);
        if ((rc = allow_lds(m, (const void*)summary_kernel<true>, lds))) return rc;
        hipLaunchKernelGGL((summary_kernel<true>), dim3(R), dim3(SUM_NT), lds, m->stream, cols, cols_ld, (long)B, R, (const double*)h_io, (double*)nullptr, npad, h_out, h_used);
    } else {
        hipLaunchKernelGGL((summary_kernel<false>), dim3(R), dim3(SUM_NT), 0, m->stream, cols, cols_ld, (long)B, R, (const double*)h_io, (double*)m->sum_buf.p, npad, h_out, h_used);
    }
    HIPCHK(m, hipGetLastError());
    HIPCHK(m, hipStreamSynchronize(m->stream));
    memcpy(summary, h_io + R, sizeof(double) * R * 6);
    if (n_used) *n_used = *(const int*)(h_io + R + (size_t)R * 6);
#ifdef PLSPM_DEBUG_MARKS
    {
        long long h[16];
        HIPCHK(m, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_summary_marks), sizeof(h)));
        fprintf(stderr, "[plspm summary clocks] compaction %lld  mean+var %lld  select %lld  successors %lld  total %lld\n", h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[4] - h[0]);
    }
#endif
    return 0;
}

extern "C" {

int plspm_bootstrap_moments(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* idx, double* out) {
    if (!m || !out || B < 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_moments: bad arguments");
    if (!m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_moments: no data uploaded");
    if (m->stage1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap_moments: not on an attached second stage");
    HIPCHK(m, hipSetDevice(m->device));
    const int32_t* d_idx = nullptr;
    int rc;
    if (idx) {
        const size_t bytes = (size_t)B * m->N * sizeof(int32_t);
        if ((rc = ensure(m, m->idx, bytes))) return rc;
        if ((rc = plspm_detail_h2d(m, m->idx.p, idx, bytes))) return rc;
        d_idx = (const int32_t*)m->idx.p;
    }
    const size_t C = (size_t)m->Pg + 1, bytes = (size_t)B * C * C * sizeof(double);
    double* d_out = nullptr;
    HIPCHK(m, plspm_dmalloc((void**)&d_out, bytes));
    m->moments_out = d_out;
    rc = plspm_detail_bootstrap(m, B, seed, rep_offset, d_idx, nullptr);
    m->moments_out = nullptr;
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, m->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
        if (e != hipSuccess) rc = fail(m, -(int)e, std::string("plspm_bootstrap_moments: ") + hipGetErrorString(e));
    } else hipStreamSynchronize(m->stream);
    plspm_dfree(d_out);
    m->rows_B = 0;
    return rc;
}

}  // extern "C"
