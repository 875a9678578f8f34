// plspm_nonmetric.hip -- host side, part 2b: the non-metric iteration (Scale.NUM / RAW, ORD / NOM, incomplete rows, HOC second stages):
// prepare -> (step, stop-rule pass)* -> finish, the host reading one counter per iteration.  Kernels: kernels_nonmetric.h.
#include "host_internal.h"

#include "wave_ops.h"
#include "device_exec.h"
#include "kernels_nonmetric.h"
#include "kernels_nmw.h"
#include "kernels_nmp.h"

// Non-metric solve of `nproblems` problems whose packed scatter matrices are at Mp: prepare -> (step, convergence pass)* ->
// finish.  The host only reads one counter per iteration (how many problems are still active).
// doubles of per-problem solver state of a non-metric handle (NmState head + what its solver keeps behind it)
size_t nm_state_doubles_of(const plspm_model* m) {
    return m->categorical ? (size_t)nmg_state_doubles(m->P, m->Pm, m->L, m->cmax, m->kmv)
                          : m->nmx_K > 0 ? (size_t)nmx_state_doubles(m->P, m->L, m->n_chol, m->nmx_K) : (size_t)nm_state_doubles(m->P, m->L, m->n_chol);
}

// LDS footprint of the dense stop-rule pass (nm_conv_dense_kernel) for this handle: the coefficient tile of 64 replicates whole, or one LV
// block at a time; 0 when neither fits or the option forbids the pass
size_t nm_dense_lds(const plspm_model* m, bool* whole, int* kb_out) {
    const plspm_model* src = m->stage1 ? m->stage1 : m;
    const int table_rows = 2 * src->P + 2 * m->L + 1;
    const size_t dense_lds = (size_t)table_rows * 64 * sizeof(double);
    const std::vector<int>& conv_blocks = m->stage1 ? m->lv_cols : m->boff;
    int kb = 1;
    for (int l = 0; l < m->L; ++l) kb = std::max(kb, conv_blocks[l + 1] - conv_blocks[l]);
    const bool w = dense_lds <= kMaxLds && m->tune.conv_pass != 2;            // (option conv_pass = 2 forces the blocked variant: tests)
    const size_t use = w ? dense_lds : (size_t)(2 * kb + 2) * 64 * sizeof(double);
    if (whole) *whole = w;
    if (kb_out) *kb_out = kb;
    return (use <= kMaxLds && m->tune.conv_pass != 1) ? use : 0;
}

// round 5: the iteration as one WAVE per problem (kernels_nmw.h nmw_step_kernel) for all-indicator models of at most 65,535 rows whose blocks are all Mode A,
// of at most 64 MVs with at most 16 categories each (nmw::CMAX_MAX), 8 LVs and 511 indicator columns
bool nm_wave_step_planned(const plspm_model* m) {
    if (!m->categorical || !m->cat_pure || m->nmx_K > 0 || m->N > 65535 || m->tune.nm_k16 == 0 || m->tune.nm_wave == 0) return false;
    for (int l = 0; l < m->L; ++l) if (m->mode[l] != PLSPM_MODE_A) return false;
    const size_t wave_lds = (size_t)nmw::lds_doubles(m->P, m->Pm, m->L, m->kmax) * sizeof(double);
    return m->Pm <= 64 && m->L <= nmw::LMAX_MAX && m->cmax <= nmw::CMAX_MAX && m->P + 1 <= 512 && wave_lds <= kMaxLds;
}

// cd8 / cd8_MT: the int8 row multiplicities of THESE problems (the counts the digit-plane Gram consumed; bootstrap only), or null
// counts16_ready: the upper triangles of the problems' uint16 count matrices are in m->gK16 already (run_gram_i8 wrote them: no packed matrices at Mp)
int run_nonmetric(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so_in, const int2* ent, const int* nent,
                         long ent_stride, int threads, bool finish, const void* cd8, int cd8_MT, bool counts16_ready) {
    SolverOut so = so_in;
    const int P = m->P, L = m->L;
    plspm_model* src = m->stage1 ? m->stage1 : m;                // an attached second stage streams its first stage's data (solver_hoc.h)
    const long N = src->N;
    const bool cat = m->categorical != 0, nmx = m->nmx_K > 0;
    int rc;
    const size_t s_bytes = (size_t)cov_doubles(P) * sizeof(double);
    const size_t st_doubles = nm_state_doubles_of(m);
    // bootstrap: dense stop-rule pass (nm_conv_dense_kernel) when the replicates' uint16 histograms are at hand and the coefficient
    // tile of 64 replicates fits LDS; otherwise (and for a single fit) the gathering pass
    const long ntiles16 = (N + 15) / 16;
    const int table_rows = 2 * src->P + 2 * L + 1;
    // coefficient tile of 64 replicates: whole in LDS when it fits, else one LV block at a time (kb = widest block of the map the pass uses)
    bool dense_whole = false;
    int kb = 1;
    const size_t dense_use_lds = nm_dense_lds(m, &dense_whole, &kb);
    // the replicates' row multiplicities: the int8 counts of the digit-plane Gram (round 3) or the uint16 histograms of resample_kernel
    const bool counts8 = cd8 != nullptr;
    const bool dense = dense_use_lds != 0 && (counts8 || (ent && src->dcnt_ready));
    if (counts8 && !dense) return fail(m, PLSPM_E_STATE, "non-metric bootstrap: the dense stop-rule pass does not fit and no (row,count) lists were built");
    int nparts = dense ? (int)ntiles16 : (int)std::max<long>(1, std::min<long>(nproblems == 1 ? 1024 : 8, (N + 1023) / 1024));
    const int ngroups = (int)((nproblems + 63) / 64);
    if (dense) {
        if ((rc = ensure(m, src->Xt, (size_t)ntiles16 * 16 * src->PA * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->ctable, (size_t)ngroups * table_rows * 64 * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->nmlist, 2 * ((size_t)nproblems + 1) * sizeof(int)))) return rc;      // [count | live problems] + [count | those that ask for the pass over all rows]
        const void* ck = counts8 ? (dense_whole ? (const void*)nm_conv_dense_kernel<16, 8, false, true> : (const void*)nm_conv_dense_kernel<16, 8, true, true>)
                                 : (dense_whole ? (const void*)nm_conv_dense_kernel<16, 8, false, false> : (const void*)nm_conv_dense_kernel<16, 8, true, false>);
        if ((rc = allow_lds(m, ck, dense_use_lds))) return rc;
        if (!src->Xt_valid) {
            hipLaunchKernelGGL(tile_transpose_kernel, dim3((unsigned)ntiles16), dim3(256), 0, m->stream, (const double*)src->d_Xa, N, src->PA, (double*)src->Xt.p);
            src->Xt_valid = true;
        }
    }
    // all-indicator categorical data on the dense pass with the Gram's int8 counts: the pass on category codes (kernels_nonmetric.h
    // nm_conv_codes_kernel; one table of 16 codes per (row tile, MV), built once per upload -- a second HOC stage streams its first
    // stage's rows under its own blocks and keeps its own table)
    const int* codes_base = m->stage1 ? m->d_mv_base2 : m->d_mv_base;
    const int* codes_lmv = m->stage1 ? m->d_lmv2_off : m->d_lmv_off;
    const size_t codes_lds = (size_t)(2 * (kb + 1) + 2) * 64 * sizeof(double);      // one block's coefficients + the zero slot + the two constants
    const bool use_codes = dense && counts8 && src->categorical && src->cat_pure && (m->stage1 || cat) && codes_base && codes_lmv && !nmx && m->tune.nm_codes != 0 && kb < 65535 &&
                           codes_lds <= kMaxLds;
    if (use_codes) {
        if ((rc = allow_lds(m, (const void*)nm_conv_codes_kernel<8>, codes_lds))) return rc;
        if (!m->codes_valid) {
            if ((rc = ensure(m, m->codes, (size_t)ntiles16 * src->Pm * 16 * sizeof(unsigned short)))) return rc;
            const long total = ntiles16 * src->Pm * 16;
            hipLaunchKernelGGL(cat_codes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->stream, (const double*)src->d_Xa, N, src->PA, src->Pm, (const int*)src->d_mv_off,
                               codes_base, kb, ntiles16, (unsigned short*)m->codes.p);
            m->codes_valid = true;
        }
    }
    // round 5: that pass as an exact int8 matrix product (kernels_nmp.h) -- indicator bytes x digit planes of the score maps, blocks of at most 64 columns
    // (one k-step of the instruction).  Row chunks of `tpc` tiles: enough waves to fill the device at the batch's first passes, whole tiles of work each.
    const bool use_mfma = use_codes && kb <= 128 && m->tune.nm_mfma != 0;
    const int KS = kb > 64 ? 2 : 1;                              // k-steps of a block (64 columns per instruction)
    const long ng16 = (nproblems + 15) / 16;
    int tpc = 0;
    if (use_mfma) {
        const long waves = m->tune.conv_gy > 0 ? (long)m->tune.conv_gy * 256 : 8192;      // (option conv_gy n: n x 256 waves aimed at; 4,096 / 8,192 / 16,384 measured 4.45 / 3.90 / 3.88 ms of passes per 5,000-replicate step)
        const long want = std::max<long>(1, std::min<long>((waves + ng16 - 1) / ng16, (ntiles16 + 7) / 8));
        tpc = (int)((ntiles16 + want - 1) / want);
        nparts = (int)((ntiles16 + tpc - 1) / tpc);
        if (!m->ind8_valid) {
            if ((rc = ensure(m, m->ind8, (size_t)L * ntiles16 * KS * 64 * sizeof(uint4)))) return rc;
            const long total = (long)L * ntiles16 * KS * 64;
            hipLaunchKernelGGL(nmp::ind8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->stream, (const unsigned short*)m->codes.p, ntiles16, src->Pm, L, KS, codes_lmv,
                               (uint4*)m->ind8.p);
            m->ind8_valid = true;
        }
        if ((rc = ensure(m, m->tab8, (size_t)ng16 * L * 2 * nmp::S * KS * 64 * sizeof(uint4)))) return rc;
        if ((rc = ensure(m, m->scl8, (size_t)ng16 * L * 2 * 16 * sizeof(double2)))) return rc;
    }
    m->last_nm_codes = use_codes ? 1 : 0;
    m->last_nm_problems = nproblems;
    m->last_nm_exact = 0;
    m->last_nm_mfma = use_mfma ? 1 : 0;
    if (cat && (rc = ensure(m, m->gSm, (size_t)nproblems * cov_doubles(m->Pm) * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->nmstate, (size_t)nproblems * st_doubles * sizeof(double)))) return rc;
    // all-indicator categorical models of at most 65,535 rows: a uint16 copy of every problem's count matrix for the streaming product of the step
    const int ld16 = (P + 1 + 7) & ~7;                           // (whole 16-byte groups: the wave step loads eight counts per lane and row)
    const bool k16 = cat && m->cat_pure && N <= 65535 && m->tune.nm_k16 != 0;
    const size_t wave_lds = (size_t)nmw::lds_doubles(P, m->Pm, L, m->kmax) * sizeof(double);
    const bool wave_step = k16 && nm_wave_step_planned(m);
    // round 6: the wave step stops on its own upper bound and needs the pass only to hear "go on" -- a lower bound from the first row chunks says that as surely as the
    // exact sum (kernels_nmw.h); what it leaves open gets the full pass in a launch of its own.  The int8-product pass of all-indicator models only.
    // (launch by launch, data sets of a few hundred rows: a pass is a few row chunks and a launch floor either way -- nothing to save, one more list to file; the
    //  one-launch form below has no pass per step and takes them as well)
    const bool bound_ok = use_mfma && !m->stage1 && m->tune.nm_subset != 0 && wave_step;
    const bool one_launch = bound_ok && counts8 && cd8 && m->tune.nm_cat_one != 0 && nproblems <= 0x7fffffffL;
    const bool sub_pass = bound_ok && (N >= 1024 || one_launch);
    const int nsub = sub_pass ? std::max(1, m->tune.nm_subset) : 0;      // the safety factor of the rows a problem asks for (kernels_nmw.h); 0: every pass over all rows
    if (counts16_ready && !wave_step) return fail(m, PLSPM_E_STATE, "non-metric solver: uint16 counts without the wave step");
    m->last_nm_wave = wave_step ? 1 : 0;
    m->last_nm_direct16 = counts16_ready ? 1 : 0;
    // the fp64 square of every problem (730 KB at 300 indicator columns): not for the wave step, which reads the uint16 counts only
    if (!wave_step && (rc = ensure(m, m->gS, (size_t)nproblems * s_bytes))) return rc;
    if (k16 && (rc = ensure(m, m->gK16, (size_t)nproblems * (P + 1) * ld16 * sizeof(unsigned short) + 64))) return rc;      // (+ 64: the six-column lanes of the last row read a dword past it)
    if ((rc = ensure(m, m->nmpartial, (size_t)nproblems * nparts * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->nmactive, sizeof(int)))) return rc;
    size_t lds = (size_t)workspace_small_doubles(cat ? m->Pm : P, L, m->kmax, m->n_chol) * sizeof(double) + desc_lds_bytes(P, L, m->n_eff, (int)m->pred_idx.size());
    if (cat) lds += (size_t)workspace_small_doubles(m->Pm, L, m->kmax, 0) * sizeof(double);
    if (lds > kMaxLds) return fail(m, PLSPM_E_LIMIT, "non-metric solver: workspace exceeds LDS");
    // categorical problems: the small arrays of the iteration in LDS when they fit beside the workspaces (kernels_nonmetric.h nmg_kernel)
    const size_t cat_fast_bytes = cat ? (size_t)((nmg_fast_doubles(P, m->Pm, L, m->cmax, m->kmv) + 1) & ~1L) * sizeof(double) : 0;
    const int cat_fast = (cat && m->tune.nm_fast_lds != 0 && lds + cat_fast_bytes <= kMaxLds) ? 1 : 0;
    if (cat_fast) lds += cat_fast_bytes;
    if (cat) {
        if ((rc = allow_lds(m, (const void*)nmg_kernel<0>, lds)) || (rc = allow_lds(m, (const void*)nmg_kernel<1>, lds)) || (rc = allow_lds(m, (const void*)nmg_kernel<2>, lds)) ||
            (rc = allow_lds(m, (const void*)nmg_kernel<3>, lds)) || (rc = allow_lds(m, (const void*)nmg_kernel<4>, lds)))
            return rc;
    } else if (nmx) {
        if ((rc = allow_lds(m, (const void*)nmx_kernel<0>, lds)) || (rc = allow_lds(m, (const void*)nmx_kernel<1>, lds)) || (rc = allow_lds(m, (const void*)nmx_kernel<2>, lds)))
            return rc;
    } else if ((rc = allow_lds(m, (const void*)nm_kernel<0>, lds)) || (rc = allow_lds(m, (const void*)nm_kernel<1>, lds)) || (rc = allow_lds(m, (const void*)nm_kernel<2>, lds)))
        return rc;
    const size_t conv_lds = ((size_t)SCORE_ROWS * (src->PA + 1) + 2 * (size_t)src->P + 2 * (size_t)L + SCORE_ROWS + 256) * sizeof(double) + (size_t)(L + 2) * sizeof(int);
    const long ps_stride = 8 + 4L * src->P + 2L * L;
    if (m->stage1 && (rc = ensure(m, m->pseudo, (size_t)nproblems * ps_stride * sizeof(double)))) return rc;
    if ((rc = allow_lds(m, (const void*)nm_conv_kernel, conv_lds))) return rc;
    const ModelDesc md = make_desc(m);
    CatDesc cd{};
    ModelDesc mdm = md;
    if (cat) {
        cd.Pm = m->Pm; cd.cmax = m->cmax; cd.kmv = m->kmv; cd.mv_off = m->d_mv_off; cd.mv_kind = m->d_mv_kind; cd.lmv_off = m->d_lmv_off;
        mdm.P = m->Pm; mdm.boff = m->d_lmv_off; mdm.lvof = m->d_mv_lv; mdm.chol_off = m->d_no_chol; mdm.n_chol = 0;      // shift: zeros (upload)
    }
    double* gS = (double*)m->gS.p;
    double* gSm = (double*)m->gSm.p;
    double* gst = (double*)m->nmstate.p;
    double* part = (double*)m->nmpartial.p;
    int* nact = (int*)m->nmactive.p;
    const dim3 grid((unsigned)nproblems);
    const int fuse = finish ? 1 : 0;           // the finish of a problem runs inside the step launch that decides its stop
#ifdef PLSPM_DEBUG_MARKS
    long long* d_nm_marks = nullptr;
    if (cat) { HIPCHK(m, plspm_dmalloc((void**)&d_nm_marks, 32 * sizeof(long long))); so.marks = d_nm_marks; }
#endif
    // round 5 (second half): from the second step on, a launch of the dense route covers the problems that were still iterating after the previous step --
    // the list the previous stop-rule pass built and the count the host has just read -- instead of every problem of the batch: the late iterations of
    // a batch are a handful of stragglers (up to max_iter + 1 trips) among thousands of problems whose workgroups did nothing but find their flag cleared
    // (HOC on ordinal items: ~90 of ~100 trips per stage; 31 us per step launch, 79 us per compose launch at 5,000 problems)
    dim3 lgrid = grid;
    const int* live = nullptr;
    auto launch = [&](int mode_op) {
        ProfScope ps(m, PLSPM_K_SOLVER);
        if (cat) {
            auto k = mode_op == 0 ? nmg_kernel<0> : mode_op == 1 ? nmg_kernel<1> : nmg_kernel<2>;
            hipLaunchKernelGGL(k, lgrid, dim3(threads), lds, m->stream, md, cd, mdm, Mp, mp_stride, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact, fuse, cat_fast,
                               k16 ? (unsigned short*)m->gK16.p : (unsigned short*)nullptr, ld16, live);
        } else if (nmx) {
            auto k = mode_op == 0 ? nmx_kernel<0> : mode_op == 1 ? nmx_kernel<1> : nmx_kernel<2>;
            const MissDesc xd{m->nmx_raw, m->nmx_K, m->d_Xk, m->d_Mk};
            hipLaunchKernelGGL(k, lgrid, dim3(threads), lds, m->stream, md, xd, (const int*)m->d_rowid, Mp, mp_stride, so, gS, gst, (long)st_doubles, (const double*)part, nparts,
                               nact, ent, nent, ent_stride, fuse, live);
        } else {
            auto k = mode_op == 0 ? nm_kernel<0> : mode_op == 1 ? nm_kernel<1> : nm_kernel<2>;
            hipLaunchKernelGGL(k, lgrid, dim3(threads), lds, m->stream, md, Mp, mp_stride, so, gS, gst, (const double*)part, nparts, nact, fuse, live);
        }
    };
    // (dense stop-rule pass: the list kernel of the pass counts the live problems anyway and writes the count to the pinned flag itself -- no
    //  counter to clear, no copy operation: two tiny launches and their gaps less per iteration, 35 of ~590 us at three iterations)
    const bool flag_from_list = dense && !m->stage1;
    // (instantiations: LMAX 2 / 4 / 6 / 8 LVs x at most 8 categories per MV -- two waves per SIMD -- or at most 16 -- ten-point items; one wave per SIMD, 512 registers)
    using nmw::nmw_step_kernel;
    // (round 6: six columns per lane where they cover the model -- at most 383 aug columns of items with at most eight categories: 300 columns keep 51 lanes busy
    //  instead of 38; option nm_cpl 8: eight per lane as before)
    // (the finish files an MV's columns from at most three neighbouring lanes: 13 categories at six columns per lane)
    // (not for 7 / 8 LVs with items of 11 ... 13 categories: nmw_step_kernel<8, 16, false, false, 6> -- the launch-by-launch form without the step's own bound, i.e. every
    //  single FIT of such a model -- returns NaN inner weights under the PATH scheme and faults on the fit's one-problem buffers, while the same source with SUB, with eight columns
    //  per lane or with LMAX 6 is right (found by the large categorical fuzz, tests/fuzz_cases.make_cat_big_case seeds 34 / 124 / 133 / ...; centroid and factorial runs of the same
    //  binary are right too: DESIGN 6).  That class keeps eight columns per lane in every form.)
    const bool cpl6 = m->cmax <= 13 && P + 1 <= 6 * 64 && m->tune.nm_cpl != 8 && (m->tune.nm_cpl == 6 || !(m->cmax > 10 && L > 6));      // (nm_cpl 6: six wherever the layout allows -- probes)
    // (round 6, last: items of nine or ten categories -- the reference's own mobi data -- on an instantiation of their own: its register arrays leave room for TWO waves
    //  per SIMD like the eight-category form, where the sixteen-category one runs alone; six columns per lane only)
    const bool c10 = m->cmax > 8 && m->cmax <= 10 && cpl6 && m->tune.nm_c10 != 0;
#define NMW_PICK(SUBV, ONEV)                                                                                                                                            \
    (c10 ? (L <= 2 ? nmw_step_kernel<2, 10, SUBV, ONEV, 6> : L <= 4 ? nmw_step_kernel<4, 10, SUBV, ONEV, 6> : L <= 6 ? nmw_step_kernel<6, 10, SUBV, ONEV, 6> : nmw_step_kernel<8, 10, SUBV, ONEV, 6>) : \
     m->cmax <= 8 ? (cpl6 ? (L <= 2 ? nmw_step_kernel<2, 8, SUBV, ONEV, 6> : L <= 4 ? nmw_step_kernel<4, 8, SUBV, ONEV, 6> : L <= 6 ? nmw_step_kernel<6, 8, SUBV, ONEV, 6> : nmw_step_kernel<8, 8, SUBV, ONEV, 6>) \
                          : (L <= 2 ? nmw_step_kernel<2, 8, SUBV, ONEV, 8> : L <= 4 ? nmw_step_kernel<4, 8, SUBV, ONEV, 8> : L <= 6 ? nmw_step_kernel<6, 8, SUBV, ONEV, 8> : nmw_step_kernel<8, 8, SUBV, ONEV, 8>)) \
                  : (cpl6 ? (L <= 2 ? nmw_step_kernel<2, 16, SUBV, ONEV, 6> : L <= 4 ? nmw_step_kernel<4, 16, SUBV, ONEV, 6> : L <= 6 ? nmw_step_kernel<6, 16, SUBV, ONEV, 6> : nmw_step_kernel<8, 16, SUBV, ONEV, 6>) \
                          : (L <= 2 ? nmw_step_kernel<2, 16, SUBV, ONEV, 8> : L <= 4 ? nmw_step_kernel<4, 16, SUBV, ONEV, 8> : L <= 6 ? nmw_step_kernel<6, 16, SUBV, ONEV, 8> : nmw_step_kernel<8, 16, SUBV, ONEV, 8>)))
    auto wave_kernel = sub_pass ? NMW_PICK(true, false) : NMW_PICK(false, false);
    if (wave_step && (rc = allow_lds(m, (const void*)wave_kernel, wave_lds))) return rc;
    // ---- round 6: the whole batch in ONE solver launch + verification (kernels_nmw.h ONE; the categorical counterpart of run_nonmetric_wave) --------------------------------
    // All-indicator, all-Mode-A models on the wave step with the int8 stop-rule product, not a stage of a HOC pair.  The solver iterates on its own upper bound and
    // leaves every step's score map behind; the verification evaluates the criterion of every step a replicate continued behind on the row chunks that step asks for
    // (a lower bound), the exact pass takes what that leaves open, a replicate whose exact criterion was below the tolerance is replayed with the reference's stop.
    // (the first stage of a HOC pair too -- nothing to finish there: the final state is what the second stage's moments are composed from)
    m->last_nm_one = one_launch ? 1 : 0;
    if (one_launch) {
        constexpr int JR = 8;                                // steps verified per round (six to nine iterations is the rule: one round, one host read-back)
        const long capV = nproblems * JR, ng16V = (capV + 15) / 16;
        const long cstride = (long)(m->max_iter + 2) * P, kstride = (long)(m->max_iter + 2) * (L + 1);
        if ((rc = ensure(m, m->nmw_maps, (size_t)nproblems * (cstride + kstride) * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->nmw_ints, (size_t)(3 * nproblems + 5 * capV + 16) * sizeof(int)))) return rc;
        if ((rc = ensure(m, m->nmw_vsum, (size_t)capV * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->tab8, (size_t)ng16V * L * 2 * nmp::S * KS * 64 * sizeof(uint4)))) return rc;
        if ((rc = ensure(m, m->scl8, (size_t)ng16V * L * 2 * 16 * sizeof(double2)))) return rc;
        double* cmaps = (double*)m->nmw_maps.p;
        double* kmaps = cmaps + nproblems * cstride;
        int* ip = (int*)m->nmw_ints.p;
        int* steps = ip; ip += nproblems;
        int* force = ip; ip += nproblems;
        int* fixlist = ip; ip += nproblems;
        int* vb = ip; ip += capV;
        int* vj = ip; ip += capV;
        int* fb = ip; ip += capV;
        int* fj = ip; ip += capV;
        int* vneed = ip; ip += capV;
        int* cnt = ip;
        double* vsum = (double*)m->nmw_vsum.p;
        int* h = (int*)m->h_flag;                                // pinned: [0] most steps of a replicate, [1] flagged, [2] to replay
        m->last_nm_flagged = 0; m->last_nm_replayed = 0;
        auto one_kernel = NMW_PICK(true, true);
        if ((rc = allow_lds(m, (const void*)one_kernel, wave_lds))) return rc;
        auto pass_kernel = KS == 1 ? nmp::conv_mfma_kernel<4, 1> : nmp::conv_mfma_kernel<4, 2>;
        auto solve = [&](long count, const int* list, const int* forced) {
            ProfScope ps(m, PLSPM_K_SOLVER);
            const dim3 g1((unsigned)count);
            if (counts16_ready)
                hipLaunchKernelGGL(nmg_kernel<4>, g1, dim3(256), lds, m->stream, md, cd, mdm, (const double*)nullptr, 0L, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact,
                                   0, cat_fast, (unsigned short*)m->gK16.p, ld16, list);
            else
                hipLaunchKernelGGL(nmg_kernel<3>, g1, dim3(threads), lds, m->stream, md, cd, mdm, Mp, mp_stride, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact,
                                   0, cat_fast, (unsigned short*)m->gK16.p, ld16, list);
            hipLaunchKernelGGL(one_kernel, g1, dim3(64), wave_lds, m->stream, md, cd, mdm, so, gSm, gst, (long)st_doubles, (const double*)nullptr, nparts, nact,
                               (const unsigned short*)m->gK16.p, ld16, fuse, list, nsub, nmw::NmwMaps{forced ? nullptr : cmaps, cstride, forced ? nullptr : kmaps, kstride, steps, forced, std::ldexp(1.0, m->tune.nm_bound_shift)});
        };
        solve(nproblems, nullptr, nullptr);
        bool any_flagged = false;
        // (last session of round 6: behind the fourth round the replicates that are still iterating are the few that never converge -- 101 steps each, nine more rounds of four
        //  launches and a host read-back for a handful of slots -- so the fifth round takes JRB steps at once where their slots fit the buffers of a short round: the list
        //  kernel checks that itself and files nothing otherwise; option nm_vlong 0: short rounds only)
        constexpr int JRB = 72;
        bool long_ok = m->tune.nm_vlong != 0;
        for (int j0 = 1;;) {
            const bool long_round = long_ok && j0 > 4 * JR;
            const int jr = long_round ? JRB : JR;
            {
                ProfScope ps(m, PLSPM_K_SCORES);
                if (long_round)
                    hipLaunchKernelGGL(nm_vlist_kernel<JRB>, dim3(1), dim3(1024), 0, m->stream, (const int*)steps, (long)nproblems, j0, vb, vj, cnt, vsum, force, h, (int)std::min<long>(capV, 0x7fffffffL));
                else
                    hipLaunchKernelGGL(nm_vlist_kernel<JR>, dim3(1), dim3(1024), 0, m->stream, (const int*)steps, (long)nproblems, j0, vb, vj, cnt, vsum, force, h, 0);
                hipLaunchKernelGGL(nmp::planes_kernel, dim3((unsigned)(ng16V * 16)), dim3(64), 0, m->stream, (const double*)cmaps, cstride, P, L, KS, (const int*)m->d_boff, (const int*)vb,
                                   (const int*)cnt, (uint4*)m->tab8.p, (double2*)m->scl8.p, (const int*)vj, (const double*)kmaps, kstride, vneed, m->tol, nsub, nparts);
                hipLaunchKernelGGL(pass_kernel, dim3((unsigned)(nparts * ((ng16V + 3) / 4))), dim3(256), 0, m->stream, (const uint4*)m->ind8.p, ntiles16, L, (const unsigned*)cd8, (long)cd8_MT,
                                   (const uint4*)m->tab8.p, (const double2*)m->scl8.p, (const int*)vb, (const int*)cnt, (double*)nullptr, nparts, tpc, nparts, (const double*)nullptr, 0L,
                                   (const int*)vneed, vsum, 1);
                hipLaunchKernelGGL(nm_vflag_kernel, dim3(1), dim3(1024), 0, m->stream, (const double*)vsum, (const int*)vb, (const int*)vj, (const int*)cnt, m->tol, fb, fj, cnt + 1, h + 1);
            }
            HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
            HIPCHK(m, hipEventSynchronize(m->ev_flag));
            if (long_round && h[3] == 1) { long_ok = false; continue; }      // (too many slots: nothing was filed; the same steps again, eight at a time)
            const int most = h[0], flagged = h[1];
            if (flagged > 0) {
                ProfScope ps(m, PLSPM_K_SCORES);
                any_flagged = true;
                m->last_nm_flagged += flagged;
                const long ngf = ((long)flagged + 15) / 16;
                if ((rc = ensure(m, m->nmpartial, (size_t)std::max<long>(flagged, nproblems) * nparts * sizeof(double)))) return rc;
                hipLaunchKernelGGL(nmp::planes_kernel, dim3((unsigned)(ngf * 16)), dim3(64), 0, m->stream, (const double*)cmaps, cstride, P, L, KS, (const int*)m->d_boff, (const int*)fb,
                                   (const int*)(cnt + 1), (uint4*)m->tab8.p, (double2*)m->scl8.p, (const int*)fj, (const double*)kmaps, kstride, (int*)nullptr, m->tol, 0, nparts);
                hipLaunchKernelGGL(pass_kernel, dim3((unsigned)(nparts * ((ngf + 3) / 4))), dim3(256), 0, m->stream, (const uint4*)m->ind8.p, ntiles16, L, (const unsigned*)cd8, (long)cd8_MT,
                                   (const uint4*)m->tab8.p, (const double2*)m->scl8.p, (const int*)fb, (const int*)(cnt + 1), (double*)m->nmpartial.p, nparts, tpc, nparts, (const double*)nullptr, 0L,
                                   (const int*)nullptr, (double*)nullptr, 1);
                hipLaunchKernelGGL(nm_vcheck_kernel, dim3((unsigned)flagged), dim3(64), 0, m->stream, (const double*)m->nmpartial.p, nparts, (const int*)fb, (const int*)fj, (const int*)(cnt + 1),
                                   m->tol, force);
            }
            if (j0 + jr > most - 1) break;
            j0 += jr;
        }
        if (any_flagged) {
            hipLaunchKernelGGL(nm_vfix_kernel, dim3(1), dim3(1024), 0, m->stream, (const int*)steps, (const int*)force, (long)nproblems, fixlist, cnt + 2, h + 2);
            HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
            HIPCHK(m, hipEventSynchronize(m->ev_flag));
            if (h[2] > 0) { m->last_nm_replayed = h[2]; solve(h[2], fixlist, force); }
        }
        HIPCHK(m, hipGetLastError());
        return 0;
    }
    // (round 6: a problem whose lower bound decided nothing sits one launch out while the pass over all rows runs for it -- at most once per step)
    for (int it = 0; it <= (sub_pass ? 2 : 1) * (m->max_iter + 1); ++it) {
        if (it >= 1 && dense && m->tune.nm_live != 0) {            // (*h_flag: the count behind the previous step == the length of the list its pass built)
            const long nlive = *m->h_flag;
            if (nlive >= 1 && nlive < nproblems) { lgrid = dim3((unsigned)nlive); live = (const int*)m->nmlist.p + 1; }
        }
        if (!flag_from_list) HIPCHK(m, hipMemsetAsync(nact, 0, sizeof(int), m->stream));
        if (wave_step) {
            // prepare: the uint16 counts + the initial state only (nmg_kernel<3>); every step, the first one included, one wave per problem; the
            // finish of a problem inside the launch that decides its stop
            ProfScope ps(m, PLSPM_K_SOLVER);
            if (it == 0) {
                if (counts16_ready)       // the Gram wrote the upper triangles: mirror them, set the initial state (no packed fp64 matrix exists)
                    hipLaunchKernelGGL(nmg_kernel<4>, grid, dim3(256), lds, m->stream, md, cd, mdm, (const double*)nullptr, 0L, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact,
                                       0, cat_fast, (unsigned short*)m->gK16.p, ld16, (const int*)nullptr);
                else
                    hipLaunchKernelGGL(nmg_kernel<3>, grid, dim3(threads), lds, m->stream, md, cd, mdm, Mp, mp_stride, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact,
                                       0, cat_fast, (unsigned short*)m->gK16.p, ld16, (const int*)nullptr);
            }
            hipLaunchKernelGGL(wave_kernel, lgrid, dim3(64), wave_lds, m->stream, md, cd, mdm, so, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact,
                               (const unsigned short*)m->gK16.p, ld16, fuse, live, nsub, nmw::NmwMaps{});
        } else
        launch(it == 0 ? 0 : 1);                   // launch 0 = prepare + first step
        // The stop-rule pass is enqueued right behind the step, BEFORE the host knows whether any problem is still active: finished
        // problems / replicate groups return at once on the device, and the 4-byte read-back of the counter overlaps with the pass
        // instead of leaving the GPU idle for a host round trip per iteration.
        if (!flag_from_list) {
            HIPCHK(m, hipMemcpyAsync(m->h_flag, nact, sizeof(int), hipMemcpyDeviceToHost, m->stream));
            HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
        }
        {
            ProfScope ps(m, PLSPM_K_SCORES);
            const double* conv_state = gst;
            long conv_stride = (long)st_doubles;
            const int* conv_boff = m->d_boff;
            if (m->stage1) {
                hipLaunchKernelGGL(hoc_compose_kernel, lgrid, dim3(64), 0, m->stream, make_hoc_desc(m), (const double*)m->stage1->nmstate.p,
                                   (long)nm_state_doubles_of(src), gst, (long)st_doubles, m->n_chol, (double*)m->pseudo.p, ps_stride, live);
                conv_state = (const double*)m->pseudo.p; conv_stride = ps_stride; conv_boff = m->d_lv_cols;
            }
            if (dense) {
                int* live_list = (int*)m->nmlist.p;                            // [count | ids of the problems still iterating, in problem order]
                int* full_list = live_list + nproblems + 1;                    // [count | ids of the live problems that ask for the pass over all rows] (round 6)
                int* h_full = (int*)m->h_flag + 1;
                if (sub_pass) *h_full = 0;
                hipLaunchKernelGGL(active_list_kernel, dim3(1), dim3(1024), 0, m->stream, conv_state, conv_stride, nproblems, live_list + 1, live_list,
                                   flag_from_list ? (int*)m->h_flag : (int*)nullptr, sub_pass ? full_list + 1 : (int*)nullptr, sub_pass ? full_list : (int*)nullptr,
                                   (sub_pass && flag_from_list) ? h_full : (int*)nullptr);
                if (flag_from_list) HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
                if (use_mfma) {
                    auto pass_kernel = KS == 1 ? nmp::conv_mfma_kernel<4, 1> : nmp::conv_mfma_kernel<4, 2>;
                    hipLaunchKernelGGL(nmp::planes_kernel, dim3((unsigned)(ng16 * 16)), dim3(64), 0, m->stream, conv_state, conv_stride, src->P, L, KS, conv_boff, (const int*)(live_list + 1),
                                       (const int*)live_list, (uint4*)m->tab8.p, (double2*)m->scl8.p, (const int*)nullptr, (const double*)nullptr, 0L, (int*)nullptr, 0.0, 0, 0);
                    hipLaunchKernelGGL(pass_kernel, dim3((unsigned)(nparts * ((ng16 + 3) / 4))), dim3(256), 0, m->stream, (const uint4*)m->ind8.p, ntiles16, L, (const unsigned*)cd8,
                                       (long)cd8_MT, (const uint4*)m->tab8.p, (const double2*)m->scl8.p, (const int*)(live_list + 1), (const int*)live_list, part, nparts, tpc, nparts,
                                       sub_pass ? conv_state : (const double*)nullptr, conv_stride, (const int*)nullptr, (double*)nullptr, 0);
                    if (sub_pass && flag_from_list) {
                        // the problems whose lower bound decided nothing (few, as a rule none): all row chunks, fixed-order sums -- the host knows their number
                        // by now (the list kernel wrote it to pinned memory; the passes above run meanwhile)
                        HIPCHK(m, hipEventSynchronize(m->ev_flag));
                        const long nfull = *h_full;
                        if (nfull > 0) {
                            m->last_nm_exact += (int)nfull;
                            const long ngf = (nfull + 15) / 16;
                            hipLaunchKernelGGL(nmp::planes_kernel, dim3((unsigned)(ngf * 16)), dim3(64), 0, m->stream, conv_state, conv_stride, src->P, L, KS, conv_boff, (const int*)(full_list + 1),
                                               (const int*)full_list, (uint4*)m->tab8.p, (double2*)m->scl8.p, (const int*)nullptr, (const double*)nullptr, 0L, (int*)nullptr, 0.0, 0, 0);
                            hipLaunchKernelGGL(pass_kernel, dim3((unsigned)(nparts * ((ngf + 3) / 4))), dim3(256), 0, m->stream, (const uint4*)m->ind8.p, ntiles16, L, (const unsigned*)cd8,
                                               (long)cd8_MT, (const uint4*)m->tab8.p, (const double2*)m->scl8.p, (const int*)(full_list + 1), (const int*)full_list, part, nparts, tpc, nparts, (const double*)nullptr, 0L, (const int*)nullptr, (double*)nullptr, 0);
                        }
                    }
                } else {
                hipLaunchKernelGGL(coef_table_kernel, dim3((unsigned)ngroups, (unsigned)((2 * src->P + 2 * L + 1 + 63) / 64)), dim3(256), 0, m->stream, conv_state, conv_stride, src->P, L,
                                   (const int*)(live_list + 1), (const int*)live_list, (double*)m->ctable.p);
                const int gx = (int)((ntiles16 + 7) / 8);                      // row blocks of 128 rows (8 tiles: 8 x 16-row or 16 x 8-row waves)
                const int rbx = (gx + 7) / 8;                                  // row blocks per XCD
                // replicate slices: one group of 64 replicates per workgroup measured best (1.06 ms for three passes against 1.17 / 1.21 /
                // 1.28 with 12 / 6 / 13 slices): many small workgroups let the dispatcher balance the CUs
                const int gy = m->tune.conv_gy > 0 ? m->tune.conv_gy : ngroups;
                auto conv_kernel = counts8 ? (dense_whole ? nm_conv_dense_kernel<16, 8, false, true> : nm_conv_dense_kernel<16, 8, true, true>)
                                           : (dense_whole ? nm_conv_dense_kernel<16, 8, false, false> : nm_conv_dense_kernel<16, 8, true, false>);
                if (use_codes)
                    hipLaunchKernelGGL(nm_conv_codes_kernel<8>, dim3((unsigned)(8 * rbx * gy)), dim3(512), codes_lds, m->stream, (const unsigned short*)m->codes.p, ntiles16, src->Pm, src->P, L,
                                       conv_boff, codes_lmv, (const uint4*)cd8, (long)cd8_MT, (const double*)m->ctable.p,
                                       (const int*)((int*)m->nmlist.p + 1), (const int*)m->nmlist.p, part, nparts, rbx, gy, kb);
                else
                hipLaunchKernelGGL(conv_kernel, dim3((unsigned)(8 * rbx * gy)), dim3(512), dense_use_lds, m->stream, (const double*)src->Xt.p, ntiles16, src->PA, src->P, L,
                                   conv_boff, counts8 ? (const unsigned short*)cd8 : (const unsigned short*)src->dcnt.p, counts8 ? (long)cd8_MT : src->dcnt_stride,
                                   (const double*)m->ctable.p, (const int*)((int*)m->nmlist.p + 1), (const int*)m->nmlist.p, part, nparts, rbx, gy, kb, 0);
                }
            } else {
                hipLaunchKernelGGL(nm_conv_kernel, dim3(nparts, (unsigned)nproblems), dim3(256), conv_lds, m->stream, src->d_Xa, N, src->PA, src->P, L, 0, conv_boff, ent, nent,
                                   ent_stride, conv_state, conv_stride, part);
            }
        }
        HIPCHK(m, hipEventSynchronize(m->ev_flag));
#ifdef PLSPM_DEBUG_MARKS
        if (cat && it == 1) {
            long long h[32];
            HIPCHK(m, hipStreamSynchronize(m->stream));
            HIPCHK(m, hipMemcpy(h, d_nm_marks, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[plspm nmg_step clocks] V=Mn.c %lld  YY+G %lld  inner weights %lld  MZ+a %lld  quantify(par) %lld  LV loop %lld  score map %lld  total %lld\n", h[21] - h[20],
                    h[22] - h[21], h[23] - h[22], h[24] - h[23], h[25] - h[24], h[26] - h[25], h[27] - h[26], h[27] - h[20]);
        }
#endif
        if (*m->h_flag == 0) break;
    }
#ifdef PLSPM_DEBUG_MARKS
    if (d_nm_marks) plspm_dfree(d_nm_marks);
#endif
    HIPCHK(m, hipGetLastError());
    return 0;
}
#undef NMW_PICK

// Round 6: a bootstrap batch of a Scale.NUM / RAW model (no missing cells, at most 64 MVs and 16 LVs) on the int8 Gram route as ONE solver launch + a
// verification pass (kernels_solver.h solver_nmwave_kernel; kernels_nonmetric.h nm_vlist_kernel ...).  The legacy loop (run_nonmetric) launches one step
// kernel + one stop-rule pass over all rows per iteration, with a host round trip each: 0.435 ms of solver launches and 0.61 ms of passes per 5,000
// replicates of the headline's shape (profiles/r05_nonmetric_kernels.txt).  Here: 1 launch that iterates on the bound, then -- for the steps it continued
// behind -- an eighth of the rows (a lower bound of the criterion that only has to clear the tolerance), ONE host read-back, and the exact pass + replay for
// whatever the lower bound could not confirm (nothing, as a rule).  Dense moment matrices at m->gram; cd8 / cd8_MT: the int8 counts the Gram consumed.
bool nm_wave_route_planned(const plspm_model* m) {
    bool whole = false;
    int kb = 1;
    return m->tune.nm_wave16 != 0 && nm_wave_solver_covers(m) && m->N <= 0x7fffffffL && nm_dense_lds(m, &whole, &kb) != 0;
}

int run_nonmetric_wave(plspm_model* m, long nb, const SolverOut& so, const void* cd8, int cd8_MT) {
    const int P = m->P, L = m->L, W = P + L + 1;                 // a map: c_p | k_l | the bound of its step
    const long N = m->N, ntiles16 = (N + 15) / 16;
    int rc;
    bool dense_whole = false;
    int kb = 1;
    const size_t dense_use_lds = nm_dense_lds(m, &dense_whole, &kb);
    if (!dense_use_lds || !cd8) return fail(m, PLSPM_E_STATE, "non-metric wave route: no dense stop-rule pass / no int8 counts");
    constexpr int JR = 4;                                        // steps verified per round (three iterations -- two continued steps -- is the rule)
    const long capV = nb * JR;
    const long maps_stride = (long)(m->max_iter + 2) * W;
    const int table_rows = 2 * P + 2 * L + 1;
    const long ngroupsV = (capV + 63) / 64;
    if ((rc = ensure(m, m->nmw_maps, (size_t)nb * maps_stride * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->nmw_ints, (size_t)(3 * nb + 5 * capV + 16) * sizeof(int)))) return rc;
    if ((rc = ensure(m, m->nmw_vsum, (size_t)capV * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->Xt, (size_t)ntiles16 * 16 * m->PA * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->ctable, (size_t)ngroupsV * table_rows * 64 * sizeof(double)))) return rc;
    auto conv_kernel = dense_whole ? nm_conv_dense_kernel<16, 8, false, true> : nm_conv_dense_kernel<16, 8, true, true>;
    if ((rc = allow_lds(m, (const void*)conv_kernel, dense_use_lds))) return rc;
    if (!m->Xt_valid) {
        hipLaunchKernelGGL(tile_transpose_kernel, dim3((unsigned)ntiles16), dim3(256), 0, m->stream, (const double*)m->d_Xa, N, m->PA, (double*)m->Xt.p);
        m->Xt_valid = true;
    }
    double* maps = (double*)m->nmw_maps.p;
    int* ip = (int*)m->nmw_ints.p;
    int* steps = ip; ip += nb;
    int* force = ip; ip += nb;
    int* fixlist = ip; ip += nb;
    int* vb = ip; ip += capV;
    int* vj = ip; ip += capV;
    int* fb = ip; ip += capV;
    int* fj = ip; ip += capV;
    int* vneed = ip; ip += capV;
    int* cnt = ip;                                               // [0] virtual problems of the round, [1] flagged, [2] replicates to replay
    double* vsum = (double*)m->nmw_vsum.p;
    int* h = (int*)m->h_flag;                                    // pinned: [0] most steps of a replicate, [1] flagged, [2] to replay
    m->last_nm_wave16 = 1; m->last_nm_problems = 0; m->last_nm_codes = 0; m->last_nm_mfma = 0; m->last_nm_wave = 0; m->last_nm_direct16 = 0;
    m->last_nm_flagged = 0; m->last_nm_replayed = 0;
    if ((rc = launch_nm_wave_solver(m, nb, so, maps, maps_stride, steps, nullptr, nullptr))) return rc;
    // the rows pass A looks at: per slot by the bound of its step (nm_vlist_kernel), at most the first eighth of the row blocks of 128 rows -- or, option
    // nm_verify_rows, that percentage for every slot
    const int nblocks_all = (int)((ntiles16 + 7) / 8);
    const int fixed_blocks = m->tune.nm_verify_rows > 0 ? (int)std::min<long>(nblocks_all, std::max<long>(1, ((long)nblocks_all * m->tune.nm_verify_rows + 99) / 100)) : 0;
    const int cap_blocks = fixed_blocks > 0 ? fixed_blocks : std::max(1, (nblocks_all + 7) / 8);
    const long nsub = std::min<long>(ntiles16, 8L * cap_blocks);
    const int gyV = (int)std::max<long>(1, (2 * nb + 63) / 64);
    const size_t verify_lds = ((size_t)table_rows * 64 + (size_t)8 * P * 16) * sizeof(double);
    if ((rc = allow_lds(m, (const void*)nm_verify_kernel, verify_lds))) return rc;
    auto exact_pass = [&](const int* list, const int* count, double* partial) {      // all rows, fixed-order partial sums filed under the (virtual) slot
        const int gx = (int)((ntiles16 + 7) / 8), rbx = (gx + 7) / 8;
        hipLaunchKernelGGL(conv_kernel, dim3((unsigned)(8 * rbx * gyV)), dim3(512), dense_use_lds, m->stream, (const double*)m->Xt.p, ntiles16, m->PA, P, L, (const int*)m->d_boff,
                           (const unsigned short*)cd8, (long)cd8_MT, (const double*)m->ctable.p, list, count, partial, (int)ntiles16, rbx, gyV, kb, 1);
    };
    bool any_flagged = false;
    for (int j0 = 1;; j0 += JR) {
        {
            ProfScope ps(m, PLSPM_K_SCORES);
            hipLaunchKernelGGL(nm_vlist_kernel<JR>, dim3(1), dim3(1024), 0, m->stream, (const int*)steps, nb, j0, vb, vj, cnt, vsum, force, h, 0);
            hipLaunchKernelGGL(nm_vtable_kernel, dim3((unsigned)ngroupsV, (unsigned)((table_rows + 63) / 64)), dim3(256), 0, m->stream, (const double*)maps, maps_stride, P, L, (const int*)vb,
                               (const int*)vj, (const int*)cnt, (double*)m->ctable.p, vneed, m->tol, nblocks_all, cap_blocks, fixed_blocks);
            hipLaunchKernelGGL(nm_verify_kernel, dim3((unsigned)(cap_blocks * gyV)), dim3(512), verify_lds, m->stream, (const double*)m->Xt.p, nsub, m->PA, P, L, (const int*)m->d_boff,
                               (const uint4*)cd8, (long)cd8_MT, (const double*)m->ctable.p, (const int*)vb, (const int*)cnt, (const int*)vneed, vsum, cap_blocks, gyV);
            hipLaunchKernelGGL(nm_vflag_kernel, dim3(1), dim3(1024), 0, m->stream, (const double*)vsum, (const int*)vb, (const int*)vj, (const int*)cnt, m->tol, fb, fj, cnt + 1, h + 1);
        }
        HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
        HIPCHK(m, hipEventSynchronize(m->ev_flag));
        const int most = h[0], flagged = h[1];
        if (flagged > 0) {
            // the exact criterion of what the lower bound left open: all rows, fixed-order sums; a value below the tolerance moves that replicate's stop
            ProfScope ps(m, PLSPM_K_SCORES);
            any_flagged = true;
            m->last_nm_flagged += flagged;
            if ((rc = ensure(m, m->nmpartial, (size_t)flagged * ntiles16 * sizeof(double)))) return rc;
            hipLaunchKernelGGL(nm_vtable_kernel, dim3((unsigned)((flagged + 63) / 64), (unsigned)((table_rows + 63) / 64)), dim3(256), 0, m->stream, (const double*)maps, maps_stride, P, L,
                               (const int*)fb, (const int*)fj, (const int*)(cnt + 1), (double*)m->ctable.p, (int*)nullptr, m->tol, 0, 0, 0);
            exact_pass(fb, cnt + 1, (double*)m->nmpartial.p);
            hipLaunchKernelGGL(nm_vcheck_kernel, dim3((unsigned)flagged), dim3(64), 0, m->stream, (const double*)m->nmpartial.p, (int)ntiles16, (const int*)fb, (const int*)fj,
                               (const int*)(cnt + 1), m->tol, force);
        }
        if (j0 + JR > most - 1) break;                           // every step a replicate continued behind has been looked at
    }
    if (any_flagged) {
        hipLaunchKernelGGL(nm_vfix_kernel, dim3(1), dim3(1024), 0, m->stream, (const int*)steps, (const int*)force, nb, fixlist, cnt + 2, h + 2);
        HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
        HIPCHK(m, hipEventSynchronize(m->ev_flag));
        if (h[2] > 0) {
            m->last_nm_replayed = h[2];
            if ((rc = launch_nm_wave_solver(m, h[2], so, nullptr, 0, nullptr, force, fixlist))) return rc;
        }
    }
    HIPCHK(m, hipGetLastError());
    return 0;
}

// include/plspm_hip_test.h
extern "C" int plspm_nonmetric_criteria(plspm_model_t* m, int64_t B, double* out) {
    if (!m || !out || B < 0) return PLSPM_E_ARG;
    const size_t st = nm_state_doubles_of(m);
    if (!m->nmstate.p || m->nmstate.cap < (size_t)B * st * sizeof(double) || !m->last_nm_problems || B > m->last_nm_problems) return fail(m, PLSPM_E_STATE, "plspm_nonmetric_criteria: no non-metric run of that many problems");
    hipSetDevice(m->device);
    HIPCHK(m, hipStreamSynchronize(m->stream));
    HIPCHK(m, hipMemcpy2D(out, sizeof(double), (const double*)m->nmstate.p + 4, st * sizeof(double), sizeof(double), (size_t)B, hipMemcpyDeviceToHost));
    return 0;
}
