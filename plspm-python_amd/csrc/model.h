// Host-side state of one libplspm_hip handle (include/plspm_hip.h), shared by the translation units of the library:
// the .hip translation units (kernels + single-device entry points; host_internal.h lists them) and plspm_group.cpp (multi-GPU groups over RCCL).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/plspm_hip_test.h"

inline thread_local std::string g_create_error;

// Caching allocator of the library (defined in plspm_hip.hip).  A Plspm() call creates a handle, uploads, fits and destroys it:
// without a cache every call pays ~40 hipMalloc / hipHostMalloc / hipFree round trips (measured 12 of the 17 ms of a 10k x 60
// Plspm() fit).  Freed blocks are kept per device (and for pinned host memory) and handed out again best-fit; a block is only
// ever returned by an owner whose stream has been synchronised, so a cached block has no work in flight.
// plspm_release_cached_memory() (C-ABI) gives everything back to the runtime.
hipError_t plspm_dmalloc(void** p, size_t bytes);       // device memory on the CURRENT device
void plspm_dfree(void* p);
hipError_t plspm_hmalloc(void** p, size_t bytes);       // pinned host memory
void plspm_hfree(void* p);
// Streams are cached the same way (hipStreamCreate / hipStreamDestroy set up and tear down a hardware queue: ~1-2 ms each):
// a stream goes back only after it has been synchronised.  Non-blocking streams of the CURRENT device.
hipError_t plspm_stream_acquire(hipStream_t* s);
void plspm_stream_release(hipStream_t s);

// Kernel timing (plspm_profile_*): event pairs are recycled through `pool`, so a profiled launch costs two hipEventRecord only.
struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev, pool; double total_ms = 0.0; int64_t launches = 0; };

struct plspm_model {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;      // an attached second stage runs on its first stage's stream
    int P = 0, L = 0, PA = 0, T = 0, scheme = 0, scaled = 1, max_iter = 100, kmax = 0, n_eff = 0, n_chol = 0;
    double tol = 1e-6;
    std::vector<int> boff, lvof, mode, chol_off, eff_from, eff_to, pred_off, pred_idx, succ_off, succ_idx;
    std::vector<uint8_t> C;
    int *d_boff = nullptr, *d_lvof = nullptr, *d_mode = nullptr, *d_chol_off = nullptr, *d_eff_from = nullptr, *d_eff_to = nullptr;
    int *d_pred_off = nullptr, *d_pred_idx = nullptr, *d_succ_off = nullptr, *d_succ_idx = nullptr;
    uint8_t* d_C = nullptr;
    double* d_shift = nullptr;
    int64_t N = 0;
    double* d_Xa = nullptr;
    // grow-only device scratch
    struct Buf { void* p = nullptr; size_t cap = 0; };
    Buf ent, nent, gram, gram_partial, rows, status, iters, gS, gsmall, fitout, idx, err, ghist, nmstate, nmpartial, nmactive, nmlist, gK16, sum_buf, cols;
    int nonmetric = 0;           // Scale.NUM / Scale.RAW data: population-standardised MVs, score-based stop rule
    bool codes_valid = false;    // m->codes holds the category codes of the uploaded rows (nm_conv_codes_kernel)
    bool ind8_valid = false;     // m->ind8 holds their indicator bytes in MFMA fragment order (kernels_nmp.h)
    long last_nm_problems = 0;   // problems of the last run_nonmetric (plspm_nonmetric_criteria)
    int last_i8_nibbles = 0;     // 1: the last int8 resample drew into 4-bit counters (data sets beyond one 16-bit window)
    int last_nm_direct16 = 0;    // 1: the last categorical bootstrap's count matrices were written by the int8 product itself (uint16, no scatter pass)
    int last_nm_mfma = 0;        // 1: the last non-metric bootstrap's stop-rule passes ran as int8 matrix products (kernels_nmp.h)
    int last_nm_wave = 0;        // 1: the last categorical iteration ran one wave per problem (kernels_nmw.h)
    int last_nm_wave16 = 0;      // 1: the last Scale.NUM / RAW bootstrap batch ran as one solver launch + verification (plspm_nonmetric.hip run_nonmetric_wave)
    int last_nm_one = 0;         // 1: the last categorical bootstrap batch ran as one solver launch + verification (plspm_nonmetric.hip, round 6)
    int last_nm_exact = 0;       // categorical wave step (round 6): (problem, step) pairs of the last bootstrap whose lower bound decided nothing and that took the pass over all rows
    int last_nm_flagged = 0, last_nm_replayed = 0;      // ... (replicate, step) pairs its lower-bound pass left to the exact pass / replicates whose stop moved
    Buf nmw_maps, nmw_ints, nmw_vsum;
    int last_nm_codes = 0;       // 1: the last non-metric bootstrap's stop-rule passes ran on category codes
    bool cat_pure = false;       // every logical MV is ORD / NOM: all device columns are 0/1 indicators
    int categorical = 0;         // Scale.ORD / NOM present: device columns are aug columns (solver_nmg.h); Pm logical MVs
    int Pm = 0, cmax = 1, kmv = 1;
    std::vector<int> mv_off, mv_kind, lmv_off, mv_lv, no_chol, mv_base, mv_base2, lmv2_off;
    int *d_mv_off = nullptr, *d_mv_kind = nullptr, *d_lmv_off = nullptr, *d_mv_lv = nullptr, *d_no_chol = nullptr, *d_mv_base = nullptr, *d_mv_base2 = nullptr, *d_lmv2_off = nullptr;
    Buf gSm;
    // metric data with missing values: Pg = P + n_ind device columns (data | missing indicators); PA / T describe the Gram of
    // those, PAs / Ts the P-column moment matrix the solver reads (impute_collapse maps one to the other).  Pg == P otherwise.
    int Pg = 0, PAs = 0, Ts = 0, n_ind = 0;
    std::vector<int> ind_of;
    int* d_ind_of = nullptr;
    Buf gram2;
    // two-stage higher order constructs (solver_hoc.h): `stage2` of a data-holding handle / `stage1` of its attached second stage
    plspm_model* stage2 = nullptr;
    plspm_model* stage1 = nullptr;
    std::vector<int> lv_first, col2_lv1, col2_p1, hcol, hidx;
    std::vector<int> lv_cols;
    int* d_lv_cols = nullptr;
    int *d_lv_first = nullptr, *d_col2_lv1 = nullptr, *d_col2_p1 = nullptr, *d_hcol = nullptr, *d_hidx = nullptr;
    Buf pseudo;
    // non-metric data with missing values (solver_nmx.h): K incomplete rows live in side tables, their rows of Xa are zero
    Buf dcnt, ctable, Xt;        // dense stop-rule pass of the non-metric bootstrap: uint16 histograms, coefficient table, tiled copy of Xa
    long dcnt_stride = 0;
    bool Xt_valid = false, dcnt_ready = false;
    int nmx_K = 0, nmx_raw = 0;
    double *d_Xk = nullptr, *d_Mk = nullptr;
    int* d_rowid = nullptr;
    int* h_flag = nullptr;        // pinned: active-problem counter of the non-metric iteration
    hipEvent_t ev_flag = nullptr;
    void* h_stage = nullptr;      // pinned host staging for plspm_fit results
    size_t h_stage_cap = 0;
    // result bookkeeping: `rows` holds the records of the last plspm_bootstrap(_device) only (the fit's scores have their own buffer)
    Buf scores;
    Buf xa, up_raw, up_ci, up_partial;   // resident matrix (d_Xa points into `xa` while data are uploaded) and the upload staging
    int64_t rows_B = 0;           // number of valid records in `rows` (0: none)
    bool err_clean = false;       // the device error word is zero and no call since could have raised it (plspm_detail_bootstrap)
    // launch-geometry options (plspm_model_set_option); validated there, never read from the environment
    struct Tune { int wide_nw = 4, fit_chunks = 0, conv_pass = 0, conv_gy = 0, nm_threads = 0, solver_threads = 128, scores_tile = 0, gram_lds_kb = 0;
                  int gram_path = 0, i8_slices = 0, i8_min_batch = 1, i8_waves = 8, i8_rt = 0, i8_short = -1, i8_cus = 0, upload_direct = 0, i8_dma = 0, i8_variant = -1, solver_rows = 1, solver_wave = 1, solver_quad = 1, nm_counts8 = 1, nm_fast_lds = 1, nm_k16 = 1, nm_codes = 1, i8_ind = 1, i8_shape = 16, resample_aux = 0, i8_sched = 0, i8_priv = 1, boot_chunks = 0, boot_ratio = 60, boot_align = 0, i8_persist = 0, i8_min_slices = 0, i8_nostore = 0, nm_wave = 1, nm_live = 1, nm_mfma = 1, nm_direct16 = 1, i8_nibbles = 1, nm_wave16 = 1, nm_verify_rows = 0, nm_bound_shift = 0, wide_ring = 4, nm_subset = 4, nm_cat_one = 1, nm_cpl = 0, nm_c10 = 1, nm_vlong = 1; } tune;
    // int8 digit-plane Gram of bootstrap batches (kernels_gram_i8.h): per data set the digit planes `zs` of all pair products and the
    // pair tables (p, q, k, slot in the packed matrix | 2^-k); per call the dense int8 multiplicities `cd`
    Buf zs, cd, cd1, err2, pair_tab, pair_scale, zs_stat, codes, ind8, tab8, scl8, pp_ctl;      // pp_ctl: tile counters of the persistent Gram (kernels_gram_i8p.h GramI8PPCtl)
    // set_option("resample_aux", 1..3): the int8 counts are drawn on a second stream (1 lowest / 2 default / 3 highest priority) into
    // alternating buffers, so that the draws of call k+1 -- enqueued while the Gram / solver of call k still run -- take the CUs those
    // leave idle (the Gram's last, partial round of workgroups first of all); the Gram waits for its counts by event.  Measured
    // (tools/aux_ab.py, 5,000 replicates per step): 0.647 ms off, 0.635 lowest, 0.654 default, 0.634 highest priority.  OFF by default:
    // 2 % is inside the box-to-box spread, and with it every Gram launch shares the chip with another kernel (0.455 -> 0.49 ms per
    // launch), so per-kernel timings would no longer describe the kernel alone.
    // persistent ("stream-K") int8 Gram, set_option("i8_sched", 1): one accumulator slot + one flag per wave of every workgroup, launch serial
    Buf sk_partial, sk_flags;
    unsigned sk_epoch = 0;
    int cu_count = 0;
    hipStream_t aux = nullptr;
    hipEvent_t ev_counts[2] = {nullptr, nullptr}, ev_cdfree[2] = {nullptr, nullptr};
    bool cdfree_set[2] = {false, false};
    int cd_slot = 0;
    bool zs_valid = false;
    bool zs_stats_ready = false;  // phase 1 of the digit planes is enqueued (pair tables, column statistics on their way to h_zstat): plspm_gram_i8.hip prepare_zs_stats
    int zs_stats_S = 0;           // ... for this value of the "i8_slices" option
    void* h_zstat = nullptr;      // pinned copy of the column statistics (automatic plane count)
    size_t h_zstat_cap = 0;
    hipEvent_t ev_zstat = nullptr;
    int zs_S = 0, zs_KB = 0, zs_NT = 0, zs_npair = 0, zs_npg = 0;
    bool zs_ind = false;        // one plane per pair group, launched through the seven-plane main loop (gram_i8_kernel<.., IND>)
    double zs_ratio = 0.0;      // smallest sum|z| / max|z| over the pair columns (automatic plane count; 0: not evaluated)
    double* moments_out = nullptr; // plspm_bootstrap_moments: dense moment matrices go here and the solver is skipped
    int last_gram_path = 0;       // 1 fp64 MFMA, 2 int8 digit planes: what the last bootstrap call used (plspm_model_get_info)
    int last_i8_dma = 0;          // 1 global_load_lds, 2 buffer_load ... lds: the LDS-DMA form of the last int8 Gram launch
    int last_i8_persist = 0;      // 1: ... as one persistent workgroup per CU (gram_i8pp_kernel)
    int last_i8_priv = 0;         // 1: the last int8 Gram launch was gram_i8p_kernel (private count fragments)
    int last_i8_rt = 0;           // count tiles (16 replicates each) per workgroup of the last int8 Gram launch: 16, 20 or 8
    bool mix_valid = false, mix_wide = false; long mix_key[4] = {0, 0, 0, 0}; int mix_tall = 0, mix_short = 0;      // plspm_gram_i8.hip i8_mix_plan: the last tile-row cut
    int last_i8_mt = 0;           // count tiles (padded) of the last int8 Gram launch: 16 x this many replicate slots went through the matrix pipe
    int last_i8_short = 0;        // ... and, with 20, how many of its tile rows were short ones (16 count tiles: plspm_gram_i8.hip i8_mix_plan)
    int last_solver = 0;          // 1 LDS solver (solver_kernel), 2 rows solver (solver_rows_kernel), 3 wave solver (solver_wave_kernel), 4 split rows solver (solver_rows_split_kernel), 5 quad solver (solver_quad_kernel), 6 wave solver for 9 .. 16 LVs (solver_wave16_kernel<16>), 7 its LMAX = 8 form (solver_wave16_kernel<8>: models of at most 8 LVs since round 5), 8 its LMAX = 32 form (17 .. 32 LVs, all Mode A): the last metric bootstrap's
    // grow-only pinned host staging for uploads / row downloads (two halves: copy-in of chunk k+1 overlaps the DMA of chunk k)
    void* h_pin = nullptr;
    size_t h_pin_cap = 0;
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    hipEvent_t ev_pin_async = nullptr;   // behind copies that left the staging area without a host wait (plspm_hip.hip pin_leave_async)
    bool pin_pending = false;
    enum { BLOB_MODEL = 0, BLOB_CATEGORICAL = 1, BLOB_HOC = 2, BLOB_COUNT = 3 };
    void* blobs[BLOB_COUNT] = {nullptr, nullptr, nullptr};     // descriptor blocks, one per call site (several small arrays uploaded as one: plspm_hip.hip upload_blob)
    void* group = nullptr;        // the plspm_group this handle currently belongs to (plspm_group.cpp)
    hipEvent_t stop_event = nullptr;   // set by a caller of plspm_detail_bootstrap: the LAST kernel of a one-chunk metric batch signals it on completion (taken = reset to null)
    // plspm_bootstrap as sub-batches (plspm_bootstrap.hip): the copy stream the records of sub-batch k leave on while sub-batch k + 1 computes,
    // one event per sub-batch
    hipStream_t dl = nullptr;
    hipEvent_t ev_part[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool profiling = false;
    int prof_only = -1;          // >= 0: only this kernel id is bracketed by events (plspm_profile_enable(m, 2 + id))
    ProfSlot prof[PLSPM_K_COUNT];
    std::string error;
};

// Core of plspm_bootstrap_device (plspm_bootstrap.hip): enqueue B replicates on the handle's stream, records written at `rows_out`
// (pitch plspm_row_stride) or into the handle's own `rows` buffer when rows_out is NULL.  No host synchronisation for metric models.
int plspm_detail_bootstrap(plspm_model* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* d_idx, double* rows_out);
// plspm_group.cpp: a handle that is destroyed while bound to a group takes the group's hold on every handle with it.
void plspm_detail_group_orphan(void* group);
// Host -> device copy through the handle's pinned staging halves (chunked; returns when the source may be re-used).
int plspm_detail_h2d(plspm_model* m, void* dst, const void* src, size_t bytes);
// Summary statistics of device records (plspm_bootstrap_summary without the argument checks on `rows`).
int plspm_detail_summary(plspm_model* m, const double* rows, int64_t B, int32_t stride, const double* original, double* summary, int64_t* n_used);
// Host copy of device records [B x stride] -> rows [B x R], status, iters (any may be NULL), through the pinned staging buffer.
int plspm_detail_fetch_records(plspm_model* m, const double* d_records, int64_t B, int32_t stride, double* out, int32_t* status, int32_t* iters);

// Sub-batches of ONE call of B units (replicates) whose results leave the device behind the kernels -- over PCIe (plspm_bootstrap) or through the
// collective (plspm_group_bootstrap): sizes in a geometric progression (ratio_pct / 100: what moving a unit costs relative to computing it), so that
// the transfer of sub-batch k hides under the kernels of sub-batch k + 1 and only the last, smallest transfer is exposed; every part but the last
// a multiple of `align` units (64: whole count tiles of the int8 Gram; plspm_detail_round_units: whole rounds of the machine).  chunks_opt 0: automatic (one part below 2 MiB of results, else up to three),
// n >= 1: n parts.  Returns the number of parts (<= kBootChunksMax), sizes in parts[].  Host arithmetic only (plspm_group.cpp).
static constexpr int kBootChunksMax = 8;
int plspm_detail_chunk_plan(int64_t B, int64_t bytes_per_unit, int chunks_opt, int ratio_pct, int64_t* parts, int64_t align = 64);
// Replicates that fill ONE round of the device with tall tiles of the int8 Gram on this handle's model (whole tile rows: a row of tall tiles
// is 320 / 256 replicates x every pair tile) -- the alignment of the sub-batches of a call: a part that ends inside a round pays for the
// whole round (5,000 replicates of the headline model as 2,560 + 1,536 + 904: Gram 0.42 ms; as 2,560 + 1,280 + 1,160: 0.35; one batch 0.334).
// 64 when the handle's bootstrap does not take that Gram.  plspm_gram_i8.hip.
int64_t plspm_detail_round_units(plspm_model* m);
// The same figure from the handle's state AS IT IS -- no allocation, no launch, no host wait: 64 while the digit planes of the current upload
// have not been built (read-only queries: plspm_model_get_option "boot_round_units", plspm_group_plan).
int64_t plspm_detail_round_units_peek(const plspm_model* m);
// plspm_group.cpp: the sub-batch alignment a group's ranks agreed on is void after an upload / option change on one of its handles (calls
// every rank of a job makes alike); the next plspm_group_bootstrap agrees again.
void plspm_detail_group_plan_changed(void* group);
struct FetchSeg { int64_t b0, nb; hipEvent_t ready; };

// Same-device record exchange of a group (plspm_bootstrap.hip): send[i] -> recv[d] + i * doubles for all i, d < n, one launch on `stream`.
#define PLSPM_GATHER_LOCAL_MAX 16
// ndst: the first `ndst` receive buffers are filled (n: all-gather; 1: the gather to rank 0 of group option "gather_root").
int plspm_detail_gather_local(hipStream_t stream, int n, const double* const* send, double* const* recv, size_t doubles, int ndst);

inline int fail(plspm_model* m, int code, const std::string& msg) {
    if (m) m->error = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(m, call)                                                                                         \
    do {                                                                                                        \
        hipError_t e__ = (call);                                                                                \
        if (e__ != hipSuccess) { (void)hipGetLastError(); /* (the runtime's last-error word is per thread and sticky: a later hipGetLastError() check of ANOTHER handle must not inherit it) */ \
            return fail((m), -(int)e__, std::string(#call) + ": " + hipGetErrorString(e__)); }                  \
    } while (0)

inline int ensure(plspm_model* m, plspm_model::Buf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { HIPCHK(m, hipStreamSynchronize(m->stream)); plspm_dfree(b.p); }     // nothing of this handle may still be using the old block
    b.p = nullptr; b.cap = 0;
    HIPCHK(m, plspm_dmalloc(&b.p, bytes));
    b.cap = bytes;
    return 0;
}

static constexpr size_t kMaxLds = 160 * 1024;
// Dynamic LDS beyond the 64 KiB default needs an explicit opt-in per kernel.
inline int allow_lds(plspm_model* m, const void* fn, size_t bytes) {
    if (bytes > kMaxLds) return fail(m, PLSPM_E_LIMIT, "kernel needs more than 160 KiB of LDS");
    if (bytes > 48 * 1024) {
        // (a kernel with static LDS of its own can be refused below 160 KiB of dynamic LDS: the same limit, the same code -- found by the many-LV fuzz, a 34-LV / 179-MV model)
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(m, PLSPM_E_LIMIT, "kernel needs " + std::to_string(bytes) + " bytes of dynamic LDS beside its static share: more than the device grants (" + hipGetErrorString(e) + ")");
        }
    }
    return 0;
}

struct ProfScope {
    plspm_model* m; int id; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(plspm_model* m_, int id_, hipStream_t stream = nullptr) : m(m_), id(id_), s(stream ? stream : m_->stream), on(m_->profiling && (m_->prof_only < 0 || m_->prof_only == id_)) {
        if (on) {
            auto& pool = m->prof[id].pool;
            if (pool.empty()) { hipEventCreate(&a); hipEventCreate(&b); }
            else { a = pool.back().first; b = pool.back().second; pool.pop_back(); }
            hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (on) { hipEventRecord(b, s); m->prof[id].ev.emplace_back(a, b); }
    }
};
inline void prof_collect(plspm_model* m) {
    for (int k = 0; k < PLSPM_K_COUNT; ++k) {
        for (auto& pr : m->prof[k].ev) {
            float ms = 0.f;
            hipEventSynchronize(pr.second);
            hipEventElapsedTime(&ms, pr.first, pr.second);
            m->prof[k].total_ms += ms; m->prof[k].launches += 1;
            m->prof[k].pool.push_back(pr);
        }
        m->prof[k].ev.clear();
    }
}
