// libplspm_hip.so -- MI355X (gfx950 / CDNA4) backend for the PLS-PM weight solver + bootstrap hot path.
// C-ABI: include/plspm_hip.h.  Design, data layout and per-kernel rooflines: DESIGN.md.
//
// Pipeline (all fp64, one HIP stream per handle):
//   upload   : raw X -> column means (two-stage reduce) -> Xa = [X - mean | 1 | 0-pad], N x PA, PA % 32 == 0
//   bootstrap: resample_kernel  (Philox indices or explicit idx -> LDS histogram -> ordered (row,count) list)
//              gram_rows/_wide  (fp64 MFMA 16x16x4 weighted Gram  sum_i c_i xa_i xa_i^T, upper tiles, tile-packed)
//              solver_kernel    (csrc/solver_core.h, LDS-resident PLS iteration + inner model + effects + loadings)
//   fit      : gram (dense rows, split over workgroups) -> reduce -> solver -> scores (LDS-staged X tile . W)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <condition_variable>

#include "../../include/plspm_hip.h"
#include "solver_core.h"
#include "solver_nmg.h"
#include "solver_hoc.h"
#include "solver_nmx.h"
#include "solver_ops.h"
#include "solver_wave.h"

using namespace plspm;

typedef double d4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ long lmin(long a, long b) { return a < b ? a : b; }

#include "wave_ops.h"
#include "kernels_input.h"
#include "kernels_gram.h"
#include "kernels_gram_i8.h"
#include "kernels_gram_i8p.h"
#include "kernels_solver.h"
#include "kernels_nonmetric.h"
#include "kernels_post.h"

// ================================================================================================ host side
#include "model.h"

// ------------------------------------------------------------------------------------------------ caching allocator (model.h)
namespace {
struct MemPool {
    std::mutex mu;
    std::multimap<size_t, void*> idle[17];                  // [device], 16 = pinned host
    std::unordered_map<void*, std::pair<int, size_t>> live; // block -> (pool index, capacity)
    size_t idle_bytes[17] = {};
};
MemPool& pool() { static MemPool* p = new MemPool(); return *p; }      // leaked on purpose: must outlive every static destructor
constexpr size_t kPoolKeepPerDevice = (size_t)8 << 30, kPoolMaxBlock = (size_t)1 << 30, kPoolKeepHost = (size_t)1 << 30;
// size classes of 12.5 % (a re-fit of a slightly larger data set still finds its blocks), 256-byte granularity below 2 KiB
size_t size_class(size_t bytes) {
    if (bytes <= 2048) return (bytes + 255) & ~(size_t)255;
    int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (lg - 3);
    return (bytes + step - 1) & ~(step - 1);
}
hipError_t pool_get(int idx, size_t bytes, void** out) {
    const size_t want = size_class(std::max<size_t>(bytes, 1));
    MemPool& P = pool();
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.idle[idx].lower_bound(want);
        if (it != P.idle[idx].end() && it->first <= want + want / 4) {
            *out = it->second;
            P.live[*out] = {idx, it->first};
            P.idle_bytes[idx] -= it->first;
            P.idle[idx].erase(it);
            return hipSuccess;
        }
    }
    void* p = nullptr;
    hipError_t e = (idx == 16) ? hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped) : hipMalloc(&p, want);
    if (e != hipSuccess) {                                  // give the cache back to the runtime and try once more
        plspm_release_cached_memory();
        e = (idx == 16) ? hipHostMalloc(&p, want, hipHostMallocPortable | hipHostMallocMapped) : hipMalloc(&p, want);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lock(P.mu);
    P.live[p] = {idx, want};
    *out = p;
    return hipSuccess;
}
void pool_put(void* p) {
    if (!p) return;
    MemPool& P = pool();
    int idx; size_t cap;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) return;                     // not ours
        idx = it->second.first; cap = it->second.second;
        P.live.erase(it);
        const size_t keep = (idx == 16) ? kPoolKeepHost : kPoolKeepPerDevice;
        if (cap <= kPoolMaxBlock && P.idle_bytes[idx] + cap <= keep) {
            P.idle[idx].emplace(cap, p);
            P.idle_bytes[idx] += cap;
            return;
        }
    }
    if (idx == 16) (void)hipHostFree(p); else (void)hipFree(p);          // hipFree works from any current device
}
}  // namespace

hipError_t plspm_dmalloc(void** p, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipMalloc(p, bytes);
    return pool_get(dev, bytes, p);
}
void plspm_dfree(void* p) { pool_put(p); }
hipError_t plspm_hmalloc(void** p, size_t bytes) { return pool_get(16, bytes, p); }
void plspm_hfree(void* p) { pool_put(p); }

namespace {
struct StreamPool { std::mutex mu; std::vector<hipStream_t> idle[16]; std::unordered_map<hipStream_t, int> live; };
StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }
}  // namespace
hipError_t plspm_stream_acquire(hipStream_t* s) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    StreamPool& P = stream_pool();
    if (dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> lock(P.mu);
        if (!P.idle[dev].empty()) { *s = P.idle[dev].back(); P.idle[dev].pop_back(); P.live[*s] = dev; return hipSuccess; }
    }
    e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    if (e == hipSuccess && dev >= 0 && dev < 16) { std::lock_guard<std::mutex> lock(P.mu); P.live[*s] = dev; }
    return e;
}
void plspm_stream_release(hipStream_t s) {
    if (!s) return;
    StreamPool& P = stream_pool();
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.live.find(s);
        if (it != P.live.end()) {
            const int dev = it->second;
            P.live.erase(it);
            if (P.idle[dev].size() < 32) { P.idle[dev].push_back(s); return; }
        }
    }
    (void)hipStreamDestroy(s);
}

extern "C" int plspm_release_cached_memory(void) {
    MemPool& P = pool();
    std::vector<std::pair<int, void*>> victims;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        for (int i = 0; i < 17; ++i) { for (auto& kv : P.idle[i]) victims.emplace_back(i, kv.second); P.idle[i].clear(); P.idle_bytes[i] = 0; }
    }
    for (auto& v : victims) { if (v.first == 16) (void)hipHostFree(v.second); else (void)hipFree(v.second); }
    return 0;
}

static int prepare_zs(plspm_model* m);      // digit planes of the resident data (below; plspm_fit cuts them in its tail when asked to)

static ModelDesc make_desc(const plspm_model* m) {
    ModelDesc md{};
    md.P = m->P; md.L = m->L; md.PA = m->PAs; md.T = m->Ts; md.scheme = m->scheme; md.scaled = m->scaled; md.max_iter = m->max_iter;
    md.kmax = m->kmax; md.n_eff = m->n_eff; md.n_chol = m->n_chol; md.tol = m->tol;
    md.boff = m->d_boff; md.lvof = m->d_lvof; md.C = m->d_C; md.mode = m->d_mode; md.chol_off = m->d_chol_off;
    md.eff_from = m->d_eff_from; md.eff_to = m->d_eff_to; md.shift = m->d_shift;
    md.pred_off = m->d_pred_off; md.pred_idx = m->d_pred_idx; md.succ_off = m->d_succ_off; md.succ_idx = m->d_succ_idx;
    md.n_edges = (int)m->pred_idx.size();
    md.tile_tu = nullptr;
    return md;
}

static HocDesc make_hoc_desc(const plspm_model* m2) {
    const plspm_model* m1 = m2->stage1;
    HocDesc hd{};
    hd.P1 = m1->P; hd.L1 = m1->L; hd.P2 = m2->P; hd.L2 = m2->L; hd.T1 = m1->T; hd.T2 = m2->Ts;
    hd.boff1 = m1->d_boff; hd.boff2 = m2->d_boff; hd.lv_first = m2->d_lv_first; hd.col2_lv1 = m2->d_col2_lv1; hd.col2_p1 = m2->d_col2_p1;
    hd.nh = (int)m2->hcol.size(); hd.hcol = m2->d_hcol; hd.hidx = m2->d_hidx;
    return hd;
}

// Padded width of the resident matrix [columns | 1 | 0-pad] and its tile count.  Pairs of 16-column tiles (32-column groups, one
// 16-byte load per lane) by default; METRIC models whose width leaves the last group half empty run an odd tile count on the
// gram_wide kernels (5 <= T <= 15): P + 1 = 201 columns -> 13 tiles instead of 14, i.e. 91 instead of 105 MFMAs per k-group and
// 7 % fewer bytes per row for every streaming pass.  The non-metric kernels (tiled copies, stop-rule passes) keep whole groups.
static int tiles_for(const plspm_model* m, int cols) {
    const int t16 = (cols + 1 + 15) / 16;
    const bool odd_ok = !m->nonmetric && (t16 & 1) && t16 >= 5 && t16 <= 15;
    return odd_ok ? t16 : 2 * ((cols + 1 + 31) / 32);
}
// T / PA: the Gram of the uploaded columns (Pg = data + missing-indicator columns); Ts / PAs: the P-column matrix the solver reads.
static void set_geometry(plspm_model* m) {
    m->T = tiles_for(m, m->Pg); m->PA = 16 * m->T;
    m->Ts = tiles_for(m, m->P); m->PAs = 16 * m->Ts;
}

// Side tables of plspm_model_set_incomplete_rows: freed (and nulled) on every re-upload and on a failed set call.
static void drop_incomplete_rows(plspm_model* m) {
    if (m->d_Xk) plspm_dfree(m->d_Xk);
    if (m->d_Mk) plspm_dfree(m->d_Mk);
    if (m->d_rowid) plspm_dfree(m->d_rowid);
    m->d_Xk = m->d_Mk = nullptr; m->d_rowid = nullptr; m->nmx_K = 0;
}

template <class Tv>
static int upload_vec(plspm_model* m, Tv** dst, const std::vector<Tv>& v) {
    const size_t bytes = std::max<size_t>(1, v.size()) * sizeof(Tv);
    HIPCHK(m, plspm_dmalloc((void**)dst, bytes));
    if (!v.empty()) HIPCHK(m, hipMemcpy(*dst, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice));
    return 0;
}

extern "C" {

int plspm_abi_version(void) { return PLSPM_ABI_VERSION; }

int plspm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* plspm_last_error(const plspm_model_t* m) { return m ? m->error.c_str() : g_create_error.c_str(); }

plspm_model_t* plspm_model_create(int32_t P, int32_t L, const int32_t* block_offset, const uint8_t* path, const int32_t* mode, int32_t scheme,
                                  int32_t scaled, int32_t max_iter, double tol, int32_t device_id) {
    g_create_error.clear();
    if (!block_offset || !path || !mode) { fail(nullptr, PLSPM_E_ARG, "null argument"); return nullptr; }
    if (L < 1 || L > 64 || P < L || P > 1022) { fail(nullptr, PLSPM_E_LIMIT, "limits: 1 <= L <= 64, L <= P <= 1022"); return nullptr; }
    if (scheme < 0 || scheme > 2 || !(tol > 0.0) || max_iter < 1) { fail(nullptr, PLSPM_E_ARG, "bad scheme / tolerance / max_iter"); return nullptr; }
    if (block_offset[0] != 0 || block_offset[L] != P) { fail(nullptr, PLSPM_E_ARG, "block_offset must run from 0 to P"); return nullptr; }
    for (int l = 0; l < L; ++l) {
        if (block_offset[l + 1] <= block_offset[l]) { fail(nullptr, PLSPM_E_ARG, "every LV needs at least one MV"); return nullptr; }
        if (mode[l] != PLSPM_MODE_A && mode[l] != PLSPM_MODE_B) { fail(nullptr, PLSPM_E_ARG, "mode must be A(0) or B(1)"); return nullptr; }
        for (int j = 0; j < L; ++j) {
            if (path[l * L + j] > 1) { fail(nullptr, PLSPM_E_ARG, "path entries must be 0/1"); return nullptr; }
            if (j >= l && path[l * L + j]) { fail(nullptr, PLSPM_E_ARG, "path matrix must be strictly lower triangular"); return nullptr; }
        }
    }
    int ndev = plspm_device_count();
    if (ndev <= 0) { fail(nullptr, PLSPM_E_STATE, "no HIP device visible: libplspm_hip has no CPU fallback"); return nullptr; }
    if (device_id < 0 || device_id >= ndev) { fail(nullptr, PLSPM_E_ARG, "device_id out of range"); return nullptr; }
    plspm_model* m = new (std::nothrow) plspm_model();
    if (!m) { fail(nullptr, PLSPM_E_STATE, "out of host memory"); return nullptr; }
    m->device = device_id; m->P = P; m->L = L; m->scheme = scheme; m->scaled = scaled ? 1 : 0; m->max_iter = max_iter; m->tol = tol;
    m->Pg = P;
    set_geometry(m);
    m->boff.assign(block_offset, block_offset + L + 1);
    m->mode.assign(mode, mode + L);
    m->C.assign(path, path + (size_t)L * L);
    m->lvof.resize(P); m->chol_off.assign(L, -1);
    for (int l = 0; l < L; ++l) {
        for (int p = m->boff[l]; p < m->boff[l + 1]; ++p) m->lvof[p] = l;
        int k = 0;
        for (int j = 0; j < L; ++j) k += m->C[l * L + j] ? 1 : 0;
        m->kmax = std::max(m->kmax, k);
        if (mode[l] == PLSPM_MODE_B) { const int kb = m->boff[l + 1] - m->boff[l]; m->chol_off[l] = m->n_chol; m->n_chol += (int)chol_block_doubles(kb); }
    }
    m->pred_off.assign(L + 1, 0); m->succ_off.assign(L + 1, 0);
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < L; ++j) if (m->C[i * L + j]) m->pred_idx.push_back(j);
        m->pred_off[i + 1] = (int)m->pred_idx.size();
        for (int s2 = 0; s2 < L; ++s2) if (m->C[s2 * L + i]) m->succ_idx.push_back(s2);
        m->succ_off[i + 1] = (int)m->succ_idx.size();
    }
    // transitive closure -> effect rows (from-major), inner_model.py:46-52
    std::vector<uint8_t> reach(m->C);
    for (int k = 0; k < L; ++k) for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j)
        if (reach[i * L + k] && reach[k * L + j]) reach[i * L + j] = 1;
    for (int f = 0; f < L; ++f) for (int t = 0; t < L; ++t)
        if (f != t && reach[t * L + f]) { m->eff_from.push_back(f); m->eff_to.push_back(t); }
    m->n_eff = (int)m->eff_from.size();

    auto bail = [&](const std::string& why) { g_create_error = why + (m->error.empty() ? "" : (": " + m->error)); plspm_model_destroy(m); return (plspm_model_t*)nullptr; };
    if (hipSetDevice(device_id) != hipSuccess) return bail("hipSetDevice failed");
    if (plspm_stream_acquire(&m->stream) != hipSuccess) return bail("hipStreamCreate failed");
    if (upload_vec(m, &m->d_boff, m->boff) || upload_vec(m, &m->d_lvof, m->lvof) || upload_vec(m, &m->d_mode, m->mode) ||
        upload_vec(m, &m->d_chol_off, m->chol_off) || upload_vec(m, &m->d_eff_from, m->eff_from) || upload_vec(m, &m->d_eff_to, m->eff_to) ||
        upload_vec(m, &m->d_C, m->C) || upload_vec(m, &m->d_pred_off, m->pred_off) || upload_vec(m, &m->d_pred_idx, m->pred_idx) ||
        upload_vec(m, &m->d_succ_off, m->succ_off) || upload_vec(m, &m->d_succ_idx, m->succ_idx))
        return bail("descriptor upload failed");
    if (plspm_dmalloc((void**)&m->d_shift, sizeof(double) * P) != hipSuccess) return bail("hipMalloc failed");
    if (plspm_hmalloc((void**)&m->h_flag, 64) != hipSuccess || hipEventCreateWithFlags(&m->ev_flag, hipEventDisableTiming) != hipSuccess)
        return bail("pinned flag / event creation failed");
    return m;
}

void plspm_model_destroy(plspm_model_t* m) {
    if (!m) return;
    if (m->group) plspm_detail_group_orphan(m->group);                                // host objects die in arbitrary order
    if (m->stage2) { m->stage2->stage1 = nullptr; m->stage2->stream = nullptr; }      // the pair is dissolved; the survivor is inert
    if (m->stage1) m->stage1->stage2 = nullptr;
    hipSetDevice(m->device);
    if (m->aux) hipStreamSynchronize(m->aux);
    if (m->stream) hipStreamSynchronize(m->stream);
    prof_collect(m);
    void* ptrs[] = {m->d_boff, m->d_lvof, m->d_mode, m->d_chol_off, m->d_eff_from, m->d_eff_to, m->d_C, m->d_shift, m->xa.p, m->up_raw.p, m->up_ci.p, m->up_partial.p, m->scores.p,
                    m->d_pred_off, m->d_pred_idx, m->d_succ_off, m->d_succ_idx, m->d_mv_off, m->d_mv_kind, m->d_lmv_off, m->d_mv_lv, m->d_mv_base, m->d_mv_base2, m->d_lmv2_off, m->d_no_chol, m->gSm.p, m->d_ind_of, m->gram2.p, m->d_lv_cols, m->d_lv_first, m->d_col2_lv1, m->d_col2_p1, m->d_hcol, m->d_hidx, m->pseudo.p, m->d_Xk, m->d_Mk, m->d_rowid, m->dcnt.p, m->ctable.p, m->Xt.p,
                    m->ent.p, m->nent.p, m->gram.p, m->gram_partial.p, m->rows.p, m->status.p, m->iters.p, m->gS.p, m->gsmall.p,
                    m->fitout.p, m->idx.p, m->err.p, m->ghist.p, m->nmstate.p, m->nmpartial.p, m->nmactive.p, m->nmlist.p, m->gK16.p, m->sum_buf.p, m->cols.p,
                    m->zs.p, m->cd.p, m->cd1.p, m->codes.p, m->err2.p, m->sk_partial.p, m->sk_flags.p, m->pair_tab.p, m->pair_scale.p, m->zs_stat.p};
    for (void* p : ptrs) if (p) plspm_dfree(p);
    if (m->h_stage) plspm_hfree(m->h_stage);
    if (m->h_pin) plspm_hfree(m->h_pin);
    for (int k = 0; k < 2; ++k) if (m->ev_pin[k]) hipEventDestroy(m->ev_pin[k]);
    for (int k = 0; k < PLSPM_K_COUNT; ++k) for (auto& pr : m->prof[k].pool) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (m->h_flag) plspm_hfree(m->h_flag);
    if (m->h_zstat) plspm_hfree(m->h_zstat);
    if (m->ev_zstat) hipEventDestroy(m->ev_zstat);
    if (m->ev_flag) hipEventDestroy(m->ev_flag);
    for (int k = 0; k < 2; ++k) { if (m->ev_counts[k]) hipEventDestroy(m->ev_counts[k]); if (m->ev_cdfree[k]) hipEventDestroy(m->ev_cdfree[k]); }
    if (m->aux) hipStreamDestroy(m->aux);                                   // (synchronised above; a low-priority stream of its own, not from the cache)
    if (m->stream && m->owns_stream) plspm_stream_release(m->stream);       // (synchronised above)
    delete m;
}

int32_t plspm_effect_pairs(const plspm_model_t* m, int32_t* from, int32_t* to) {
    if (!m) return 0;
    for (int e = 0; e < m->n_eff; ++e) { if (from) from[e] = m->eff_from[e]; if (to) to[e] = m->eff_to[e]; }
    return m->n_eff;
}
int32_t plspm_row_width(const plspm_model_t* m) {
    if (!m) return 0;
    if (m->stage2) m = m->stage2;               // two-stage handles report the second stage's rows
    return 2 * (m->categorical ? m->Pm : m->P) + m->L + 2 * m->n_eff;
}
int32_t plspm_row_stride(const plspm_model_t* m) { return m ? plspm_row_width(m) + 2 : 0; }

int plspm_upload(plspm_model_t* m, const double* X, int64_t N, int32_t src_cols, int32_t layout, const int32_t* col_index) {
    if (!m) return PLSPM_E_ARG;
    if (!X || N < 2 || src_cols < 1 || (layout != 0 && layout != 1)) return fail(m, PLSPM_E_ARG, "plspm_upload: bad arguments");
    if (N > 0x7fffffffLL - 64) return fail(m, PLSPM_E_LIMIT, "plspm_upload: N must fit in int32");
    const int Pg = m->Pg;                  // data columns (+ missing-indicator columns, plspm_model_set_missing)
    std::vector<int> ci(Pg);
    for (int p = 0; p < Pg; ++p) {
        ci[p] = col_index ? col_index[p] : p;
        if (ci[p] < 0 || ci[p] >= src_cols) return fail(m, PLSPM_E_ARG, "plspm_upload: col_index out of range");
    }
    HIPCHK(m, hipSetDevice(m->device));
    if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    // whatever was resident is gone from here on (a failed upload leaves an empty, re-usable handle)
    m->N = 0; m->d_Xa = nullptr; m->Xt_valid = false; m->codes_valid = false; if (m->stage2) m->stage2->codes_valid = false; m->rows_B = 0; m->dcnt_ready = false; m->zs_valid = false; m->zs_stats_ready = false;
    drop_incomplete_rows(m);
    // persistent grow-only buffers: a repeated upload of the same shape allocates nothing
    const size_t raw_bytes = (size_t)N * src_cols * sizeof(double);
    const int nblk = (int)std::min<int64_t>(1024, (N + 255) / 256);
    int rc;
    if ((rc = ensure(m, m->up_raw, raw_bytes))) return rc;
    if ((rc = ensure(m, m->up_ci, sizeof(int) * (size_t)Pg))) return rc;
    if ((rc = ensure(m, m->up_partial, sizeof(double) * (size_t)nblk * Pg))) return rc;
    if ((rc = ensure(m, m->xa, (size_t)(N + 1) * m->PA * sizeof(double)))) return rc;       // + one all-zero pad row (dense Gram walks read it past the end)
    double* d_raw = (double*)m->up_raw.p;
    int* d_ci = (int*)m->up_ci.p;
    double* d_partial = (double*)m->up_partial.p;
    double* d_Xa = (double*)m->xa.p;
    // host -> device through the handle's pinned staging halves (no page pinning per call); from 64 MB on several host threads fill
    // every half (plspm_detail_h2d).  The runtime's own pageable path ("upload_direct" 1) moved configs[4]'s 1.6 GB at 52 GB/s on one
    // box and at 13-15 GB/s on two others (its staging copy is one thread's memcpy)
    if (raw_bytes <= ((size_t)64 << 20) || !m->tune.upload_direct) { if ((rc = plspm_detail_h2d(m, d_raw, X, raw_bytes))) return rc; }
    else HIPCHK(m, hipMemcpyAsync(d_raw, X, raw_bytes, hipMemcpyHostToDevice, m->stream));
    if ((rc = plspm_detail_h2d(m, d_ci, ci.data(), sizeof(int) * (size_t)Pg))) return rc;
    {
        ProfScope ps(m, PLSPM_K_PACK);
        if (layout == 0) {
            hipLaunchKernelGGL(colsum_rowmajor_kernel, dim3(nblk), dim3(256), 0, m->stream, d_raw, (long)N, (int)src_cols, d_ci, Pg, d_partial);
        } else {
            hipLaunchKernelGGL(colsum_colmajor_kernel, dim3(nblk, Pg), dim3(256), 0, m->stream, d_raw, (long)N, d_ci, Pg, d_partial);
        }
        hipLaunchKernelGGL(colmean_kernel, dim3((Pg + 63) / 64), dim3(64), 0, m->stream, d_partial, nblk, Pg, (long)N, m->d_shift);
        if (m->categorical) hipMemsetAsync(m->d_shift, 0, sizeof(double) * Pg, m->stream);         // aug columns stay raw (solver_nmg.h)
        if (m->n_ind) hipMemsetAsync(m->d_shift + m->P, 0, sizeof(double) * m->n_ind, m->stream);   // so do the 0/1 missing indicators
        if (layout == 0) {
            const long total = (long)N * m->PA;
            const int grid = (int)std::min<long>(4096, (total + 255) / 256);
            hipLaunchKernelGGL(pack_rowmajor_kernel, dim3(grid), dim3(256), 0, m->stream, d_raw, (long)N, (int)src_cols, d_ci, Pg, m->PA, m->d_shift, d_Xa);
        } else {
            hipLaunchKernelGGL(pack_colmajor_kernel, dim3((unsigned)((N + 63) / 64), (m->PA + 31) / 32), dim3(256), 0, m->stream, d_raw, (long)N, d_ci, Pg, m->PA,
                               m->d_shift, d_Xa);
        }
    }
    HIPCHK(m, hipMemsetAsync(d_Xa + (size_t)N * m->PA, 0, (size_t)m->PA * sizeof(double), m->stream));
    HIPCHK(m, hipGetLastError());
    HIPCHK(m, hipStreamSynchronize(m->stream));          // X may be released by the caller on return
    if (raw_bytes > ((size_t)1 << 30)) {                 // do not keep a multi-GB staging copy alive next to the resident matrix
        plspm_dfree(m->up_raw.p);
        m->up_raw.p = nullptr; m->up_raw.cap = 0;
    }
    m->d_Xa = d_Xa;
    m->N = N;
    return 0;
}

}  // extern "C"

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
        case 13: WIDE(13, 4, 4) break;
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

static size_t desc_lds_bytes(int P, int L, int ne, int nedge) {
    const size_t T = ((size_t)P + 1 + 31) / 32 * 2, ntile = T * (T + 1) / 2;
    return (size_t)P * 8 + (3 * (size_t)(L + 1) + P + 2 * (size_t)L + 2 * (size_t)ne + 2 * (size_t)nedge + (ntile + 1) / 2 + 4) * 4 + (((size_t)L * L + 15) & ~(size_t)15) + 16;
}

// Missing-data models: collapse the aug Gram(s) at `Min` into mean-imputed P-column moments; returns the matrix the solver reads.
static int run_impute(plspm_model* m, long nproblems, const double* Min, const double** Mp, long* mp_stride) {
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

static int launch_solver(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so, int threads) {
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
static int dense_moments(plspm_model* m) {
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

// Non-metric solve of `nproblems` problems whose packed scatter matrices are at Mp: prepare -> (step, convergence pass)* ->
// finish.  The host only reads one counter per iteration (how many problems are still active).
// doubles of per-problem solver state of a non-metric handle (NmState head + what its solver keeps behind it)
static size_t nm_state_doubles_of(const plspm_model* m) {
    return m->categorical ? (size_t)nmg_state_doubles(m->P, m->Pm, m->L, m->cmax, m->kmv)
                          : m->nmx_K > 0 ? (size_t)nmx_state_doubles(m->P, m->L, m->n_chol, m->nmx_K) : (size_t)nm_state_doubles(m->P, m->L, m->n_chol);
}

// LDS footprint of the dense stop-rule pass (nm_conv_dense_kernel) for this handle: the coefficient tile of 64 replicates whole, or one LV
// block at a time; 0 when neither fits or the option forbids the pass
static size_t nm_dense_lds(const plspm_model* m, bool* whole, int* kb_out) {
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

// cd8 / cd8_MT: the int8 row multiplicities of THESE problems (the counts the digit-plane Gram consumed; bootstrap only), or null
static int run_nonmetric(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so_in, const int2* ent, const int* nent,
                         long ent_stride, int threads, bool finish = true, const void* cd8 = nullptr, int cd8_MT = 0) {
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
    const int nparts = dense ? (int)ntiles16 : (int)std::max<long>(1, std::min<long>(nproblems == 1 ? 1024 : 8, (N + 1023) / 1024));
    const int ngroups = (int)((nproblems + 63) / 64);
    if (dense) {
        if ((rc = ensure(m, src->Xt, (size_t)ntiles16 * 16 * src->PA * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->ctable, (size_t)ngroups * table_rows * 64 * sizeof(double)))) return rc;
        if ((rc = ensure(m, m->nmlist, ((size_t)nproblems + 1) * sizeof(int)))) return rc;
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
    m->last_nm_codes = use_codes ? 1 : 0;
    if ((rc = ensure(m, m->gS, (size_t)nproblems * s_bytes))) return rc;
    if (cat && (rc = ensure(m, m->gSm, (size_t)nproblems * cov_doubles(m->Pm) * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->nmstate, (size_t)nproblems * st_doubles * sizeof(double)))) return rc;
    // all-indicator categorical models of at most 65,535 rows: a uint16 copy of every problem's count matrix for the streaming product of the step
    const int ld16 = (P + 1 + 3) & ~3;
    const bool k16 = cat && m->cat_pure && N <= 65535 && m->tune.nm_k16 != 0;
    if (k16 && (rc = ensure(m, m->gK16, (size_t)nproblems * (P + 1) * ld16 * sizeof(unsigned short)))) return rc;
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
        if ((rc = allow_lds(m, (const void*)nmg_kernel<0>, lds)) || (rc = allow_lds(m, (const void*)nmg_kernel<1>, lds)) || (rc = allow_lds(m, (const void*)nmg_kernel<2>, lds)))
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
    auto launch = [&](int mode_op) {
        ProfScope ps(m, PLSPM_K_SOLVER);
        if (cat) {
            auto k = mode_op == 0 ? nmg_kernel<0> : mode_op == 1 ? nmg_kernel<1> : nmg_kernel<2>;
            hipLaunchKernelGGL(k, grid, dim3(threads), lds, m->stream, md, cd, mdm, Mp, mp_stride, so, gS, gSm, gst, (long)st_doubles, (const double*)part, nparts, nact, fuse, cat_fast,
                               k16 ? (unsigned short*)m->gK16.p : (unsigned short*)nullptr, ld16);
        } else if (nmx) {
            auto k = mode_op == 0 ? nmx_kernel<0> : mode_op == 1 ? nmx_kernel<1> : nmx_kernel<2>;
            const MissDesc xd{m->nmx_raw, m->nmx_K, m->d_Xk, m->d_Mk};
            hipLaunchKernelGGL(k, grid, dim3(threads), lds, m->stream, md, xd, (const int*)m->d_rowid, Mp, mp_stride, so, gS, gst, (long)st_doubles, (const double*)part, nparts,
                               nact, ent, nent, ent_stride, fuse);
        } else {
            auto k = mode_op == 0 ? nm_kernel<0> : mode_op == 1 ? nm_kernel<1> : nm_kernel<2>;
            hipLaunchKernelGGL(k, grid, dim3(threads), lds, m->stream, md, Mp, mp_stride, so, gS, gst, (const double*)part, nparts, nact, fuse);
        }
    };
    // (dense stop-rule pass: the list kernel of the pass counts the live problems anyway and writes the count to the pinned flag itself -- no
    //  counter to clear, no copy operation: two tiny launches and their gaps less per iteration, 35 of ~590 us at three iterations)
    const bool flag_from_list = dense && !m->stage1;
    for (int it = 0; it <= m->max_iter + 1; ++it) {
        if (!flag_from_list) HIPCHK(m, hipMemsetAsync(nact, 0, sizeof(int), m->stream));
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
                hipLaunchKernelGGL(hoc_compose_kernel, grid, dim3(64), 0, m->stream, make_hoc_desc(m), (const double*)m->stage1->nmstate.p,
                                   (long)nm_state_doubles_of(src), gst, (long)st_doubles, m->n_chol, (double*)m->pseudo.p, ps_stride);
                conv_state = (const double*)m->pseudo.p; conv_stride = ps_stride; conv_boff = m->d_lv_cols;
            }
            if (dense) {
                int* live_list = (int*)m->nmlist.p;                            // [count | ids of the problems still iterating, in problem order]
                hipLaunchKernelGGL(active_list_kernel, dim3(1), dim3(1024), 0, m->stream, conv_state, conv_stride, nproblems, live_list + 1, live_list,
                                   flag_from_list ? (int*)m->h_flag : (int*)nullptr);
                if (flag_from_list) HIPCHK(m, hipEventRecord(m->ev_flag, m->stream));
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
                                   (const double*)m->ctable.p, (const int*)((int*)m->nmlist.p + 1), (const int*)m->nmlist.p, part, nparts, rbx, gy, kb);
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

extern "C" {

int plspm_model_set_nonmetric(plspm_model_t* m, int32_t on) {
    if (!m) return PLSPM_E_ARG;
    if (m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_model_set_nonmetric: call before plspm_upload");
    m->nonmetric = on ? 1 : 0;
    set_geometry(m);                       // the non-metric kernels keep whole 32-column groups
    return 0;
}

void* plspm_stream(plspm_model_t* m) { return m ? (void*)m->stream : nullptr; }

int plspm_model_set_option(plspm_model_t* m, const char* key, int32_t value) {
    if (!m || !key) return fail(m, PLSPM_E_ARG, "plspm_model_set_option: bad arguments");
    const std::string k(key);
    auto bad = [&]() { return fail(m, PLSPM_E_ARG, "plspm_model_set_option: value out of range for '" + k + "'"); };
    if (k == "solver_threads") { if (value != 64 && value != 128 && value != 256) return bad(); m->tune.solver_threads = value; }
    else if (k == "nm_threads") { if (value != 0 && value != 64 && value != 128 && value != 256) return bad(); m->tune.nm_threads = value; }
    else if (k == "fit_chunks") { if (value < 0 || value > 65535) return bad(); m->tune.fit_chunks = value; }
    else if (k == "wide_nw") { if (value != 4 && value != 8 && value != 16) return bad(); m->tune.wide_nw = value; }
    else if (k == "conv_pass") { if (value < 0 || value > 2) return bad(); m->tune.conv_pass = value; }
    else if (k == "scores_tile") { if (value != 0 && value != 16 && value != 32) return bad(); m->tune.scores_tile = value; }
    else if (k == "gram_lds_kb") { if (value < 0 || value > 160) return bad(); m->tune.gram_lds_kb = value; }
    else if (k == "gram_path") { if (value < 0 || value > 2) return bad(); m->tune.gram_path = value; }
    else if (k == "i8_slices") { if (value < 0 || value > 8) return bad(); if (value != m->tune.i8_slices) m->zs_valid = false; m->tune.i8_slices = value; }
    else if (k == "i8_min_batch") { if (value < 1) return bad(); m->tune.i8_min_batch = value; }
    else if (k == "i8_waves") { if (value != 4 && value != 8) return bad(); m->tune.i8_waves = value; }
    else if (k == "nm_fast_lds") { if (value < 0 || value > 1) return bad(); m->tune.nm_fast_lds = value; }
    else if (k == "nm_k16") { if (value < 0 || value > 1) return bad(); m->tune.nm_k16 = value; }
    else if (k == "nm_codes") { if (value < 0 || value > 1) return bad(); m->tune.nm_codes = value; }
    else if (k == "i8_ind") { if (value < 0 || value > 1) return bad(); if (value != m->tune.i8_ind) m->zs_valid = false; m->tune.i8_ind = value; }
    else if (k == "upload_direct") { if (value < 0 || value > 1) return bad(); m->tune.upload_direct = value; }
    else if (k == "i8_short_rows") { if (value < -1 || value > 4096) return bad(); m->tune.i8_short = value; }
    else if (k == "i8_rt") { if (value != 0 && value != 16 && value != 8 && value != 20) return bad(); m->tune.i8_rt = value; }
    else if (k == "solver_rows") { if (value != 0 && value != 1) return bad(); m->tune.solver_rows = value; }
    else if (k == "solver_wave") { if (value != 0 && value != 1) return bad(); m->tune.solver_wave = value; }
    else if (k == "nm_counts8") { if (value != 0 && value != 1) return bad(); m->tune.nm_counts8 = value; }
    else if (k == "resample_aux") { if (value < 0 || value > 3) return bad(); m->tune.resample_aux = value; }
    else if (k == "i8_sched") { if (value != 0 && value != 1) return bad(); m->tune.i8_sched = value; }
    else if (k == "i8_priv") { if (value < 0 || value > 1) return bad(); m->tune.i8_priv = value; }
    else if (k == "i8_shape") { if (value != 16 && value != 32) return bad(); if (value != m->tune.i8_shape) m->zs_valid = false; m->tune.i8_shape = value; }
    else if (k == "i8_variant") { if (value < -1 || value > 899) return bad(); m->tune.i8_variant = value; }
    else if (k == "i8_dma") { if (value < 0 || value > 2) return bad(); m->tune.i8_dma = value; }
    else if (k == "conv_gy") { if (value < 0 || value > 65535) return bad(); m->tune.conv_gy = value; }
    else return fail(m, PLSPM_E_ARG, "plspm_model_set_option: unknown option '" + k + "'");
    return 0;
}

int plspm_model_get_option(const plspm_model_t* m, const char* key, int32_t* value) {
    if (!m || !key || !value) return PLSPM_E_ARG;
    const std::string k(key);
    if (k == "solver_threads") *value = m->tune.solver_threads;
    else if (k == "nm_threads") *value = m->tune.nm_threads;
    else if (k == "fit_chunks") *value = m->tune.fit_chunks;
    else if (k == "wide_nw") *value = m->tune.wide_nw;
    else if (k == "conv_pass") *value = m->tune.conv_pass;
    else if (k == "scores_tile") *value = m->tune.scores_tile;
    else if (k == "gram_lds_kb") *value = m->tune.gram_lds_kb;
    else if (k == "conv_gy") *value = m->tune.conv_gy;
    else if (k == "gram_path") *value = m->tune.gram_path;
    else if (k == "i8_slices") *value = m->tune.i8_slices;
    else if (k == "last_i8_slices") *value = m->zs_valid ? m->zs_S : 0;
    else if (k == "last_i8_rt") *value = m->last_i8_rt;
    else if (k == "last_i8_short") *value = m->last_i8_short;
    else if (k == "last_i8_mt") *value = m->last_i8_mt;
    else if (k == "last_i8_ratio") *value = (m->zs_valid && m->zs_ratio < 9e18) ? (int64_t)m->zs_ratio : 0;      // floor of the smallest sum|z| / max|z| (automatic plane count)
    else if (k == "i8_min_batch") *value = m->tune.i8_min_batch;
    else if (k == "i8_waves") *value = m->tune.i8_waves;
    else if (k == "nm_fast_lds") *value = m->tune.nm_fast_lds;
    else if (k == "nm_k16") *value = m->tune.nm_k16;
    else if (k == "nm_codes") *value = m->tune.nm_codes;
    else if (k == "last_nm_codes") *value = m->last_nm_codes;
    else if (k == "i8_ind") *value = m->tune.i8_ind;
    else if (k == "i8_rt") *value = m->tune.i8_rt;
    else if (k == "i8_short_rows") *value = m->tune.i8_short;
    else if (k == "upload_direct") *value = m->tune.upload_direct;
    else if (k == "i8_dma") *value = m->tune.i8_dma;
    else if (k == "last_i8_dma") *value = m->last_i8_dma;
    else if (k == "solver_rows") *value = m->tune.solver_rows;
    else if (k == "solver_wave") *value = m->tune.solver_wave;
    else if (k == "nm_counts8") *value = m->tune.nm_counts8;
    else if (k == "last_solver") *value = m->last_solver;
    else if (k == "resample_aux") *value = m->tune.resample_aux;
    else if (k == "i8_sched") *value = m->tune.i8_sched;
    else if (k == "i8_priv") *value = m->tune.i8_priv;
    else if (k == "last_i8_priv") *value = m->last_i8_priv;
    else if (k == "i8_shape") *value = m->tune.i8_shape;
    else if (k == "last_gram_path") *value = m->last_gram_path;
    else return PLSPM_E_ARG;
    return 0;
}

int plspm_model_set_categorical(plspm_model_t* m, int32_t Pm, const int32_t* mv_off, const int32_t* mv_kind) {
    if (!m || !mv_off || !mv_kind || Pm < m->L || Pm > m->P) return fail(m, PLSPM_E_ARG, "plspm_model_set_categorical: bad arguments");
    if (m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_model_set_categorical: call before plspm_upload");
    if (mv_off[0] != 0 || mv_off[Pm] != m->P) return fail(m, PLSPM_E_ARG, "mv_off must run from 0 to P");
    m->mv_off.assign(mv_off, mv_off + Pm + 1);
    m->mv_kind.assign(mv_kind, mv_kind + Pm);
    m->lmv_off.assign(m->L + 1, 0);
    m->mv_lv.assign(Pm, 0);
    m->cmax = 1; m->kmv = 1;
    int l = 0;
    for (int p = 0; p < Pm; ++p) {
        const int C = mv_off[p + 1] - mv_off[p];
        if (C < 1 || mv_kind[p] < 0 || mv_kind[p] > 2 || (mv_kind[p] == 0 && C != 1)) return fail(m, PLSPM_E_ARG, "bad MV column range / kind");
        while (l < m->L && mv_off[p] >= m->boff[l + 1]) ++l;
        if (l >= m->L || mv_off[p + 1] > m->boff[l + 1]) return fail(m, PLSPM_E_ARG, "an MV's columns must lie inside one LV block");
        m->mv_lv[p] = l;
        m->lmv_off[l + 1] = p + 1;
        m->cmax = std::max(m->cmax, C);
    }
    for (int k = 1; k <= m->L; ++k) { if (m->lmv_off[k] < m->lmv_off[k - 1]) m->lmv_off[k] = m->lmv_off[k - 1]; m->kmv = std::max(m->kmv, m->lmv_off[k] - m->lmv_off[k - 1]); }
    for (int k = 0; k < m->L; ++k) if (m->lmv_off[k + 1] == m->lmv_off[k]) return fail(m, PLSPM_E_ARG, "every LV needs at least one MV");
    m->no_chol.assign(m->L, -1);
    HIPCHK(m, hipSetDevice(m->device));
    m->mv_base.assign(Pm, 0);
    for (int p = 0; p < Pm; ++p) m->mv_base[p] = m->boff[m->mv_lv[p]];          // (category codes are filed relative to the MV's LV block: nm_conv_codes_kernel)
    if (upload_vec(m, &m->d_mv_base, m->mv_base)) return fail(m, PLSPM_E_STATE, "descriptor upload failed: " + m->error);
    if (upload_vec(m, &m->d_mv_off, m->mv_off) || upload_vec(m, &m->d_mv_kind, m->mv_kind) || upload_vec(m, &m->d_lmv_off, m->lmv_off) ||
        upload_vec(m, &m->d_mv_lv, m->mv_lv) || upload_vec(m, &m->d_no_chol, m->no_chol))
        return fail(m, PLSPM_E_STATE, "descriptor upload failed");
    m->Pm = Pm; m->categorical = 1; m->nonmetric = 1; m->n_chol = 0;
    m->cat_pure = true;
    for (int p = 0; p < Pm; ++p) if (mv_kind[p] == 0) m->cat_pure = false;             // a NUM / RAW column: real-valued moments
    set_geometry(m);
    for (auto& c : m->chol_off) c = -1;
    HIPCHK(m, hipMemcpy(m->d_chol_off, m->chol_off.data(), sizeof(int) * m->L, hipMemcpyHostToDevice));
    return 0;
}

int plspm_model_set_missing(plspm_model_t* m, int32_t n_ind, const int32_t* ind_of) {
    if (!m || !ind_of || n_ind < 1 || n_ind > m->P) return fail(m, PLSPM_E_ARG, "plspm_model_set_missing: bad arguments");
    if (m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_model_set_missing: call before plspm_upload");
    if (m->nonmetric) return fail(m, PLSPM_E_STATE, "plspm_model_set_missing: mean imputation is the METRIC treatment (config.py:300)");
    if (m->P + n_ind > 1022) return fail(m, PLSPM_E_LIMIT, "limits: P + n_ind <= 1022");
    std::vector<char> seen(n_ind, 0);
    for (int p = 0; p < m->P; ++p) {
        const int c = ind_of[p];
        if (c == -1) continue;
        if (c < m->P || c >= m->P + n_ind || seen[c - m->P]) return fail(m, PLSPM_E_ARG, "ind_of: -1 or a distinct column in [P, P + n_ind)");
        seen[c - m->P] = 1;
    }
    for (char f : seen) if (!f) return fail(m, PLSPM_E_ARG, "ind_of: every indicator column needs a data column");
    HIPCHK(m, hipSetDevice(m->device));
    m->ind_of.assign(ind_of, ind_of + m->P);
    if (upload_vec(m, &m->d_ind_of, m->ind_of)) return fail(m, PLSPM_E_STATE, "descriptor upload failed");
    m->n_ind = n_ind; m->Pg = m->P + n_ind;
    set_geometry(m);
    plspm_dfree(m->d_shift);
    m->d_shift = nullptr;
    HIPCHK(m, plspm_dmalloc((void**)&m->d_shift, sizeof(double) * m->Pg));
    return 0;
}

int plspm_model_attach_second_stage(plspm_model_t* first, plspm_model_t* second, const int32_t* lv_first) {
    if (!first || !second || !lv_first || first == second) return fail(first, PLSPM_E_ARG, "plspm_model_attach_second_stage: bad arguments");
    plspm_model* m1 = first; plspm_model* m2 = second;
    if (m1->stage1 || m1->stage2 || m2->stage1 || m2->stage2) return fail(m1, PLSPM_E_STATE, "a handle takes part in one two-stage pair only");
    if (!m1->nonmetric || !m2->nonmetric || m1->n_ind || m2->n_ind || m1->nmx_K)
        return fail(m1, PLSPM_E_STATE, "two-stage estimation needs non-metric handles on complete data (the reference's metric solver cannot run HOCs)");
    // Scale.ORD / NOM data: both stages work on aug columns (solver_nmg.h); "column" below then means aug column -- a plain LV keeps its
    // indicator / data columns, a HOC's MVs are the constituents' stage-1 scores (one NUM column each), affine in the stage-1 aug columns
    if (m1->categorical != m2->categorical) return fail(m1, PLSPM_E_STATE, "both stages are categorical handles (plspm_model_set_categorical) or neither is");
    if (m2->d_Xa) return fail(m1, PLSPM_E_STATE, "the second stage takes no data of its own");
    if (m1->device != m2->device) return fail(m1, PLSPM_E_ARG, "both stages must live on one device");
    const int L1 = m1->L, L2 = m2->L;
    if (lv_first[0] != 0 || lv_first[L2] != L1) return fail(m1, PLSPM_E_ARG, "lv_first must run from 0 to the first stage's L");
    std::vector<int> col2_lv1(m2->P, -1), col2_p1(m2->P, -1), hidx(m2->P, -1), hcol, lv_cols(L2 + 1, 0);
    for (int l = 0; l < L2; ++l) {
        const int j0 = lv_first[l], j1 = lv_first[l + 1], a0 = m2->boff[l], k2 = m2->boff[l + 1] - a0;
        if (j1 <= j0) return fail(m1, PLSPM_E_ARG, "every second-stage LV stands for at least one first-stage LV");
        lv_cols[l] = m1->boff[j0]; lv_cols[l + 1] = m1->boff[j1];
        if (j1 - j0 == 1 && k2 == m1->boff[j1] - m1->boff[j0]) {            // plain LV: the same columns
            for (int a = 0; a < k2; ++a) col2_p1[a0 + a] = m1->boff[j0] + a;
        } else if (k2 == j1 - j0) {                                           // HOC: one MV per constituent, its stage-1 score
            for (int a = 0; a < k2; ++a) { col2_lv1[a0 + a] = j0 + a; hidx[a0 + a] = (int)hcol.size(); hcol.push_back(a0 + a); }
        } else return fail(m1, PLSPM_E_ARG, "a second-stage block is either the first-stage block itself or one column per constituent LV");
    }
    if (hcol.empty()) hcol.push_back(0), hcol.pop_back();
    HIPCHK(m1, hipSetDevice(m1->device));
    m2->lv_first.assign(lv_first, lv_first + L2 + 1); m2->col2_lv1 = col2_lv1; m2->col2_p1 = col2_p1; m2->hidx = hidx; m2->hcol = hcol; m2->lv_cols = lv_cols;
    if (upload_vec(m2, &m2->d_lv_first, m2->lv_first) || upload_vec(m2, &m2->d_col2_lv1, m2->col2_lv1) || upload_vec(m2, &m2->d_col2_p1, m2->col2_p1) ||
        upload_vec(m2, &m2->d_hidx, m2->hidx) || upload_vec(m2, &m2->d_hcol, m2->hcol) || upload_vec(m2, &m2->d_lv_cols, m2->lv_cols))
        return fail(m1, PLSPM_E_STATE, "descriptor upload failed: " + m2->error);
    if (m1->categorical && (int)m1->mv_lv.size() == m1->Pm && (int)m1->lmv_off.size() == L1 + 1) {
        // category codes of the first stage's rows under the SECOND stage's blocks (nm_conv_codes_kernel): per first-stage MV the first
        // column of the second-stage LV that stands for its LV, per second-stage LV its range of first-stage MVs
        std::vector<int> base2(m1->Pm, 0), lmv2(L2 + 1, 0);
        for (int l = 0; l <= L2; ++l) lmv2[l] = m1->lmv_off[lv_first[l]];
        for (int p = 0; p < m1->Pm; ++p) {
            int l2 = 0;
            while (l2 + 1 < L2 && m1->mv_lv[p] >= lv_first[l2 + 1]) ++l2;
            base2[p] = lv_cols[l2];
        }
        m2->mv_base2 = base2; m2->lmv2_off = lmv2;
        if (upload_vec(m2, &m2->d_mv_base2, m2->mv_base2) || upload_vec(m2, &m2->d_lmv2_off, m2->lmv2_off)) return fail(m1, PLSPM_E_STATE, "descriptor upload failed: " + m2->error);
    }
    HIPCHK(m1, hipMemset(m2->d_shift, 0, sizeof(double) * m2->P));
    HIPCHK(m1, hipStreamSynchronize(m2->stream));
    plspm_stream_release(m2->stream);
    m2->stream = m1->stream; m2->owns_stream = false;
    m1->stage2 = m2; m2->stage1 = m1;
    return 0;
}

int plspm_model_set_incomplete_rows(plspm_model_t* m, int32_t K, const int32_t* row_index, const uint8_t* present, int32_t raw_scale) {
    if (!m || K < 1 || !row_index || !present) return fail(m, PLSPM_E_ARG, "plspm_model_set_incomplete_rows: bad arguments");
    if (!m->d_Xa) return fail(m, PLSPM_E_STATE, "plspm_model_set_incomplete_rows: call after plspm_upload");
    if (!m->nonmetric || m->categorical || m->n_ind || m->stage1 || m->stage2 || m->nmx_K)
        return fail(m, PLSPM_E_STATE, "plspm_model_set_incomplete_rows: Scale.NUM handles only, once per upload, no two-stage pair");
    for (int j = 0; j < K; ++j) {
        if (row_index[j] < 0 || row_index[j] >= m->N || (j && row_index[j] <= row_index[j - 1])) return fail(m, PLSPM_E_ARG, "row_index must be ascending and inside [0, N)");
        int cells = 0;
        for (int l = 0; l < m->L; ++l) {
            int in_block = 0;
            for (int p = m->boff[l]; p < m->boff[l + 1]; ++p) in_block += present[(size_t)j * m->P + p] ? 1 : 0;
            if (!in_block) return fail(m, PLSPM_E_ARG, "a row with an entirely missing LV block must be dropped before the upload (config.py:273-285)");
            cells += in_block;
        }
        if (cells == m->P) return fail(m, PLSPM_E_ARG, "a listed row has no missing cell");
    }
    HIPCHK(m, hipSetDevice(m->device));
    unsigned char* d_mask = nullptr;
    const size_t cells = (size_t)K * m->P;
    // any failure below leaves the handle as it was before the call: no side tables, nmx_K == 0
#define NMXCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { if (d_mask) plspm_dfree(d_mask); drop_incomplete_rows(m); \
        return fail(m, -(int)e__, std::string(#call) + ": " + hipGetErrorString(e__)); } } while (0)
    NMXCHK(plspm_dmalloc((void**)&m->d_Xk, cells * sizeof(double)));
    NMXCHK(plspm_dmalloc((void**)&m->d_Mk, cells * sizeof(double)));
    NMXCHK(plspm_dmalloc((void**)&m->d_rowid, (size_t)K * sizeof(int)));
    NMXCHK(plspm_dmalloc((void**)&d_mask, cells));
    NMXCHK(hipMemcpyAsync(m->d_rowid, row_index, (size_t)K * sizeof(int), hipMemcpyHostToDevice, m->stream));
    NMXCHK(hipMemcpyAsync(d_mask, present, cells, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(extract_rows_kernel, dim3((unsigned)K), dim3(256), 0, m->stream, m->d_Xa, m->PA, m->P, (const int*)m->d_rowid, (const unsigned char*)d_mask, m->d_Xk,
                       m->d_Mk);
    NMXCHK(hipGetLastError());
    NMXCHK(hipStreamSynchronize(m->stream));
#undef NMXCHK
    plspm_dfree(d_mask);
    // the rows of Xa were rewritten: everything derived from them is stale (a caller may have run plspm_bootstrap_prepare before this call)
    m->nmx_K = K; m->nmx_raw = raw_scale ? 1 : 0; m->Xt_valid = false; m->zs_valid = false; m->zs_stats_ready = false; m->codes_valid = false; m->dcnt_ready = false;
    return 0;
}

int plspm_sync(plspm_model_t* m) {
    if (!m) return PLSPM_E_ARG;
    HIPCHK(m, hipSetDevice(m->device));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux));
    return 0;
}

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

}  // extern "C"

// dense [C x C] symmetric moment matrix of every replicate out of the tile-packed one (plspm_bootstrap_moments)
__global__ void __launch_bounds__(256) moments_unpack_kernel(const double* __restrict__ gram, long psize, int T, int C, double* __restrict__ out) {
    const double* g = gram + (long)blockIdx.x * psize;
    double* o = out + (long)blockIdx.x * C * C;
    for (int e = threadIdx.x; e < C * C; e += 256) { const int p = e / C, q = e - p * C; o[e] = g[packed_index(T, p, q)]; }
}

// ------------------------------------------------------------------------------------------------ int8 digit-plane Gram (kernels_gram_i8.h)
static constexpr size_t kZsBudget = (size_t)24 << 30;       // digit planes of one data set: at most 24 GiB of the 288 GiB
static inline int i8_kblocks(long N) { return (int)((N + 127) / 128) * 2; }      // k-blocks of 64 rows, an even number
static inline long i8_pairs(const plspm_model* m) { const long C = m->Pg + 1; return C * (C + 1) / 2; }
// Which Gram a bootstrap call of B replicates takes: 1 = fp64 MFMA on the (row,count) lists, 2 = int8 digit planes.
// Can a non-metric bootstrap with on-device draws take its stop-rule passes' row multiplicities from the int8 counts of the digit-plane Gram
// (plspm_detail_bootstrap counts8_plan)?
static bool nm_counts8_possible(const plspm_model* m) {
    return m->nonmetric && m->tune.nm_counts8 != 0 && m->tune.resample_aux == 0 && !m->aux && m->tune.i8_shape == 16 && m->nmx_K == 0 &&
           nm_dense_lds(m, nullptr, nullptr) != 0 && (!m->stage2 || nm_dense_lds(m->stage2, nullptr, nullptr) != 0);
}
static int choose_gram_path(const plspm_model* m, int64_t B) {
    if (m->tune.gram_path == 1) return 1;
    // every model's replicates start from the moment matrix of the uploaded columns (metric, mean-imputed, non-metric, categorical
    // indicator columns, incomplete rows zeroed, first stage of a HOC pair).  The LDS histogram bounds N; int32 accumulators need
    // 128 N < 2^31.
    // (int32 accumulators: |sum_i c_bi d_is| <= 128 sum_i c_bi = 128 N < 2^31, i.e. N < 2^24; the resample counts come from an LDS
    // histogram of 65,536 rows per workgroup, larger data sets take several windows per replicate)
    if (m->stage1 || m->N >= (1 << 24) || m->N < 2) return 1;
    // non-metric models beyond one 16-bit histogram window: the int8 route when their stop-rule passes can read the Gram's int8 counts
    // (on-device draws, and -- round 4 -- explicit index lists too: windowed 16-bit histograms, resample_i8_kernel; a chunk that carries a
    // multiplicity above 127 falls back to row lists from the global histogram, the fp64 Gram and the gathering pass, as for metric models);
    // else the fp64 route
    if (m->nonmetric && m->N > 65535 && !nm_counts8_possible(m)) return 1;
    const size_t zs_bytes = (size_t)(i8_kblocks(m->N) + I8_SLACK_KB) * (size_t)(((i8_pairs(m) + 31) / 32) * 2 * (m->tune.i8_slices ? m->tune.i8_slices : 7)) * 1024;
    if (zs_bytes > kZsBudget) return 1;
    if (m->tune.gram_path == 2) return 2;
    return B >= m->tune.i8_min_batch ? 2 : 1;
}

// Digit planes + pair tables of the resident data (once per upload / digit count).
// How many digit planes ("i8_slices" 0 = automatic).  S planes represent every product with an absolute error of at most
// 2^-(8S-1) max|z| of its column (zs_scale_kernel), so a replicate's sum is off by at most N 2^-(8S-1) max|z| (the multiplicities add up
// to N) -- the integer sum itself is exact.  A sequential fp64 accumulation of the same N terms carries the a-priori bound gamma_N sum|z|
// ~ N 2^-53 sum|z|.  S planes are therefore within the error bound of fp64 arithmetic on the same data whenever
//       sum_i |z_i|  >=  2^(54 - 8S) max_i |z_i|        in every pair column:      S = 7: always,   S = 6: sum >= 64 max.
// Automatic = 6 planes if every column clears that bar with a factor 4 to spare (sum >= 256 max; a resample re-weights the terms:
// sum_i c_i |z_i| scatters by a few percent around sum_i |z_i|), else 7.  Data sets of a few hundred rows stay at 7; 10,000 rows of
// anything bell-shaped reach several hundred.  Measured against 80-bit sums on the 10k x 60 benchmark data (tests/test_gpu_gram_i8.py,
// error relative to sqrt(M_pp M_qq)): seven planes 1e-16 (correctly rounded), six planes 3e-15, the blocked fp64 MFMA accumulation
// 1.6e-15 -- all nine orders below the 1e-6 the records are held to.
static int choose_slices(plspm_model* m, const unsigned long long* h, long npair, int* S_out) {      // h: [max bits | fixed-point sums | OR of the scaled integers], on the host
    double worst = 1e300;
    unsigned long long any = 0ull;
    for (long j = 0; j < npair; ++j) any |= h[2 * npair + j];
    for (long j = 0; j < npair; ++j) {
        const unsigned long long mb = h[j];
        if (mb == 0) continue;                                            // an all-zero column has nothing to round
        const int ef = (int)((mb >> 52) & 0x7ffull);
        if (ef < 64 || ef >= 0x7ff) { worst = 0.0; break; }               // tiny or non-finite maximum: not evaluated, full plane count
        double zmax;
        memcpy(&zmax, &mb, sizeof(double));
        const double sum = std::ldexp((double)h[npair + j], (ef - 1022) - 40);
        worst = std::min(worst, sum / zmax);
    }
    m->zs_ratio = worst;
    int S = worst >= 256.0 ? 6 : 7;
    // planes that would be identically zero in the seven-plane decomposition carry nothing: dropping them changes no sum
    int zero_planes = 0;
    if (worst > 0.0) {
        zero_planes = 6;
        if (any) { int tz = 0; while (!((any >> tz) & 1ull)) ++tz; zero_planes = std::min(6, tz / 8); }
    }
    S = std::min(S, 7 - zero_planes);
    if (m->tune.i8_shape == 32) S = std::max(S, 5);                            // (the 32x32x32 layout is instantiated for 5 .. 8 planes)
    *S_out = S;
    return 0;
}

// Phase 1 of the digit planes (enqueue only -- plspm_bootstrap_prepare): pair tables, column maxima of the pair products and, for the
// automatic plane count, the two column statistics, copied to a pinned block behind an event.  Nothing here waits for the device: the
// plane count is read in phase 2 (prepare_zs), by which time the fit that was enqueued behind this has long synchronised the stream.
static int prepare_zs_stats(plspm_model* m) {
    if (m->zs_valid || m->zs_stats_ready) return 0;
    const int C = m->Pg + 1;
    const long npair = i8_pairs(m);
    std::vector<int> tab(6 * (size_t)npair);
    int* hp = tab.data(); int* hq = hp + npair; int* hd = hq + 2 * npair;       // [p | q | k (device) | packed slot | dense slot | mirrored dense slot]
    int* hd1 = hd + npair; int* hd2 = hd1 + npair;
    const int PSd = cov_ld(m->Pg);
    long j = 0;
    for (int p = 0; p < C; ++p)
        for (int q = p; q < C; ++q, ++j) {
            hp[j] = p; hq[j] = q; hd[j] = (int)packed_index(m->T, p, q);
            hd1[j] = p * PSd + q; hd2[j] = (p == q) ? -1 : q * PSd + p;
        }
    int rc;
    if ((rc = ensure(m, m->pair_tab, tab.size() * sizeof(int)))) return rc;
    if ((rc = ensure(m, m->pair_scale, (size_t)npair * sizeof(double)))) return rc;
    if ((rc = ensure(m, m->zs_stat, 3 * (size_t)npair * sizeof(unsigned long long)))) return rc;
    if ((rc = plspm_detail_h2d(m, m->pair_tab.p, tab.data(), tab.size() * sizeof(int)))) return rc;
    int* d_p = (int*)m->pair_tab.p; int* d_q = d_p + npair;
    ProfScope ps(m, PLSPM_K_PACK);
    unsigned long long* d_max = (unsigned long long*)m->zs_stat.p;
    HIPCHK(m, hipMemsetAsync(d_max, 0, 3 * (size_t)npair * sizeof(unsigned long long), m->stream));
    const int RB = (int)std::max<size_t>(1, std::min<size_t>(64, (kMaxLds - 1024) / ((size_t)(C | 1) * sizeof(double))));
    const size_t lds = (size_t)RB * (C | 1) * sizeof(double);
    if ((rc = allow_lds(m, (const void*)zs_max_kernel, lds))) return rc;
    hipLaunchKernelGGL(zs_max_kernel, dim3((unsigned)((m->N + RB - 1) / RB)), dim3(256), lds, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, C, d_p, d_q, (int)npair, RB, d_max);
    if (m->tune.i8_slices == 0) {
        if ((rc = allow_lds(m, (const void*)zs_abssum_kernel, lds))) return rc;
        hipLaunchKernelGGL(zs_abssum_kernel, dim3((unsigned)((m->N + RB - 1) / RB)), dim3(256), lds, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, C, d_p, d_q, (int)npair, RB,
                           (const unsigned long long*)d_max, d_max + npair, d_max + 2 * npair);
        HIPCHK(m, hipGetLastError());
        const size_t bytes = 3 * (size_t)npair * sizeof(unsigned long long);
        if (m->h_zstat_cap < bytes) {
            if (m->h_zstat) plspm_hfree(m->h_zstat);
            m->h_zstat = nullptr; m->h_zstat_cap = 0;
            HIPCHK(m, plspm_hmalloc(&m->h_zstat, bytes));
            m->h_zstat_cap = bytes;
        }
        if (!m->ev_zstat) HIPCHK(m, hipEventCreateWithFlags(&m->ev_zstat, hipEventDisableTiming));
        HIPCHK(m, hipMemcpyAsync(m->h_zstat, d_max, bytes, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipEventRecord(m->ev_zstat, m->stream));
    }
    HIPCHK(m, hipGetLastError());
    m->zs_stats_ready = true; m->zs_stats_S = m->tune.i8_slices;
    return 0;
}

static int prepare_zs(plspm_model* m) {
    if (m->zs_valid) return 0;
    int rc;
    if (m->zs_stats_ready && m->zs_stats_S != m->tune.i8_slices) m->zs_stats_ready = false;      // the option changed in between
    if ((rc = prepare_zs_stats(m))) return rc;
    int S = m->tune.i8_slices;
    const long npair = i8_pairs(m);
    const int npg = (int)((npair + 31) / 32) * 2;             // pair groups of 16, padded to whole workgroup tiles (two groups)
    const int KB = i8_kblocks(m->N);
    int* d_p = (int*)m->pair_tab.p; int* d_q = d_p + npair; int* d_k = d_q + npair;
    unsigned long long* d_max = (unsigned long long*)m->zs_stat.p;
    ProfScope ps(m, PLSPM_K_PACK);
    if (S == 0) {
        HIPCHK(m, hipEventSynchronize(m->ev_zstat));          // (long done when a fit ran behind plspm_bootstrap_prepare)
        if ((rc = choose_slices(m, (const unsigned long long*)m->h_zstat, npair, &S))) return rc;
    }
    hipLaunchKernelGGL(zs_scale_kernel, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, m->stream, d_max, (int)npair, S, d_k, (double*)m->pair_scale.p);
    // one plane (0/1 data): the product runs through the seven-plane main loop with the planes of a wave standing for seven consecutive
    // pair groups (gram_i8_kernel<.., IND>): the buffer is padded to whole tiles of 2 x 7 groups
    const bool ind = S == 1 && m->tune.i8_shape == 16 && m->tune.i8_ind != 0;
    const int npg_built = ind ? ((npg + 13) / 14) * 14 : npg;
    const int NT = npg_built * S;
    if ((rc = ensure(m, m->zs, (size_t)(KB + I8_SLACK_KB) * (size_t)NT * 1024))) return rc;      // (sized once the plane count is known: 0/1 data take one plane)
    const dim3 grid((unsigned)KB, (unsigned)((npg_built + 3) / 4));
#define ZSB(SS) hipLaunchKernelGGL((zs_build_kernel<SS>), grid, dim3(256), 0, m->stream, (const double*)m->d_Xa, (long)m->N, m->PA, d_p, d_q, d_k, (int)npair, npg_built, NT, m->tune.i8_shape, (uint4*)m->zs.p)
    switch (S) { case 1: ZSB(1); break; case 2: ZSB(2); break; case 3: ZSB(3); break; case 4: ZSB(4); break; case 5: ZSB(5); break; case 6: ZSB(6); break; case 7: ZSB(7); break; default: ZSB(8); break; }
#undef ZSB
    HIPCHK(m, hipGetLastError());
    m->zs_S = S; m->zs_KB = KB; m->zs_NT = NT; m->zs_npair = (int)npair; m->zs_npg = npg_built; m->zs_ind = ind;
    m->zs_valid = true; m->zs_stats_ready = false;
    return 0;
}

// Tile rows of the six-plane int8 Gram for `ct` count tiles (16 replicates each) x `ntx` pair tiles on `cus` CUs in 8 XCDs: `tall` rows of
// 20 count tiles and -- `mix` -- `shrt` rows of 16 in one launch of gram_i8_kernel<6, 4, ., 16, 20>, or (return false) the 256-replicate
// kernel.  Cost model = what the device does: an XCD's workgroups go in order to the CU that is free first (one workgroup per CU), so the
// makespan of a cut is that of list scheduling its tall tiles first, then its short ones, per XCD.  Costs in count-tile rows: 20 per
// tall tile, 16.6 per short one (the same DMA ring for 4/5 of the MFMAs), 16.35 per tile of the 256-replicate kernel (measured on 960 tiles
// of each kind, tools/i8_mix_calib.py: 0.391 / 0.3245 / 0.3195 ms).  Deterministic in (ct, ntx, cus): every rank of a job cuts alike -- and the sums are exact
// integers, so the cut never shows in a result.
// (profiles/r04_i8_mix_calib.jsonl: 960 tiles of each height, 0.3628 / 0.2990 ms with six planes, 0.3435 / 0.2700 with seven)
static constexpr double kI8pTall6 = 20.0, kI8pShort6 = 16.5, kI8pTall7 = 16.0, kI8pShort7 = 12.6;
// Tile heights and their costs: `rt_tall` / `rt_short` count tiles per row, `ca` / `cb` what a tile of each costs (any common unit).  The
// round-3 kernel: 20 / 16 at 20 / 16.6; gram_i8p_kernel: i8p_costs() below.
static bool i8_mix_plan(long ct, long ntx, int cus, bool mix, int* tall, int* shrt, int rt_tall = 20, int rt_short = 16, double ca = 20.0, double cb = 16.6, bool vs_rt16 = true) {
    const int per_xcd = std::max(1, cus / 8);
    // (tiles of one kind are interchangeable: after the tall ones the CUs of an XCD sit on at most two load levels, and the short ones raise
    //  the lowest level a whole group of CUs at a time -- a handful of steps per XCD instead of one per tile)
    auto xcd_span = [&](long na, long nb2, double ca, double cb) {
        const long c = per_xcd, q = na / c, r = na % c;
        double lv[3] = {q * ca, (q + 1) * ca, 0.0};
        long cnt[3] = {c - r, r, 0};
        int n = r ? 2 : 1;
        long left = nb2;
        while (left > 0) {
            int lo = 0;
            for (int k = 1; k < n; ++k) if (lv[k] < lv[lo]) lo = k;
            if (left >= cnt[lo]) { left -= cnt[lo]; lv[lo] += cb; }
            else { lv[n] = lv[lo] + cb; cnt[n] = left; cnt[lo] -= left; left = 0; ++n; }
            for (int k = 0; k < n; ++k)                      // merge equal levels (keeps n <= 2 before the last step)
                for (int k2 = k + 1; k2 < n; ++k2)
                    if (lv[k2] == lv[k]) { cnt[k] += cnt[k2]; lv[k2] = lv[n - 1]; cnt[k2] = cnt[n - 1]; --n; --k2; }
        }
        double worst = 0.0;
        for (int k = 0; k < n; ++k) if (cnt[k] > 0) worst = std::max(worst, lv[k]);
        return worst;
    };
    auto makespan = [&](long a, long b, double ca, double cb) {
        double worst = 0.0;
        const long ta = a * ntx, tb = b * ntx, pa = (ta + 7) / 8, pb = (tb + 7) / 8;
        long seen_a = -1, seen_b = -1;
        for (int x = 0; x < 8; ++x) {
            const long na = std::max(0L, std::min(pa, ta - x * pa)), nb2 = std::max(0L, std::min(pb, tb - x * pb));
            if (na == seen_a && nb2 == seen_b) continue;
            seen_a = na; seen_b = nb2;
            worst = std::max(worst, xcd_span(na, nb2, ca, cb));
        }
        return worst;
    };
    const long rows16 = (ct + 15) / 16, rows_s = (ct + rt_short - 1) / rt_short;
    const double ref16 = vs_rt16 ? makespan(rows16, 0, 16.35, 0.0) : 1e300;
    double best = 1e300;
    long ba = 0, bb = 0;
    for (long b = 0; b <= (mix ? std::min(rows_s, 48L) : 0L); ++b) {
        const long a = std::max(0L, (ct - (long)rt_short * b + rt_tall - 1) / rt_tall);
        if (a == 0 && b * rt_short < ct) continue;
        const double t = makespan(a, b, ca, cb);
        if (t < best - 1e-9) { best = t; ba = a; bb = b; }
        if (a == 0) break;
    }
    *tall = (int)ba; *shrt = (int)bb;
    return best <= ref16;
}

// Resample nb replicates into dense int8 counts and multiply with the digit planes: the nb moment matrices land at `out`.
// Explicit indices can carry a multiplicity above 127 (Philox draws of N >= 128 rows cannot, P < 1e-200): the host looks at the
// flag before the product and reports *fallback so that the caller takes the fp64 Gram for this chunk.
static int run_gram_i8(plspm_model* m, int64_t nb, uint64_t seed, int64_t rep0, const int32_t* d_idx, double* out, bool dense, bool* fallback,
                       const void** counts = nullptr, int* counts_MT = nullptr) {
    *fallback = false;
    if (counts) *counts = nullptr;
    const int S = m->zs_S, KB = m->zs_KB, NT = m->zs_NT;
    // workgroup tile of the product: 16 RT replicates x 32 pairs; narrow tiles (RT 12 / 8) only in the plain four-wave 16x16x64 launch
    // Six planes leave registers for a taller workgroup tile: 320 replicates x 32 pairs (`RT` 20: 30 accumulator tiles per wave with eight
    // waves) moves 9 % fewer LDS-DMA bytes and reads 12 % fewer fragments per MFMA than 256 x 32 -- 2.2 % on the step when the tile grid
    // fills the machine equally well (5,000 replicates: 960 tiles = 3.75 rounds against 1,200 = 4.69, both five tile-units per CU).  "i8_rt"
    // 0 (default) takes it when its rounds cost no more than those of the 256-replicate tile; the sums are exact either way.
#ifdef PLSPM_I8_EXPERIMENTS
    const bool var20 = m->tune.i8_variant < 0 || m->tune.i8_rt == 20;      // experiments build: the schedule variants exist for the 320-replicate tile too
#else
    const bool var20 = m->tune.i8_variant < 0;
#endif
    // private count fragments (kernels_gram_i8p.h, "i8_priv"): four waves, the counts straight into registers, only the digit blocks through LDS;
    // six planes: tile rows of 320 (tall) / 256 (short) replicates, seven planes: 256 / 192
#ifdef PLSPM_I8_EXPERIMENTS
    const bool priv_var = true;                                             // schedule variants / ablation probes of the kernel (tools/i8p_bench.py)
#else
    const bool priv_var = m->tune.i8_variant < 0;
#endif
    const bool priv = m->tune.i8_priv != 0 && m->tune.i8_shape == 16 && m->tune.i8_sched == 0 && priv_var && (S == 6 || S == 7) && !m->zs_ind &&
                      (m->tune.i8_rt == 0 || m->tune.i8_rt == (S == 6 ? 20 : 16));      // (an explicit other tile height names a round-3 kernel)
    bool wide20 = !priv && m->tune.i8_shape == 16 && m->tune.i8_sched == 0 && var20 && S == 6 && !m->zs_ind && (m->tune.i8_rt == 20 || m->tune.i8_rt == 0);
    const int RTtall = priv ? (S == 6 ? 20 : 16) : 20, RTshort = RTtall - 4;
    // "i8_rt" 0: the cut of the replicates into tile rows is planned (i8_mix_plan below): rows of 320 and -- eight-wave kernel -- rows of 256
    // in ONE launch, so that the last round of the machine is as full as the others (5,000 replicates x 60 pair tiles: 11 + 6 rows = 1,020
    // tiles, every CU three tall + one short = 76 count-tile rows, against 16 rows of 320 = 960 tiles, 80 on three CUs of four)
    int nty_tall = 0, nty_short = 0;
    if ((wide20 || priv) && m->tune.i8_rt == 0 && m->tune.i8_short < 0) {
        if (!m->cu_count) { hipDeviceProp_t pr; HIPCHK(m, hipGetDeviceProperties(&pr, m->device)); m->cu_count = pr.multiProcessorCount; }
        // (the plan of the last shape is kept: a bootstrap calls with the same B again and again, and the search costs of the order of a millisecond)
        const long key[4] = {(long)((nb + 15) / 16), (long)(m->zs_npg / 2), (long)std::max(8, m->cu_count),
                             priv ? 2L + S : (long)(m->tune.i8_waves == 8 && var20 && m->tune.i8_dma != 2)};
        if (!(m->mix_valid && std::equal(key, key + 4, m->mix_key))) {
            // (tile costs of gram_i8p_kernel, tools/i8_mix_calib.py on 960 tiles of each height: six planes 320 / 256 replicates, seven planes 256 / 192)
            if (priv) m->mix_wide = i8_mix_plan(key[0], key[1], (int)key[2], true, &m->mix_tall, &m->mix_short, RTtall, RTshort, S == 6 ? kI8pTall6 : kI8pTall7, S == 6 ? kI8pShort6 : kI8pShort7, false);
            else m->mix_wide = i8_mix_plan(key[0], key[1], (int)key[2], key[3] != 0, &m->mix_tall, &m->mix_short);
            std::copy(key, key + 4, m->mix_key); m->mix_valid = true;
        }
        if (!priv) wide20 = m->mix_wide;
        nty_tall = m->mix_tall; nty_short = m->mix_short;
    } else if (wide20 || priv) {
        // tall rows only, or -- "i8_short_rows" n >= 0 (test seam) -- n short rows behind as many tall ones as it takes
        const long ct = (nb + 15) / 16;
        if (m->tune.i8_short > 0 && (priv || (m->tune.i8_waves == 8 && var20 && m->tune.i8_dma != 2))) nty_short = (int)std::min<long>(m->tune.i8_short, (ct + RTshort - 1) / RTshort);
        nty_tall = (int)std::max(0L, (ct - (long)RTshort * nty_short + RTtall - 1) / RTtall);
    }
    const bool rows2 = wide20 || priv;                 // launches whose grid holds tile rows of two heights
    const bool narrow = wide20 || (!priv && m->tune.i8_rt == 8 && m->tune.i8_shape == 16 && m->tune.i8_waves == 4 && m->tune.i8_sched == 0 && m->tune.i8_variant < 0 && S == 7);
    const int RTg = rows2 ? RTtall : narrow ? 8 : 16;
    const bool ind = m->zs_ind && m->tune.i8_sched == 0 && m->tune.i8_variant < 0;
    const int nty = rows2 ? nty_tall + nty_short : (int)((nb + 16 * RTg - 1) / (16 * RTg)), MT = rows2 ? nty_tall * RTtall + nty_short * RTshort : nty * RTg, ntx = ind ? m->zs_npg / 14 : m->zs_npg / 2;
    // resample counts from an LDS histogram per (replicate, window of rows): 65,536 rows of 16-bit counters, or -- Philox draws of a data
    // set that would need more than one such window -- 131,072 rows of 8-bit counters (kernels_gram_i8.h resample_i8_kernel)
    const bool hist_byte = KB > I8_HIST_KB && !d_idx;
    const int hist_kb = hist_byte ? I8_HIST_KB_BYTES : I8_HIST_KB;
    const size_t hist_bytes = (size_t)std::min(KB, hist_kb) * (hist_byte ? 16 : 32) * sizeof(unsigned);
    const unsigned hist_windows = (unsigned)((KB + hist_kb - 1) / hist_kb);
    int rc;
    if ((rc = allow_lds(m, hist_byte ? (const void*)resample_i8_kernel<true> : (const void*)resample_i8_kernel<false>, hist_bytes))) return rc;
    const auto resample_k = hist_byte ? resample_i8_kernel<true> : resample_i8_kernel<false>;
    // threads per workgroup: the histogram decides how many workgroups share a CU (160 KB of LDS); the VALU-bound Philox loop wants the
    // CU's wave slots filled either way (N = 100,000: one 128 KB histogram per CU -- 256 threads left three quarters of the SIMD time idle)
    const unsigned resample_threads = (unsigned)std::min(1024, std::max(256, 256 * (int)(8 / std::max<size_t>(1, (160 * 1024) / std::max<size_t>(1, hist_bytes)))));
    if (m->tune.resample_aux && !m->aux) {
        int lo = 0, hi = 0;
        HIPCHK(m, hipDeviceGetStreamPriorityRange(&lo, &hi));                // (numerically: lowest priority first)
        HIPCHK(m, hipStreamCreateWithPriority(&m->aux, hipStreamNonBlocking, m->tune.resample_aux == 2 ? 0 : (m->tune.resample_aux == 3 ? hi : lo)));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(m, hipEventCreateWithFlags(&m->ev_counts[k], hipEventDisableTiming));
            HIPCHK(m, hipEventCreateWithFlags(&m->ev_cdfree[k], hipEventDisableTiming));
        }
        if ((rc = ensure(m, m->err2, sizeof(int)))) return rc;
        HIPCHK(m, hipMemsetAsync(m->err2.p, 0, sizeof(int), m->aux));
    }
    // counts of this chunk: the buffer the Gram before last read; grown only with both streams idle
    const int slot = m->aux ? (m->cd_slot ^= 1) : 0;
    plspm_model::Buf& cd = slot ? m->cd1 : m->cd;
    const size_t cd_bytes = (size_t)MT * 16 * ((size_t)KB + I8_SLACK_KB) * 64;
    if (cd_bytes > cd.cap) { if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux)); if ((rc = ensure(m, cd, cd_bytes))) return rc; m->cdfree_set[slot] = false; }
    if (!d_idx && m->aux) {
        // Philox draws: on the low-priority stream, as soon as the Gram that last read this buffer is done -- i.e. beside the Gram and the
        // solver of the PREVIOUS call when the host runs ahead; this call's Gram waits for the counts by event
        if (m->cdfree_set[slot]) HIPCHK(m, hipStreamWaitEvent(m->aux, m->ev_cdfree[slot], 0));
        {
            ProfScope ps(m, PLSPM_K_RESAMPLE, m->aux);
            hipLaunchKernelGGL(resample_k, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->aux, (int)m->N, KB, MT, m->tune.i8_shape, d_idx, seed, rep0, (uint4*)cd.p, (int*)m->err2.p);
        }
        HIPCHK(m, hipEventRecord(m->ev_counts[slot], m->aux));
        HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_counts[slot], 0));
    } else {
        // explicit index lists (test / parity seam) arrive on the main stream: drawn there, and the host looks at the flag
        if (m->aux && m->cdfree_set[slot]) HIPCHK(m, hipStreamWaitEvent(m->stream, m->ev_cdfree[slot], 0));
        ProfScope ps(m, PLSPM_K_RESAMPLE);
        hipLaunchKernelGGL(resample_k, dim3((unsigned)nb, hist_windows), dim3(resample_threads), hist_bytes, m->stream, (int)m->N, KB, MT, m->tune.i8_shape, d_idx, seed, rep0, (uint4*)cd.p, (int*)m->err.p);
    }
    if (d_idx) {
        int* h_err = (int*)m->h_flag + 9;
        HIPCHK(m, hipMemcpyAsync(h_err, m->err.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (*h_err & 2) {
            const int keep = *h_err & 1;
            HIPCHK(m, hipMemcpyAsync(m->err.p, &keep, sizeof(int), hipMemcpyHostToDevice, m->stream));
            HIPCHK(m, hipStreamSynchronize(m->stream));
            *fallback = true;
            return 0;
        }
    }
    if (counts && m->tune.i8_shape == 16) { *counts = cd.p; *counts_MT = MT; }       // (16-row pieces: what nm_conv_dense_kernel<.., CNT8> reads)
    const int total = ntx * nty, per = rows2 ? (ntx * nty_tall + 7) / 8 + (ntx * nty_short + 7) / 8 : (total + 7) / 8;      // workgroups per XCD
    // packed: the tile-packed slots the LDS solver / impute kernel read; dense: [(Pg+1) x cov_ld(Pg)] row-major, upper triangle (rows solver)
    const int* d_dst = (const int*)m->pair_tab.p + (dense ? 4 : 3) * (size_t)m->zs_npair;
    // (a mirrored second store per element cost 0.08 ms per 5,000 replicates of the metric benchmark: the rows solver reads the triangle
    //  instead.  Categorical problems, whose solver wants the full square: Gram 1.40 -> 2.07 ms per 1,000 problems with the mirrored
    //  stores against 0.4 ms saved in nmg_prepare's scatter -- not taken either)
    const int* d_dst2 = nullptr;
    const long out_stride = dense ? cov_doubles(m->Pg) : packed_size(m->T);
    // persistent stream-K schedule (kernels_gram_i8.h gram_i8_sk_kernel): one workgroup per CU, whole CUs per XCD
    const bool sk = m->tune.i8_sched == 1 && m->tune.i8_shape == 16 && m->tune.i8_variant < 0 && S >= 5 && S <= 7;      // (S = 8 spills in the persistent kernel)
    int sk_grid = 0;
    if (sk) {
        if (!m->cu_count) { hipDeviceProp_t pr; HIPCHK(m, hipGetDeviceProperties(&pr, m->device)); m->cu_count = pr.multiProcessorCount; }
        sk_grid = std::max(8, (m->cu_count / 8) * 8);
        const size_t slot_bytes = (size_t)32 * S * 1024;                    // 16 count tiles x 2 pair groups x S planes x 1 KB of int32 per workgroup
        if ((size_t)sk_grid * slot_bytes > m->sk_partial.cap && (rc = ensure(m, m->sk_partial, (size_t)sk_grid * slot_bytes))) return rc;
        if (!m->sk_flags.p) {
            if ((rc = ensure(m, m->sk_flags, (size_t)2048 * 8 * sizeof(unsigned)))) return rc;
            HIPCHK(m, hipMemsetAsync(m->sk_flags.p, 0, (size_t)2048 * 8 * sizeof(unsigned), m->stream));
            m->sk_epoch = 0;
        }
    }
    // the buffer form of the LDS-DMA needs every byte offset of a workgroup's walk (incl. the slack k-blocks) below 4 GiB
    const bool dma_fits = (uint64_t)(KB + I8_SLACK_KB) * (uint64_t)std::max(MT, NT) * 1024ull < (1ull << 32);
    const bool dma_buffer = !priv && m->tune.i8_dma != 1 && dma_fits && m->tune.i8_shape == 16 && !sk && (!narrow || (wide20 && m->tune.i8_waves == 4)) && m->tune.i8_variant < 0;
    m->last_i8_dma = dma_buffer ? 2 : 1;
    m->last_i8_rt = RTg;
    m->last_i8_short = rows2 ? nty_short : 0;
    m->last_i8_priv = priv ? 1 : 0;
    m->last_i8_mt = MT;
    ProfScope ps(m, PLSPM_K_GRAM);
#define GI8SK(SS, WW)                                                                                                                        \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<SS, WW, I8_DEFAULT_VAR, 16>::LDS_BYTES;                                                              \
        if ((rc = allow_lds(m, (const void*)gram_i8_sk_kernel<SS, WW>, lds_bytes))) return rc;                                               \
        hipLaunchKernelGGL((gram_i8_sk_kernel<SS, WW>), dim3((unsigned)sk_grid), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p,    \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, \
                           (i32x4*)m->sk_partial.p, (unsigned*)m->sk_flags.p, ++m->sk_epoch, (int*)m->err.p);                                \
    }
#define GI8P(SS, MM, VV)                                                                                                                     \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8P<SS, MM, VV>::LDS_BYTES;                                                                             \
        auto kfn = nty_short ? gram_i8p_kernel<SS, MM, VV, true> : gram_i8p_kernel<SS, MM, VV, false>;                                       \
        if ((rc = allow_lds(m, (const void*)kfn, lds_bytes))) return rc;                                                                     \
        hipLaunchKernelGGL(kfn, dim3((unsigned)(8 * per)), dim3(256), lds_bytes, m->stream, (const uint4*)cd.p,                               \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, nty_short); \
    }
    if (priv) {
#define GI8PX(VV) case VV: if (S == 6) GI8P(6, 5, VV) else GI8P(7, 4, VV) break;
#ifdef PLSPM_I8_EXPERIMENTS
        // "i8_variant" v >= 0: the template's VAR itself -- bits 0-3 ablations, bit 4 digit blocks through staging registers, bits 5-6 the filler
        // schedule (0 strides / 1 one per gap, DMAs last / 2 VMEM evenly spaced = the release kernel), bit 7 one barrier per two k-steps
        switch (m->tune.i8_variant < 0 ? 64 : m->tune.i8_variant) {
            GI8PX(0) GI8PX(1) GI8PX(2) GI8PX(4) GI8PX(8) GI8PX(15) GI8PX(16) GI8PX(32) GI8PX(128) GI8PX(192)
            GI8PX(64) GI8PX(65) GI8PX(66) GI8PX(68) GI8PX(72) GI8PX(69) GI8PX(77) GI8PX(79)
            default: return fail(m, PLSPM_E_ARG, "i8_variant: not a built variant of the private-count kernel"); }
#else
        switch (64) { GI8PX(64) }
#endif
    } else
    if (sk) {
        if (m->tune.i8_waves == 4) { switch (S) { case 5: GI8SK(5, 2) break; case 6: GI8SK(6, 2) break; default: GI8SK(7, 2) break; } }
        else { switch (S) { case 5: GI8SK(5, 4) break; case 6: GI8SK(6, 4) break; default: GI8SK(7, 4) break; } }
    } else
#define GI8VS(SS, WW, VV, SH)                                                                                                                \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<SS, WW, VV, SH>::LDS_BYTES;                                                                          \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<SS, WW, VV, SH>, lds_bytes))) return rc;                                          \
        hipLaunchKernelGGL((gram_i8_kernel<SS, WW, VV, SH>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, 0); \
    }
#define GI8V(SS, WW, VV) GI8VS(SS, WW, VV, 16)
#define GI8(SS, WW) GI8V(SS, WW, I8_DEFAULT_VAR)
#define GI8B(SS, WW) GI8V(SS, WW, I8_DEFAULT_VAR + 800)
#ifdef PLSPM_I8_EXPERIMENTS       // every schedule variant of the 7-plane kernel (tools/i8_bench.py --variants; not in the release library)
#define GI8X(WW) switch (m->tune.i8_variant) { case 0: GI8V(7, WW, 0) break; case 3: GI8V(7, WW, 3) break; case 6: GI8V(7, WW, 6) break; case 4: GI8V(7, WW, 4) break; \
        case 803: GI8V(7, WW, 803) break; case 103: GI8V(7, WW, 103) break; case 203: GI8V(7, WW, 203) break; case 303: GI8V(7, WW, 303) break; case 403: GI8V(7, WW, 403) break; case 503: GI8V(7, WW, 503) break; case 703: GI8V(7, WW, 703) break; \
        case 12: GI8V(7, WW, 12) break; case 18: GI8V(7, WW, 18) break; case 21: GI8V(7, WW, 21) break; case 24: GI8V(7, WW, 24) break; case 30: GI8V(7, WW, 30) break; default: GI8V(7, WW, 33) break; }
    if (S == 7 && m->tune.i8_variant >= 0) { if (m->tune.i8_waves == 4) GI8X(2) else GI8X(4) } else
#endif
#define GI8RT(RR)                                                                                                                            \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<7, 2, I8_DEFAULT_VAR, 16, RR>::LDS_BYTES;                                                            \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<7, 2, I8_DEFAULT_VAR, 16, RR>, lds_bytes))) return rc;                             \
        hipLaunchKernelGGL((gram_i8_kernel<7, 2, I8_DEFAULT_VAR, 16, RR>), dim3((unsigned)(8 * per)), dim3(256), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, 0); \
    }
#define GI8IND(WW, VV)                                                                                                                       \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<7, WW, VV, 16, 16>::LDS_BYTES;                                                                       \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<7, WW, VV, 16, 16, true>, lds_bytes))) return rc;                                 \
        hipLaunchKernelGGL((gram_i8_kernel<7, WW, VV, 16, 16, true>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, 0); \
    }
    if (ind) {                   // one plane per pair group, seven groups per wave
        if (dma_buffer) { if (m->tune.i8_waves == 4) GI8IND(2, I8_DEFAULT_VAR + 800) else GI8IND(4, I8_DEFAULT_VAR + 800) }
        else { if (m->tune.i8_waves == 4) GI8IND(2, I8_DEFAULT_VAR) else GI8IND(4, I8_DEFAULT_VAR) }
    } else
#define GI8RT20(WW, VV)                                                                                                                      \
    {                                                                                                                                        \
        const size_t lds_bytes = GramI8<6, WW, VV, 16, 20>::LDS_BYTES;                                                                       \
        if ((rc = allow_lds(m, (const void*)gram_i8_kernel<6, WW, VV, 16, 20>, lds_bytes))) return rc;                                       \
        if (nty_short && !GramI8<6, WW, VV, 16, 20>::MIX) return fail(m, PLSPM_E_STATE, "int8 Gram: short tile rows planned for a kernel form without them"); \
        hipLaunchKernelGGL((gram_i8_kernel<6, WW, VV, 16, 20>), dim3((unsigned)(8 * per)), dim3(128 * WW), lds_bytes, m->stream, (const uint4*)cd.p, \
                           (const uint4*)m->zs.p, KB, MT, NT, ntx, nty, d_dst, d_dst2, (const double*)m->pair_scale.p, m->zs_npair, (long)nb, out, out_stride, nty_short); \
    }
#ifdef PLSPM_I8_EXPERIMENTS
#define GI8X20(VV) case VV: GI8RT20(4, VV) break;
    if (wide20 && m->tune.i8_variant >= 0 && m->tune.i8_waves == 8) {
        switch (m->tune.i8_variant) { GI8X20(0) GI8X20(1) GI8X20(3) GI8X20(4) GI8X20(5) GI8X20(6) GI8X20(7) GI8X20(12) GI8X20(13) GI8X20(18) GI8X20(21) GI8X20(103) GI8X20(203) GI8X20(403) GI8X20(703) default: return fail(m, PLSPM_E_ARG, "i8_variant: not built for the 320-replicate tile"); }
    } else
#endif
    if (wide20) { if (m->tune.i8_waves == 4) { if (dma_buffer) GI8RT20(2, I8_DEFAULT_VAR + 800) else GI8RT20(2, I8_DEFAULT_VAR) } else GI8RT20(4, I8_DEFAULT_VAR) } else
    if (narrow) GI8RT(8) else
#undef GI8RT_DUMMY
    if (m->tune.i8_shape == 32 && S >= 5) {    // v_mfma_i32_32x32x32_i8: four waves (64 replicates x 32 pairs x S planes each)
        switch (S) { case 5: GI8VS(5, 2, I8_DEFAULT_VAR, 32) break; case 6: GI8VS(6, 2, I8_DEFAULT_VAR, 32) break; case 7: GI8VS(7, 2, I8_DEFAULT_VAR, 32) break; default: GI8VS(8, 2, I8_DEFAULT_VAR, 32) break; }
    } else
    if (dma_buffer) {            // LDS-DMA as buffer_load ... lds (32-bit offsets from per-workgroup descriptors)
        if (m->tune.i8_waves == 4) { switch (S) { case 1: GI8B(1, 2) break; case 2: GI8B(2, 2) break; case 3: GI8B(3, 2) break; case 4: GI8B(4, 2) break; case 5: GI8B(5, 2) break; case 6: GI8B(6, 2) break; case 7: GI8B(7, 2) break; default: GI8B(8, 2) break; } }
        else { switch (S) { case 1: GI8B(1, 4) break; case 2: GI8B(2, 4) break; case 3: GI8B(3, 4) break; case 4: GI8B(4, 4) break; case 5: GI8B(5, 4) break; case 6: GI8B(6, 4) break; case 7: GI8B(7, 4) break; default: GI8B(8, 4) break; } }
    } else
    if (m->tune.i8_waves == 4) { switch (S) { case 1: GI8(1, 2) break; case 2: GI8(2, 2) break; case 3: GI8(3, 2) break; case 4: GI8(4, 2) break; case 5: GI8(5, 2) break; case 6: GI8(6, 2) break; case 7: GI8(7, 2) break; default: GI8(8, 2) break; } }
    else { switch (S) { case 1: GI8(1, 4) break; case 2: GI8(2, 4) break; case 3: GI8(3, 4) break; case 4: GI8(4, 4) break; case 5: GI8(5, 4) break; case 6: GI8(6, 4) break; case 7: GI8(7, 4) break; default: GI8(8, 4) break; } }
#undef GI8V
#undef GI8VS
#undef GI8
#undef GI8B
    HIPCHK(m, hipGetLastError());
    if (m->aux) { HIPCHK(m, hipEventRecord(m->ev_cdfree[slot], m->stream)); m->cdfree_set[slot] = true; }     // the counts buffer is free once this Gram has run
    return 0;
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
    const bool rows_solver = gpath == 2 && m->tune.solver_rows != 0 && m->P <= 64 && !m->n_ind && !m->nonmetric && !m->moments_out && rows_lds <= kMaxLds / 4;
    // the fp64 Gram walks (row,count) lists (explicit indices may fall back to it); so do the stop-rule passes of the non-metric solvers
    const bool need_lists = gpath == 1 || d_idx != nullptr || (m->nonmetric && !counts8_plan);
    const size_t kpad = (size_t)i8_kblocks(N) * 64;
    // (the global-scratch histogram serves the (row,count) lists only: the int8 route on Philox draws never builds them)
    const bool need_ghist = !lds_hist && need_lists;
    const size_t per_rep = (need_lists ? (size_t)ent_stride * sizeof(int2) : 0) + (size_t)std::max<long>(psize, cov_doubles(m->Pg)) * sizeof(double) + (need_ghist ? (size_t)N * sizeof(unsigned) : 0) +
                           (want_dcnt ? (size_t)dcnt_stride * sizeof(unsigned short) : 0) + (gpath == 2 ? kpad : 0);
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(B, (int64_t)((2ull << 30) / per_rep)));
    if (gpath == 2 && chunk < B) chunk = std::max<int64_t>(256, chunk & ~(int64_t)255);      // whole 256-replicate tiles per pass
    int rc;
    if (gpath == 2) {
        if ((rc = prepare_zs(m))) return rc;
    }
    if (need_lists) {
        if ((rc = ensure(m, m->ent, (size_t)chunk * ent_stride * sizeof(int2)))) return rc;
        if ((rc = ensure(m, m->nent, (size_t)chunk * sizeof(int)))) return rc;
    }
    if ((rc = ensure(m, m->gram, (size_t)chunk * std::max<long>(psize, rows_solver ? cov_doubles(m->Pg) : 0) * sizeof(double)))) return rc;
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
    // indices or runs the persistent Gram: a Philox call on the tiled launch neither raises nor needs to clear it (the 4-byte memset is a
    // kernel of its own on the stream: ~8 us with its gaps, 1.5 % of a 5,000-replicate step)
    const bool may_raise = d_idx != nullptr || m->tune.i8_sched != 0 || N < 128;
    if (!m->err_clean || may_raise) HIPCHK(m, hipMemsetAsync(m->err.p, 0, sizeof(int), m->stream));
    m->err_clean = !may_raise;
    double* const gram_buf = (double*)m->gram.p;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = std::min<int64_t>(chunk, B - b0);
        bool f64_gram = gpath == 1;
        const void* cd8 = nullptr;
        int cd8_MT = 0;
        if (gpath == 2) {
            bool fallback = false;
            if ((rc = run_gram_i8(m, nb, seed, rep_offset + b0, d_idx ? d_idx + b0 * N : nullptr, gram_buf, rows_solver, &fallback, &cd8, &cd8_MT))) return rc;
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
            if ((rc = launch_gram<false>(m, nb, 1, (const int2*)m->ent.p, (const int*)m->nent.p, ent_stride, (double*)m->gram.p))) return rc;
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
            if ((rc = run_nonmetric(m, nb, (const double*)m->gram.p, psize, SolverOut{}, ent_l, nent_l, ent_stride, 128, false, cd8, cd8_MT))) return rc;
            if ((rc = ensure(m, m2->gram, (size_t)nb * psize2 * sizeof(double)))) return rc;
            const HocDesc hd = make_hoc_desc(m2);
            const size_t vlds = std::max<size_t>(1, (size_t)hd.nh * (hd.P1 + 1)) * sizeof(double);
            if ((rc = allow_lds(m, (const void*)hoc_moments_kernel, vlds))) return rc;
            {
                ProfScope ps(m, PLSPM_K_REDUCE);
                hipLaunchKernelGGL(hoc_moments_kernel, dim3((unsigned)nb), dim3(256), vlds, m->stream, hd, (const double*)m->gram.p, psize, (const double*)m->nmstate.p,
                                   (long)nm_state_doubles_of(m), (double*)m2->gram.p, psize2);
            }
            rc = run_nonmetric(m2, nb, (const double*)m2->gram.p, psize2, so, ent_l, nent_l, ent_stride, 128, true, cd8, cd8_MT);
            if (rc) return fail(m, rc, "second stage: " + m2->error);
            continue;
        }
        if (m->nonmetric) {
            // threads per problem by model width (measured: 60 columns 0.60 / 0.64 / 0.81 ms with 64 / 128 / 256 threads; 300 indicator
            // columns 21.0 / 13.5 / 10.0 ms)
            const int nm_threads = m->tune.nm_threads > 0 ? m->tune.nm_threads : (m->P > 128 ? 256 : (m->P > 64 ? 128 : 64));
            if ((rc = run_nonmetric(m, nb, (const double*)m->gram.p, psize, so, (need_lists && !cd8) ? (const int2*)m->ent.p : nullptr, (need_lists && !cd8) ? (const int*)m->nent.p : nullptr, ent_stride,
                                    nm_threads, true, cd8, cd8_MT))) return rc;
            continue;
        }
#ifdef PLSPM_DEBUG_MARKS      // phase clocks of one solver problem (tools/gpu_marks.sh builds with -DPLSPM_DEBUG_MARKS); never in the release library
        long long* d_marks = nullptr;
        HIPCHK(m, plspm_dmalloc((void**)&d_marks, 32 * sizeof(long long))); so.marks = d_marks;
#endif
        if (rows_solver && !f64_gram && m->tune.solver_wave != 0 && wave_solver_covers<8>(m->P, m->L, m->n_chol)) {
            // one wave per problem with fixed lane roles (solver_wave.h): Mode-A models of at most 64 MVs and 8 LVs
            const size_t lds = (size_t)wave_ws_doubles<8>() * sizeof(double);
            ProfScope ps(m, PLSPM_K_SOLVER);
            hipLaunchKernelGGL(solver_wave_kernel<8>, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), (const double*)gram_buf, (long)cov_doubles(m->Pg), so);
            m->last_solver = 3;
        } else if (rows_solver && !f64_gram) {
            m->last_solver = 2;
            const size_t lds = desc_lds_bytes(m->P, m->L, m->n_eff, (int)m->pred_idx.size()) + (size_t)workspace_small_doubles(m->P, m->L, m->kmax, m->n_chol) * sizeof(double);
            if ((rc = allow_lds(m, (const void*)solver_rows_kernel, lds))) return rc;
            ProfScope ps(m, PLSPM_K_SOLVER);
            hipLaunchKernelGGL(solver_rows_kernel, dim3((unsigned)nb), dim3(64), lds, m->stream, make_desc(m), (const double*)m->gram.p, (long)cov_doubles(m->Pg), so);
        } else {
            const double* Mp; long mp_stride;
            if ((rc = run_impute(m, nb, (const double*)m->gram.p, &Mp, &mp_stride))) return rc;
            ProfScope ps(m, PLSPM_K_SOLVER);
            m->last_solver = 1;
            if ((rc = launch_solver(m, nb, Mp, mp_stride, so, m->tune.solver_threads))) return rc;
        }
#ifdef PLSPM_DEBUG_MARKS
        {
            long long h[32];
            HIPCHK(m, hipStreamSynchronize(m->stream));
            HIPCHK(m, hipMemcpy(h, d_marks, sizeof(h), hipMemcpyDeviceToHost));
            if (m->last_solver == 3) {
                fprintf(stderr, "[plspm wave clocks] load %lld  treat %lld  init %lld  iterations %lld  finalize %lld  inner %lld  effects %lld  outputs %lld  total %lld\n",
                        h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[13] - h[7], h[13] - h[0]);
                fprintf(stderr, "[plspm wave last iterate] apply_cov %lld  a/G/E %lld  regress %lld  outer+conv %lld\n", h[9] - h[8], h[10] - h[9], h[11] - h[10], h[12] - h[11]);
                fprintf(stderr, "[plspm wave last apply_cov] seg_products+T %lld  sync %lld  Q %lld\n", h[17] - h[16], h[18] - h[17], h[19] - h[18]);
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
    }
    HIPCHK(m, hipGetLastError());
    if (rows_out == (double*)m->rows.p) m->rows_B = B;
    return 0;
}

extern "C" {

int plspm_bootstrap_prepare(plspm_model_t* m) {
    if (!m) return PLSPM_E_ARG;
    if (!m->d_Xa || m->N < 2) return fail(m, PLSPM_E_STATE, "plspm_bootstrap_prepare: no data uploaded");
    HIPCHK(m, hipSetDevice(m->device));
    // what the first bootstrap call on this data would build before its first replicate: the digit planes of the pair products
    // (enqueue only; a model that takes the fp64 Gram has nothing to prepare)
    // (automatic plane count: the column statistics are enqueued and copied to pinned memory behind an event -- no host wait here; the
    //  planes are cut by the first bootstrap call, which finds the statistics on the host.  A fixed plane count has nothing to read back:
    //  everything is enqueued now)
    if (choose_gram_path(m, (int64_t)1 << 20) == 2) return m->tune.i8_slices == 0 ? prepare_zs_stats(m) : prepare_zs(m);
    return 0;
}

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
    if (!m || !out || B < 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap: bad arguments");
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
    if ((rc = plspm_detail_bootstrap(m, B, seed, rep_offset, d_idx, nullptr))) return rc;
    if ((rc = plspm_detail_fetch_records(m, (const double*)m->rows.p, B, plspm_row_stride(m), out, status, iters))) return rc;
    int* h_err = (int*)m->h_flag + 8;
    HIPCHK(m, hipMemcpyAsync(h_err, m->err.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (*h_err & 4) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: the persistent Gram gave up waiting for a partial tile (device shared with a long-running kernel?); set_option i8_sched 0");
    if (*h_err & 1) return fail(m, PLSPM_E_ARG, "plspm_bootstrap: resample index outside [0, N)");
    if (*h_err & 2) return fail(m, PLSPM_E_LIMIT, "plspm_bootstrap: a resample multiplicity exceeded 127 on the int8 Gram path (set_option gram_path 1)");
    if (*h_err) return fail(m, PLSPM_E_STATE, "plspm_bootstrap: the device reported error bits " + std::to_string(*h_err));
    if (m->err2.p) {                       // Philox draws on the int8 path: a multiplicity above 127 (P < 1e-200) would have wrapped
        HIPCHK(m, hipMemcpyAsync(h_err, m->err2.p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (*h_err) return fail(m, PLSPM_E_LIMIT, "plspm_bootstrap: a resample multiplicity exceeded 127 on the int8 Gram path (set_option gram_path 1)");
    }
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

// ---- pinned staging: pageable host buffers never meet the DMA engines directly --------------------------------------------------
static constexpr size_t kPinHalf = (size_t)8 << 20;       // two halves: the host copy of chunk k+1 overlaps the DMA of chunk k
static int pin_ready(plspm_model* m) {
    if (m->h_pin) return 0;
    HIPCHK(m, plspm_hmalloc(&m->h_pin, 2 * kPinHalf));
    m->h_pin_cap = 2 * kPinHalf;
    for (int k = 0; k < 2; ++k) HIPCHK(m, hipEventCreateWithFlags(&m->ev_pin[k], hipEventDisableTiming));
    return 0;
}

// Copy-in of a large pageable source by several host threads: one thread's memcpy into the pinned half runs at 13-50 GB/s depending on the
// host (profiles/r02c_fit_bench.jsonl / r03_fit_bench.jsonl: configs[4]'s 1.6 GB took 105 / 30 ms on two boxes), below what the DMA behind
// it moves; kCopyThreads stripes of every 8 MB chunk keep the staging ahead of the link on either.  The helpers live for one call.
static constexpr int kCopyThreads = 8;
static constexpr size_t kPinHalfBig = (size_t)32 << 20;
static constexpr size_t kCopyParallelFrom = (size_t)64 << 20;
struct CopyCrew {
    std::atomic<uint64_t> seq{0};
    std::atomic<int> done{0};
    std::atomic<bool> stop{false};
    const char* src = nullptr;
    char* dst = nullptr;
    size_t n = 0;
    int T = 1;
    std::vector<std::thread> helpers;
    static void stripe(const char* src, char* dst, size_t n, int t, int T) {
        const size_t per = ((n / T) + 4095) & ~(size_t)4095, lo = std::min(n, per * t), hi = (t == T - 1) ? n : std::min(n, per * (t + 1));
        if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
    }
    void start(int threads) {
        T = threads;
        for (int t = 1; t < T; ++t)
            helpers.emplace_back([this, t]() {
                uint64_t seen = 0;
                for (;;) {
                    uint64_t s;
                    int spins = 0;
                    while ((s = seq.load(std::memory_order_acquire)) == seen && !stop.load(std::memory_order_acquire))
                        if (++spins > 2000) std::this_thread::yield();
                    if (s == seen) return;                       // stop without new work
                    seen = s;
                    stripe(src, dst, n, t, T);
                    done.fetch_add(1, std::memory_order_release);
                }
            });
    }
    void copy(const char* s, char* d, size_t bytes) {
        if (T <= 1) { memcpy(d, s, bytes); return; }
        src = s; dst = d; n = bytes;
        done.store(0, std::memory_order_relaxed);
        seq.fetch_add(1, std::memory_order_release);
        stripe(s, d, bytes, 0, T);
        while (done.load(std::memory_order_acquire) < T - 1) std::this_thread::yield();
    }
    ~CopyCrew() {
        stop.store(true, std::memory_order_release);
        for (auto& h : helpers) h.join();
    }
};

int plspm_detail_h2d(plspm_model* m, void* dst, const void* src, size_t bytes) {
    int rc = pin_ready(m);
    if (rc) return rc;
    CopyCrew crew;
    // large transfers: halves of 32 MB from the library's pinned cache for the duration of the call (a chunk's fixed costs -- event wait,
    // copy enqueue, the crew's hand-shake -- are ~30 us against 0.15 ms of DMA per 8 MB)
    struct Big { void* p = nullptr; ~Big() { if (p) plspm_hfree(p); } } big;
    size_t half = kPinHalf;
    char* base = (char*)m->h_pin;
    // ... and on a stream of their own: behind a kernel on the handle's stream the runtime moves host -> device copies at about half the
    // rate it reaches on a stream that only ever copied (configs[4]'s upload right after a fit: 54 ms against 29.5; tools/experiments/upload_seq.py)
    struct Lane { hipStream_t s = nullptr; ~Lane() { if (s) { (void)hipStreamSynchronize(s); plspm_stream_release(s); } } } lane;      // (error paths: no copy may still read the halves freed below)
    hipStream_t cs = m->stream;
    if (bytes >= kCopyParallelFrom) {
        crew.start((int)std::min<unsigned>(kCopyThreads, std::max(1u, std::thread::hardware_concurrency() / 2)));
        if (plspm_hmalloc(&big.p, 2 * kPinHalfBig) == hipSuccess && big.p) { half = kPinHalfBig; base = (char*)big.p; }
        else big.p = nullptr;
        HIPCHK(m, hipStreamSynchronize(m->stream));                  // whatever still reads or writes `dst` there has finished
        if (plspm_stream_acquire(&lane.s) == hipSuccess && lane.s) cs = lane.s; else lane.s = nullptr;
    }
    size_t done = 0;
    for (int k = 0; done < bytes; ++k) {
        const int h = k & 1;
        const size_t n = std::min(half, bytes - done);
        char* stage = base + h * half;
        if (k >= 2) HIPCHK(m, hipEventSynchronize(m->ev_pin[h]));            // the DMA that last read this half has finished
        crew.copy((const char*)src + done, stage, n);
        HIPCHK(m, hipMemcpyAsync((char*)dst + done, stage, n, hipMemcpyHostToDevice, cs));
        HIPCHK(m, hipEventRecord(m->ev_pin[h], cs));
        done += n;
    }
    // the staging halves are re-used by the next call: wait for the (at most two) copies still in flight
    HIPCHK(m, hipEventSynchronize(m->ev_pin[0]));
    if (bytes > half) HIPCHK(m, hipEventSynchronize(m->ev_pin[1]));
    return 0;
}

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

int plspm_detail_fetch_records(plspm_model* m, const double* d_records, int64_t B, int32_t stride, double* out, int32_t* status, int32_t* iters) {
    const int R = stride - 2;
    int rc = pin_ready(m);
    if (rc) return rc;
    // chunks of at most half the staging area -- and of at most a quarter of the records, so that the host's unpacking of one chunk runs
    // beside the DMA of the next even when everything would fit a single chunk (5,000 x 158 records = 6.3 MB)
    int64_t per = std::max<int64_t>(1, (int64_t)(kPinHalf / ((size_t)stride * sizeof(double))));
    per = std::min<int64_t>(per, std::max<int64_t>(256, (B + 3) / 4));
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
    if ((size_t)B * stride * sizeof(double) >= ((size_t)1 << 20)) crew.on = unpack_crew().begin();
    auto unpack = [&](int h, int64_t b0, int64_t nb) {
        Job j{(const double*)((const char*)m->h_pin + h * kPinHalf), b0, nb, stride, R, out, status, iters};
        if (crew.on) unpack_crew().run(unpack_part, &j); else unpack_part(&j, 0, 1);
    };
    int64_t prev_b0 = 0, prev_nb = 0;
    int k = 0;
    for (int64_t b0 = 0; b0 < B; b0 += per, ++k) {
        const int h = k & 1;
        const int64_t nb = std::min<int64_t>(per, B - b0);
        HIPCHK(m, hipMemcpyAsync((char*)m->h_pin + h * kPinHalf, d_records + b0 * stride, (size_t)nb * stride * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(m, hipEventRecord(m->ev_pin[h], m->stream));
        if (prev_nb) { HIPCHK(m, hipEventSynchronize(m->ev_pin[h ^ 1])); unpack(h ^ 1, prev_b0, prev_nb); }
        prev_b0 = b0; prev_nb = nb;
    }
    if (prev_nb) { HIPCHK(m, hipEventSynchronize(m->ev_pin[(k - 1) & 1])); unpack((k - 1) & 1, prev_b0, prev_nb); }
    return 0;
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

int plspm_gram_tile_plan(int64_t count_tiles, int64_t pair_tiles, int32_t cus, int32_t mix, int32_t* tall, int32_t* shrt) {
    if (!tall || !shrt || count_tiles < 1 || pair_tiles < 1 || cus < 1 || count_tiles > ((int64_t)1 << 26) || pair_tiles > ((int64_t)1 << 20)) return PLSPM_E_ARG;
    int a = 0, b = 0;
    const bool wide = i8_mix_plan((long)count_tiles, (long)pair_tiles, std::max(8, (int)cus), mix != 0, &a, &b);
    *tall = a; *shrt = b;
    return wide ? 1 : 0;
}

int plspm_bootstrap_indices(uint64_t seed, int64_t rep, int64_t N, int32_t* idx) {
    if (!idx || N < 1 || N > 0x7fffffffLL || rep < 0) return PLSPM_E_ARG;
    for (int64_t q = 0; q < (N + 3) / 4; ++q) {
        const u32x4 u = resample_quad(seed, (uint64_t)rep, (uint32_t)q);
        for (int j = 0; j < 4; ++j) if (4 * q + j < N) idx[4 * q + j] = to_index(u.v[j], (uint32_t)N);
    }
    return 0;
}

int plspm_profile_enable(plspm_model_t* m, int32_t on) {
    if (!m) return PLSPM_E_ARG;
    if (on) {                                              // event pairs are created here, not inside a profiled (timed) region
        hipSetDevice(m->device);
        for (int k = 0; k < PLSPM_K_COUNT; ++k)
            while (m->prof[k].pool.size() < 128) { hipEvent_t a = nullptr, b = nullptr; hipEventCreate(&a); hipEventCreate(&b); m->prof[k].pool.emplace_back(a, b); }
    }
    m->profiling = on != 0;
    m->prof_only = (on >= 2 && on < 2 + PLSPM_K_COUNT) ? on - 2 : -1;
    return 0;
}
int plspm_profile_reset(plspm_model_t* m) {
    if (!m) return PLSPM_E_ARG;
    hipSetDevice(m->device);
    if (m->aux) hipStreamSynchronize(m->aux);
    hipStreamSynchronize(m->stream);
    prof_collect(m);
    for (int k = 0; k < PLSPM_K_COUNT; ++k) { m->prof[k].total_ms = 0.0; m->prof[k].launches = 0; }
    return 0;
}
int plspm_profile_read(plspm_model_t* m, int32_t kernel_id, double* total_ms, int64_t* launches) {
    if (!m || kernel_id < 0 || kernel_id >= PLSPM_K_COUNT) return PLSPM_E_ARG;
    hipSetDevice(m->device);
    if (m->aux) hipStreamSynchronize(m->aux);
    hipStreamSynchronize(m->stream);
    prof_collect(m);
    if (total_ms) *total_ms = m->prof[kernel_id].total_ms;
    if (launches) *launches = m->prof[kernel_id].launches;
    return 0;
}

}  // extern "C"
