// libplspm_hip.so -- MI355X (gfx950 / CDNA4) backend for the PLS-PM weight solver + bootstrap hot path.
// C-ABI: include/plspm_hip.h.  Design, data layout and per-kernel rooflines: DESIGN.md.
// This unit: caching allocator, stream cache, handles (create / destroy / options / model variants), upload, pinned staging, kernel timing.
// The other host units: plspm_fit.hip, plspm_gram_i8.hip, plspm_bootstrap.hip, plspm_group.cpp (host_internal.h names the seams).
#include "host_internal.h"

#include "philox.h"
#include "kernels_upload.h"

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

ModelDesc make_desc(const plspm_model* m) {
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

HocDesc make_hoc_desc(const plspm_model* m2) {
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
void set_geometry(plspm_model* m) {
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

// Several small descriptor arrays as ONE device block and ONE staged copy: a blocking hipMemcpy costs a host round trip each (~20 us on a busy
// device, 50-90 us on one coming out of idle), and plspm_model_create made eleven of them -- most of what a Plspm() call waited for
// besides its kernels (tools/api_pyprofile.py).  One block per call site (`slot`: plspm_model::BLOB_*): a repeated call on the same handle
// (plspm_model_set_categorical, a second plspm_model_attach_second_stage) replaces its block instead of stacking another one; the part
// pointers are only assigned once the copy is on the stream, so a failed upload leaves the previous descriptors in place.
struct BlobPart { void** dst; const void* src; size_t bytes; };
static int pin_leave_async(plspm_model* m);
static int upload_blob(plspm_model* m, int slot, const std::vector<BlobPart>& parts) {
    size_t total = 0;
    std::vector<size_t> off(parts.size());
    for (size_t i = 0; i < parts.size(); ++i) { off[i] = total; total += (std::max<size_t>(parts[i].bytes, 1) + 63) & ~(size_t)63; }
    if (total > kPinHalf) return fail(m, PLSPM_E_LIMIT, "descriptor block exceeds the staging area");
    int rc = pin_ready(m);
    if (rc) return rc;
    void* blob = nullptr;
    HIPCHK(m, plspm_dmalloc(&blob, total));
    char* host = (char*)m->h_pin;
    memset(host, 0, total);
    for (size_t i = 0; i < parts.size(); ++i)
        if (parts[i].bytes) memcpy(host + off[i], parts[i].src, parts[i].bytes);
    hipError_t e = hipMemcpyAsync(blob, host, total, hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) { plspm_dfree(blob); return fail(m, -(int)e, std::string("descriptor upload: ") + hipGetErrorString(e)); }
    if ((rc = pin_leave_async(m))) { hipStreamSynchronize(m->stream); plspm_dfree(blob); return rc; }      // no host wait otherwise: whoever uses the staging area or the descriptors next is ordered behind the copy
    if (m->blobs[slot]) { hipStreamSynchronize(m->stream); plspm_dfree(m->blobs[slot]); }                   // (a kernel enqueued earlier may still read the old block)
    m->blobs[slot] = blob;
    for (size_t i = 0; i < parts.size(); ++i) *parts[i].dst = (char*)blob + off[i];
    return 0;
}
template <class Tv> static BlobPart blob_part(Tv** dst, const std::vector<Tv>& v) { return BlobPart{(void**)dst, v.data(), v.size() * sizeof(Tv)}; }

template <class Tv>
static int upload_vec(plspm_model* m, Tv** dst, const std::vector<Tv>& v) {
    const size_t bytes = std::max<size_t>(1, v.size()) * sizeof(Tv);
    if (*dst) { HIPCHK(m, hipStreamSynchronize(m->stream)); plspm_dfree(*dst); *dst = nullptr; }      // a repeated call replaces its block
    HIPCHK(m, plspm_dmalloc((void**)dst, bytes));
    // (on the handle's stream: its descriptor blocks travel there without a host wait, and a copy on the null stream is not ordered behind them)
    if (!v.empty()) { HIPCHK(m, hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(Tv), hipMemcpyHostToDevice, m->stream)); HIPCHK(m, hipStreamSynchronize(m->stream)); }
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
    if (upload_blob(m, plspm_model::BLOB_MODEL, {blob_part(&m->d_boff, m->boff), blob_part(&m->d_lvof, m->lvof), blob_part(&m->d_mode, m->mode), blob_part(&m->d_chol_off, m->chol_off),
                        blob_part(&m->d_eff_from, m->eff_from), blob_part(&m->d_eff_to, m->eff_to), blob_part(&m->d_C, m->C), blob_part(&m->d_pred_off, m->pred_off),
                        blob_part(&m->d_pred_idx, m->pred_idx), blob_part(&m->d_succ_off, m->succ_off), blob_part(&m->d_succ_idx, m->succ_idx)}))
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
    // (the descriptor arrays d_boff ... d_lv_cols are parts of the blobs below, not blocks of their own)
    void* ptrs[] = {m->d_shift, m->xa.p, m->up_raw.p, m->up_ci.p, m->up_partial.p, m->scores.p,
                    m->d_mv_base2, m->d_lmv2_off, m->gSm.p, m->d_ind_of, m->gram2.p, m->pseudo.p, m->d_Xk, m->d_Mk, m->d_rowid, m->dcnt.p, m->ctable.p, m->Xt.p,
                    m->ent.p, m->nent.p, m->gram.p, m->gram_partial.p, m->rows.p, m->status.p, m->iters.p, m->gS.p, m->gsmall.p,
                    m->fitout.p, m->idx.p, m->err.p, m->ghist.p, m->nmstate.p, m->nmpartial.p, m->nmactive.p, m->nmlist.p, m->gK16.p, m->sum_buf.p, m->cols.p,
                    m->nmw_maps.p, m->nmw_ints.p, m->nmw_vsum.p, m->zs.p, m->cd.p, m->cd1.p, m->codes.p, m->ind8.p, m->tab8.p, m->scl8.p, m->err2.p, m->pp_ctl.p, m->sk_partial.p, m->sk_flags.p, m->pair_tab.p, m->pair_scale.p, m->zs_stat.p};
    for (void* p : ptrs) if (p) plspm_dfree(p);
    for (void* p : m->blobs) if (p) plspm_dfree(p);
    if (m->h_stage) plspm_hfree(m->h_stage);
    if (m->h_pin) plspm_hfree(m->h_pin);
    for (int k = 0; k < 2; ++k) if (m->ev_pin[k]) hipEventDestroy(m->ev_pin[k]);
    for (auto& e : m->ev_part) if (e) hipEventDestroy(e);
    if (m->dl) { hipStreamSynchronize(m->dl); plspm_stream_release(m->dl); }
    if (m->ev_pin_async) hipEventDestroy(m->ev_pin_async);
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
    if (m->group) plspm_detail_group_plan_changed(m->group);
    if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    // whatever was resident is gone from here on (a failed upload leaves an empty, re-usable handle)
    m->N = 0; m->d_Xa = nullptr; m->Xt_valid = false; m->codes_valid = false; m->ind8_valid = false; if (m->stage2) { m->stage2->codes_valid = false; m->stage2->ind8_valid = false; } m->rows_B = 0; m->dcnt_ready = false; m->zs_valid = false; m->zs_stats_ready = false;
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
    const size_t ci_bytes = sizeof(int) * (size_t)Pg;
    bool fast_upload = false;
    if (raw_bytes + ci_bytes + 64 <= kPinHalf) {
        // small data sets (the 4.8 MB of the 10k x 60 headline): matrix and column index through ONE staging half, both copies enqueued without a
        // host wait -- the synchronise at the end of this call is the only round trip of the upload (it also covers the staging area's re-use)
        if ((rc = pin_ready(m))) return rc;
        char* stage = (char*)m->h_pin;
        const size_t ci_off = (raw_bytes + 63) & ~(size_t)63;
        memcpy(stage, X, raw_bytes);
        memcpy(stage + ci_off, ci.data(), ci_bytes);
        HIPCHK(m, hipMemcpyAsync(d_raw, stage, raw_bytes, hipMemcpyHostToDevice, m->stream));
        HIPCHK(m, hipMemcpyAsync(d_ci, stage + ci_off, ci_bytes, hipMemcpyHostToDevice, m->stream));
        if ((rc = pin_leave_async(m))) return rc;                 // (the caller's X has been read: nothing below needs the host to wait)
        fast_upload = true;
    } else {
        if (raw_bytes <= ((size_t)64 << 20) || !m->tune.upload_direct) { if ((rc = plspm_detail_h2d(m, d_raw, X, raw_bytes))) return rc; }
        else HIPCHK(m, hipMemcpyAsync(d_raw, X, raw_bytes, hipMemcpyHostToDevice, m->stream));
        if ((rc = plspm_detail_h2d(m, d_ci, ci.data(), ci_bytes))) return rc;
    }
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
    // X may be released by the caller on return: the small path has copied it into the staging area (the pack kernels run behind the copies
    // on the handle's stream, like everything a later call enqueues); the chunked path waits for its last DMA
    if (!fast_upload) HIPCHK(m, hipStreamSynchronize(m->stream));
    if (raw_bytes > ((size_t)1 << 30)) {                 // do not keep a multi-GB staging copy alive next to the resident matrix
        plspm_dfree(m->up_raw.p);
        m->up_raw.p = nullptr; m->up_raw.cap = 0;
    }
    m->d_Xa = d_Xa;
    m->N = N;
    return 0;
}

}  // extern "C"

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
    // measured-neutral / slower kernel variants and timing probes exist in the experiments build only (make -C plspm-python_amd/csrc experiments)
#ifdef PLSPM_I8_EXPERIMENTS
    const bool experiments = true;
#else
    const bool experiments = false;
#endif
    auto exp_only = [&]() { return fail(m, PLSPM_E_ARG, "plspm_model_set_option: this value of '" + k + "' names a variant of the experiments build (make experiments; PLSPM_HIP_LIB)"); };
    if (k == "solver_threads") { if (value != 64 && value != 128 && value != 256) return bad(); m->tune.solver_threads = value; }
    else if (k == "nm_threads") { if (value != 0 && value != 64 && value != 128 && value != 256) return bad(); m->tune.nm_threads = value; }
    else if (k == "fit_chunks") { if (value < 0 || value > 65535) return bad(); m->tune.fit_chunks = value; }
    else if (k == "wide_ring") { if (value != 0 && value != 4 && value != 6) return bad(); m->tune.wide_ring = value; }      // row buffers of the dense wide Gram's ring (0: two-stage ping-pong)
    else if (k == "wide_nw") { if (value != 4 && value != 8 && value != 16) return bad(); m->tune.wide_nw = value; }
    else if (k == "conv_pass") { if (value < 0 || value > 2) return bad(); m->tune.conv_pass = value; }
    else if (k == "scores_tile") { if (value != 0 && value != 16 && value != 32) return bad(); m->tune.scores_tile = value; }
    else if (k == "gram_lds_kb") { if (value < 0 || value > 160) return bad(); m->tune.gram_lds_kb = value; }
    else if (k == "gram_path") { if (value < 0 || value > 2) return bad(); m->tune.gram_path = value; }
    else if (k == "i8_slices") { if (value < 0 || value > 8) return bad(); if (value != m->tune.i8_slices) m->zs_valid = false; m->tune.i8_slices = value; }
    else if (k == "i8_min_slices") { if (value != 0 && (value < 6 || value > 8)) return bad(); if (value != m->tune.i8_min_slices) { m->zs_valid = false; m->zs_stats_ready = false; } m->tune.i8_min_slices = value; }
    else if (k == "i8_min_batch") { if (value < 1) return bad(); m->tune.i8_min_batch = value; }
    else if (k == "i8_waves") { if (value != 4 && value != 8) return bad(); if (value != 8 && !experiments) return exp_only(); m->tune.i8_waves = value; }
    else if (k == "nm_fast_lds") { if (value < 0 || value > 1) return bad(); m->tune.nm_fast_lds = value; }
    else if (k == "nm_k16") { if (value < 0 || value > 1) return bad(); m->tune.nm_k16 = value; }
    else if (k == "nm_codes") { if (value < 0 || value > 1) return bad(); m->tune.nm_codes = value; }
    else if (k == "nm_mfma") { if (value < 0 || value > 1) return bad(); m->tune.nm_mfma = value; }
    else if (k == "nm_direct16") { if (value < 0 || value > 1) return bad(); m->tune.nm_direct16 = value; }
    else if (k == "i8_nibbles") { if (value < 0 || value > 2) return bad(); m->tune.i8_nibbles = value; }
    else if (k == "nm_wave") { if (value < 0 || value > 1) return bad(); m->tune.nm_wave = value; }
    else if (k == "i8_ind") { if (value < 0 || value > 1) return bad(); if (value != m->tune.i8_ind) m->zs_valid = false; m->tune.i8_ind = value; }
    else if (k == "upload_direct") { if (value < 0 || value > 1) return bad(); m->tune.upload_direct = value; }
    else if (k == "i8_short_rows") { if (value < -1 || value > 4096) return bad(); m->tune.i8_short = value; }
    else if (k == "i8_cus") { if (value != 0 && (value < 8 || value > 4096)) return bad(); m->tune.i8_cus = value; }
    else if (k == "i8_rt") { if (value != 0 && value != 16 && value != 8 && value != 20) return bad(); if (value == 8 && !experiments) return exp_only(); m->tune.i8_rt = value; }
    else if (k == "solver_rows") { if (value != 0 && value != 1) return bad(); m->tune.solver_rows = value; }
    else if (k == "solver_wave") { if (value < 0 || value > 3) return bad(); m->tune.solver_wave = value; }
    else if (k == "solver_quad") { if (value != 0 && value != 1) return bad(); m->tune.solver_quad = value; }
    else if (k == "nm_live") { if (value != 0 && value != 1) return bad(); m->tune.nm_live = value; }
    else if (k == "nm_wave16") { if (value != 0 && value != 1) return bad(); m->tune.nm_wave16 = value; }            // 0: Scale.NUM / RAW batches on the per-iteration launches of rounds 1-5
    else if (k == "nm_subset") { if (value < 0 || value > 100) return bad(); m->tune.nm_subset = value; }      // categorical wave step: 0 every stop-rule pass over all rows (rounds 1-5) | n >= 1: the step stops on its own upper bound and its pass reads n tol / bound of the rows (lower bound; default 4)
    else if (k == "nm_cpl") { if (value != 0 && value != 8 && value != 6) return bad(); m->tune.nm_cpl = value; }      // categorical wave step: 0 automatic (six aug columns per lane where they cover the model) | 8 eight per lane (round 5) | 6 six wherever the layout allows (probes: plspm_nonmetric.hip `cpl6`)
    else if (k == "nm_c10") { if (value != 0 && value != 1) return bad(); m->tune.nm_c10 = value; }      // categorical wave step, items of nine / ten categories: 1 the ten-category instantiation (two waves per SIMD) | 0 the sixteen-category one
    else if (k == "nm_vlong") { if (value != 0 && value != 1) return bad(); m->tune.nm_vlong = value; }      // one-launch categorical batch, verification: 1 one long round for the stragglers behind the fourth short one
    else if (k == "nm_cat_one") { if (value != 0 && value != 1) return bad(); m->tune.nm_cat_one = value; }      // 0: the categorical wave step launch by launch (round 5 / the bound + row-subset form of round 6)
    else if (k == "nm_bound_shift") { if (value < 0 || value > 200) return bad(); m->tune.nm_bound_shift = value; }     // test seam: the solver's stop bound times 2^value (solver_wave16.h NmWaveIo)
    else if (k == "nm_verify_rows") { if (value < 0 || value > 100) return bad(); m->tune.nm_verify_rows = value; }  // percent of the rows the lower-bound pass reads (0: an eighth)
    else if (k == "nm_counts8") { if (value != 0 && value != 1) return bad(); m->tune.nm_counts8 = value; }
    else if (k == "resample_aux") { if (value < 0 || value > 3) return bad(); if (value && !experiments) return exp_only(); m->tune.resample_aux = value; }
    else if (k == "i8_sched") { if (value != 0 && value != 1) return bad(); if (value && !experiments) return exp_only(); m->tune.i8_sched = value; }
    else if (k == "i8_priv") { if (value < 0 || value > 1) return bad(); m->tune.i8_priv = value; }
    else if (k == "i8_persist") { if (value < 0 || value > 1) return bad(); m->tune.i8_persist = value; }
    else if (k == "i8_nostore") { if (value < 0 || value > 1) return bad(); if (value && !experiments) return exp_only(); m->tune.i8_nostore = value; }
    else if (k == "i8_shape") { if (value != 16 && value != 32) return bad(); if (value != 16 && !experiments) return exp_only(); if (value != m->tune.i8_shape) m->zs_valid = false; m->tune.i8_shape = value; }
    else if (k == "i8_variant") { if (value < -1 || value > 899) return bad(); if (value >= 0 && !experiments) return exp_only(); m->tune.i8_variant = value; }
    else if (k == "i8_dma") { if (value < 0 || value > 2) return bad(); m->tune.i8_dma = value; }
    else if (k == "conv_gy") { if (value < 0 || value > 65535) return bad(); m->tune.conv_gy = value; }
    else if (k == "boot_chunks") { if (value < 0 || value > kBootChunksMax) return bad(); m->tune.boot_chunks = value; }
    else if (k == "boot_ratio") { if (value < 10 || value > 100) return bad(); m->tune.boot_ratio = value; }
    else if (k == "boot_align") { if (value < 0 || value > (1 << 20)) return bad(); m->tune.boot_align = value; }
    else return fail(m, PLSPM_E_ARG, "plspm_model_set_option: unknown option '" + k + "'");
    if (m->group) plspm_detail_group_plan_changed(m->group);
    return 0;
}

int plspm_model_get_option(const plspm_model_t* m, const char* key, int32_t* value) {
    if (!m || !key || !value) return PLSPM_E_ARG;
    const std::string k(key);
    if (k == "solver_threads") *value = m->tune.solver_threads;
    else if (k == "nm_threads") *value = m->tune.nm_threads;
    else if (k == "fit_chunks") *value = m->tune.fit_chunks;
    else if (k == "wide_nw") *value = m->tune.wide_nw;
    else if (k == "wide_ring") *value = m->tune.wide_ring;
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
    else if (k == "nm_mfma") *value = m->tune.nm_mfma;
    else if (k == "last_nm_mfma") *value = m->last_nm_mfma;
    else if (k == "nm_direct16") *value = m->tune.nm_direct16;
    else if (k == "last_nm_direct16") *value = m->last_nm_direct16;
    else if (k == "i8_nibbles") *value = m->tune.i8_nibbles;
    else if (k == "last_i8_nibbles") *value = m->last_i8_nibbles;
    else if (k == "i8_ind") *value = m->tune.i8_ind;
    else if (k == "i8_rt") *value = m->tune.i8_rt;
    else if (k == "i8_short_rows") *value = m->tune.i8_short;
    else if (k == "i8_cus") *value = m->tune.i8_cus;
    else if (k == "upload_direct") *value = m->tune.upload_direct;
    else if (k == "i8_dma") *value = m->tune.i8_dma;
    else if (k == "last_i8_dma") *value = m->last_i8_dma;
    else if (k == "solver_rows") *value = m->tune.solver_rows;
    else if (k == "solver_wave") *value = m->tune.solver_wave;
    else if (k == "solver_quad") *value = m->tune.solver_quad;
    else if (k == "nm_live") *value = m->tune.nm_live;
    else if (k == "nm_wave16") *value = m->tune.nm_wave16;
    else if (k == "nm_verify_rows") *value = m->tune.nm_verify_rows;
    else if (k == "nm_bound_shift") *value = m->tune.nm_bound_shift;
    else if (k == "nm_subset") *value = m->tune.nm_subset;
    else if (k == "nm_cat_one") *value = m->tune.nm_cat_one;
    else if (k == "nm_cpl") *value = m->tune.nm_cpl;
    else if (k == "nm_c10") *value = m->tune.nm_c10;
    else if (k == "nm_vlong") *value = m->tune.nm_vlong;
    else if (k == "last_nm_one") *value = m->last_nm_one;
    else if (k == "last_nm_exact") *value = m->last_nm_exact;
    else if (k == "last_nm_wave16") *value = m->last_nm_wave16;
    else if (k == "last_nm_flagged") *value = m->last_nm_flagged;
    else if (k == "last_nm_replayed") *value = m->last_nm_replayed;
    else if (k == "nm_counts8") *value = m->tune.nm_counts8;
    else if (k == "last_solver") *value = m->last_solver;
    else if (k == "resample_aux") *value = m->tune.resample_aux;
    else if (k == "i8_sched") *value = m->tune.i8_sched;
    else if (k == "i8_priv") *value = m->tune.i8_priv;
    else if (k == "last_i8_priv") *value = m->last_i8_priv;
    else if (k == "i8_persist") *value = m->tune.i8_persist;
    else if (k == "nm_wave") *value = m->tune.nm_wave;
    else if (k == "last_nm_wave") *value = m->last_nm_wave;
    else if (k == "i8_min_slices") *value = m->tune.i8_min_slices;
    else if (k == "last_i8_persist") *value = m->last_i8_persist;
    else if (k == "i8_shape") *value = m->tune.i8_shape;
    else if (k == "boot_chunks") *value = m->tune.boot_chunks;
    else if (k == "boot_ratio") *value = m->tune.boot_ratio;
    else if (k == "boot_align") *value = m->tune.boot_align;
    else if (k == "boot_round_units") *value = (int32_t)plspm_detail_round_units_peek(m);      // (64 until the digit planes of this upload exist: a query builds nothing)
    else if (k == "last_gram_path") *value = m->last_gram_path;
    else if (k == "build_experiments") {
#ifdef PLSPM_I8_EXPERIMENTS
        *value = 1;
#else
        *value = 0;
#endif
    }
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
    if (upload_blob(m, plspm_model::BLOB_CATEGORICAL, {blob_part(&m->d_mv_base, m->mv_base), blob_part(&m->d_mv_off, m->mv_off), blob_part(&m->d_mv_kind, m->mv_kind), blob_part(&m->d_lmv_off, m->lmv_off),
                        blob_part(&m->d_mv_lv, m->mv_lv), blob_part(&m->d_no_chol, m->no_chol)}))
        return fail(m, PLSPM_E_STATE, "descriptor upload failed: " + m->error);
    m->Pm = Pm; m->categorical = 1; m->nonmetric = 1; m->n_chol = 0;
    m->cat_pure = true;
    for (int p = 0; p < Pm; ++p) if (mv_kind[p] == 0) m->cat_pure = false;             // a NUM / RAW column: real-valued moments
    set_geometry(m);
    for (auto& c : m->chol_off) c = -1;
    HIPCHK(m, hipMemcpyAsync(m->d_chol_off, m->chol_off.data(), sizeof(int) * m->L, hipMemcpyHostToDevice, m->stream));      // (behind the descriptor block of plspm_model_create)
    HIPCHK(m, hipStreamSynchronize(m->stream));
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
    if (upload_blob(m2, plspm_model::BLOB_HOC, {blob_part(&m2->d_lv_first, m2->lv_first), blob_part(&m2->d_col2_lv1, m2->col2_lv1), blob_part(&m2->d_col2_p1, m2->col2_p1),
                         blob_part(&m2->d_hidx, m2->hidx), blob_part(&m2->d_hcol, m2->hcol), blob_part(&m2->d_lv_cols, m2->lv_cols)}))
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
    HIPCHK(m1, hipMemsetAsync(m2->d_shift, 0, sizeof(double) * m2->P, m2->stream));
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
    m->nmx_K = K; m->nmx_raw = raw_scale ? 1 : 0; m->Xt_valid = false; m->zs_valid = false; m->zs_stats_ready = false; m->codes_valid = false; m->ind8_valid = false; m->dcnt_ready = false;
    return 0;
}

int plspm_sync(plspm_model_t* m) {
    if (!m) return PLSPM_E_ARG;
    HIPCHK(m, hipSetDevice(m->device));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    if (m->aux) HIPCHK(m, hipStreamSynchronize(m->aux));
    return 0;
}

}  // extern "C"

// ---- pinned staging: pageable host buffers never meet the DMA engines directly --------------------------------------------------
int pin_ready(plspm_model* m) {
    if (!m->h_pin) {
        HIPCHK(m, plspm_hmalloc(&m->h_pin, 2 * kPinHalf));
        m->h_pin_cap = 2 * kPinHalf;
        for (int k = 0; k < 2; ++k) HIPCHK(m, hipEventCreateWithFlags(&m->ev_pin[k], hipEventDisableTiming));
        HIPCHK(m, hipEventCreateWithFlags(&m->ev_pin_async, hipEventDisableTiming));
    }
    // a copy that was enqueued out of the staging area WITHOUT a host wait (pin_leave_async: descriptor block of plspm_model_create, the small
    // upload) must have read it before the next user writes there -- by now it has, as a rule, and the wait returns at once
    if (m->pin_pending) { HIPCHK(m, hipEventSynchronize(m->ev_pin_async)); m->pin_pending = false; }
    return 0;
}

// The caller enqueued copies out of the staging area on the handle's stream and returns without waiting for them: pin_ready() of the next user waits.
static int pin_leave_async(plspm_model* m) {
    HIPCHK(m, hipEventRecord(m->ev_pin_async, m->stream));
    m->pin_pending = true;
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

extern "C" {

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
