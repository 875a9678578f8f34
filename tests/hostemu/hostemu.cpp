// TEST INFRASTRUCTURE ONLY.  CPU emulation build of the device-side solver (csrc/solver_core.h):
// the same source the HIP kernel compiles, executed by NT std::threads with a std::barrier playing
// __syncthreads().  Lets the CPU test-suite (and ThreadSanitizer) check the solver's algebra and its
// barrier discipline against the oracle without a GPU.  Never loaded by the product package.
#include <algorithm>
#include <barrier>
#include <cstring>
#include <thread>
#include <vector>

#include "../../plspm-python_amd/csrc/solver_core.h"
#include "../../plspm-python_amd/csrc/solver_nmg.h"
#include "../../plspm-python_amd/csrc/solver_hoc.h"
#include "../../plspm-python_amd/csrc/solver_nmx.h"
#include "../../plspm-python_amd/csrc/solver_ops.h"
#include "../../plspm-python_amd/csrc/solver_wave.h"
#include "../../plspm-python_amd/csrc/solver_quad.h"
#include "../../plspm-python_amd/csrc/solver_wave16.h"

using namespace plspm;

struct HostExec {
    static constexpr int kcap = 8;
    int tid, nt;
    std::barrier<>* bar;
    template <class F> void par(int n, F f) { for (int i = tid; i < n; i += nt) f(i); bar->arrive_and_wait(); }
    template <class F> void one(F f) { if (tid == 0) f(); bar->arrive_and_wait(); }
    void mark(int) {}
    void sync() { bar->arrive_and_wait(); }
    unsigned long long uniform(unsigned long long v) { return v; }
    double* sink(double*) { static thread_local double mine[64]; return mine; }       // (no shared writes: ThreadSanitizer runs this code)
    // (device: w as a DPP broadcast operand of the multiply-adds, block ends by scalar bit tests; same order of additions)
    template <int PMAX> void seg_products(const double (&s)[PMAX], const double* w, int P, unsigned long long ends, double* vrow) {
        double r0 = 0.0, r1 = 0.0;
        for (int q = 0; q < PMAX && q < P; ++q) {
            if (q & 1) r1 += s[q] * w[q]; else r0 += s[q] * w[q];
            if ((ends >> q) & 1ull) { *vrow++ = r0 + r1; r0 = 0.0; r1 = 0.0; }
        }
    }
    // (device_exec.h seg_products2: a second store per block, the sum times `scale`, OFF2 bytes behind the first)
    template <int PMAX, int OFF2> void seg_products2(const double (&s)[PMAX], const double* w, int P, unsigned long long ends, double* vrow, double scale) {
        double r0 = 0.0, r1 = 0.0;
        for (int q = 0; q < PMAX && q < P; ++q) {
            if (q & 1) r1 += s[q] * w[q]; else r0 += s[q] * w[q];
            if ((ends >> q) & 1ull) { const double v = r0 + r1; *vrow = v; vrow[OFF2 / 8] = v * scale; ++vrow; r0 = 0.0; r1 = 0.0; }
        }
    }
    // (device: rows read along the lanes + LDS transposition, device_exec.h; here the plain symmetric lookup)
    template <int PMAX> void load_cov_block(const double* Md, int PS, int, int pc, int q0, int nq, double (&s)[PMAX]) {
        for (int q = 0; q < PMAX; ++q) {
            const int qc = q0 + ((q < nq) ? q : nq - 1);
            s[q] = Md[(qc <= pc) ? qc * PS + pc : pc * PS + qc];
        }
    }
    template <class F> void par2(int n0, int n1, F f) {
        for (int e = tid; e < n0 * n1; e += nt) f(e % n0, e / n0);
        bar->arrive_and_wait();
    }
    template <class F> void par_chunks64(int nchunks, const double* src, F f) {
        for (int c = tid; c < nchunks; c += nt) for (int lane = 0; lane < 64; ++lane) f(c, lane, src[c * 64 + lane]);
        bar->arrive_and_wait();
    }
    double* red;
    template <class F> double sum(int n, F f) {
        double s = 0.0;
        for (int i = tid; i < n; i += nt) s += f(i);
        red[tid] = s;
        bar->arrive_and_wait();
        double t = 0.0;
        for (int k = 0; k < nt; ++k) t += red[k];
        bar->arrive_and_wait();
        return t;
    }
    // ---- wave executor of solver_wave.h (nt == 64: one emulated thread per lane) ----
    // butterfly sum in the device's order (v += value of lane ^ 1, 2, ... 32: wave_ops.h): bitwise the same value on every lane
    double allsum(double v) {
        for (int off = 1; off < 64; off <<= 1) {
            red[tid] = v;
            bar->arrive_and_wait();
            const double t = red[tid ^ off];
            bar->arrive_and_wait();
            v += t;
        }
        return v;
    }
    int vote_count(bool b) {
        red[tid] = b ? 1.0 : 0.0;
        bar->arrive_and_wait();
        int c = 0;
        for (int k = 0; k < nt; ++k) c += red[k] != 0.0 ? 1 : 0;
        bar->arrive_and_wait();
        return c;
    }
    bool vote_any(bool b) { return vote_count(b) > 0; }
    // ballot over the caller's 64-lane wave (solver_quad.h: nt == 256, four emulated waves)
    int wave_vote_count(bool b) {
        red[tid] = b ? 1.0 : 0.0;
        bar->arrive_and_wait();
        int c = 0;
        for (int k = tid & ~63; k < (tid & ~63) + 64 && k < nt; ++k) c += red[k] != 0.0 ? 1 : 0;
        bar->arrive_and_wait();
        return c;
    }
    void fence() {}
    double uniform_d(double v) { return v; }
    int uniform_i(int v) { return v; }
    double bcast(double, int q, const double* published) { return published[q]; }
    void opaque(unsigned&) {}
    void opaque(int&) {}
    void pin8(double&, double&, double&, double&, double&, double&, double&, double&) {}
    // column `tid` of the symmetric moment matrix out of its upper triangle (device: coalesced rows + an LDS transpose)
    template <int PMAX> void load_cov(const double* Md, int PS, int P, double (&s)[PMAX], double*, double& mu, double& diag) {
        for (int q = 0; q < PMAX; ++q) s[q] = (q < P && tid < P) ? Md[(long)(q < tid ? q : tid) * PS + (q < tid ? tid : q)] : 0.0;
        mu = tid < P ? Md[(long)tid * PS + P] : 0.0;
        diag = tid < P ? Md[(long)tid * PS + tid] : 0.0;
    }
    template <class F> bool any(int n, F f) {
        double s = 0.0;
        for (int i = tid; i < n; i += nt) if (f(i)) s = 1.0;
        red[tid] = s;
        bar->arrive_and_wait();
        bool t = false;
        for (int k = 0; k < nt; ++k) t = t || (red[k] != 0.0);
        bar->arrive_and_wait();
        return t;
    }
};

// Host-side model descriptors shared by the entry points below.
struct EmuModel {
    std::vector<int> lvof, chol_off, pred_off, pred_idx, succ_off, succ_idx;
    ModelDesc md{};
    EmuModel(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode,
             const double* shift, int n_eff, const int* eff_from, const int* eff_to)
        : lvof(P), chol_off(L, -1), pred_off(L + 1, 0), succ_off(L + 1, 0) {
        int kmax = 0, n_chol = 0;
        for (int l = 0; l < L; ++l) {
            for (int p = boff[l]; p < boff[l + 1]; ++p) lvof[p] = l;
            int k = 0;
            for (int j = 0; j < L; ++j) k += C[l * L + j] ? 1 : 0;
            kmax = k > kmax ? k : kmax;
            if (mode[l] == MODE_B) { int kb = boff[l + 1] - boff[l]; chol_off[l] = n_chol; n_chol += (int)chol_block_doubles(kb); }
        }
        for (int i = 0; i < L; ++i) {
            for (int j = 0; j < L; ++j) if (C[i * L + j]) pred_idx.push_back(j);
            pred_off[i + 1] = (int)pred_idx.size();
            for (int s2 = 0; s2 < L; ++s2) if (C[s2 * L + i]) succ_idx.push_back(s2);
            succ_off[i + 1] = (int)succ_idx.size();
        }
        md.n_edges = pred_off[L];
        pred_idx.push_back(0); succ_idx.push_back(0);
        md.P = P; md.L = L; md.PA = PA; md.T = PA / 16; md.scheme = scheme; md.scaled = scaled; md.max_iter = max_iter;
        md.kmax = kmax; md.n_eff = n_eff; md.n_chol = n_chol; md.tol = tol; md.boff = boff; md.lvof = lvof.data(); md.C = C;
        md.mode = mode; md.chol_off = chol_off.data(); md.eff_from = eff_from; md.eff_to = eff_to; md.shift = shift;
        md.pred_off = pred_off.data(); md.pred_idx = pred_idx.data(); md.succ_off = succ_off.data(); md.succ_idx = succ_idx.data();
        md.tile_tu = nullptr;
    }
};

template <class Body>
static void run_group(int nthreads_in, int P, int L, const ModelDesc& md, double* S, Body body) {
    const int nthreads = nthreads_in > 16 ? 16 : nthreads_in;
    std::vector<double> small(workspace_small_doubles(P, L, md.kmax, md.n_chol));
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Workspace ws{};
            ws.S = S; ws.PS = cov_ld(P);
            carve_small(ws, small.data(), P, L, md.kmax, md.n_chol);
            HostExec ex{t, nthreads, &bar, ws.red};
            body(ex, ws);
        });
    for (auto& x : th) x.join();
}

template <class Body>
static void run_plain(int nthreads_in, Body body) {
    const int nthreads = nthreads_in > 16 ? 16 : nthreads_in;
    std::vector<double> red(nthreads);
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back([&, t]() { HostExec ex{t, nthreads, &bar, red.data()}; body(ex); });
    for (auto& x : th) x.join();
}

extern "C" {

// ---- operator seam (csrc/solver_ops.h): Mp = tile-packed moment matrix of the SHIFTED columns + ones, shift = the column means
void hostemu_op_inner_weights(int L, int PA, int scheme, const unsigned char* C, const double* shift, const double* Mp, int nthreads, double* E) {
    std::vector<int> boff(L + 1), mode(L, MODE_A);
    for (int l = 0; l <= L; ++l) boff[l] = l;
    EmuModel em(L, L, PA, scheme, 0, 1, 1.0, boff.data(), C, mode.data(), shift, 0, nullptr, nullptr);
    run_group(nthreads, L, L, em.md, nullptr, [&](HostExec& ex, Workspace& ws) { op_inner_weights(ex, em.md, ws, Mp, E); });
}
int hostemu_op_outer_weights(int mode, int k, int PA, const double* shift, const double* Mp, int nthreads, double* w) {
    std::vector<double> scratch(3 * (size_t)k * k + 1);
    run_plain(nthreads, [&](HostExec& ex) {
        op_outer_weights(ex, mode, k, PA / 16, Mp, shift, scratch.data(), scratch.data() + (size_t)k * k, scratch.data() + 2 * (size_t)k * k, w, scratch.data() + 3 * (size_t)k * k);
    });
    return scratch[3 * (size_t)k * k] != 0.0 ? 0 : 2;
}

// ---- non-metric (NUM / RAW) entry points: S (R matrix, cov_doubles(P)) and state (nm_state_doubles) persist in caller memory
long hostemu_cov_doubles(int P) { return cov_doubles(P); }
long hostemu_nm_state_doubles(int P, int L, int n_chol) { return nm_state_doubles(P, L, n_chol); }

int hostemu_nm_prepare(int P, int L, int PA, int scheme, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode,
                       const double* shift, const double* Mp, int nthreads, double* S, double* state) {
    EmuModel em(P, L, PA, scheme, 1, max_iter, tol, boff, C, mode, shift, 0, nullptr, nullptr);
    run_group(nthreads, P, L, em.md, S, [&](HostExec& ex, Workspace& ws) {
        NmState st; nm_carve(st, state, P, L);
        nm_prepare(ex, em.md, ws, st, Mp);
    });
    return em.md.n_chol;
}
// returns 1 when the problem is still active after the call
int hostemu_nm_step(int P, int L, int PA, int scheme, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode,
                    const double* shift, int nthreads, double* S, double* state, const double* partial, int nparts) {
    EmuModel em(P, L, PA, scheme, 1, max_iter, tol, boff, C, mode, shift, 0, nullptr, nullptr);
    int active = 0;
    run_group(nthreads, P, L, em.md, S, [&](HostExec& ex, Workspace& ws) {
        NmState st; nm_carve(st, state, P, L);
        const bool a = nm_step(ex, em.md, ws, st, partial, nparts);
        if (ex.tid == 0) active = a ? 1 : 0;
    });
    return active;
}
int hostemu_nm_finish(int P, int L, int PA, int scheme, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode,
                      const double* shift, int n_eff, const int* eff_from, const int* eff_to, int nthreads, double* S, double* state, double* row,
                      double* crossloadings, double* path_coef, double* lv_cov, double* indirect, double* score_w, double* score_c, double* cov,
                      double* mean, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, 1, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    FitOutputs out{};
    out.row = row; out.crossloadings = crossloadings; out.path_coef = path_coef; out.lv_cov = lv_cov; out.indirect = indirect;
    out.score_w = score_w; out.score_c = score_c; out.cov = cov; out.mean = mean; out.iters = iters; out.status = status;
    run_group(nthreads, P, L, em.md, S, [&](HostExec& ex, Workspace& ws) {
        NmState st; nm_carve(st, state, P, L);
        nm_finish(ex, em.md, ws, st, out);
    });
    return 0;
}

// ---- categorical (ORD / NOM) non-metric entry points.  boff: aug-column block offsets per LV; mv_off / mv_kind / lmv_off: CatDesc.
struct EmuCat {
    CatDesc cd{};
    std::vector<int> mv_lv;
    EmuCat(int Pm, int L, const int* mv_off, const int* mv_kind, const int* lmv_off) : mv_lv(Pm) {
        cd.Pm = Pm; cd.mv_off = mv_off; cd.mv_kind = mv_kind; cd.lmv_off = lmv_off; cd.cmax = 1; cd.kmv = 1;
        for (int p = 0; p < Pm; ++p) cd.cmax = std::max(cd.cmax, mv_off[p + 1] - mv_off[p]);
        for (int l = 0; l < L; ++l) { cd.kmv = std::max(cd.kmv, lmv_off[l + 1] - lmv_off[l]); for (int p = lmv_off[l]; p < lmv_off[l + 1]; ++p) mv_lv[p] = l; }
    }
};
long hostemu_nmg_state_doubles(int Q, int Pm, int L, const int* mv_off, const int* mv_kind, const int* lmv_off) {
    EmuCat ec(Pm, L, mv_off, mv_kind, lmv_off);
    return nmg_state_doubles(Q, Pm, L, ec.cd.cmax, ec.cd.kmv);
}
// mode: 0 prepare, 1 step (returns active), 2 finish
int hostemu_nmg(int mode_op, int Q, int Pm, int L, int PA, int scheme, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode,
                const int* mv_off, const int* mv_kind, const int* lmv_off, const double* Mp, int nthreads, double* S, double* state,
                const double* partial, int nparts, int n_eff, const int* eff_from, const int* eff_to, double* row, double* crossloadings,
                double* path_coef, double* score_w, double* score_c, double* cov, int* iters, int* status) {
    std::vector<double> shift(Q, 0.0);
    EmuModel em(Q, L, PA, scheme, 1, max_iter, tol, boff, C, mode, shift.data(), n_eff, eff_from, eff_to);
    EmuCat ec(Pm, L, mv_off, mv_kind, lmv_off);
    std::vector<double> shift_m(Pm, 0.0);
    EmuModel emm(Pm, L, PA, scheme, 1, max_iter, tol, lmv_off, C, mode, shift_m.data(), n_eff, eff_from, eff_to);
    emm.md.n_chol = 0;
    std::vector<double> Sm((size_t)cov_doubles(Pm)), small_m(workspace_small_doubles(Pm, L, emm.md.kmax, 0));
    int active = 0;
    FitOutputs out{};
    out.row = row; out.crossloadings = crossloadings; out.path_coef = path_coef; out.score_w = score_w; out.score_c = score_c; out.cov = cov;
    out.iters = iters; out.status = status;
    em.md.n_chol = 0;
    run_group(nthreads, Q, L, em.md, S, [&](HostExec& ex, Workspace& ws) {
        NmState st; nm_carve(st, state, Q, L);
        NmgExtra x; nmg_carve(x, state + nm_state_doubles(Q, L, 0), Q, Pm, L, ec.cd.cmax, ec.cd.kmv);
        if (mode_op == 0) nmg_prepare(ex, em.md, ec.cd, ws, st, x, Mp);
        else if (mode_op == 1) { const bool a = nmg_step(ex, em.md, ec.cd, ws, st, x, partial, nparts); if (ex.tid == 0) active = a ? 1 : 0; }
        else {
            Workspace wsm{};
            wsm.S = Sm.data(); wsm.PS = cov_ld(Pm);
            carve_small(wsm, small_m.data(), Pm, L, emm.md.kmax, 0);
            nmg_finish(ex, em.md, ec.cd, emm.md, ws, wsm, st, x, out);
        }
    });
    return active;
}

// Mp: packed scatter (see packed_index); everything else mirrors ModelDesc / FitOutputs.
int hostemu_solve(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                  const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Mp,
                  int nthreads_in, double* row, double* crossloadings, double* path_coef, double* lv_cov, double* indirect,
                  double* score_w, double* score_c, double* cov, double* mean, int8_t* sign, int* iters, int* status) {
    const int nthreads = nthreads_in > 16 ? 16 : nthreads_in;
    std::vector<int> lvof(P), chol_off(L, -1);
    int kmax = 0, n_chol = 0;
    for (int l = 0; l < L; ++l) {
        for (int p = boff[l]; p < boff[l + 1]; ++p) lvof[p] = l;
        int k = 0;
        for (int j = 0; j < L; ++j) k += C[l * L + j] ? 1 : 0;
        kmax = k > kmax ? k : kmax;
        if (mode[l] == MODE_B) { int kb = boff[l + 1] - boff[l]; chol_off[l] = n_chol; n_chol += (int)chol_block_doubles(kb); }
    }
    std::vector<int> pred_off(L + 1, 0), pred_idx, succ_off(L + 1, 0), succ_idx;
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < L; ++j) if (C[i * L + j]) pred_idx.push_back(j);
        pred_off[i + 1] = (int)pred_idx.size();
        for (int s2 = 0; s2 < L; ++s2) if (C[s2 * L + i]) succ_idx.push_back(s2);
        succ_off[i + 1] = (int)succ_idx.size();
    }
    pred_idx.push_back(0); succ_idx.push_back(0);
    ModelDesc md{};
    md.pred_off = pred_off.data(); md.pred_idx = pred_idx.data(); md.succ_off = succ_off.data(); md.succ_idx = succ_idx.data();
    md.n_edges = pred_off[L];
    md.P = P; md.L = L; md.PA = PA; md.T = PA / 16; md.scheme = scheme; md.scaled = scaled; md.max_iter = max_iter;
    md.kmax = kmax; md.n_eff = n_eff; md.n_chol = n_chol; md.tol = tol; md.boff = boff; md.lvof = lvof.data(); md.C = C;
    md.mode = mode; md.chol_off = chol_off.data(); md.eff_from = eff_from; md.eff_to = eff_to; md.shift = shift;
    const int PS = cov_ld(P);
    std::vector<double> S((size_t)cov_doubles(P)), small(workspace_small_doubles(P, L, kmax, n_chol));
    FitOutputs out{};
    out.row = row; out.crossloadings = crossloadings; out.path_coef = path_coef; out.lv_cov = lv_cov; out.indirect = indirect;
    out.score_w = score_w; out.score_c = score_c; out.cov = cov; out.mean = mean; out.sign = sign; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Workspace ws{};
            ws.S = S.data(); ws.PS = PS;
            carve_small(ws, small.data(), P, L, kmax, n_chol);
            HostExec ex{t, nthreads, &bar, ws.red};
            solve_problem(ex, md, ws, Mp, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// Rows variant (solve_problem_rows<64>): Md = dense symmetric moment matrix [(P+1) x cov_ld(P)] of the shifted columns + ones; one
// emulated thread per lane of the device wave (at least P).
int hostemu_solve_rows(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                       const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                       int nthreads, double* row, double* crossloadings, double* path_coef, double* lv_cov, double* indirect,
                       double* score_w, double* score_c, double* cov, double* mean, int8_t* sign, int* iters, int* status) {
    // nthreads 256: the split form (solve_problem_rows<64, true>: 64 < P <= 128, two emulated threads per MV, no covariance output)
    const bool split = nthreads == 256;
    if (split ? (P <= 64 || P > 128 || rows_split_block(boff, L, 64) == 0) : (P > 64 || nthreads < P || nthreads > 64)) return 1;
    if (split) cov = nullptr;
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    std::vector<double> small(workspace_small_doubles(P, L, em.md.kmax, em.md.n_chol)), red(nthreads);
    FitOutputs out{};
    out.row = row; out.crossloadings = crossloadings; out.path_coef = path_coef; out.lv_cov = lv_cov; out.indirect = indirect;
    out.score_w = score_w; out.score_c = score_c; out.cov = cov; out.mean = mean; out.sign = sign; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Workspace ws{};
            ws.S = nullptr; ws.PS = cov_ld(P);
            carve_small(ws, small.data(), P, L, em.md.kmax, em.md.n_chol);
            HostExec ex{t, nthreads, &bar, red.data()};
            if (split) solve_problem_rows<64, true>(ex, em.md, ws, Md, out);
            else solve_problem_rows<64>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// The wave solver's reciprocal forms (seed of the hardware's accuracy + two Newton steps): the emulation executes the same refinement.
double hostemu_wave_rsqrt(double x) { return wave_rsqrt(x); }
double hostemu_wave_rcp(double x) { return wave_rcp(x); }

// Wave solver (solver_wave.h solve_problem_wave<8>): same inputs as the rows variant, 64 emulated lanes; returns 1 for a model it does not cover.
int hostemu_solve_wave(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                       const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                       double* row, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    if (!wave_solver_covers<8>(P, L, em.md.n_chol)) return 1;
    const int nthreads = 64;
    std::vector<double> lds(wave_ws_doubles<8>(em.md.n_chol), 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            WaveWs<8> ws{};
            wave_carve(ws, lds.data());
            HostExec ex{t, nthreads, &bar, red.data()};
            if (em.md.n_chol > 0) solve_problem_wave<8, true>(ex, em.md, ws, Md, out);
            else solve_problem_wave<8, false>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// Wave solver for 9 .. 16 LVs (solver_wave16.h solve_problem_wave16<16>): 64 emulated lanes; returns 1 for a model it does not cover.
int hostemu_solve_wave16(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                         const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                         double* row, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    if (!wave16_solver_covers<16>(P, L, em.md.n_chol, em.md.kmax)) return 1;
    const int nthreads = 64;
    std::vector<double> lds(wave16_ws_doubles<16>(L, em.md.kmax, em.md.n_chol), 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Wave16Ws<16> ws{};
            wave16_carve(ws, lds.data(), L, em.md.kmax);
            HostExec ex{t, nthreads, &bar, red.data()};
            if (em.md.n_chol > 0) solve_problem_wave16<16, true>(ex, em.md, ws, Md, out);
            else solve_problem_wave16<16, false>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// The same source at LMAX = 8 (one matrix entry per pair lane, the product stream's second copy w V: the A/B form of the wave solver's own class,
// set_option("solver_wave", 2)); returns 1 for a model it does not cover.
int hostemu_solve_wave16_l8(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                            const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                            double* row, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    if (!wave_solver_covers<8>(P, L, em.md.n_chol)) return 1;
    const int nthreads = 64;
    std::vector<double> lds(wave16_ws_doubles<8>(L, em.md.kmax, em.md.n_chol), 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Wave16Ws<8> ws{};
            wave16_carve(ws, lds.data(), L, em.md.kmax);
            HostExec ex{t, nthreads, &bar, red.data()};
            if (em.md.n_chol > 0) solve_problem_wave16<8, true>(ex, em.md, ws, Md, out);
            else solve_problem_wave16<8, false>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// The non-metric (Scale.NUM / RAW) instantiation of the same source (solver_wave16.h NM; round 6): LMAX = 8 for at most 8 LVs, else 16.  maps: [(max_iter + 2) x (P + L)]
// score maps of the steps the problem continued behind (or null), force_T > 0: stop behind exactly that many steps.  Returns 1 for a model it does not cover.
int hostemu_solve_nmwave16(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                           const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                           double* row, int* iters, int* status, double* maps, int force_T, int* steps) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    const bool small = L <= 8, big = L > 16;
    if (small ? !(P <= 64 && em.md.n_chol / 2 <= 16 * 66) : (big ? !(em.md.n_chol == 0 && wave16_solver_covers<32>(P, L, 0, em.md.kmax)) : !wave16_solver_covers<16>(P, L, em.md.n_chol, em.md.kmax))) return 1;
    const int nthreads = 64;
    const long wsd = small ? wave16_ws_doubles<8>(L, em.md.kmax, em.md.n_chol) : (big ? wave16_ws_doubles<32>(L, em.md.kmax, 0) : wave16_ws_doubles<16>(L, em.md.kmax, em.md.n_chol));
    std::vector<double> lds(wsd + 64, 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    NmWaveIo io{maps, force_T, steps, 1.0};
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            HostExec ex{t, nthreads, &bar, red.data()};
            if (small) {
                Wave16Ws<8> ws{};
                wave16_carve(ws, lds.data(), L, em.md.kmax);
                if (em.md.n_chol > 0) solve_problem_wave16<8, true, true>(ex, em.md, ws, Md, out, &io, lds.data() + wsd);
                else solve_problem_wave16<8, false, true>(ex, em.md, ws, Md, out, &io, lds.data() + wsd);
            } else if (big) {
                Wave16Ws<32> ws{};
                wave16_carve(ws, lds.data(), L, em.md.kmax);
                solve_problem_wave16<32, false, true>(ex, em.md, ws, Md, out, &io, lds.data() + wsd);
            } else {
                Wave16Ws<16> ws{};
                wave16_carve(ws, lds.data(), L, em.md.kmax);
                if (em.md.n_chol > 0) solve_problem_wave16<16, true, true>(ex, em.md, ws, Md, out, &io, lds.data() + wsd);
                else solve_problem_wave16<16, false, true>(ex, em.md, ws, Md, out, &io, lds.data() + wsd);
            }
        });
    for (auto& x : th) x.join();
    return 0;
}

// ... and at LMAX = 32 (sixteen matrix entries per pair lane: all-Mode-A models of 17 ... 32 LVs); returns 1 for a model it does not cover.
int hostemu_solve_wave16_l32(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                             const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                             double* row, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    if (em.md.n_chol != 0 || !wave16_solver_covers<32>(P, L, 0, em.md.kmax)) return 1;
    const int nthreads = 64;
    std::vector<double> lds(wave16_ws_doubles<32>(L, em.md.kmax, 0), 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            Wave16Ws<32> ws{};
            wave16_carve(ws, lds.data(), L, em.md.kmax);
            HostExec ex{t, nthreads, &bar, red.data()};
            solve_problem_wave16<32, false>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// Quad solver (solver_quad.h solve_problem_quad<16>): 256 emulated threads = four waves; returns 1 for a model it does not cover.
int hostemu_solve_quad(int P, int L, int PA, int scheme, int scaled, int max_iter, double tol, const int* boff, const unsigned char* C,
                       const int* mode, const double* shift, int n_eff, const int* eff_from, const int* eff_to, const double* Md,
                       double* row, int* iters, int* status) {
    EmuModel em(P, L, PA, scheme, scaled, max_iter, tol, boff, C, mode, shift, n_eff, eff_from, eff_to);
    if (!quad_solver_covers<16>(P, L, em.md.n_chol, em.md.kmax, boff)) return 1;
    const int nthreads = 256;
    std::vector<double> lds(quad_ws_doubles<16>(L, em.md.kmax), 0.0), red(nthreads);
    FitOutputs out{};
    out.row = row; out.iters = iters; out.status = status;
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            QuadWs<16> ws{};
            quad_carve(ws, lds.data());
            HostExec ex{t, nthreads, &bar, red.data()};
            solve_problem_quad<16>(ex, em.md, ws, Md, out);
        });
    for (auto& x : th) x.join();
    return 0;
}

// Mean imputation on the moments (solver_core.h impute_collapse): aug packed Gram (Ta tiles) -> P-column packed moments (Ts tiles).
void hostemu_impute_collapse(int P, int Qa, int Ta, int Ts, const int* ind_of, const double* Min, double* Mout, int nthreads_in) {
    const int nthreads = nthreads_in > 16 ? 16 : nthreads_in;
    std::vector<double> gam(P), red(nthreads);
    std::barrier<> bar(nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            HostExec ex{t, nthreads, &bar, red.data()};
            impute_collapse(ex, P, Qa, Ta, Ts, ind_of, Min, Mout, gam.data());
        });
    for (auto& x : th) x.join();
}

// ---- two-stage higher order constructs (solver_hoc.h)
static HocDesc emu_hoc_desc(int P1, int L1, int P2, int L2, int T1, int T2, const int* boff1, const int* boff2, const int* lv_first, const int* col2_lv1,
                            const int* col2_p1, int nh, const int* hcol, const int* hidx) {
    HocDesc hd{};
    hd.P1 = P1; hd.L1 = L1; hd.P2 = P2; hd.L2 = L2; hd.T1 = T1; hd.T2 = T2; hd.boff1 = boff1; hd.boff2 = boff2; hd.lv_first = lv_first;
    hd.col2_lv1 = col2_lv1; hd.col2_p1 = col2_p1; hd.nh = nh; hd.hcol = hcol; hd.hidx = hidx;
    return hd;
}
void hostemu_hoc_moments(int P1, int L1, int P2, int L2, int T1, int T2, const int* boff1, const int* boff2, const int* lv_first, const int* col2_lv1,
                         const int* col2_p1, int nh, const int* hcol, const int* hidx, const double* M1, const double* c1, const double* k1, int ok,
                         double* M2, int nthreads) {
    const HocDesc hd = emu_hoc_desc(P1, L1, P2, L2, T1, T2, boff1, boff2, lv_first, col2_lv1, col2_p1, nh, hcol, hidx);
    std::vector<double> V((size_t)(nh > 0 ? nh : 1) * (P1 + 1));
    run_plain(nthreads, [&](HostExec& ex) { hoc_second_stage_moments(ex, hd, M1, c1, k1, ok != 0, M2, V.data()); });
}
void hostemu_hoc_compose(int P1, int L1, int P2, int L2, const int* boff1, const int* boff2, const int* lv_first, const int* col2_lv1, const double* c1,
                         const double* k1, double* state2, double* pseudo, int nthreads) {
    const HocDesc hd = emu_hoc_desc(P1, L1, P2, L2, 0, 0, boff1, boff2, lv_first, col2_lv1, nullptr, 0, nullptr, nullptr);
    run_plain(nthreads, [&](HostExec& ex) { NmState st2; nm_carve(st2, state2, P2, L2); hoc_compose_score_maps(ex, hd, c1, k1, st2, pseudo); });
}

// ---- non-metric data with missing values (solver_nmx.h).  mode_op: 0 prepare, 1 step (returns active), 2 finish.  The caller
// writes the incomplete rows' weights ck[K] at state[nm_state_doubles(P, L, n_chol)] before the prepare call.
long hostemu_nmx_state_doubles(int P, int L, int n_chol, int K) { return nmx_state_doubles(P, L, n_chol, K); }
int hostemu_nmx(int mode_op, int raw, int P, int L, int PA, int scheme, int max_iter, double tol, const int* boff, const unsigned char* C, const int* mode, int K,
                const double* Xk, const double* Mk, const double* Mp, int nthreads, double* S, double* state, const double* partial, int nparts, int n_eff,
                const int* eff_from, const int* eff_to, double* row, double* crossloadings, double* path_coef, double* score_w, double* score_c, double* cov,
                int* iters, int* status) {
    std::vector<double> shift(P, 0.0);
    EmuModel em(P, L, PA, scheme, 1, max_iter, tol, boff, C, mode, shift.data(), n_eff, eff_from, eff_to);
    MissDesc xd{raw, K, Xk, Mk};
    int active = 0;
    FitOutputs out{};
    out.row = row; out.crossloadings = crossloadings; out.path_coef = path_coef; out.score_w = score_w; out.score_c = score_c; out.cov = cov;
    out.iters = iters; out.status = status;
    run_group(nthreads, P, L, em.md, S, [&](HostExec& ex, Workspace& ws) {
        NmState st; nm_carve(st, state, P, L);
        NmxExtra x; nmx_carve(x, state + nm_state_doubles(P, L, em.md.n_chol), P, L, K);
        if (mode_op == 0) nmx_prepare(ex, em.md, xd, ws, st, x, Mp);
        else if (mode_op == 1) { const bool a = nmx_step(ex, em.md, xd, ws, st, x, partial, nparts); if (ex.tid == 0) active = a ? 1 : 0; }
        else nmx_finish(ex, em.md, xd, ws, st, x, out);
    });
    return active;
}

long hostemu_packed_index(int T, int p, int q) { return packed_index(T, p, q); }
long hostemu_packed_size(int T) { return packed_size(T); }
}
