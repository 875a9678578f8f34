// plspm_group.cpp -- multi-GPU bootstrap groups of libplspm_hip.so: replicate shards on several MI355X + ONE RCCL all-gather
// over xGMI (include/plspm_hip.h, "Multi-GPU").  Host code only; the kernels live in plspm_hip.hip.
//
// Reference: Bootstrap.__init__ (plspm/bootstrap.py:89-111) forks `processes` workers that each run iterations / processes
// replicates and merges their five DataFrames through a multiprocessing.Queue, polling once a second.  Replicates are
// independent (bootstrap.py:54-66), so here rank r runs the contiguous replicate-id shard r of one logical Philox stream on its
// own GPU (X is resident on every GPU), the solver kernel writes the [row | status | iterations] records straight into the
// rank's send buffer, and a single ncclAllGather replaces the Queue.  There is no other data-path collective.
//
// RCCL is loaded with dlopen on the first group / unique-id call: a process that uses one GPU never maps the 570 MB library.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>

#include "model.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, ncclConfig_t*) = nullptr;      // (optional symbols: channel-capped communicators)
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;      // (optional symbols: the gather to rank 0, group option "gather_root")
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;
thread_local std::string g_group_create_error;

// Resolve librccl once per process.  Returns nullptr (and the reason) when the library or a symbol is missing: the caller fails
// loudly -- there is no substitute transport for distinct devices.
const Rccl* rccl(std::string& why) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.so) return &g_rccl;
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* so = nullptr;
    for (const char* n : names) { so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (so) break; }
    if (!so) { why = std::string("librccl.so.1 not loadable: ") + dlerror(); return nullptr; }
    Rccl r;
    r.so = so;
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(so, name); if (!p) { ok = false; why = std::string("librccl lacks ") + name; } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(so); return nullptr; }
    r.CommInitRankConfig = (decltype(r.CommInitRankConfig))dlsym(so, "ncclCommInitRankConfig");
    r.CommSplit = (decltype(r.CommSplit))dlsym(so, "ncclCommSplit");
    r.Send = (decltype(r.Send))dlsym(so, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(so, "ncclRecv");
    r.Broadcast = (decltype(r.Broadcast))dlsym(so, "ncclBroadcast");
    g_rccl = r;
    return &g_rccl;
}

// librccl prints an init banner (ROCm version / hostname / library path) on stdout; a host program's stdout is its own (bench.py
// prints ONE JSON line there), so the banner is sent to stderr for the duration of the communicator set-up.
struct StdoutToStderr {
    int saved = -1;
    StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
    ~StdoutToStderr() { if (saved >= 0) { fflush(stdout); dup2(saved, 1); close(saved); } }
};

struct Local {
    plspm_model* m = nullptr;
    ncclComm_t comm = nullptr;                     // borrowed from the group's plspm_comm
    hipStream_t cstream = nullptr;                 // the collective's stream: it runs beside the next call's kernels
    plspm_model::Buf send[2], recv[2];
    hipEvent_t computed[2][kBootChunksMax] = {};   // shard kernels of sub-batch k of slot s done (recorded on the handle's stream)
    hipEvent_t gathered[2] = {nullptr, nullptr};   // records of slot s complete in recv[s] (recorded on cstream)
    double* d_word = nullptr;                      // barrier / max scratch
    double* h_word = nullptr;                      // pinned mirror
    plspm_model::Buf bcast;                        // gather_root: the summary table on its way from rank 0 to the other ranks
};

// Resident helper threads of a group with several local handles: every handle's shard is enqueued by its own thread (handle 0 by the
// caller), so the host side of a step costs one handle's enqueue (~25 us: three launches + event) instead of their sum -- 8 handles driven
// by one thread took 0.19 ms for the launches alone (tools/group_enqueue.py, profiles/r04_group_enqueue.jsonl) of a 0.48 ms step.  The
// helpers spin for a few hundred microseconds after a job (a bootstrap loop's next call arrives within that window) and then sleep on a
// condition variable.  Non-metric handles (host read-backs per iteration) run on the same crew.
struct ShardCrew {
    int n = 0;                                     // local handles; helpers = n - 1
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> seq{0};
    std::atomic<int> done{0};
    std::atomic<bool> stop{false};
    std::function<void(int)> job;
    void helper(int i) {
        uint64_t seen = 0;
        for (;;) {
            uint64_t s2 = seq.load(std::memory_order_acquire);
            if (s2 == seen) {
                const auto t0 = std::chrono::steady_clock::now();
                while ((s2 = seq.load(std::memory_order_acquire)) == seen && !stop.load(std::memory_order_acquire)) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return seq.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_acquire); });
                    }
                }
                if (s2 == seen) return;            // stop without new work
            }
            seen = s2;
            job(i);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void start(int handles) {
        n = handles;
        for (int i = 1; i < n; ++i) th.emplace_back([this, i] { helper(i); });
    }
    void run(const std::function<void(int)>& f) {  // f(i) for every local handle i; returns when all are done
        job = f;
        done.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(mu); seq.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
        f(0);
        while (done.load(std::memory_order_acquire) < n - 1) std::this_thread::yield();
    }
    ~ShardCrew() {
        { std::lock_guard<std::mutex> lk(mu); stop.store(true, std::memory_order_release); }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

}  // namespace

// The communicator set of this process: every rank of a single-process job, or one rank of a one-process-per-GPU job.  Creating
// it is the expensive part (librccl load + ncclCommInit*: of the order of a second), so it is an object of its own that a host
// program creates once and binds to any number of groups, one after the other.
struct plspm_comm {
    int nranks = 1, first_rank = 0;
    bool use_rccl = false;
    std::vector<int> devices;                      // one per local rank
    std::vector<ncclComm_t> comms;                 // empty when use_rccl is false
    int transport = 0;                             // 1 RCCL, 2 device-to-device copies on peer-mapped buffers (copy engines), 3 ranks share a device (one copy launch)
    int max_channels = 0;                          // > 0: the communicator was created with ncclConfig_t.maxCTAs = this
    plspm_group* bound = nullptr;                  // the live group using it, if any
};

struct plspm_group {
    plspm_comm* comm = nullptr;
    int nranks = 1, first_rank = 0;
    bool use_rccl = false;
    std::vector<Local> loc;
    int next_slot = 0, last_slot = -1;
    bool pending[2] = {false, false};
    int64_t last_B = 0, last_cap = 0;              // last call: replicates, records per rank (all sub-batches)
    // sub-batches of the last call ("chunks"): sub-batch k = the global replicate ids [sub_first[k], sub_first[k] + sub_B[k]), sharded over the ranks like
    // a call of its own; its gathered records start at record nranks * sub_off[k] of the receive buffer, sub_cap[k] per rank
    int last_K = 1;
    int64_t sub_B[kBootChunksMax] = {}, sub_first[kBootChunksMax] = {}, sub_cap[kBootChunksMax] = {}, sub_off[kBootChunksMax] = {};
    int opt_chunks = 0, opt_ratio = 50;            // plspm_group_set_option
    int opt_align = 0;                             // "chunk_align": 0 = whole rounds of the device (plspm_detail_round_units), n > 0 = multiples of n replicates per rank
    // The automatic alignment is rank-LOCAL state (CU count, "i8_cus", whether the digit planes could be built), while the number and sizes of the
    // collectives must be the same on every rank: the ranks agree on it (max over ranks, one blocking all-reduce) in the first plspm_group_bootstrap
    // that consults it, and again after an upload / option change on a handle (calls every rank makes alike).  0 = not agreed.
    int64_t agreed_units = 0;
    int opt_lean_events = 1;                       // "lean_events": no wait packet for an event that has fired, `computed` signalled by the shard's last kernel
    int opt_skip_exchange = 0;                     // diagnostics (include/plspm_hip_test.h): the step without its exchange
    // "gather_root" 1: only rank 0 -- the rank whose handle summarises (bootstrap.py:96-111: only the parent merges) -- receives the shards: 1 / nranks of the
    // all-gather's bytes on the links, nothing arriving at (or read by) the other ranks; the summary table travels back to them in one small broadcast.
    int opt_gather_root = 0;
    bool last_root_only = false;                   // the last plspm_group_bootstrap gathered to rank 0 only
    int peers_checked = -1;                        // slot whose gathered shards were inspected for a failed peer (check_peer_shards)
    int peers_rc = 0;
    double t_shards_ms = 0.0, t_exchange_ms = 0.0;  // host time of the last plspm_group_bootstrap: shard enqueue / exchange enqueue
    std::unique_ptr<ShardCrew> crew;               // one enqueuing thread per local handle (groups of several handles)
    std::string error;
};

namespace {

int gfail(plspm_group* g, int code, const std::string& msg) {
    if (g) g->error = msg; else g_group_create_error = msg;
    return code;
}
#define GHIP(g, call)                                                                                         \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) return gfail((g), -(int)e__, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define GNCCL(g, r, call)                                                                                                  \
    do {                                                                                                                   \
        ncclResult_t n__ = (call);                                                                                         \
        if (n__ != ncclSuccess) return gfail((g), -(1000 + (int)n__), std::string(#call) + ": " + (r)->GetErrorString(n__)); \
    } while (0)

void shard_of(int64_t B, int nranks, int rank, int64_t* first, int64_t* count) {
    const int64_t base = B / nranks, extra = B % nranks;
    *first = rank * base + std::min<int64_t>(rank, extra);
    *count = base + (rank < extra ? 1 : 0);
}

int grow(plspm_group* g, Local& l, plspm_model::Buf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) plspm_dfree(b.p);                     // (the caller synchronised every stream of the group)
    b.p = nullptr; b.cap = 0;
    GHIP(g, plspm_dmalloc(&b.p, bytes));
    b.cap = bytes;
    return 0;
}

int sync_all(plspm_group* g) {
    for (auto& l : g->loc) {
        GHIP(g, hipSetDevice(l.m->device));
        GHIP(g, hipStreamSynchronize(l.m->stream));
        GHIP(g, hipStreamSynchronize(l.cstream));
    }
    return 0;
}

}  // namespace

// Release everything the group holds on its handles (streams, events, buffers) and unbind it; the struct itself stays (an owner may
// still call plspm_group_destroy on it).  Idempotent.
static void group_release(plspm_group* g) {
    for (auto& l : g->loc) {
        hipSetDevice(l.m->device);
        if (l.m->stream) hipStreamSynchronize(l.m->stream);
        if (l.cstream) hipStreamSynchronize(l.cstream);
    }
    for (auto& l : g->loc) {
        hipSetDevice(l.m->device);
        for (int s = 0; s < 2; ++s) {
            if (l.send[s].p) plspm_dfree(l.send[s].p);
            if (l.recv[s].p) plspm_dfree(l.recv[s].p);
            for (auto& e : l.computed[s]) if (e) hipEventDestroy(e);
            if (l.gathered[s]) hipEventDestroy(l.gathered[s]);
        }
        if (l.bcast.p) { plspm_dfree(l.bcast.p); l.bcast.p = nullptr; l.bcast.cap = 0; }
        if (l.d_word) plspm_dfree(l.d_word);
        if (l.h_word) plspm_hfree(l.h_word);
        if (l.cstream) plspm_stream_release(l.cstream);
        l.m->group = nullptr;
    }
    g->crew.reset();                                // (joins the helpers: none of them may touch a handle after this)
    g->loc.clear();
    g->last_slot = -1;
    if (g->comm && g->comm->bound == g) g->comm->bound = nullptr;
    g->comm = nullptr;
}

// A handle is being destroyed while it still belongs to a group (host objects are collected in arbitrary order): the group lets go
// of ALL its handles first, so that nothing dangles; later calls on the group report PLSPM_E_STATE.
void plspm_detail_group_orphan(void* group) { if (group) group_release((plspm_group*)group); }
void plspm_detail_group_plan_changed(void* group) { if (group) ((plspm_group*)group)->agreed_units = 0; }

int plspm_detail_chunk_plan(int64_t B, int64_t bytes_per_unit, int chunks_opt, int ratio_pct, int64_t* parts, int64_t align) {
    parts[0] = B;
    if (B < 1) return 1;
    if (align < 1) align = 64;
    int n = chunks_opt;
    if (n <= 0) n = (B * bytes_per_unit < ((int64_t)2 << 20)) ? 1 : 3;           // automatic: nothing worth hiding below 2 MiB of results
    n = std::min(n, kBootChunksMax);
    // no part below 512 units (a sub-batch's fixed costs: three launches + an event) -- nor below one alignment unit (a round of the machine)
    n = (int)std::min<int64_t>(n, std::max<int64_t>(1, (B + align / 2) / std::max<int64_t>(512, align)));
    if (n <= 1) return 1;
    const double q = std::min(100, std::max(10, ratio_pct)) / 100.0;
    double wsum = 0.0, w = 1.0;
    for (int k = 0; k < n; ++k, w *= q) wsum += w;
    int64_t left = B;
    w = 1.0;
    int k = 0;
    for (; k < n - 1; ++k, w *= q) {
        int64_t c = (int64_t)((double)B * w / wsum + 0.5);
        c = std::max<int64_t>(align, (c + align / 2) / align * align);
        if (left - c < std::min<int64_t>(align, 64)) break;                        // the remainder would be no part of its own
        parts[k] = c; left -= c;
    }
    parts[k] = left;
    return k + 1;
}

extern "C" {

const char* plspm_group_last_error(const plspm_group_t* g) { return g ? g->error.c_str() : g_group_create_error.c_str(); }

int plspm_rccl_unique_id(uint8_t* id) {
    g_group_create_error.clear();
    if (!id) return gfail(nullptr, PLSPM_E_ARG, "plspm_rccl_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == PLSPM_UNIQUE_ID_BYTES, "unique id size");
    std::string why;
    const Rccl* r = rccl(why);
    if (!r) return gfail(nullptr, PLSPM_E_STATE, why);
    ncclUniqueId uid;
    StdoutToStderr quiet;
    GNCCL(nullptr, r, r->GetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return 0;
}

void plspm_group_destroy(plspm_group_t* g) {
    if (!g) return;
    group_release(g);
    delete g;
}

void plspm_comm_destroy(plspm_comm_t* c) {
    if (!c) return;
    // A group still bound lets go of its handles, streams and buffers here, but the struct stays with its owner: a later call on it
    // reports PLSPM_E_STATE instead of touching freed memory, and the owner's plspm_group_destroy frees it.
    if (c->bound) group_release(c->bound);
    for (size_t i = 0; i < c->comms.size(); ++i)
        if (c->comms[i] && g_rccl.so) { hipSetDevice(c->devices[i]); g_rccl.CommDestroy(c->comms[i]); }
    delete c;
}

// max_channels > 0: RCCL may run at most that many workgroups ("channels", ncclConfig_t.maxCTAs) for this communicator's collectives.  Why: the
// all-gather of step k runs beside the Gram of step k + 1, a Gram workgroup needs a whole CU, and every CU an RCCL channel sits on is missing from a
// launch that was cut for all of them; 44 MB per 0.48 ms (eight ranks, 5,000 replicates each) does not need RCCL's default channel count.
// communicators a failed group of ncclCommInitRankConfig / ncclCommSplit calls did create are destroyed, not dropped (ncclCommAbort where the library has it:
// the peers of a half-built communicator may never arrive at a collective destroy)
static void drop_partial_comms(const Rccl* r, plspm_comm* c) {
    for (size_t i = 0; i < c->comms.size(); ++i)
        if (c->comms[i]) { hipSetDevice(c->devices[i]); r->CommDestroy(c->comms[i]); c->comms[i] = nullptr; }
    c->comms.clear();
}

static ncclConfig_t capped_config(int max_channels) {
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.minCTAs = 1;
    cfg.maxCTAs = max_channels;
    return cfg;
}

plspm_comm_t* plspm_comm_create_ex(const int32_t* device_ids, int32_t n_local, int32_t nranks, int32_t first_rank, const uint8_t* unique_id, int32_t transport,
                                   int32_t max_channels) {
    g_group_create_error.clear();
    auto bad = [&](int code, const std::string& why) { gfail(nullptr, code, why); return (plspm_comm_t*)nullptr; };
    if (!device_ids || n_local < 1 || nranks < n_local || first_rank < 0 || first_rank + n_local > nranks || transport < 0 || transport > 2 || max_channels < 0 || max_channels > 256)
        return bad(PLSPM_E_ARG, "plspm_comm_create: bad arguments");
    if (n_local != nranks && (n_local != 1 || !unique_id))
        return bad(PLSPM_E_ARG, "plspm_comm_create: either all ranks in one process (unique_id NULL) or one rank per process with rank 0's unique id");
    if (transport == PLSPM_TRANSPORT_COPY && n_local != nranks) return bad(PLSPM_E_ARG, "plspm_comm_create: the copy-engine exchange needs every rank in this process (peer-mapped buffers)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bad(PLSPM_E_STATE, "no HIP device visible");
    bool distinct = true;
    for (int i = 0; i < n_local; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return bad(PLSPM_E_ARG, "plspm_comm_create: device id out of range");
        for (int j = 0; j < i; ++j) if (device_ids[i] == device_ids[j]) distinct = false;
    }
    if (transport == PLSPM_TRANSPORT_RCCL && !distinct) return bad(PLSPM_E_ARG, "plspm_comm_create: RCCL refuses ranks that share a device");
    plspm_comm* c = new (std::nothrow) plspm_comm();
    if (!c) return bad(PLSPM_E_STATE, "out of host memory");
    c->nranks = nranks; c->first_rank = first_rank;
    c->devices.assign(device_ids, device_ids + n_local);
    // ranks that share a device (a 1-GPU test box; RCCL refuses duplicate devices) exchange records with device-to-device copies (all on one device: gather_local_kernel, one launch)
    c->use_rccl = distinct && transport != PLSPM_TRANSPORT_COPY;
    c->transport = c->use_rccl ? PLSPM_TRANSPORT_RCCL : (distinct ? PLSPM_TRANSPORT_COPY : 3);
    if (!c->use_rccl) {
        if (distinct && n_local > 1) {
            // copy-engine exchange: every device maps every other device's memory (idempotent; "already enabled" is not an error)
            for (int i = 0; i < n_local; ++i)
                for (int j = 0; j < n_local; ++j) {
                    if (i == j) continue;
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) != hipSuccess || !can) { delete c; return bad(PLSPM_E_STATE, "plspm_comm_create: device " + std::to_string(device_ids[i]) + " cannot map the memory of device " + std::to_string(device_ids[j])); }
                    if (hipSetDevice(device_ids[i]) != hipSuccess) { delete c; return bad(PLSPM_E_STATE, "hipSetDevice failed"); }
                    const hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { delete c; return bad(-(int)e, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); }
                    (void)hipGetLastError();
                }
        }
        return c;
    }
    std::string why;
    const Rccl* r = rccl(why);
    if (!r) { delete c; return bad(PLSPM_E_STATE, why); }
    if (max_channels > 0 && !r->CommInitRankConfig) { delete c; return bad(PLSPM_E_STATE, "this librccl has no ncclCommInitRankConfig: max_channels needs it"); }
    c->comms.assign(n_local, nullptr);
    c->max_channels = max_channels;
    ncclResult_t rc;
    StdoutToStderr quiet;
    if (n_local == nranks && max_channels == 0) {
        rc = r->CommInitAll(c->comms.data(), n_local, c->devices.data());
        if (rc != ncclSuccess) { delete c; return bad(-(1000 + (int)rc), std::string("ncclCommInitAll: ") + r->GetErrorString(rc)); }
    } else if (n_local == nranks) {
        // all ranks of this process with a configuration: what ncclCommInitAll does, by hand (one unique id, a group of ncclCommInitRankConfig)
        ncclUniqueId uid;
        rc = r->GetUniqueId(&uid);
        if (rc != ncclSuccess) { delete c; return bad(-(1000 + (int)rc), std::string("ncclGetUniqueId: ") + r->GetErrorString(rc)); }
        ncclResult_t first = ncclSuccess;
        rc = r->GroupStart();
        if (rc != ncclSuccess) { delete c; return bad(-(1000 + (int)rc), std::string("ncclGroupStart: ") + r->GetErrorString(rc)); }
        std::vector<ncclConfig_t> cfg(n_local, capped_config(max_channels));
        for (int i = 0; i < n_local; ++i) {
            if (hipSetDevice(c->devices[i]) != hipSuccess) { if (first == ncclSuccess) first = ncclUnhandledCudaError; continue; }
            rc = r->CommInitRankConfig(&c->comms[i], nranks, uid, i, &cfg[i]);
            if (rc != ncclSuccess && first == ncclSuccess) first = rc;
        }
        rc = r->GroupEnd();
        if (first == ncclSuccess) first = rc;
        if (first != ncclSuccess) { drop_partial_comms(r, c); delete c; return bad(-(1000 + (int)first), std::string("ncclCommInitRankConfig: ") + r->GetErrorString(first)); }
    } else {
        ncclUniqueId uid;
        memcpy(&uid, unique_id, sizeof(uid));
        if (hipSetDevice(c->devices[0]) != hipSuccess) { delete c; return bad(PLSPM_E_STATE, "hipSetDevice failed"); }
        if (max_channels > 0) { ncclConfig_t cfg = capped_config(max_channels); rc = r->CommInitRankConfig(&c->comms[0], nranks, uid, first_rank, &cfg); }
        else rc = r->CommInitRank(&c->comms[0], nranks, uid, first_rank);
        if (rc != ncclSuccess) { delete c; return bad(-(1000 + (int)rc), std::string("ncclCommInitRank: ") + r->GetErrorString(rc)); }
    }
    return c;
}

plspm_comm_t* plspm_comm_create(const int32_t* device_ids, int32_t n_local, int32_t nranks, int32_t first_rank, const uint8_t* unique_id) {
    return plspm_comm_create_ex(device_ids, n_local, nranks, first_rank, unique_id, PLSPM_TRANSPORT_AUTO, 0);
}

// A second communicator over the same ranks and devices with its own channel cap, split off an RCCL communicator (ncclCommSplit: collective over
// the parent, no new unique id to hand round) -- lets a job compare RCCL's default against a capped one without a second rendezvous.
plspm_comm_t* plspm_comm_split(plspm_comm_t* parent, int32_t max_channels) {
    g_group_create_error.clear();
    auto bad = [&](int code, const std::string& why) { gfail(nullptr, code, why); return (plspm_comm_t*)nullptr; };
    if (!parent || max_channels < 0 || max_channels > 256) return bad(PLSPM_E_ARG, "plspm_comm_split: bad arguments");
    if (!parent->use_rccl) return bad(PLSPM_E_STATE, "plspm_comm_split: the parent does not use RCCL");
    const Rccl* r = &g_rccl;
    if (!r->CommSplit) return bad(PLSPM_E_STATE, "this librccl has no ncclCommSplit");
    plspm_comm* c = new (std::nothrow) plspm_comm();
    if (!c) return bad(PLSPM_E_STATE, "out of host memory");
    c->nranks = parent->nranks; c->first_rank = parent->first_rank; c->devices = parent->devices; c->use_rccl = true; c->transport = PLSPM_TRANSPORT_RCCL; c->max_channels = max_channels;
    const int n_local = (int)c->devices.size();
    c->comms.assign(n_local, nullptr);
    StdoutToStderr quiet;
    std::vector<ncclConfig_t> cfg(n_local);
    for (auto& f : cfg) { ncclConfig_t init = NCCL_CONFIG_INITIALIZER; f = init; if (max_channels > 0) { f.minCTAs = 1; f.maxCTAs = max_channels; } }
    ncclResult_t first = ncclSuccess, rc = n_local > 1 ? r->GroupStart() : ncclSuccess;
    if (rc != ncclSuccess) { delete c; return bad(-(1000 + (int)rc), std::string("ncclGroupStart: ") + r->GetErrorString(rc)); }
    for (int i = 0; i < n_local; ++i) {
        if (hipSetDevice(c->devices[i]) != hipSuccess) { if (first == ncclSuccess) first = ncclUnhandledCudaError; continue; }
        rc = r->CommSplit(parent->comms[i], 0, parent->first_rank + i, &c->comms[i], &cfg[i]);
        if (rc != ncclSuccess && first == ncclSuccess) first = rc;
    }
    if (n_local > 1) { rc = r->GroupEnd(); if (first == ncclSuccess) first = rc; }
    if (first != ncclSuccess) { drop_partial_comms(r, c); delete c; return bad(-(1000 + (int)first), std::string("ncclCommSplit: ") + r->GetErrorString(first)); }
    return c;
}

int32_t plspm_comm_size(const plspm_comm_t* c) { return c ? c->nranks : 0; }
int32_t plspm_comm_uses_rccl(const plspm_comm_t* c) { return (c && c->use_rccl) ? 1 : 0; }
int32_t plspm_comm_transport(const plspm_comm_t* c) { return c ? c->transport : 0; }
int32_t plspm_comm_max_channels(const plspm_comm_t* c) { return c ? c->max_channels : 0; }

plspm_group_t* plspm_group_create(plspm_comm_t* c, plspm_model_t* const* models) {
    g_group_create_error.clear();
    auto bad = [&](int code, const std::string& why) { gfail(nullptr, code, why); return (plspm_group_t*)nullptr; };
    if (!c || !models) return bad(PLSPM_E_ARG, "plspm_group_create: bad arguments");
    if (c->bound) return bad(PLSPM_E_STATE, "plspm_group_create: the communicator serves one group at a time");
    const int n_local = (int)c->devices.size();
    for (int i = 0; i < n_local; ++i) {
        const plspm_model* m = models[i];
        if (!m) return bad(PLSPM_E_ARG, "plspm_group_create: null handle");
        if (m->group) return bad(PLSPM_E_STATE, "plspm_group_create: a handle belongs to one group at a time");
        if (m->device != c->devices[i]) return bad(PLSPM_E_ARG, "plspm_group_create: handle i must live on the communicator's device i");
        if (!m->d_Xa || m->N < 2) return bad(PLSPM_E_STATE, "plspm_group_create: every handle needs its data uploaded first");
        if (m->stage1) return bad(PLSPM_E_ARG, "plspm_group_create: pass the data-holding first stage of a two-stage pair");
        for (int j = 0; j < i; ++j) if (models[j] == m) return bad(PLSPM_E_ARG, "plspm_group_create: the same handle twice");
        if (plspm_row_stride(m) != plspm_row_stride(models[0]) || m->N != models[0]->N || m->P != models[0]->P || m->L != models[0]->L)
            return bad(PLSPM_E_ARG, "plspm_group_create: the handles must hold the same model and data");
    }
    plspm_group* g = new (std::nothrow) plspm_group();
    if (!g) return bad(PLSPM_E_STATE, "out of host memory");
    g->comm = c; g->nranks = c->nranks; g->first_rank = c->first_rank; g->use_rccl = c->use_rccl;
    g->loc.resize(n_local);
    for (int i = 0; i < n_local; ++i) { g->loc[i].m = models[i]; if (c->use_rccl) g->loc[i].comm = c->comms[i]; }
    auto bail = [&](const std::string& why) { plspm_group_destroy(g); g_group_create_error = why; return (plspm_group_t*)nullptr; };
    for (auto& l : g->loc) {
        l.m->group = g;
        if (hipSetDevice(l.m->device) != hipSuccess || plspm_stream_acquire(&l.cstream) != hipSuccess) return bail("gather stream creation failed");
        for (int s = 0; s < 2; ++s) {
            if (hipEventCreateWithFlags(&l.gathered[s], hipEventDisableTiming) != hipSuccess) return bail("event creation failed");
            for (auto& e : l.computed[s]) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail("event creation failed");
        }
        if (plspm_dmalloc((void**)&l.d_word, 64) != hipSuccess || hipMemset(l.d_word, 0, 64) != hipSuccess || plspm_hmalloc((void**)&l.h_word, 64 + 8 * (size_t)std::max(8, c->nranks)) != hipSuccess)
            return bail("scratch allocation failed");
    }
    if (n_local > 1) {
        try { g->crew.reset(new ShardCrew()); g->crew->start(n_local); }
        catch (...) { g->crew.reset(); }           // no threads: the caller's thread enqueues every shard
    }
    c->bound = g;
    return g;
}

int32_t plspm_group_size(const plspm_group_t* g) { return g ? g->nranks : 0; }

int plspm_group_shard(const plspm_group_t* g, int64_t B, int32_t rank, int64_t* first, int64_t* count) {
    if (!g || B < 0 || rank < 0 || rank >= g->nranks || !first || !count) return PLSPM_E_ARG;
    shard_of(B, g->nranks, rank, first, count);
    return 0;
}

int plspm_group_sync(plspm_group_t* g) {
    if (!g) return PLSPM_E_ARG;
    if (g->loc.empty()) return gfail(g, PLSPM_E_STATE, "the group's handles were destroyed");
    return sync_all(g);
}

// How a call of B replicates is cut into sub-batches on this group ("chunks" / "chunk_ratio"): K ranges of global replicate ids, each sharded
// over the ranks like a call of its own.  A function of rank-INVARIANT inputs only -- B, nranks, the group's options, the record width and the
// alignment the ranks agreed on (agree_units) -- so every rank of a job issues the same number of collectives of the same sizes whatever
// happened to it locally.
static bool plan_consults_units(const plspm_group* g) {
    const bool nothing_to_hide = g->nranks == 1 || (g->comm && g->comm->transport == 3);
    return !(nothing_to_hide && g->opt_chunks <= 0) && g->opt_chunks != 1 && g->opt_align <= 0;
}

static int plan_sub_batches(const plspm_group* g, int64_t B, int RS, int64_t* sub_first, int64_t* sub_B) {
    const int64_t per_rank = (B + g->nranks - 1) / g->nranks;
    int64_t parts[kBootChunksMax];
    // (one rank, or ranks that share a device -- one copy launch of microseconds: nothing travels, nothing to hide; unless sub-batches are asked for by number)
    const bool nothing_to_hide = g->nranks == 1 || (g->comm && g->comm->transport == 3);
    // (not agreed yet -- a forecast through plspm_group_plan before the first call: this rank's own figure as its handle stands, nothing built for it)
    const int64_t units = g->opt_align > 0 ? g->opt_align : (g->agreed_units > 0 ? g->agreed_units : plspm_detail_round_units_peek(g->loc[0].m));
    const int K = (nothing_to_hide && g->opt_chunks <= 0) ? 1 : plspm_detail_chunk_plan(per_rank, (int64_t)RS * (int64_t)sizeof(double), g->opt_chunks, g->opt_ratio, parts, units);
    int64_t at = 0;
    int n = 0;
    for (int k = 0; k < K && at < B; ++k) {
        const int64_t take = (k == K - 1) ? B - at : std::min<int64_t>(B - at, parts[k] * g->nranks);
        sub_first[n] = at; sub_B[n] = take; at += take; ++n;
    }
    if (at < B) sub_B[n - 1] += B - at;
    return n;
}

int plspm_chunk_plan(int64_t B, int64_t bytes_per_unit, int32_t chunks, int32_t ratio_pct, int64_t align, int64_t* parts) {
    if (!parts || B < 1 || bytes_per_unit < 1 || chunks < 0 || chunks > kBootChunksMax || ratio_pct < 10 || ratio_pct > 100 || align < 0) return -PLSPM_E_ARG;
    return plspm_detail_chunk_plan(B, bytes_per_unit, chunks, ratio_pct, parts, align);
}

int plspm_group_set_option(plspm_group_t* g, const char* key, int32_t value) {
    if (!g || !key) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: bad arguments");
    const std::string k(key);
    if (k == "chunks") { if (value < 0 || value > kBootChunksMax) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: chunks in 0 .. 8"); g->opt_chunks = value; }
    else if (k == "chunk_align") { if (value < 0 || value > (1 << 20)) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: chunk_align in 0 .. 2^20"); g->opt_align = value; }
    else if (k == "lean_events") { if (value < 0 || value > 1) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: lean_events 0 | 1"); g->opt_lean_events = value; }
    else if (k == "gather_root") {
        if (value < 0 || value > 1) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: gather_root 0 | 1");
        if (value && g->use_rccl && (!g_rccl.Send || !g_rccl.Recv || !g_rccl.Broadcast)) return gfail(g, PLSPM_E_STATE, "plspm_group_set_option: this librccl has no ncclSend / ncclRecv / ncclBroadcast");
        g->opt_gather_root = value;
    }
    else if (k == "skip_exchange") { if (value < 0 || value > 1) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: skip_exchange 0 | 1"); g->opt_skip_exchange = value; }
    else if (k == "events_device_scope") {
        if (value < 0 || value > 1) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: events_device_scope 0 | 1");
        int rc = sync_all(g);
        if (rc) return rc;
        for (auto& l : g->loc) {
            GHIP(g, hipSetDevice(l.m->device));
            for (int s2 = 0; s2 < 2; ++s2) {
                const unsigned flags = hipEventDisableTiming | (value ? hipEventReleaseToDevice : 0u);
                hipEventDestroy(l.gathered[s2]); l.gathered[s2] = nullptr;
                GHIP(g, hipEventCreateWithFlags(&l.gathered[s2], flags));
                for (auto& e : l.computed[s2]) { hipEventDestroy(e); e = nullptr; GHIP(g, hipEventCreateWithFlags(&e, flags)); }
            }
        }
        g->pending[0] = g->pending[1] = false;
    }
    else if (k == "chunk_ratio") { if (value < 10 || value > 100) return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: chunk_ratio in 10 .. 100"); g->opt_ratio = value; }
    else return gfail(g, PLSPM_E_ARG, "plspm_group_set_option: unknown option '" + k + "'");
    return 0;
}

int plspm_group_plan(const plspm_group_t* g, int64_t B, int32_t* n_sub, int64_t* sub_first, int64_t* sub_count) {
    if (!g || B < 1 || !n_sub || !sub_first || !sub_count || g->loc.empty()) return PLSPM_E_ARG;
    *n_sub = plan_sub_batches(g, B, plspm_row_stride(g->loc[0].m), sub_first, sub_count);
    return 0;
}

int plspm_group_bootstrap(plspm_group_t* g, int64_t B, uint64_t seed, int64_t rep_offset) {
    if (!g || B < 1 || rep_offset < 0 || B > ((int64_t)1 << 30)) return gfail(g, PLSPM_E_ARG, "plspm_group_bootstrap: bad arguments (1 <= B <= 2^30, rep_offset >= 0)");
    if (g->loc.empty()) return gfail(g, PLSPM_E_STATE, "the group's handles were destroyed");
    const int nl = (int)g->loc.size();
    const int RS = plspm_row_stride(g->loc[0].m);
    // sub-batches (round 5): the all-gather of sub-batch k runs on the gather streams beside the shard kernels of sub-batch k + 1, so that ONE
    // call hides its merge the way a loop of calls does; the gathered buffer holds the sub-batches one after the other, each in the layout of a
    // call of its own -- i.e. in replicate-id order throughout (what the device summaries' fixed-order sums rely on)
    int64_t sub_first[kBootChunksMax], sub_B[kBootChunksMax], sub_cap[kBootChunksMax], sub_off[kBootChunksMax];
    if (plan_consults_units(g) && g->agreed_units <= 0) {
        // every rank takes this branch alike (group options, rank count, and an agreement voided only by calls all ranks make); what a rank
        // contributes is its own business -- a rank whose planes could not be built says 64 and fails in its shard below, BEHIND the collectives
        double units = 64.0;
        for (auto& l : g->loc) units = std::max(units, (double)plspm_detail_round_units(l.m));
        int arc = plspm_group_max(g, &units);
        if (arc) return arc;
        g->agreed_units = (int64_t)units;
    }
    const int K = plan_sub_batches(g, B, RS, sub_first, sub_B);
    int64_t cap = 0;
    for (int k = 0; k < K; ++k) { sub_cap[k] = (sub_B[k] + g->nranks - 1) / g->nranks; sub_off[k] = cap; cap += sub_cap[k]; }
    const size_t send_bytes = (size_t)cap * RS * sizeof(double), recv_bytes = send_bytes * g->nranks;
    const int s = g->next_slot;
    int rc;
    bool must_grow = false;
    for (auto& l : g->loc) if (l.send[s].cap < send_bytes || l.recv[s].cap < recv_bytes) must_grow = true;
    if (must_grow) {                                   // nothing may be in flight on a buffer that is about to be replaced
        if ((rc = sync_all(g))) return rc;
        g->pending[0] = g->pending[1] = false; g->last_slot = -1;
        for (auto& l : g->loc) {
            GHIP(g, hipSetDevice(l.m->device));
            for (int k = 0; k < 2; ++k) if ((rc = grow(g, l, l.send[k], send_bytes)) || (rc = grow(g, l, l.recv[k], recv_bytes))) return rc;
        }
    }
    // handles sharing one device (tests, single-GPU runs of this path) exchange their records in ONE launch instead of G x G copies
    bool one_device = !g->use_rccl && nl <= PLSPM_GATHER_LOCAL_MAX;
    for (int i = 1; i < nl && one_device; ++i) one_device = g->loc[i].m->device == g->loc[0].m->device;
    // 1. shard kernels (enqueue only; non-metric models iterate with host read-backs): every local handle by its own resident thread
    std::vector<int> shard_rc(nl, 0), shard_done(nl, 0);      // shard_done[i]: sub-batches of handle i whose `computed` event is recorded
    auto run_shard = [&](int i) {
        Local& l = g->loc[i];
        plspm_model* m = l.m;
        if (hipSetDevice(m->device) != hipSuccess) { shard_rc[i] = fail(m, PLSPM_E_STATE, "hipSetDevice failed"); return; }
        if (g->pending[s]) {
            // slot s was last read / written by the collective of two calls ago -- long finished, as a rule: then no wait is enqueued at all (a wait
            // is a barrier packet the queue drains the device for even when its event has fired: ~5 us of a 0.48 ms step)
            auto wait_unless_done = [&](hipEvent_t e) { if (!g->opt_lean_events || hipEventQuery(e) != hipSuccess) { (void)hipGetLastError(); hipStreamWaitEvent(m->stream, e, 0); } };
            if (g->use_rccl || one_device) wait_unless_done(l.gathered[s]);            // (one launch / one collective read every send buffer)
            else for (auto& peer : g->loc) wait_unless_done(peer.gathered[s]);          // peers pull from this send buffer
        }
        for (int k = 0; k < K; ++k) {
            int64_t first = 0, count = 0;
            shard_of(sub_B[k], g->nranks, g->first_rank + i, &first, &count);
            double* send = (double*)l.send[s].p + sub_off[k] * RS;
            // (a full shard: its last kernel signals `computed` itself; a ragged one records the event behind the padding below)
            const bool fused = g->opt_lean_events && count == sub_cap[k] && count > 0;
            m->stop_event = fused ? l.computed[s][k] : nullptr;
            if (count > 0 && (shard_rc[i] = plspm_detail_bootstrap(m, count, seed, rep_offset + sub_first[k] + first, nullptr, send))) { m->stop_event = nullptr; return; }
            const bool signalled = fused && m->stop_event == nullptr;      // (taken by the solver launch; left in place by routes that do not end in it)
            m->stop_event = nullptr;
            if (signalled) { shard_done[i] = k + 1; continue; }
            if (count < sub_cap[k] && hipMemsetAsync(send + count * RS, 0xFF, (size_t)(sub_cap[k] - count) * RS * sizeof(double), m->stream) != hipSuccess) {   // NaN status: not a replicate
                shard_rc[i] = fail(m, PLSPM_E_STATE, "hipMemsetAsync failed"); return;
            }
            if (hipEventRecord(l.computed[s][k], m->stream) != hipSuccess) { shard_rc[i] = fail(m, PLSPM_E_STATE, "hipEventRecord failed"); return; }
            shard_done[i] = k + 1;
        }
    };
    const auto t_a = std::chrono::steady_clock::now();
    if (nl > 1 && g->crew) g->crew->run(run_shard);
    else for (int i = 0; i < nl; ++i) run_shard(i);
    const auto t_b = std::chrono::steady_clock::now();
    // A failed shard (out of memory, an LDS limit, a read-back error of a non-metric model) must not keep this rank out of the
    // collectives: the other ranks of the job are already inside them and would wait forever.  Its send buffer becomes NaN-status
    // records (never counted as replicates) from the failed sub-batch on, every all-gather runs as planned, and the error is reported afterwards.
    int first_bad = -1;
    for (int i = 0; i < nl; ++i) {
        if (!shard_rc[i]) continue;
        if (first_bad < 0) first_bad = i;
        Local& l = g->loc[i];
        hipSetDevice(l.m->device);
        (void)hipGetLastError();
        const int k0 = shard_done[i];
        hipMemsetAsync((double*)l.send[s].p + sub_off[k0] * RS, 0xFF, (size_t)(cap - sub_off[k0]) * RS * sizeof(double), l.m->stream);
        for (int k = k0; k < K; ++k) hipEventRecord(l.computed[s][k], l.m->stream);
    }
    // 2. the collective of every sub-batch, on the gather streams behind that sub-batch's shard kernels
    const bool root_only = g->opt_gather_root != 0 && g->nranks > 0;
    int crc = 0;
    std::string cwhy;
    for (int k = 0; k < K; ++k) {
        const size_t sub_doubles = (size_t)sub_cap[k] * RS, sub_bytes = sub_doubles * sizeof(double);
        const size_t soff = (size_t)sub_off[k] * RS, roff = soff * g->nranks;       // (in doubles)
        if (g->opt_skip_exchange) {
            for (auto& l : g->loc) { hipSetDevice(l.m->device); hipStreamWaitEvent(l.cstream, l.computed[s][k], 0); }
        } else if (g->use_rccl) {
            const Rccl* r = &g_rccl;
            for (auto& l : g->loc) {
                hipError_t e = hipSetDevice(l.m->device);
                if (e == hipSuccess) e = hipStreamWaitEvent(l.cstream, l.computed[s][k], 0);
                if (e != hipSuccess && !crc) { crc = -(int)e; cwhy = std::string("hipStreamWaitEvent: ") + hipGetErrorString(e); }
            }
            ncclResult_t n = r->GroupStart();
            if (n != ncclSuccess) { if (!crc) { crc = -(1000 + (int)n); cwhy = std::string("ncclGroupStart: ") + r->GetErrorString(n); } }
            else {
                // between GroupStart and GroupEnd nothing returns early: an open group call would poison every later RCCL call of the process
                for (int i = 0; i < nl; ++i) {
                    Local& l = g->loc[i];
                    hipSetDevice(l.m->device);
                    if (root_only) {
                        // the gather SURVEY 8(e) names: every rank sends its shard to rank 0, rank 0 posts one receive per rank (its own included) -- one group call
                        n = r->Send((const double*)l.send[s].p + soff, sub_doubles, ncclDouble, 0, l.comm, l.cstream);
                        if (n == ncclSuccess && g->first_rank + i == 0)
                            for (int src = 0; src < g->nranks && n == ncclSuccess; ++src)
                                n = r->Recv((double*)l.recv[s].p + roff + (size_t)src * sub_doubles, sub_doubles, ncclDouble, src, l.comm, l.cstream);
                        if (n != ncclSuccess && !crc) { crc = -(1000 + (int)n); cwhy = std::string("ncclSend / ncclRecv: ") + r->GetErrorString(n); }
                        continue;
                    }
                    n = r->AllGather((const double*)l.send[s].p + soff, (double*)l.recv[s].p + roff, sub_doubles, ncclDouble, l.comm, l.cstream);
                    if (n != ncclSuccess && !crc) { crc = -(1000 + (int)n); cwhy = std::string("ncclAllGather: ") + r->GetErrorString(n); }
                }
                n = r->GroupEnd();
                if (n != ncclSuccess && !crc) { crc = -(1000 + (int)n); cwhy = std::string("ncclGroupEnd: ") + r->GetErrorString(n); }
            }
        } else if (one_device) {
            Local& l0 = g->loc[0];
            hipError_t e = hipSetDevice(l0.m->device);
            const double* send[PLSPM_GATHER_LOCAL_MAX];
            double* recv[PLSPM_GATHER_LOCAL_MAX];
            for (int i = 0; i < nl; ++i) {
                if (e == hipSuccess) e = hipStreamWaitEvent(l0.cstream, g->loc[i].computed[s][k], 0);
                send[i] = (const double*)g->loc[i].send[s].p + soff; recv[i] = (double*)g->loc[i].recv[s].p + roff;
            }
            if (e == hipSuccess && plspm_detail_gather_local(l0.cstream, nl, send, recv, sub_doubles, root_only ? 1 : nl)) e = hipErrorLaunchFailure;
            if (e != hipSuccess && !crc) { crc = -(int)e; cwhy = std::string("record exchange: ") + hipGetErrorString(e); }
        } else {
            // copy-engine exchange (single process, peer-mapped buffers; plspm_comm_create_ex transport 2): every rank PULLS the peers' shards with
            // device-to-device copies on its own gather stream -- no kernel, no CU: the SDMA engines move the records over xGMI
            for (int d = 0; d < (root_only ? 1 : nl); ++d) {      // (gather_root: only rank 0's copy engines pull)
                Local& dst = g->loc[d];
                hipSetDevice(dst.m->device);
                for (int i = 0; i < nl; ++i) {
                    hipError_t e = hipStreamWaitEvent(dst.cstream, g->loc[i].computed[s][k], 0);
                    if (e == hipSuccess) e = hipMemcpyAsync((double*)dst.recv[s].p + roff + (size_t)i * sub_doubles, (const double*)g->loc[i].send[s].p + soff, sub_bytes, hipMemcpyDeviceToDevice, dst.cstream);
                    if (e != hipSuccess && !crc) { crc = -(int)e; cwhy = std::string("record exchange: ") + hipGetErrorString(e); }
                }
            }
        }
    }
    if (one_device) { for (int i = 0; i < nl; ++i) { hipError_t e = hipEventRecord(g->loc[i].gathered[s], g->loc[0].cstream); if (e != hipSuccess && !crc) { crc = -(int)e; cwhy = "hipEventRecord failed"; } } }
    else for (auto& l : g->loc) { hipSetDevice(l.m->device); hipEventRecord(l.gathered[s], l.cstream); }
    g->t_shards_ms = std::chrono::duration<double, std::milli>(t_b - t_a).count();
    g->t_exchange_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b).count();
    if (first_bad >= 0 || crc) {
        // the slot's buffers were handed to the collective: they count as in flight, but the call has no result
        g->pending[s] = true; g->next_slot = s ^ 1; g->last_slot = -1;
        if (first_bad >= 0) return gfail(g, shard_rc[first_bad], "shard of rank " + std::to_string(g->first_rank + first_bad) + ": " + g->loc[first_bad].m->error);
        return gfail(g, crc, cwhy);
    }
    g->pending[s] = true; g->last_slot = s; g->next_slot = s ^ 1; g->last_B = B; g->last_cap = cap; g->peers_checked = -1;
    g->last_root_only = root_only;
    g->last_K = K;
    for (int k = 0; k < K; ++k) { g->sub_B[k] = sub_B[k]; g->sub_first[k] = sub_first[k]; g->sub_cap[k] = sub_cap[k]; g->sub_off[k] = sub_off[k]; }
    return 0;
}

// Segment (sub-batch k, rank r) of the gathered records of the last call: its first record in the receive buffer, the global id of its first
// replicate within the call, and how many replicates it holds.
static void segment_of(const plspm_group* g, int k, int r, int64_t* rec0, int64_t* first, int64_t* count) {
    int64_t f = 0, c = 0;
    shard_of(g->sub_B[k], g->nranks, r, &f, &c);
    *rec0 = g->nranks * g->sub_off[k] + (int64_t)r * g->sub_cap[k];
    *first = g->sub_first[k] + f; *count = c;
}

// A rank whose shard failed still joins every all-gather (its records are NaN-status filler from the failed sub-batch on, so the other ranks never
// wait for it) and reports the error from ITS plspm_group_bootstrap.  The other ranks learn of it here, before any result leaves the group: the
// status word of the first record of every rank's shard of the LAST sub-batch is read back once per bootstrap (nranks 8-byte copies); NaN where
// replicates were due = that rank had nothing to contribute.  Called with local handle 0's device current, by everything that hands out
// records or statistics of the last bootstrap (all of which wait for the collective anyway).
static int check_peer_shards(plspm_group* g) {
    const int s = g->last_slot;
    if (g->peers_checked == s) return g->peers_rc ? gfail(g, g->peers_rc, g->error) : 0;
    Local& l = g->loc[0];
    const int RS = plspm_row_stride(l.m);
    double* h = l.h_word + 8;
    const int kl = g->last_K - 1;
    GHIP(g, hipStreamWaitEvent(l.cstream, l.gathered[s], 0));
    for (int r = 0; r < g->nranks; ++r) {
        int64_t rec0 = 0, first = 0, count = 0;
        segment_of(g, kl, r, &rec0, &first, &count);
        GHIP(g, hipMemcpyAsync(h + r, (const double*)l.recv[s].p + rec0 * RS + (RS - 2), sizeof(double), hipMemcpyDeviceToHost, l.cstream));
    }
    GHIP(g, hipStreamSynchronize(l.cstream));
    g->peers_checked = s; g->peers_rc = 0;
    for (int r = 0; r < g->nranks; ++r) {
        int64_t rec0 = 0, first = 0, count = 0;
        segment_of(g, kl, r, &rec0, &first, &count);
        if (count > 0 && h[r] != h[r]) {
            g->peers_rc = PLSPM_E_STATE;
            return gfail(g, PLSPM_E_STATE, "the shard of rank " + std::to_string(r) + " failed on its rank (no replicates arrived from it): the bootstrap has no complete result");
        }
    }
    return 0;
}

int plspm_group_records(plspm_group_t* g, int32_t local, void** d_records, int64_t* n_records, int32_t* stride) {
    if (!g || local < 0 || local >= (int)g->loc.size()) return gfail(g, PLSPM_E_ARG, "plspm_group_records: bad arguments");
    if (g->last_slot < 0) return gfail(g, PLSPM_E_STATE, "plspm_group_records: no bootstrap on this group yet");
    if (g->last_root_only && g->first_rank + local != 0) return gfail(g, PLSPM_E_STATE, "plspm_group_records: the records of the last bootstrap were gathered to rank 0 only (gather_root)");
    Local& l = g->loc[local];
    GHIP(g, hipSetDevice(l.m->device));
    GHIP(g, hipEventSynchronize(l.gathered[g->last_slot]));
    { int prc; GHIP(g, hipSetDevice(g->loc[0].m->device)); if ((prc = check_peer_shards(g))) return prc; GHIP(g, hipSetDevice(l.m->device)); }
    if (d_records) *d_records = l.recv[g->last_slot].p;
    if (n_records) *n_records = g->last_cap * g->nranks;
    if (stride) *stride = plspm_row_stride(l.m);
    return 0;
}

int plspm_group_summary(plspm_group_t* g, const double* original, double* summary, int64_t* n_used) {
    if (!g || !original || !summary) return gfail(g, PLSPM_E_ARG, "plspm_group_summary: bad arguments");
    if (g->last_slot < 0) return gfail(g, PLSPM_E_STATE, "plspm_group_summary: no bootstrap on this group yet");
    Local& l = g->loc[0];
    GHIP(g, hipSetDevice(l.m->device));
    const bool other_processes = g->use_rccl && (int)g->loc.size() != g->nranks;
    if (g->last_root_only && other_processes) {
        // gather_root in a one-process-per-GPU job: rank 0 holds the records, computes the table and sends [rc | n_used | table] to every rank in one
        // broadcast -- every rank makes this call (as with the all-gather, where every rank computed the same table from its own copy)
        const int R = plspm_row_width(l.m);
        const size_t words = 2 + (size_t)R * 6, bytes = words * sizeof(double);
        if (l.bcast.cap < bytes) { GHIP(g, hipStreamSynchronize(l.cstream)); int grc = grow(g, l, l.bcast, bytes); if (grc) return grc; }
        std::vector<double> host(words, 0.0);
        int rc0 = 0;
        if (g->first_rank == 0) {
            GHIP(g, hipStreamWaitEvent(l.m->stream, l.gathered[g->last_slot], 0));
            int64_t used = 0;
            rc0 = check_peer_shards(g);
            if (!rc0) { rc0 = plspm_detail_summary(l.m, (const double*)l.recv[g->last_slot].p, g->last_cap * g->nranks, plspm_row_stride(l.m), original, host.data() + 2, &used); if (rc0) g->error = l.m->error; }
            host[0] = (double)rc0; host[1] = (double)used;
            GHIP(g, hipMemcpyAsync(l.bcast.p, host.data(), bytes, hipMemcpyHostToDevice, l.cstream));
        }
        const Rccl* r = &g_rccl;
        GNCCL(g, r, r->Broadcast(l.bcast.p, l.bcast.p, words, ncclDouble, 0, l.comm, l.cstream));
        if (g->first_rank != 0) GHIP(g, hipMemcpyAsync(host.data(), l.bcast.p, bytes, hipMemcpyDeviceToHost, l.cstream));
        GHIP(g, hipStreamSynchronize(l.cstream));
        if (g->first_rank == 0 && rc0) return gfail(g, rc0, g->error);
        if (host[0] != 0.0) return gfail(g, (int)host[0], "plspm_group_summary: rank 0 reports a failed bootstrap (see its error)");
        memcpy(summary, host.data() + 2, (size_t)R * 6 * sizeof(double));
        if (n_used) *n_used = (int64_t)host[1];
        return 0;
    }
    GHIP(g, hipStreamWaitEvent(l.m->stream, l.gathered[g->last_slot], 0));
    int rc = check_peer_shards(g);
    if (rc) return rc;
    rc = plspm_detail_summary(l.m, (const double*)l.recv[g->last_slot].p, g->last_cap * g->nranks, plspm_row_stride(l.m), original, summary, n_used);
    if (rc) return gfail(g, rc, l.m->error);
    return 0;
}

int plspm_group_rows(plspm_group_t* g, double* out, int32_t* status, int32_t* iters) {
    if (!g) return PLSPM_E_ARG;
    if (g->last_slot < 0) return gfail(g, PLSPM_E_STATE, "plspm_group_rows: no bootstrap on this group yet");
    if (g->last_root_only && g->first_rank != 0) return gfail(g, PLSPM_E_STATE, "plspm_group_rows: the records of the last bootstrap were gathered to rank 0 only (gather_root)");
    Local& l = g->loc[0];
    const int RS = plspm_row_stride(l.m), R = RS - 2;
    GHIP(g, hipSetDevice(l.m->device));
    GHIP(g, hipStreamWaitEvent(l.m->stream, l.gathered[g->last_slot], 0));
    { int prc = check_peer_shards(g); if (prc) return prc; }
    const double* rec = (const double*)l.recv[g->last_slot].p;
    for (int k = 0; k < g->last_K; ++k)
        for (int r = 0; r < g->nranks; ++r) {
            int64_t rec0 = 0, first = 0, count = 0;
            segment_of(g, k, r, &rec0, &first, &count);
            if (!count) continue;
            int rc = plspm_detail_fetch_records(l.m, rec + (size_t)rec0 * RS, count, RS, out ? out + first * R : nullptr, status ? status + first : nullptr,
                                                iters ? iters + first : nullptr);
            if (rc) return gfail(g, rc, l.m->error);
        }
    return 0;
}

// The records of the last bootstrap move into local handle 0's own record buffer, compacted to replicate-id order: afterwards
// plspm_bootstrap_fetch / plspm_bootstrap_summary on that handle serve them and the group (with its communicator slot) is free.
int plspm_group_adopt(plspm_group_t* g) {
    if (!g) return PLSPM_E_ARG;
    if (g->last_slot < 0 || g->loc.empty()) return gfail(g, PLSPM_E_STATE, "plspm_group_adopt: no bootstrap result on this group");
    if (g->last_root_only && g->first_rank != 0) return gfail(g, PLSPM_E_STATE, "plspm_group_adopt: the records of the last bootstrap were gathered to rank 0 only (gather_root)");
    Local& l = g->loc[0];
    plspm_model* m = l.m;
    const int RS = plspm_row_stride(m);
    GHIP(g, hipSetDevice(m->device));
    m->rows_B = 0;
    int rc = check_peer_shards(g);             // (no extra round trip in the Plspm flow: plspm_group_summary ran -- and waited -- before)
    if (rc) return rc;
    rc = ensure(m, m->rows, (size_t)g->last_B * RS * sizeof(double));
    if (rc) return gfail(g, rc, m->error);
    GHIP(g, hipStreamWaitEvent(m->stream, l.gathered[g->last_slot], 0));
    const double* rec = (const double*)l.recv[g->last_slot].p;
    for (int k = 0; k < g->last_K; ++k)
        for (int r = 0; r < g->nranks; ++r) {
            int64_t rec0 = 0, first = 0, count = 0;
            segment_of(g, k, r, &rec0, &first, &count);
            if (count) GHIP(g, hipMemcpyAsync((double*)m->rows.p + first * RS, rec + (size_t)rec0 * RS, (size_t)count * RS * sizeof(double), hipMemcpyDeviceToDevice, m->stream));
        }
    m->rows_B = g->last_B;
    return 0;
}

int plspm_group_barrier(plspm_group_t* g) {
    if (!g) return PLSPM_E_ARG;
    if (g->loc.empty()) return gfail(g, PLSPM_E_STATE, "the group's handles were destroyed");
    int rc = sync_all(g);
    if (rc || !g->use_rccl) return rc;
    const Rccl* r = &g_rccl;
    GNCCL(g, r, r->GroupStart());
    for (auto& l : g->loc) {
        GHIP(g, hipSetDevice(l.m->device));
        GNCCL(g, r, r->AllReduce(l.d_word, l.d_word + 1, 1, ncclDouble, ncclMax, l.comm, l.cstream));
    }
    GNCCL(g, r, r->GroupEnd());
    for (auto& l : g->loc) { GHIP(g, hipSetDevice(l.m->device)); GHIP(g, hipStreamSynchronize(l.cstream)); }
    return 0;
}

int plspm_group_enqueue_times(const plspm_group_t* g, double* shards_ms, double* exchange_ms) {
    if (!g) return PLSPM_E_ARG;
    if (shards_ms) *shards_ms = g->t_shards_ms;
    if (exchange_ms) *exchange_ms = g->t_exchange_ms;
    return 0;
}

int plspm_group_max(plspm_group_t* g, double* value) {
    if (!g || !value) return PLSPM_E_ARG;
    if (g->loc.empty()) return gfail(g, PLSPM_E_STATE, "the group's handles were destroyed");
    if (!g->use_rccl || (int)g->loc.size() == g->nranks) return 0;         // every rank lives in this process: the caller's value is the job's
    const Rccl* r = &g_rccl;
    Local& l = g->loc[0];
    GHIP(g, hipSetDevice(l.m->device));
    l.h_word[0] = *value;
    GHIP(g, hipMemcpyAsync(l.d_word + 2, l.h_word, sizeof(double), hipMemcpyHostToDevice, l.cstream));
    GNCCL(g, r, r->AllReduce(l.d_word + 2, l.d_word + 3, 1, ncclDouble, ncclMax, l.comm, l.cstream));
    GHIP(g, hipMemcpyAsync(l.h_word + 1, l.d_word + 3, sizeof(double), hipMemcpyDeviceToHost, l.cstream));
    GHIP(g, hipStreamSynchronize(l.cstream));
    *value = l.h_word[1];
    return 0;
}

}  // extern "C"
