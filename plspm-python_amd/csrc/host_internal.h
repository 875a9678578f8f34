// host_internal.h -- what the translation units of libplspm_hip.so share on the HOST side (round 4: the former one-file host is split by
// concern so that the units compile side by side): plspm_hip.hip (allocator, handles, options, upload, staging), plspm_fit.hip (fp64 Gram,
// solvers, non-metric iteration, single fit, operator seam), plspm_gram_i8.hip (digit planes + the int8 Gram of bootstrap batches),
// plspm_bootstrap.hip (bootstrap driver, record download, summaries), plspm_group.cpp (multi-GPU groups).  Every device kernel lives in
// exactly one unit's headers; the functions below are the seams between them.  Nothing here is part of the C-ABI (include/plspm_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/plspm_hip_test.h"
#include "solver_core.h"
#include "solver_nmg.h"
#include "solver_hoc.h"
#include "solver_nmx.h"
#include "solver_ops.h"
#include "solver_wave.h"
#include "solver_quad.h"
#include "solver_wave16.h"

using namespace plspm;

typedef double d4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ long lmin(long a, long b) { return a < b ? a : b; }

#include "model.h"

// where a solver launch writes: per-problem strides; null base pointers are skipped
struct SolverOut {
    double* row; long row_stride;
    int* status; int* iters;
    long long* marks;
    FitOutputs fit;     // single-fit extras (problem 0 only)
};

// ---- plspm_hip.hip
ModelDesc make_desc(const plspm_model* m);
HocDesc make_hoc_desc(const plspm_model* m2);
void set_geometry(plspm_model* m);
static constexpr size_t kPinHalf = (size_t)8 << 20;       // two halves of the handle's pinned staging area: the host copy of chunk k+1 overlaps the DMA of chunk k
int pin_ready(plspm_model* m);

// ---- plspm_fit.hip
size_t desc_lds_bytes(int P, int L, int ne, int nedge);
// moment matrix of ALL uploaded rows -> m->gram (dense fp64 MFMA Gram over row chunks + fixed-order reduce)
int dense_moments(plspm_model* m);
// fp64 MFMA Gram of `nproblems` replicates over their (row,count) lists -> tile-packed matrices at `out`
int launch_gram_lists(plspm_model* m, long nproblems, const int2* ent, const int* nent, long ent_stride, double* out);
int run_impute(plspm_model* m, long nproblems, const double* Min, const double** Mp, long* mp_stride);
int launch_solver(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so, int threads);
// the metric solver of a bootstrap batch: wave / rows solver on dense matrices (`dense`: the int8 Gram wrote that layout) or the LDS solver
int launch_batch_solver(plspm_model* m, long nb, bool dense, const SolverOut& so);
// Scale.NUM / RAW non-metric bootstrap batches as ONE solver launch (round 6; solver_wave16.h NM): does this model have such a kernel, and the launch itself
// (dense moment matrices at m->gram; maps / steps as kernels_solver.h solver_nmwave_kernel takes them; force + live: the replay of `nb` listed replicates)
bool nm_wave_solver_covers(const plspm_model* m);
int launch_nm_wave_solver(plspm_model* m, long nb, const SolverOut& so, double* maps, long maps_stride, int* steps, const int* force, const int* live);
size_t nm_state_doubles_of(const plspm_model* m);
size_t nm_dense_lds(const plspm_model* m, bool* whole, int* kb_out);
int run_nonmetric(plspm_model* m, long nproblems, const double* Mp, long mp_stride, const SolverOut& so_in, const int2* ent, const int* nent, long ent_stride, int threads,
                  bool finish = true, const void* cd8 = nullptr, int cd8_MT = 0, bool counts16_ready = false);
// second-stage moments of a HOC pair by congruence with the first stage's score maps: m->gram (stage 1) -> m2->gram
int run_hoc_moments(plspm_model* m, plspm_model* m2, long nb);

// ---- plspm_gram_i8.hip
static inline int i8_kblocks(long N) { return (int)((N + 127) / 128) * 2; }      // k-blocks of 64 rows, an even number
static inline long i8_pairs(const plspm_model* m) { const long C = m->Pg + 1; return C * (C + 1) / 2; }
bool nm_counts8_possible(const plspm_model* m);
int choose_gram_path(const plspm_model* m, int64_t B);      // 1 fp64 MFMA on (row,count) lists, 2 int8 digit planes
int prepare_zs_stats(plspm_model* m);
int prepare_zs(plspm_model* m);
int run_gram_i8(plspm_model* m, int64_t nb, uint64_t seed, int64_t rep0, const int32_t* d_idx, double* out, bool dense, bool* fallback, const void** counts = nullptr,
                int* counts_MT = nullptr, unsigned short* out16 = nullptr, bool* wrote16 = nullptr);
bool nm_wave_route_planned(const plspm_model* m);     // plspm_nonmetric.hip: Scale.NUM / RAW batches as one solver launch + verification (run_nonmetric_wave)
int run_nonmetric_wave(plspm_model* m, long nb, const SolverOut& so, const void* cd8, int cd8_MT);
bool nm_wave_step_planned(const plspm_model* m);      // plspm_nonmetric.hip: the categorical iteration of this handle runs one wave per problem (kernels_nmw.h)
