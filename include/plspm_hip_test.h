/*
 * plspm_hip_test.h -- test seams of libplspm_hip.so: entry points and option values that exist for the parity tests, the benches and the
 * experiments build, NOT part of the production ABI a reference maintainer binds (include/plspm_hip.h).  The release library exports them
 * (tests/ and bench.py call them through ctypes like everything else); nothing in plspm-python_amd/plspm's estimator path depends on them.
 */
#ifndef PLSPM_HIP_TEST_H
#define PLSPM_HIP_TEST_H

#include "plspm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Options of plspm_model_set_option that only tests use (release library):
 *   "i8_short_rows"   -1 (default) | n   n short tile rows behind as many tall ones as it takes (forced cuts of the int8 Gram's tile rows:
 *                     the matrices must be bit-identical for every cut)
 *   "gram_lds_kb"     0 .. 160   pads the dynamic LDS of gram_rows_kernel (occupancy experiments)
 *
 * Experiments build only (make -C plspm-python_amd/csrc experiments, loaded through PLSPM_HIP_LIB; the release library answers PLSPM_E_ARG):
 * "i8_waves" 4 (four-wave forms of the round-3 kernel: measured equal), "i8_shape" 32 (v_mfma_i32_32x32x32_i8 layout: 16 % slower), "i8_sched" 1
 * (persistent stream-K launch: kernel -3 %, step unchanged, and two such launches sharing a device can starve each other), "i8_rt" 8 (128-replicate
 * tile: 2.6 % slower), "resample_aux" 1 .. 3 (counts drawn on a second stream: inside the spread), "i8_variant" (schedule variants and ablation
 * probes of both Gram kernels).  Read-only "build_experiments" tells which library is loaded.  DESIGN.md 7b has the measurements.
 *
 */

/* Test seam: only the resample + Gram stages of plspm_bootstrap (on the Gram path the handle's "gram_path" option selects).
 *   idx   NULL (on-device Philox draws) or [B*N] explicit row indices;   out [B * C * C], C = device columns + 1 (after the data
 *   columns the missing indicators of plspm_model_set_missing, last the ones column): the replicate's full symmetric matrix
 *   sum_i c_i [x_i - shift, 1][x_i - shift, 1]' of the uploaded (mean-shifted) columns.  Metric handles only. */
int plspm_bootstrap_moments(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* idx, double* out);

/* The resample indices the on-device RNG uses for replicate `rep` (host-side mirror, for tests). */
int plspm_bootstrap_indices(uint64_t seed, int64_t rep, int64_t N, int32_t* idx);

/* Test seam, host arithmetic only (no device is touched): how the six-plane int8 Gram cuts `count_tiles` (16 replicates each) x
 * `pair_tiles` (32 pair columns each) into tile rows on `cus` CUs ("i8_rt" 0).  *tall rows of 20 count tiles and, with `mix` != 0, *shrt
 * rows of 16 in one launch; returns 1 when that launch is taken, 0 when the 256-replicate kernel is no slower, PLSPM_E_ARG on bad sizes. */
int plspm_gram_tile_plan(int64_t count_tiles, int64_t pair_tiles, int32_t cus, int32_t mix, int32_t* tall, int32_t* shrt);

/* Test seam, host arithmetic only: the sub-batch sizes of ONE call of B units whose results leave the device behind the kernels (PCIe download of
 * plspm_bootstrap, "boot_chunks"; gather of plspm_group_bootstrap, "chunks", per rank): parts [8]; returns the number of parts.  chunks 0 =
 * automatic (one part below 2 MiB = B * bytes_per_unit of results, else up to three), ratio_pct: size of part k + 1 relative to part k; align: every part
 * but the last a multiple of it (0: 64 -- whole count tiles; the library passes the replicates of one ROUND of the device, read-only option
 * "boot_round_units" of a handle: 1,280 for the headline model on 256 CUs).  Handle option "boot_align" n > 0 overrides the alignment (A/B).
 * Group options for diagnosis: "skip_exchange" 1 (the step's events and kernels without moving the records -- NOT a result), "events_device_scope" 1
 * (the group's events release at device scope). */
int plspm_chunk_plan(int64_t B, int64_t bytes_per_unit, int32_t chunks, int32_t ratio_pct, int64_t align, int64_t* parts);

/* Test seam: the stop-rule value (weights.py:120) each of the first B problems of the handle's LAST non-metric run was decided on -- the criterion of
 * the iteration that stopped it (or of its last iteration) -- out[B].  For holding the three forms of the stop-rule pass ("nm_mfma", "nm_codes") against
 * each other on the VALUE, not only on the iteration counts.  PLSPM_E_STATE when no such run has happened. */
int plspm_nonmetric_criteria(plspm_model_t* m, int64_t B, double* out);

#ifdef __cplusplus
}
#endif
#endif /* PLSPM_HIP_TEST_H */
