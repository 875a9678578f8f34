/*
 * plspm_hip.h -- C-ABI of libplspm_hip.so, the MI355X (gfx950) estimator backend for the PLS-PM
 * weight-solver + bootstrap hot path of GoogleCloudPlatform/plspm-python.
 *
 * The reference is pure Python and has no FFI of its own; this ABI is the seam a maintainer would bind
 * with ctypes in place of two internal call sites (paths relative to the reference repository):
 *
 *   (i)  WeightsCalculatorFactory.calculate(data, path)      plspm/weights.py:172-187
 *        called from Estimator.estimate                      plspm/estimator.py:39,52
 *        -> plspm_model_create + plspm_upload + plspm_fit
 *   (ii) Bootstrap.__init__ / BootstrapProcess.run           plspm/bootstrap.py:81-117, 45-73
 *        called from Plspm.__init__                          plspm/plspm.py:78-82
 *        -> plspm_bootstrap (replicates in [rep_offset, rep_offset + B) of one logical stream, so the
 *           result does not depend on how replicates are sharded over GPUs / processes)
 *        -> plspm_group_* : the fan-out over `processes` workers and the Queue merge (bootstrap.py:89-111)
 *           as replicate shards on several MI355X with ONE RCCL all-gather over xGMI at the end
 *
 * Conventions
 *   - plain C types only; all host buffers caller-allocated, row-major, IEEE fp64 unless noted.
 *   - "device column order": MVs grouped by LV in path-matrix order, blocks contiguous
 *     (block_offset[l] .. block_offset[l+1]); the caller supplies the map from its own columns.
 *   - every function returns 0 on success, > 0 for a numerical/argument condition, < 0 for a
 *     HIP runtime error; plspm_last_error() returns the text.  Nothing throws across the boundary.
 *   - one opaque handle per (model, device); handles are independent and may be used from different
 *     threads; a single handle is not re-entrant.
 *   - there is NO CPU fallback: without a usable HIP device plspm_model_create fails.
 */
#ifndef PLSPM_HIP_H
#define PLSPM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plspm_model plspm_model_t;

/* Scheme ids (reference plspm/scheme.py:57-63) and Mode ids (plspm/mode.py:64-69). */
enum { PLSPM_SCHEME_CENTROID = 0, PLSPM_SCHEME_FACTORIAL = 1, PLSPM_SCHEME_PATH = 2 };
enum { PLSPM_MODE_A = 0, PLSPM_MODE_B = 1 };
/* Per-problem status written by plspm_fit / plspm_bootstrap. */
enum {
    PLSPM_OK = 0,
    PLSPM_NOT_CONVERGED = 1, /* reference raises Exception("Could not converge ...") weights.py:185-186 */
    PLSPM_SINGULAR = 2,      /* a regression the reference cannot solve either (e.g. Mode B on a block with missing cells, mode.py:55-56) */
    PLSPM_NONFINITE = 3      /* zero-variance MV or NaN input                                            */
};
/* Argument errors returned by the API functions themselves. */
enum { PLSPM_E_ARG = 100, PLSPM_E_STATE = 101, PLSPM_E_LIMIT = 102 };

/* ABI version of this header; plspm_abi_version() of the loaded library must match. */
#define PLSPM_ABI_VERSION 4
int plspm_abi_version(void);

/* Number of HIP devices visible to the process (0 when there is none; never negative). */
int plspm_device_count(void);

/* The library keeps freed device / pinned-host blocks for re-use (a Plspm() call creates, fills and destroys a handle; without the
 * cache that is ~40 allocator round trips per call).  This hands every cached block back to the HIP runtime. */
int plspm_release_cached_memory(void);

/* Text of the last error on this handle (or of the last failed plspm_model_create when NULL). */
const char* plspm_last_error(const plspm_model_t* m);

/*
 * Options of a handle.  The defaults are the measured optima; tests and benches use the rest to force a route or a kernel form.  The library
 * reads NO environment variables.  Unknown keys / out-of-range values return PLSPM_E_ARG.
 *
 * Routes and precision
 *   "gram_path"       0 auto | 1 fp64 MFMA Gram over (row,count) lists | 2 int8 digit-plane Gram (exact integer product of the dense resample
 *                     multiplicities with the base-256 digit planes of the pair products x_p x_q).  Auto = 2 for every bootstrap whose planes
 *                     fit the memory budget and whose sums fit int32 (N < 2^24 rows): metric, mean-imputed, non-metric and categorical models,
 *                     on-device draws and explicit index lists alike (a chunk of an index list that carries a multiplicity above 127 takes
 *                     route 1); second stages of HOC pairs work on their first stage's matrices
 *   "i8_slices"       0 | 1 .. 8   digit planes per pair product.  0 (default) = automatic: 6 planes when every pair column of the uploaded data has
 *                     sum|z| >= 257 max|z| -- the worst-case error N 2^-47 max|z| of a replicate's sum is then below the a-priori bound N 2^-53 sum|z|
 *                     of an fp64 accumulation of the same terms, with a factor 4 to spare, even for a replicate that misses the column's largest
 *                     row -- else 7 (>= 54 significant bits of the column maximum), else -- sum|z| < 2 max|z|: one gross outlier row carries
 *                     the column, and a replicate without it sums products that are tiny against the scale of the planes -- 8; and never more
 *                     planes than carry anything: planes that are identically zero in the
 *                     seven-plane decomposition are dropped (0/1 indicator columns: ONE plane, bit-identical sums).  Read-only "last_i8_slices" /
 *                     "last_i8_ratio" report the choice and floor(min sum / max)
 *   "i8_min_slices"   0 (default) | 6 .. 8   the automatic choice, but never fewer planes (Plspm(precision="strict"): 7 -- the rate bench.py reports
 *                     as `value_strict`)
 *   "i8_min_batch"    auto mode takes the int8 route from this many replicates per call (default 1: always -- the route must not
 *                     depend on how a job is sharded, or shards of different size would differ in the last bits)
 * Int8 Gram kernels (all forms give bit-identical matrices: exact int32 sums)
 *   "i8_priv"         1 (default) | 0   six / seven planes: gram_i8p_kernel (round 4: four waves, count fragments straight from global memory into
 *                     registers, only the digit blocks through the LDS ring; tile rows of 320 / 256 replicates with six planes, 256 / 192 with
 *                     seven) or, 0, the round-3 gram_i8_kernel (eight waves, both operands through LDS: 9-10 % slower, kept as the A/B reference
 *                     and for the other plane counts).  Read-only "last_i8_priv"
 *   "i8_rt"           0 (default) | 16 | 20   count tiles (16 replicates each) per tall workgroup tile.  0 = automatic: the replicates are cut into
 *                     tall and short tile rows in ONE launch so that the machine's last round is as full as the others (list-scheduling model of
 *                     the 8 XCDs, tile costs measured per kernel: profiles/r04_i8_mix_calib.jsonl); 20 / 16 with six / seven planes: tall rows
 *                     only; 16 with six planes: the 256-replicate round-3 kernel.  Read-only "last_i8_rt" / "last_i8_short" / "last_i8_mt": tall
 *                     tile height, short rows and padded count tiles of the last launch
 *   "i8_cus"          0 (default: every CU of the device) | 8 ... the device's count: the tile-row cut is planned for that many CUs -- for launches
 *                     that share the chip with the kernels of a collective (a Gram workgroup needs a whole CU; bench.py tries a few values when it
 *                     runs more than one rank).  Results do not depend on it
 *   "i8_dma"          0 (auto) | 1 | 2   LDS-DMA form of the round-3 kernel: 1 global_load_lds_dwordx4 (64-bit base per block), 2 buffer_load_dwordx4
 *                     ... lds (per-workgroup descriptors + 32-bit offsets; auto takes it whenever an operand's walk stays below 4 GiB)
 *   "i8_ind"          1 (default) | 0   one-plane data run through the seven-plane main loop, the planes of a wave standing for seven pair groups
 *   "i8_nibbles"      1 (default) | 0 | 2   (round 5) Philox draws of data sets beyond 65,536 rows: 4-bit LDS counters (three workgroups per CU at 100,000
 *                     rows) with an exact overflow test -- sum of the counts == draws -- and an 8-bit second try for a replicate that fails it; 0: the
 *                     8-bit histogram of round 3; 2: every replicate through the second try (tests).  Same counts.  Read-only "last_i8_nibbles"
 * Solvers
 *   "solver_rows"     1 (default) | 0   bootstrap of metric models with at most 64 MVs on the int8 route: one wave per replicate with the
 *                     covariance columns in registers (solver_rows_kernel) instead of one workgroup with the covariance in LDS; models of
 *                     65 ... 128 MVs whose blocks divide into two runs of at most 64 MVs: four waves per replicate, two threads per MV
 *                     (solver_rows_split_kernel)
 *   "solver_wave"     1 (default) | 0 | 3   among those, models with at most 16 LVs (round 5; Mode-B blocks whose inverses fit 1,056 doubles) and all-Mode-A
 *                     models with at most 32: the
 *                     wave-native formulation (one wave per replicate with fixed lane roles, coalesced triangle load + LDS transpose:
 *                     solver_wave16_kernel<8> / <16>) instead of solver_rows_kernel; 3: models of at most 8 LVs on the round-3 / round-4
 *                     kernel (solver_wave_kernel<8>; A/B)
 *   "solver_quad"     1 (default) | 0   (round 5) Mode-A models of 65 ... 128 MVs and at most 16 LVs whose blocks divide into two runs of at most 64
 *                     MVs: the wave solver's fixed lane roles on four waves per replicate (solver_quad_kernel) instead of solver_rows_split_kernel
 *   "solver_threads"  64 | 128 | 256      threads per problem of the LDS solver (default 128)
 *   "nm_threads"      0 (by model width) | 64 | 128 | 256   threads per problem of the non-metric solvers
 *   "nm_live"         1 (default) | 0   (round 5) non-metric iteration on the dense stop-rule route: from the second step on, the step / compose launches cover the
 *                     problems still iterating after the previous step (the list that step's stop-rule pass built) instead of every problem of the batch
 *   "nm_counts8"      1 (default) | 0   non-metric bootstrap on the int8 route: the dense stop-rule pass takes the replicates' row
 *                     multiplicities from the int8 counts of the Gram (no second resample kernel / uint16 histograms / (row,count) lists)
 *   "nm_codes"        1 (default) | 0   all-indicator categorical models (every MV ORD / NOM), bootstrap on the int8 route: the dense
 *                     stop-rule pass adds the coefficient of the one column a row has set per MV (16 category codes per row tile and MV,
 *                     nm_conv_codes_kernel) instead of multiplying every 0/1 column through -- bit-identical partial sums, 2.3 x faster
 *                     passes on 300 indicator columns.  Read-only "last_nm_codes"
 *   "nm_mfma"         1 (default) | 0   (round 5) among those, LV blocks of at most 128 indicator columns: the stop-rule pass as an exact int8 matrix product --
 *                     indicator bytes of the rows x seven base-256 digit planes of the replicates' score maps (nmp::conv_mfma_kernel), the digits put
 *                     together again per (row, replicate); same decisions, criterion values equal to ~1e-12 relative, 2 x faster than the pass on
 *                     category codes.  Read-only "last_nm_mfma"
 *   "nm_wave"         1 (default) | 0   (round 5) all-indicator categorical models, every block Mode A, at most 64 MVs of at most 16 categories, 8 LVs,
 *                     511 indicator columns, 65,535 rows: the iteration as one wave per problem (nmw::nmw_step_kernel) instead of one workgroup
 *                     (nmg_kernel<1>); records equal to ~1e-13, equal iteration counts.  Read-only "last_nm_wave"
 *   "nm_c10"          1 (default) | 0   (round 6) among those, items of nine or ten categories (the reference's mobi / ECSI data): the ten-category instantiation
 *                     of the wave step, two waves per SIMD, instead of the sixteen-category one, which runs alone on its SIMD -- the same arithmetic, identical records
 *   "nm_vlong"        1 (default) | 0   (round 6) verification of the one-launch categorical batch: behind the fourth round of eight steps the replicates still
 *                     iterating are the few that never converge; their remaining steps (up to 72) are verified in ONE round where the slots fit, instead of
 *                     nine more rounds with a host read-back each -- the same slots, the same decisions
 *   "nm_direct16"     1 (default) | 0   (round 5) bootstrap of such models on the int8 route: the product writes the replicates' co-occurrence counts
 *                     as uint16 matrices itself (upper triangle; mirrored through LDS by nmg_kernel<4>) instead of fp64 moment matrices that a scatter
 *                     pass turns into the same integers -- bit-identical records.  Read-only "last_nm_direct16"
 *   "nm_fast_lds"     1 (default) | 0   categorical (ORD / NOM) solver: the small arrays of the iteration in LDS for the duration of a launch
 *   "nm_k16"          1 (default) | 0   all-indicator categorical models of at most 65,535 rows: uint16 copy of the count matrix for the
 *                     streaming product of every step (bit-identical steps, a quarter of the bytes)
 *   "conv_pass"       0 auto | 1 gathering stop-rule pass | 2 dense pass with block-staged coefficients (non-metric bootstrap)
 *   "conv_gy"         0 (default) .. 65535   replicate slices of the dense stop-rule pass (0: one replicate group per workgroup); matrix-product pass:
 *                     the waves it aims at, in units of 256 (0: 8,192)
 * Host-buffer bootstrap (plspm_bootstrap)
 *   "boot_chunks"     0 (default: automatic) | 1 .. 8   plspm_bootstrap() of a metric model on Philox draws runs as sub-batches: the records of
 *                     sub-batch k cross PCIe on a copy stream -- and are unpacked into the caller's buffers -- while the kernels of sub-batch
 *                     k + 1 run.  Automatic: one below 2 MiB of records, else three, sizes falling by "boot_ratio" percent (default 60: what
 *                     moving a record costs relative to computing it on the headline model), every one but the last a multiple of 64
 *                     replicates.  Records do not depend on it
 * Single fit / upload
 *   "fit_chunks"      0 (auto) .. 65535   row chunks (workgroups) of the single-fit Gram
 *   "wide_nw"         4 | 8 | 16          waves sharing one row walk in gram_wide_kernel<14>
 *   "scores_tile"     0 (by LDS footprint) | 16 | 32   rows per tile of the scores kernel
 *   "upload_direct"   0 (default) | 1   plspm_upload of more than 64 MB: 0 through the handle's pinned staging halves, filled by several host
 *                     threads; 1 the runtime's pageable copy (one staging thread: 13-52 GB/s depending on the host)
 * (Test seams and the option values of the experiments build: include/plspm_hip_test.h.)
 *
 * plspm_model_get_option reads a value back; the read-only keys "last_gram_path" (1 fp64 MFMA, 2 int8 digit planes), "last_i8_dma" (1 / 2) and "last_solver"
 * (1 LDS solver, 2 rows solver, 3 wave solver of round 3 / Mode-B blocks, 4 split rows solver, 5 quad solver, 6 / 7 / 8 wave solver for 9 ... 16 / at most 8 / 17 ... 32 LVs) tell what the last bootstrap call took.
 */
int plspm_model_set_option(plspm_model_t* m, const char* key, int32_t value);
int plspm_model_get_option(const plspm_model_t* m, const char* key, int32_t* value);

/*
 * Compile a model specification (reference Config + path matrix + Plspm kwargs, plspm/config.py:89-160,
 * plspm/plspm.py:35-67) into device descriptors.
 *   P, L           manifest / latent variable counts (1 <= L <= 64, L <= P <= 1022).  Metric handles fall back to global scratch
 *                  when a problem's workspace exceeds LDS; the non-metric / categorical / incomplete-rows solvers keep their
 *                  small workspace (about (7P + P L + 8 L^2 + L (2 k^2 + k) + 2 sum k_b^2) * 8 bytes, k = most predecessors of an
 *                  LV, k_b = Mode-B block sizes) in the 160 KiB of LDS and return PLSPM_E_LIMIT from plspm_fit / plspm_bootstrap
 *                  beyond that (P ~ 1000 with L = 6 still fits; L = 64 with P = 1022 does not)
 *   block_offset   [L+1] device-column ranges of the LV blocks
 *   path           [L*L] row-major 0/1, path[i*L+j] = 1 iff LV j -> LV i; must be strictly lower triangular
 *   mode           [L]   PLSPM_MODE_A / PLSPM_MODE_B per LV
 *   scheme         PLSPM_SCHEME_*
 *   scaled         Config(scaled=...) : divide the centred data by ONE global scalar (config.py:302-303)
 *   max_iter, tol  already clamped by the caller the way Plspm.__init__ does (plspm.py:54-56)
 *   device_id      HIP device ordinal
 * Returns NULL on failure (see plspm_last_error(NULL)).
 */
plspm_model_t* plspm_model_create(int32_t P, int32_t L, const int32_t* block_offset, const uint8_t* path, const int32_t* mode,
                                  int32_t scheme, int32_t scaled, int32_t max_iter, double tol, int32_t device_id);
void plspm_model_destroy(plspm_model_t* m);

/*
 * Non-metric data with Scale.NUM / Scale.RAW on every MV (reference _NonmetricWeights, plspm/weights.py:73-133 with
 * scale.py:22-39; `Config(default_scale=Scale.NUM)`): every MV is population-standardised (config.py:314), the stop rule is
 * sum (|Y_old| - |Y_new|)^2 over the LV scores (weights.py:120, evaluated by a streaming pass over the observations after
 * every iteration), weights are rescaled as weights.py:130-132 and there is no sign rule.  `scaled` of plspm_model_create is
 * ignored in this mode.  Scale.ORD / Scale.NOM need plspm_model_set_categorical below.
 */
int plspm_model_set_nonmetric(plspm_model_t* m, int32_t on);

/*
 * Non-metric data with Scale.ORD / Scale.NOM MVs (optimal scaling: scale.py:41-56 with util.py rank / dummy / list_to_dummy,
 * weights.py:96-118).  The handle's P device columns become AUGMENTED columns: a NUM / RAW MV keeps its one data column, an
 * ORD / NOM MV is uploaded as its 0/1 indicator columns, one per category in ascending value order (what util.dummy builds
 * per fit).  Pm logical MVs, grouped by LV in path order like the columns:
 *   mv_off   [Pm+1] aug-column range of every MV (mv_off[0] = 0, mv_off[Pm] = P; a range never crosses an LV block)
 *   mv_kind  [Pm]   PLSPM_MV_NUM (one column) / PLSPM_MV_ORD / PLSPM_MV_NOM
 * Every per-MV output (weights, loadings, crossloadings, cov, bootstrap rows) then has Pm entries instead of P; the score
 * map (scores = Xaug . score_w + score_c) stays on the aug columns.  `mean` is reported as zeros.  Implies set_nonmetric(1).
 * Call before plspm_upload.  Bootstrap replicates that miss a category re-rank the present ones (rank of a resampled
 * column, util.py rank).
 */
#define PLSPM_MV_NUM 0
#define PLSPM_MV_ORD 1
#define PLSPM_MV_NOM 2
int plspm_model_set_categorical(plspm_model_t* m, int32_t Pm, const int32_t* mv_off, const int32_t* mv_kind);

/*
 * Metric data with missing values (reference util.impute, plspm/util.py:61-68, applied by Config.treat, config.py:300 -- to
 * the full data for the fit and to every RESAMPLED data set in the bootstrap, bootstrap.py:57): n_ind of the P data columns
 * have NaNs.  The caller uploads P + n_ind columns: the data columns with every NaN replaced by that column's mean over its
 * present values, then one 0/1 column per incomplete data column marking its NaN cells.
 *   ind_of  [P]  upload column (in [P, P + n_ind)) of the indicator of data column p, or -1 when p is complete
 * The fit then sees the mean-imputed data; every bootstrap replicate is re-imputed with its own column means, on the moments
 * (solver_core.h impute_collapse) -- no per-replicate pass over the data.  A replicate in which some column lost all its
 * present cells reports PLSPM_NONFINITE.  Metric handles only; call before plspm_upload.  Limit: P + n_ind <= 1022.
 */
int plspm_model_set_missing(plspm_model_t* m, int32_t n_ind, const int32_t* ind_of);

/*
 * Two-stage estimation of higher order constructs in the bootstrap (reference Estimator.estimate, plspm/estimator.py:29-55, run
 * per resampled data set by bootstrap.py:57).  `first` is the stage-1 model (every HOC replaced by its constituent LVs,
 * estimator.py:60-74) and holds the data; `second` is the original path model whose HOC blocks have ONE column per
 * constituent LV -- that LV's stage-1 score (Scale.NUM, estimator.py:43-52) -- and takes no upload.  The stage-1 LV order
 * must be the stage-2 order with every HOC expanded in place:
 *   lv_first  [L2+1]  stage-2 LV l stands for the stage-1 LVs [lv_first[l], lv_first[l+1]): one LV with the same block
 *                     (an ordinary LV) or its constituents (a HOC)
 * After attaching, plspm_bootstrap* on `first` runs both stages per replicate -- the stage-2 moment matrix is a congruence of
 * the replicate's stage-1 Gram with the stage-1 score maps (solver_hoc.h), no second pass over the data -- and reports the
 * SECOND stage's rows (plspm_row_width(first) becomes the second stage's; effect pairs: plspm_effect_pairs(second)).
 * plspm_fit(first) still fits stage 1 alone.  Both handles: plspm_model_set_nonmetric(.., 1), same device.  Destroy in any order.
 */
int plspm_model_attach_second_stage(plspm_model_t* first, plspm_model_t* second, const int32_t* lv_first);

/*
 * Upload the filtered raw observation matrix (what Config.filter returns, config.py:247-285; no NaNs).
 *   X          host pointer, dense fp64, src_cols columns x N rows
 *   layout     0: row-major (element (i,c) at X[i*src_cols + c]);  1: column-major (X[c*N + i])
 *   col_index  [P] source column of every device column, or NULL for the identity
 * The matrix stays resident in HBM (shifted by its column means, padded, with a ones column) until the
 * next upload / destroy.
 */
int plspm_upload(plspm_model_t* m, const double* X, int64_t N, int32_t src_cols, int32_t layout, const int32_t* col_index);

/*
 * Non-metric (Scale.NUM) data with missing values (reference weights.py:88-98, mode.py:35-41, scale.py:27-30; pairwise-complete
 * loadings outer_model.py:26).  Call AFTER plspm_upload: K rows of the uploaded matrix have missing cells (their NaNs replaced by
 * any finite value before the upload); rows in which a whole LV block is missing must have been dropped (config.py:273-285).
 *   row_index  [K]    ascending row numbers
 *   present    [K*P]  1 = cell present, 0 = missing (device column order)
 *   raw_scale  1 for a Scale.RAW-only model: the MVs then keep the treated values (scale.py:38-39), which are scaled by
 *              sqrt((f-1)/f) / sqrt((N-1)/N) against Scale.NUM in a column with f < N present cells; 0 for Scale.NUM
 * The rows move into a side table and become all-zero rows of the resident matrix, so the Gram kernels see the complete rows only;
 * the solver adds the incomplete rows explicitly in every sum (solver_nmx.h), weighted by their bootstrap counts.  Mode B blocks
 * must be complete in the data set at hand (mode.py:55-56): otherwise the fit / replicate reports PLSPM_SINGULAR.
 * plspm_model_set_nonmetric(.., 1) handles only; not combinable with set_categorical / set_missing / a two-stage pair.
 */
int plspm_model_set_incomplete_rows(plspm_model_t* m, int32_t K, const int32_t* row_index, const uint8_t* present, int32_t raw_scale);

/* Number of (from,to) effect rows = ordered LV pairs joined by a directed path; from-major order
 * (reference inner_model.py:46-52).  from/to may be NULL. */
int32_t plspm_effect_pairs(const plspm_model_t* m, int32_t* from, int32_t* to);

/* Width R = 2P + L + 2*n_eff of one bootstrap row:  weights[P] | r2[L] | total[n_eff] | direct[n_eff] | loadings[P]
 * (device column order; the reference collects the same five vectors per replicate, bootstrap.py:58-64). */
int32_t plspm_row_width(const plspm_model_t* m);

/* Row pitch (in doubles) of the DEVICE result buffer of plspm_bootstrap_device: R + 2.  Columns R and R+1 of every
 * device row hold that replicate's status and iteration count as doubles, so that one collective moves everything. */
int32_t plspm_row_stride(const plspm_model_t* m);

/* Outputs of one fit; every pointer may be NULL. */
typedef struct plspm_fit_result {
    double* weights;       /* [P]    outer weights, never sign-flipped (weights.py:69)                      */
    double* loadings;      /* [P]    cor(x_p, score of own LV) after the sign rule (outer_model.py:27)      */
    double* crossloadings; /* [P*L]  cor(x_p, score_l)                    (outer_model.py:26)               */
    double* path_coef;     /* [L*L]  inner-model slopes, [i*L+j] = j -> i (inner_model.py:70)               */
    double* r2;            /* [L]    R^2 per LV, 0 for exogenous          (inner_model.py:71-72)            */
    double* lv_cov;        /* [L*L]  population covariance of the LV scores                                */
    double* total;         /* [n_eff] total effects                       (inner_model.py:45-52)            */
    double* direct;        /* [n_eff]                                                                       */
    double* indirect;      /* [n_eff]                                                                       */
    double* scores;        /* [N*L]  LV scores, row-major, sign-corrected (weights.py:60-68)               */
    double* cov;           /* [P*P]  population covariance of the treated data (device column order)       */
    double* mean;          /* [P]    raw column means                                                      */
    int8_t* sign;          /* [L]    the sign vector of weights.py:62-64                                   */
    int32_t* iterations;   /* [1]    value of the reference's iteration counter when the loop stopped      */
    int32_t* status;       /* [1]    PLSPM_OK / PLSPM_NOT_CONVERGED / ...                                  */
} plspm_fit_result_t;

/* One estimate on the uploaded data (Estimator.estimate, estimator.py:29-55, without higher-order constructs). */
int plspm_fit(plspm_model_t* m, const plspm_fit_result_t* out);

/*
 * B bootstrap replicates (the loop of BootstrapProcess.run, bootstrap.py:54-66).
 *   rep_offset  global id of the first replicate; replicate r draws its N row indices from a counter-based
 *               Philox4x32-10 stream keyed by (seed, r), so any sharding reproduces the same rows
 *   idx         NULL -> on-device RNG;  else [B*N] int32 explicit resample indices in [0, N) (parity testing:
 *               the reference draws np.random.randint(N, size=N), bootstrap.py:56)
 *   out         [B*R] host buffer, R = plspm_row_width();  status/iters [B] (may be NULL)
 * Replicates whose status != PLSPM_OK are the ones the reference silently drops (bootstrap.py:65-66).
 */
int plspm_bootstrap(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* idx, double* out,
                    int32_t* status, int32_t* iters);

/* Same, leaving the results in device memory owned by the handle (valid until the next call on it):
 * *d_out -> [B * plspm_row_stride()] fp64, *d_status / *d_iters -> [B] int32.  Work is enqueued on the handle's stream;
 * plspm_sync() waits for it.  Lets a caller hand the buffers to RCCL without a host round trip. */
int plspm_bootstrap_device(plspm_model_t* m, int64_t B, uint64_t seed, int64_t rep_offset, const int32_t* d_idx, void** d_out,
                           void** d_status, void** d_iters);
int plspm_sync(plspm_model_t* m);
/* Enqueue now what the first bootstrap call on the uploaded data would have to build before its first replicate (the int8 digit planes
 * of the pair products, ~0.1 ms at 10k x 60): a host that knows a bootstrap follows the fit (Plspm(bootstrap=True): plspm.py:78-82 calls
 * Bootstrap right behind the estimate) calls this after plspm_upload.  Enqueue only -- the call never waits for the device: with the
 * automatic plane count it starts the column statistics and their copy to pinned memory, the plspm_fit that follows finds them on the host
 * when its own synchronisation returns and cuts the planes in its tail, beside the caller's unpacking of the fit.  Optional.  Anything that
 * rewrites the resident rows afterwards (plspm_upload, plspm_model_set_incomplete_rows) discards what was prepared. */
int plspm_bootstrap_prepare(plspm_model_t* m);
/* Host copy of replicates [first, first + count) of the LAST plspm_bootstrap(_device) call on this handle, whose records are still
 * in HBM: out [count*R], status / iters [count] (each may be NULL).  Lets a caller keep the rows on the device (summaries:
 * plspm_bootstrap_summary) and pay for the PCIe transfer only when individual replicates are asked for.  PLSPM_E_STATE when the
 * handle holds no records (no bootstrap yet, or a later upload replaced the data). */
int plspm_bootstrap_fetch(plspm_model_t* m, int64_t first, int64_t count, double* out, int32_t* status, int32_t* iters);
/* The handle's HIP stream (a hipStream_t), so that a caller can order its own work -- e.g. an RCCL collective -- behind the
 * enqueued kernels with stream/event semantics instead of a host synchronisation. */
void* plspm_stream(plspm_model_t* m);

/* Replace the handle's records by B host records [B * plspm_row_stride()] in the device layout [row | status | iterations] -- for a
 * caller that merged the shards of several processes with a transport of its own (MPI, gloo, files) and wants the device summary
 * (plspm_bootstrap_summary with d_rows = NULL) of the merged set. */
int plspm_bootstrap_store(plspm_model_t* m, const double* records, int64_t B);

/*
 * Summary statistics of a bootstrap on the device (reference _create_summary, plspm/bootstrap.py:24-32): for every result
 * column c: summary[c*6 + 0..5] = original, mean, std.error (ddof 1), perc.025, perc.975 (linear interpolation), t stat. --
 * over the replicates whose status is PLSPM_OK.
 *   d_rows   NULL: the rows of the last plspm_bootstrap(_device) call on this handle (stride = plspm_row_stride(); B must be
 *            that call's B, else PLSPM_E_ARG; PLSPM_E_STATE when the handle holds no records);
 *            else a device buffer [B * stride] in the same record layout (e.g. the all-gathered records of all GPUs);
 *            records whose status column is not exactly 0 (failed replicates, NaN-marked padding of ragged shards) are skipped
 *   B        1 <= B <= 2^30
 *   original [R] host: the full-sample estimates;  summary [R*6] host;  n_used: number of OK replicates (may be NULL)
 */
int plspm_bootstrap_summary(plspm_model_t* m, const void* d_rows, int64_t B, int32_t stride, const double* original, double* summary,
                            int64_t* n_used);

/*
 * ---- Multi-GPU: replicate shards + ONE RCCL all-gather --------------------------------------------------------------------------
 * Reference: Bootstrap.__init__ forks `processes` workers, each running iterations / processes replicates, and merges their
 * frames through a Queue (plspm/bootstrap.py:89-111; `processes` kwarg plspm/plspm.py:35-37,60-61).  Here a GROUP of handles --
 * the same compiled model with the same data uploaded, one handle per MI355X -- runs the replicate ids
 * [rep_offset, rep_offset + B) of one logical stream: rank r of nranks takes the contiguous shard plspm_group_shard(B, r), every
 * rank's [row | status | iterations] records are written into its send buffer by the solver kernel, and ONE ncclAllGather over
 * xGMI (librccl.so.1, loaded on first use) leaves all records on every rank.  Results are bit-identical for every nranks
 * (Philox stream keyed by (seed, replicate id)).  Nothing here needs PyTorch.
 *
 * Two objects: a COMMUNICATOR (plspm_comm_t: the RCCL communicators of this process's ranks -- expensive, of the order of a
 * second, create once per process) and a GROUP (plspm_group_t: binds one uploaded handle per local rank to the communicator --
 * cheap: a gather stream, four events and the record buffers per handle).  Two ways to form a communicator:
 *   single process   n_local == nranks ranks on DISTINCT devices (ncclCommInitAll); unique_id = NULL, first_rank = 0.
 *                    Ranks that share a device (testing on a 1-GPU box; RCCL refuses duplicate devices) exchange their records
 *                    with device-to-device copies (one copy launch when every handle shares a device) instead of RCCL -- same buffers, same ordering, same results.
 *   one process per GPU (torchrun / mpirun)   n_local == 1, first_rank = this process's rank, unique_id = the 128 bytes that
 *                    rank 0 obtained from plspm_rccl_unique_id() and handed to every rank (ncclCommInitRank: collective call).
 * A communicator serves one group at a time; a handle belongs to at most one group at a time and must outlive it.  Group calls
 * are not re-entrant.
 *
 * Stream semantics: plspm_group_bootstrap only ENQUEUES (shard kernels on each handle's stream, the all-gather on a second
 * stream per handle behind an event); records are double-buffered so that the collective of call k overlaps the kernels of call
 * k+1.  plspm_group_summary / _rows / _sync wait for the last call's collective.
 */
typedef struct plspm_comm plspm_comm_t;
typedef struct plspm_group plspm_group_t;
#define PLSPM_UNIQUE_ID_BYTES 128
int plspm_rccl_unique_id(uint8_t* id /* [PLSPM_UNIQUE_ID_BYTES] */);
/* device_ids [n_local]: HIP device of every local rank; local rank i is global rank first_rank + i.  NULL on failure. */
plspm_comm_t* plspm_comm_create(const int32_t* device_ids, int32_t n_local, int32_t nranks, int32_t first_rank, const uint8_t* unique_id);
void plspm_comm_destroy(plspm_comm_t* c);                  /* a group still bound to it is released (streams, buffers, its hold on the handles): later
                                                              calls on that group return PLSPM_E_STATE; its owner still calls plspm_group_destroy */
/* The same with the transport and RCCL's footprint chosen (round 5):
 *   transport     PLSPM_TRANSPORT_AUTO   RCCL between distinct devices, one copy launch among ranks that share a device (plspm_comm_create)
 *                 PLSPM_TRANSPORT_RCCL   RCCL or failure
 *                 PLSPM_TRANSPORT_COPY   single-process jobs only (n_local == nranks): no RCCL -- every rank pulls its peers' shards with
 *                                        device-to-device copies on peer-mapped buffers (hipDeviceEnablePeerAccess): the SDMA engines move the
 *                                        records over xGMI and no kernel of the exchange takes a CU from the next step's Gram
 *   max_channels  0: RCCL's default; n > 0: the communicator's collectives run at most n workgroups (ncclConfig_t.maxCTAs) -- a Gram workgroup
 *                 needs a whole CU, so every CU an RCCL channel occupies is missing from a launch cut for all of them; the one all-gather of
 *                 44 MB per 0.48 ms step (eight ranks) does not need RCCL's default channel count.  The NCCL_* environment variables of the
 *                 caller are not touched.
 * plspm_comm_split: a second communicator over the same ranks with its own max_channels, split off an RCCL communicator (ncclCommSplit --
 * collective over the parent, every rank calls it; no new unique id): lets a job time RCCL's default against a capped one (bench.py). */
enum { PLSPM_TRANSPORT_AUTO = 0, PLSPM_TRANSPORT_RCCL = 1, PLSPM_TRANSPORT_COPY = 2 };
plspm_comm_t* plspm_comm_create_ex(const int32_t* device_ids, int32_t n_local, int32_t nranks, int32_t first_rank, const uint8_t* unique_id, int32_t transport,
                                   int32_t max_channels);
plspm_comm_t* plspm_comm_split(plspm_comm_t* parent, int32_t max_channels);
int32_t plspm_comm_size(const plspm_comm_t* c);            /* nranks */
int32_t plspm_comm_uses_rccl(const plspm_comm_t* c);       /* 1: records travel through RCCL; 0: device-to-device copies */
int32_t plspm_comm_transport(const plspm_comm_t* c);       /* PLSPM_TRANSPORT_RCCL / PLSPM_TRANSPORT_COPY / 3: ranks share a device (one copy launch) */
int32_t plspm_comm_max_channels(const plspm_comm_t* c);    /* the cap the communicator was created with (0: RCCL's default) */
/* models [n_local of the communicator]: handle i lives on the communicator's device i, data uploaded.  NULL on failure. */
plspm_group_t* plspm_group_create(plspm_comm_t* c, plspm_model_t* const* models);
void plspm_group_destroy(plspm_group_t* g);
/* Text of the last error on this group (or of the last failed plspm_comm_create / plspm_group_create / plspm_rccl_unique_id
 * when NULL). */
const char* plspm_group_last_error(const plspm_group_t* g);
int32_t plspm_group_size(const plspm_group_t* g);          /* nranks */
/* Shard of rank `rank`: balanced contiguous ranges, the first B % nranks ranks hold one replicate more. */
int plspm_group_shard(const plspm_group_t* g, int64_t B, int32_t rank, int64_t* first, int64_t* count);
/* Enqueue B replicates split over the group + the all-gather.  Returns without host synchronisation for metric models.
 * ONE call hides its merge (round 5): the call runs as up to three SUB-BATCHES -- consecutive ranges of the replicate ids, each sharded over the
 * ranks like a call of its own (plspm_group_plan) -- and the all-gather of sub-batch k runs on the gather streams beside the shard kernels of
 * sub-batch k + 1; only the last, smallest gather is exposed.  Results do not depend on the cut (Philox stream keyed by (seed, replicate id)).
 * Options (plspm_group_set_option): "chunks" 0 (default: automatic -- one sub-batch below 2 MiB of records per rank, on a one-rank group and among
 * ranks that share a device, else up to three, sizes falling by "chunk_ratio" percent) | 1 .. 8 sub-batches -- a caller that issues calls back to back (bench.py's
 * step loop) sets 1: the gather of call k then overlaps the kernels of call k + 1 anyway; "chunk_ratio" 10 .. 100 (default 50); "chunk_align"
 * 0 (default: every sub-batch but the last fills whole ROUNDS of the device per rank -- a part that ends inside a round of Gram tiles pays for
 * the whole round; models whose round holds more replicates than the call has stay in one piece) | n: multiples of n replicates per rank;
 * "gather_root" 0 (default: every rank receives every shard -- ncclAllGather) | 1: only rank 0 -- the rank whose handle summarises, as only the
 * reference's parent process merges (bootstrap.py:96-111) -- receives them: grouped ncclSend / ncclRecv to rank 0 (or its copy engines alone pulling, in a
 * one-process job on the copy transport): 1 / nranks of the bytes on the links, nothing arriving at the other ranks.  plspm_group_summary then computes the
 * table on rank 0 and hands it to every rank in one small ncclBroadcast (every rank calls it, as before); plspm_group_records / _rows / _adopt report
 * PLSPM_E_STATE on the ranks that hold no records.  The automatic sub-batch alignment is agreed across the ranks (max) in the first call that needs it. */
int plspm_group_bootstrap(plspm_group_t* g, int64_t B, uint64_t seed, int64_t rep_offset);
int plspm_group_set_option(plspm_group_t* g, const char* key, int32_t value);
/* The sub-batches a call of B replicates is cut into: *n_sub (<= 8) ranges [sub_first[k], sub_first[k] + sub_count[k]) (arrays of 8), in
 * replicate-id order; rank r computes plspm_group_shard(sub_count[k], r) of each, offset by sub_first[k]. */
int plspm_group_plan(const plspm_group_t* g, int64_t B, int32_t* n_sub, int64_t* sub_first, int64_t* sub_count);
/* Wait for everything enqueued on the group's streams (every local handle). */
int plspm_group_sync(plspm_group_t* g);
/* Gathered records of the last plspm_group_bootstrap on local handle `local`: *d_records -> [*n_records * *stride] fp64 in HBM.  One sub-batch
 * (plspm_group_plan): n_records = nranks * ceil(B / nranks); rank r's shard starts at record r * ceil(B / nranks); the unused tail records of
 * a ragged split carry NaN in the status column.  Several sub-batches: one such block per sub-batch, one after the other (the replicates stay
 * in id order; n_records = nranks * sum_k ceil(sub_count[k] / nranks)).  Valid until the next-but-one plspm_group_bootstrap. */
int plspm_group_records(plspm_group_t* g, int32_t local, void** d_records, int64_t* n_records, int32_t* stride);
/* _create_summary (bootstrap.py:24-32) of the last plspm_group_bootstrap, on local handle 0's copy of the gathered records:
 * same outputs as plspm_bootstrap_summary. */
int plspm_group_summary(plspm_group_t* g, const double* original, double* summary, int64_t* n_used);
/* Host copy of the last plspm_group_bootstrap's replicates in replicate-id order: out [B*R], status / iters [B] (may be NULL). */
int plspm_group_rows(plspm_group_t* g, double* out, int32_t* status, int32_t* iters);
/* Hand the last plspm_group_bootstrap's records to local handle 0 (device-to-device, replicate-id order): the handle then answers
 * plspm_bootstrap_fetch / plspm_bootstrap_summary(d_rows = NULL) for them like for its own plspm_bootstrap_device, and the group --
 * and with it the communicator's one group slot -- can be destroyed (the merge of Bootstrap.__init__, bootstrap.py:96-111, keeps
 * nothing of the workers alive either). */
int plspm_group_adopt(plspm_group_t* g);
/* Collective helpers for a caller's timing protocol (bench.py): barrier = all-reduce of one word on the gather stream + host
 * wait; max = all-reduce(max) of one double over the ranks. */
int plspm_group_barrier(plspm_group_t* g);
int plspm_group_max(plspm_group_t* g, double* value);
/* Host time of the last plspm_group_bootstrap call on this process, in milliseconds: enqueue of the shard kernels of all local handles
 * (resident threads, one per handle) and enqueue of the exchange (event waits + ONE ncclAllGather per handle inside a group call, or the
 * device copies of ranks that share a GPU).  Diagnostics for tools/group_enqueue.py. */
int plspm_group_enqueue_times(const plspm_group_t* g, double* shards_ms, double* exchange_ms);

/*
 * ---- Operator seam ------------------------------------------------------------------------------------------------------------
 * The reference's strategy objects, callable on their own; each call uploads its operands, forms their moment
 * matrix with the MFMA Gram kernel and finishes in one small kernel (csrc/solver_ops.h).  Errors: plspm_last_error(NULL).
 *   plspm_op_inner_weights  Scheme.X.value.calculate(path, y)            plspm/scheme.py:27-28, 36-37, 45-54
 *       scheme PLSPM_SCHEME_*; path [L*L] row-major 0/1 strictly lower triangular; y [N*L] row-major LV scores; E [L*L] row-major:
 *       centroid sign(corrcoef(y) * (path + path')), factorial cov1(y) * (path + path'), path: OLS coefficients (no intercept,
 *       minimum norm) of every LV on its predecessors / correlations with its successors, column i = LV i
 *   plspm_op_outer_weights  Mode.X.value.outer_weights_metric(data, Z, lv, mvs)   plspm/mode.py:28-29, 50-52
 *       Xk [N*k] row-major: the block's (treated) MVs; z [N]: the LV's inner estimate; w [k]:
 *       Mode A  X_k' z / N;  Mode B  least squares of z on X_k (minimum norm when X_k is rank deficient, as scipy.linalg.lstsq)
 *   plspm_op_outer_weights_nonmetric  Mode.X.value.outer_weights_nonmetric(mv_grouped_by_lv, mv_grouped_by_lv_missing, Z, lv, correction)
 *       plspm/mode.py:31-42, 54-61.  Xk [N*k] row-major: the block's quantified MVs (NaN where missing); present [N*k] 0/1 or NULL when
 *       the block has no missing cell (the reference's mv_grouped_by_lv_missing[lv]); z [N]; returns w [k] and the LV's new scores Y [N]:
 *       Mode A  w = X'z / sum z^2 (NaN-aware ratios with a mask), Mode B least squares of z on X (no mask allowed); Y = X w (row-wise
 *       normalised with a mask), then (Y - mean) / std1 * correction (util.py:43-53)
 */
int plspm_op_inner_weights(int32_t device_id, int32_t scheme, int32_t L, const uint8_t* path, const double* y, int64_t N, double* E);
int plspm_op_outer_weights(int32_t device_id, int32_t mode, const double* Xk, const double* z, int64_t N, int32_t k, double* w);
int plspm_op_outer_weights_nonmetric(int32_t device_id, int32_t mode, const double* Xk, const uint8_t* present, const double* z, int64_t N, int32_t k,
                                     double correction, double* w, double* Y);

/* Kernel timing with HIP events on the handle's own stream (for the roofline figures in bench.py).
 * kernel ids: 0 resample/compact, 1 gram (MFMA), 2 solver, 3 scores, 4 upload/pack, 5 gram reduce. */
enum { PLSPM_K_RESAMPLE = 0, PLSPM_K_GRAM = 1, PLSPM_K_SOLVER = 2, PLSPM_K_SCORES = 3, PLSPM_K_PACK = 4, PLSPM_K_REDUCE = 5, PLSPM_K_COUNT = 6 };
/* on: 0 off, 1 every kernel, 2 + id only kernel `id` (an event pair costs dispatch latency on both sides: bracketing one kernel of a
 * step perturbs the step less than bracketing all of them). */
int plspm_profile_enable(plspm_model_t* m, int32_t on);
int plspm_profile_read(plspm_model_t* m, int32_t kernel_id, double* total_ms, int64_t* launches);
int plspm_profile_reset(plspm_model_t* m);

#ifdef __cplusplus
}
#endif
#endif /* PLSPM_HIP_H */
