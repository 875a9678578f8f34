// PLS-PM iterative weight solver on second moments -- the LDS-resident per-problem stage.
//
// One cooperating thread group (a HIP workgroup; std::threads in the CPU emulation build used by
// tests/test_solver_hostemu.py) solves ONE problem: it turns the augmented raw scatter matrix
//     M = sum_i c_i [x'_i, 1] [x'_i, 1]^T          (x' = x - shift, shift = full-sample column means)
// produced by the Gram kernels into everything the reference computes per fit / per bootstrap
// replicate.  Reference call sites restated here (paths relative to the reference repo):
//   Config.treat            plspm/config.py:299-305  + util.treat plspm/util.py:33-39   -> moments_to_cov
//   _MetricWeights.__init__ plspm/weights.py:28-39                                        -> init_weights
//   _MetricWeights.iterate  plspm/weights.py:41-54                                        -> iterate
//   Scheme.*.calculate      plspm/scheme.py:27-28, 36-37, 45-54                           -> inner_weights
//   Mode.*.outer_weights_metric plspm/mode.py:28-29, 50-52                                -> outer step
//   WeightsCalculatorFactory.calculate plspm/weights.py:172-187                           -> solve loop
//   _MetricWeights.calculate plspm/weights.py:56-70                                       -> finalize
//   InnerModel.__init__ / _effects plspm/inner_model.py:58-75, 33-53                      -> inner_model, effects
//   bootstrap row           plspm/bootstrap.py:58-64                                      -> emit_row
// The algebra: with S = X^T X / N of the treated data,
//   var1(Xw) = w'Sw N/(N-1);  cov0(Yhat) = Wn' S Wn;  X'Z/N = S Wn E;  Mode B w = S_bb^-1 (S Wn E)_b;
//   PATH OLS beta = G_ff^-1 G_fi;  cor(x_p, score_l) = (S W)_pl / sqrt(S_pp).
//
// Execution model: every thread of the group runs solve_problem(); `ex.par(n, f)` distributes
// indices over the group and ends with a group barrier, `ex.one(f)` runs f on thread 0 and
// barriers, `ex.par2(n0, n1, f)` is par over the n0 x n1 index grid (first index fastest, no integer division),
// `ex.sum(n, f)` / `ex.any(n, f)` are group-wide reductions whose result every thread
// receives (wave shuffles + LDS on the GPU).  Every other value that crosses threads lives in the
// workspace (never in a local).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PLSPM_HD __host__ __device__ __forceinline__
#else
#define PLSPM_HD inline
#endif

namespace plspm {

enum { SCHEME_CENTROID = 0, SCHEME_FACTORIAL = 1, SCHEME_PATH = 2 };
enum { MODE_A = 0, MODE_B = 1 };
enum { ST_OK = 0, ST_NOT_CONVERGED = 1, ST_SINGULAR = 2, ST_NONFINITE = 3 };

// Tile-packed layout of the symmetric (PA x PA) scatter matrix written by the MFMA Gram kernels:
// columns are split into T = PA/16 tiles, tile t holds columns 32*(t/2) + 2*i + (t&1), i = 0..15;
// only tiles (t,u) with t <= u are stored, 256 doubles each, in v_mfma_f64_16x16x4_f64 C/D register
// order: element (row, col) of a tile sits at ((reg = row/4) * 64 + lane), lane = (row%4)*16 + col.
// An ODD tile count (16-column granularity of the padded width: P + 1 = 201 columns take 13 tiles, not 14) leaves the last tile
// without a partner: it holds the 16 consecutive columns 16(T-1) .. 16T-1.
PLSPM_HD int packed_tile_of(int T, int p) { return ((T & 1) && p >= 16 * (T - 1)) ? T - 1 : 2 * (p >> 5) + (p & 1); }
PLSPM_HD int packed_pos_of(int T, int p) { return ((T & 1) && p >= 16 * (T - 1)) ? p - 16 * (T - 1) : (p & 31) >> 1; }
PLSPM_HD int packed_col_of(int T, int tile, int pos) { return ((T & 1) && tile == T - 1) ? 16 * (T - 1) + pos : 32 * (tile >> 1) + 2 * pos + (tile & 1); }
PLSPM_HD long packed_index(int T, int p, int q) {
    int tp = packed_tile_of(T, p), tq = packed_tile_of(T, q), ip = packed_pos_of(T, p), iq = packed_pos_of(T, q);
    int t, u, r, c;
    if (tp < tq || (tp == tq && ip <= iq)) { t = tp; u = tq; r = ip; c = iq; } else { t = tq; u = tp; r = iq; c = ip; }
    long tile = (long)t * T - (long)t * (t - 1) / 2 + (u - t);
    return (tile * 4 + (r >> 2)) * 64 + (r & 3) * 16 + c;
}
PLSPM_HD long packed_size(int T) { return (long)T * (T + 1) / 2 * 256; }

struct ModelDesc {
    int P, L, PA, T;            // MVs, LVs, padded augmented width (multiple of 32), tiles per side
    int scheme, scaled, max_iter, kmax, n_eff, n_chol;
    double tol;
    const int* boff;            // [L+1] first device column of every block (device columns are in path-LV order)
    const int* lvof;            // [P]   LV of every device column
    const unsigned char* C;     // [L*L] C[i*L+j] = 1 iff LV j -> LV i
    const int* mode;            // [L]
    const int* chol_off;        // [L]   offset of the Mode-B Cholesky factor of block l inside ws.chol (-1 for Mode A)
    const int* pred_off;        // [L+1] CSR of predecessors: pred_idx[pred_off[i] .. pred_off[i+1]) = { j : C[i,j] = 1 }, ascending
    const int* pred_idx;        // [n_edges]
    const int* succ_off;        // [L+1] CSR of successors:   succ_idx[succ_off[i] .. ) = { s : C[s,i] = 1 }, ascending
    const int* succ_idx;        // [n_edges]
    int n_edges;
    const int* eff_from;        // [n_eff]
    const int* eff_to;          // [n_eff]
    const double* shift;        // [P]   column shift applied at upload
    const unsigned short* tile_tu;   // [T(T+1)/2] (t | u << 8) of every stored tile, or null (decode on the fly)
};

struct Workspace {
    double* S;                  // [(P+1)*PS] treated population covariance S[q*PS+p], p,q < P; row/column P: raw column sums and n
    int PS;
    double *w, *wn, *cv, *dv, *sd, *mu, *cs;    // [P] each (cs: factor from an uploaded column to its treated value)
    double* V;                  // [P*L]
    double *Q, *G, *E, *Bm, *Pw, *Pw2, *Ind, *Cs;   // [L*L] each
    double *a, *wf, *sgn, *r2;  // [L] each
    double* scr;                // [L*regression_scratch_doubles(kmax)]
    double* chol;               // [n_chol] = sum over the Mode-B blocks of chol_block_doubles(k)
    double* scal;               // [8]  1 n, 2 1/(n g^2) or 1/n, 3 status
    double* red;                // [16] scratch of the group reductions (Ex::sum / Ex::any), one slot per wave
};

// per-LV scratch of the small regressions: the k x k normal matrix + k right-hand sides, and a second k x k matrix for the
// eigenvectors of the minimum-norm fallback (jacobi_pinv)
PLSPM_HD long regression_scratch_doubles(int kmax) { return 2L * kmax * kmax + kmax; }
// cached factor of one Mode-B block of k MVs: the k x k factor (Cholesky, or the pseudo-inverse of a rank-deficient block) + k x k scratch
PLSPM_HD long chol_block_doubles(int k) { return 2L * k * k; }
PLSPM_HD long workspace_small_doubles(int P, int L, int kmax, int n_chol) {
    return 7L * P + (long)P * L + 8L * L * L + 4L * L + (long)L * regression_scratch_doubles(kmax) + n_chol + 8 + 16;
}
PLSPM_HD int cov_ld(int P) { return (P + 1) | 1; }      // S carries the ones row/column P as well (column sums, n)
PLSPM_HD long cov_doubles(int P) { return (long)(P + 1) * cov_ld(P); }

PLSPM_HD void carve_small(Workspace& ws, double* base, int P, int L, int kmax, int n_chol) {
    double* p = base;
    ws.w = p; p += P; ws.wn = p; p += P; ws.cv = p; p += P; ws.dv = p; p += P; ws.sd = p; p += P; ws.mu = p; p += P; ws.cs = p; p += P;
    ws.V = p; p += (long)P * L;
    ws.Q = p; p += L * L; ws.G = p; p += L * L; ws.E = p; p += L * L; ws.Bm = p; p += L * L;
    ws.Pw = p; p += L * L; ws.Pw2 = p; p += L * L; ws.Ind = p; p += L * L; ws.Cs = p; p += L * L;
    ws.a = p; p += L; ws.wf = p; p += L; ws.sgn = p; p += L; ws.r2 = p; p += L;
    ws.scr = p; p += (long)L * regression_scratch_doubles(kmax);
    ws.chol = p; p += n_chol;
    ws.scal = p; p += 8;
    ws.red = p;
}

// A pivot below PIVOT_RTOL * (its diagonal entry) means the column is a linear combination of the previous ones to working
// precision: the matrix is treated as rank deficient and the caller switches to the minimum-norm solution, which is what the
// reference's solvers return there (scipy.linalg.lstsq / gelsd in mode.py:51,58; statsmodels OLS.fit() = pinv in scheme.py:50,
// inner_model.py:69).  Eigenvalues below EIG_RTOL * (largest eigenvalue) count as zero in that solution.
#define PLSPM_PIVOT_RTOL 1e-13
#define PLSPM_EIG_RTOL 1e-12

// Scale.NUM / RAW: population std of a column from its raw moments (a = mean of x^2, m = mean of x).  A column that is CONSTANT in this replicate is 0 / 0 in the reference's
// standardisation (config.py:314): NaN scores, "could not converge" / a failed regression, and the bootstrap's bare except drops the replicate (bootstrap.py:65-66).  On the
// moments its variance is rounding residue of either sign (a finite "standardised" column of noise with a weight of 1e-9, or NaN, by the luck of the rounding): NaN here
// whatever the residue, with treated_sd's threshold (the int8 route's digit-plane residue is 1e-10 of the second moment).
PLSPM_HD double nm_column_sd(double a, double m) {
    const double var = a - m * m;
    return sqrt((var > 1e-9 * a) ? var : -1.0);
}

// Standard deviation of a treated (centred, scaled) metric column from its raw second moment about the upload's shift `dpp`, its sum `mup`, 1 / n and the scale factor.
// A column that is CONSTANT in this data set -- in a bootstrap: an item whose resample holds one value only (a rare binary indicator, a small sample) -- is centred to exact
// zeros by the reference: its Mode-A weight is 0, pandas' corrwith gives NaN for its cross-loadings and `(crossloadings * odm).sum(axis=1)` (plspm.py / bootstrap.py:62) skips
// that NaN: LOADING 0, and the estimate -- or the replicate -- counts.  On second moments that variance is `dpp - mup^2 / n` of two equal numbers: rounding noise of either
// sign -- on the int8 digit-plane route up to N 2^-(8S-1) max|z_pp| of it, i.e. ~3e-17 (M / c)^2 of the second moment at seven planes for a column that sits at c in this
// replicate and reaches M elsewhere (a rare indicator: M / c = n / k).  Below 1e-9 of the second moment it is called zero here: a real column's variance about the upload's
// shift is ALL of its second moment up to (replicate mean - shift)^2, which resampling keeps within a few variances / n -- nothing real comes within nine orders of magnitude;
// an indicator rarer than ~1 in 6,000 can fall short of the threshold on the int8 route and is then standardised like any column (as before this rule).  The outputs of a zero
// follow the reference (loading 0, cross-loadings NaN), the status stays PLSPM_OK.  NaN / inf data keep their NaN / inf: PLSPM_NONFINITE as before.
PLSPM_HD double treated_sd(double dpp, double mup, double inv_n, double fac) {
    const double var = dpp - (mup * mup) * inv_n;
    if (var > 1e-9 * dpp) return sqrt(var * fac);
    return (var == var && dpp == dpp && fac == fac) ? 0.0 : var + dpp + fac;       // (NaN stays NaN)
}

// In-place Cholesky A = R^T R of a k x k SPD matrix (row-major, ld k, upper part used/overwritten).
// Returns false at the first pivot that is not safely positive (rank deficient to working precision).
PLSPM_HD bool chol_factor(double* A, int k) {
    for (int j = 0; j < k; ++j) {
        const double ajj = A[j * k + j];
        double d = ajj;
        for (int r = 0; r < j; ++r) d -= A[r * k + j] * A[r * k + j];
        if (!(d > PLSPM_PIVOT_RTOL * ajj)) return false;
        d = sqrt(d);
        A[j * k + j] = d;
        for (int c = j + 1; c < k; ++c) {
            double s = A[j * k + c];
            for (int r = 0; r < j; ++r) s -= A[r * k + j] * A[r * k + c];
            A[j * k + c] = s / d;
        }
    }
    return true;
}
// Solve R^T R x = b in place (b -> x).
PLSPM_HD void chol_solve(const double* R, int k, double* b) {
    for (int i = 0; i < k; ++i) {
        double s = b[i];
        for (int r = 0; r < i; ++r) s -= R[r * k + i] * b[r];
        b[i] = s / R[i * k + i];
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = b[i];
        for (int c = i + 1; c < k; ++c) s -= R[i * k + c] * b[c];
        b[i] = s / R[i * k + i];
    }
}

// Moore-Penrose inverse of a symmetric positive SEMI-definite k x k matrix, in place: cyclic Jacobi eigendecomposition
// A = V diag(lambda) V^T (V: k x k scratch), then A <- sum_{lambda_i > EIG_RTOL * lambda_max} v_i v_i^T / lambda_i.  A x = b then has
// the minimum-norm least-squares solution x = A^+ b -- for the normal equations X'X w = X'z exactly the vector LAPACK gelsd /
// numpy pinv return for min |X w - z| with a rank-deficient X.  One thread; the rare path (k is a block / predecessor count).
// Returns false when the sweeps do not converge (never observed; the caller flags ST_SINGULAR).
PLSPM_HD bool jacobi_pinv(double* A, int k, double* V) {
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) V[i * k + j] = (i == j) ? 1.0 : 0.0;
    bool converged = false;
    for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < k; ++p) { diag += A[p * k + p] * A[p * k + p]; for (int q = p + 1; q < k; ++q) off += A[p * k + q] * A[p * k + q]; }
        if (!(off > 1e-36 * diag)) { converged = true; break; }
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = A[p * k + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * k + q] - A[p * k + p]) / (2.0 * apq);
                const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int r = 0; r < k; ++r) {                       // A <- A J  (columns p, q)
                    const double arp = A[r * k + p], arq = A[r * k + q];
                    A[r * k + p] = c * arp - sn * arq;
                    A[r * k + q] = sn * arp + c * arq;
                }
                for (int r = 0; r < k; ++r) {                       // A <- J^T A (rows p, q)
                    const double apr = A[p * k + r], aqr = A[q * k + r];
                    A[p * k + r] = c * apr - sn * aqr;
                    A[q * k + r] = sn * apr + c * aqr;
                }
                for (int r = 0; r < k; ++r) {                       // V <- V J
                    const double vrp = V[r * k + p], vrq = V[r * k + q];
                    V[r * k + p] = c * vrp - sn * vrq;
                    V[r * k + q] = sn * vrp + c * vrq;
                }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < k; ++i) lmax = (A[i * k + i] > lmax) ? A[i * k + i] : lmax;
    for (int i = 0; i < k; ++i) {                                    // V[:, i] <- v_i / sqrt(lambda_i), or 0 for a null direction
        const double l = A[i * k + i];
        const double f = (l > PLSPM_EIG_RTOL * lmax) ? 1.0 / sqrt(l) : 0.0;
        for (int r = 0; r < k; ++r) V[r * k + i] *= f;
    }
    for (int r = 0; r < k; ++r)                                      // A^+ = (V D^-1/2)(V D^-1/2)^T, both triangles
        for (int c = r; c < k; ++c) {
            double s2 = 0.0;
            for (int i = 0; i < k; ++i) s2 += V[r * k + i] * V[c * k + i];
            A[r * k + c] = s2;
            A[c * k + r] = s2;
        }
    return converged;
}

// Cached factor of a Mode-B block (k x k at F, k x k scratch behind it): `fill(F)` writes the full symmetric matrix.  Afterwards
// F holds either the Cholesky factor (upper triangle) or, for a rank-deficient block, the pseudo-inverse (upper triangle);
// F[k] -- an entry of the unused strictly lower triangle -- says which (+1 / -1; k == 1 needs no tag).  psd_solve applies it.
template <class Fill>
PLSPM_HD bool psd_factor(double* F, int k, Fill fill) {
    fill(F);
    if (chol_factor(F, k)) { if (k > 1) F[k] = 1.0; return true; }
    if (k == 1) return false;                                        // a zero-variance column: nothing to solve for
    fill(F);
    const bool ok = jacobi_pinv(F, k, F + (long)k * k);
    F[k] = -1.0;
    return ok;
}
PLSPM_HD void psd_solve(double* F, int k, double* b) {
    if (k == 1 || F[k] > 0.0) { chol_solve(F, k, b); return; }
    double* tmp = F + (long)k * k;                                   // the scratch half is free once the factor exists
    for (int i = 0; i < k; ++i) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += ((i <= j) ? F[i * k + j] : F[j * k + i]) * b[j];
        tmp[i] = s;
    }
    for (int i = 0; i < k; ++i) b[i] = tmp[i];
}

// One-off solve A x = b (b -> x) of a symmetric PSD k x k system that must stay intact: F is a k x k work copy, V k x k scratch.
// Cholesky when A is safely positive definite, else the minimum-norm solution.
PLSPM_HD bool psd_solve_once(const double* A, int k, double* F, double* V, double* b) {
    for (int e = 0; e < k * k; ++e) F[e] = A[e];
    if (chol_factor(F, k)) { chol_solve(F, k, b); return true; }
    if (k == 1) return false;
    for (int e = 0; e < k * k; ++e) F[e] = A[e];
    const bool ok = jacobi_pinv(F, k, V);
    for (int i = 0; i < k; ++i) { double s = 0.0; for (int j = 0; j < k; ++j) s += F[i * k + j] * b[j]; V[i] = s; }
    for (int i = 0; i < k; ++i) b[i] = V[i];
    return ok;
}

// Normal equations  M[idx, idx] x = M[idx, col]  for a short index list (the <= kmax predecessors of an LV), M an
// L x L covariance matrix in the workspace.  k <= 8 runs entirely in registers (fully unrolled, predicated
// Cholesky: the workspace accesses are the k*k + k independent gathers only); larger k uses `scratch`.
template <int K>
PLSPM_HD bool spd_solve_fixed(const double* M, int L, const int* idx, int k, int col, double* x) {
    double A[K][K], b[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int ir = (r < k) ? idx[r] : 0;
#pragma unroll
        for (int c = 0; c < K; ++c) {
            const int ic = (c < k) ? idx[c] : 0;
            A[r][c] = (r < k && c < k) ? M[ir * L + ic] : ((r == c) ? 1.0 : 0.0);   // identity padding keeps the factorisation well defined
        }
        b[r] = (r < k) ? M[ir * L + col] : 0.0;
    }
    // square-root-free factorisation A = U' D U (U unit upper triangular): K reciprocals are the only long-latency operations
    // (fp64 sqrt and divide are ~150-cycle dependent chains on the device; the Cholesky form needed 4 sqrt + 12 divides for K = 4).
    // T[r][c] = U[r][c] * D[r] is kept beside U so that the updates are plain multiply-adds.
    bool ok = true;
    double T[K][K], invd[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int r = 0; r < j; ++r) d -= A[r][j] * T[r][j];
        ok = ok && (d > PLSPM_PIVOT_RTOL * A[j][j]);
        invd[j] = 1.0 / d;
#pragma unroll
        for (int c = j + 1; c < K; ++c) {
            double t = A[j][c];
#pragma unroll
            for (int r = 0; r < j; ++r) t -= A[r][j] * T[r][c];
            T[j][c] = t;
            A[j][c] = t * invd[j];                                  // U[j][c] overwrites the upper triangle of A
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) {                                   // U' z = b
        double t = b[i];
#pragma unroll
        for (int r = 0; r < i; ++r) t -= A[r][i] * b[r];
        b[i] = t;
    }
#pragma unroll
    for (int i = 0; i < K; ++i) b[i] *= invd[i];                    // y = D^-1 z
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {                              // U x = y
        double t = b[i];
#pragma unroll
        for (int c = i + 1; c < K; ++c) t -= A[i][c] * b[c];
        b[i] = t;
    }
#pragma unroll
    for (int r = 0; r < K; ++r) if (r < k) x[r] = b[r];
    return ok;
}
// Minimum-norm solution of the same normal equations when they are rank deficient (collinear predecessor scores): what
// statsmodels' OLS.fit() (pinv) returns in scheme.py:50 / inner_model.py:69.  scratch: 2 k^2 doubles.
PLSPM_HD bool pinv_solve(const double* M, int L, const int* idx, int k, int col, double* x, double* scratch) {
    double* A = scratch;
    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) A[r * k + c] = M[idx[r] * L + idx[c]];
    const bool ok = jacobi_pinv(A, k, scratch + (long)k * k);
    for (int r = 0; r < k; ++r) {
        double s = 0.0;
        for (int c = 0; c < k; ++c) s += A[r * k + c] * M[idx[c] * L + col];
        x[r] = s;
    }
    return ok;
}
// scratch: regression_scratch_doubles(kmax) doubles (x may be its last kmax entries).
// KCAP: largest k solved in registers (the 8 x 8 instance alone holds 128 doubles: executors whose threads carry other register
// state -- the rows solver keeps a covariance column -- cap it at 4 and send 5 <= k <= 8 through `scratch` as well).
template <int KCAP = 8>
PLSPM_HD bool spd_solve(const double* M, int L, const int* idx, int k, int col, double* x, double* scratch) {
    bool ok;
    if (k <= 2) ok = spd_solve_fixed<2>(M, L, idx, k, col, x);
    else if (k <= 4) ok = spd_solve_fixed<4>(M, L, idx, k, col, x);
    else if (KCAP >= 8 && k <= 8) ok = spd_solve_fixed<(KCAP >= 8 ? 8 : 4)>(M, L, idx, k, col, x);
    else {
        double* A = scratch;
        for (int r = 0; r < k; ++r) {
            for (int c = 0; c < k; ++c) A[r * k + c] = M[idx[r] * L + idx[c]];
            x[r] = M[idx[r] * L + col];
        }
        ok = chol_factor(A, k);
        if (ok) chol_solve(A, k, x);
    }
    if (ok) return true;
    if (k == 1) return false;
    return pinv_solve(M, L, idx, k, col, x, scratch);
}
// sum_{q in [a, b)} S[q*PS + p] * w[q]  with four independent accumulators (the LDS reads of one step overlap)
PLSPM_HD double dot_col(const double* S, int PS, int p, const double* w, int a, int b) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int q = a;
    for (; q + 3 < b; q += 4) {
        s0 += S[q * PS + p] * w[q];
        s1 += S[(q + 1) * PS + p] * w[q + 1];
        s2 += S[(q + 2) * PS + p] * w[q + 2];
        s3 += S[(q + 3) * PS + p] * w[q + 3];
    }
    for (; q < b; ++q) s0 += S[q * PS + p] * w[q];
    return (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------------
// Stage 1: packed raw scatter -> treated population covariance S (config.py:299-305, util.py:33-39)
// Inverse of packed_index for one stored element: tile (t,u), register r, lane -> (p, q) with p the row column.
PLSPM_HD void packed_coords(int T, int t, int u, int r, int lane, int& p, int& q) {
    p = packed_col_of(T, t, (lane >> 4) + 4 * r);
    q = packed_col_of(T, u, lane & 15);
}

template <class Ex>
PLSPM_HD void moments_to_cov(Ex& ex, const ModelDesc& md, Workspace& ws, const double* Mp) {
    const int P = md.P, PS = ws.PS, T = md.T;
    // 1. sweep the stored tiles in memory order (coalesced 512-byte rows, loads batched): raw second moments -> S,
    //    the ones-column entries -> column sums mu and the row count n.  Diagonal tiles hold both triangles: only
    //    their row <= col half is used so that S is exactly symmetric.
    const int ntile = T * (T + 1) / 2;
    ex.par_chunks64(ntile * 4, Mp, [&](int chunk, int lane, double m) {
        const int tile = chunk >> 2, r = chunk & 3;
        int t, u;
        if (md.tile_tu) { const int tu = md.tile_tu[tile]; t = tu & 255; u = tu >> 8; }
        else { t = 0; int rem = tile; while (rem >= T - t) { rem -= T - t; ++t; } u = t + rem; }
        const int p = packed_col_of(T, t, 4 * r + (lane >> 4));
        const int q = packed_col_of(T, u, lane & 15);
        if ((t != u || p <= q) && p <= P && q <= P) { ws.S[q * PS + p] = m; ws.S[p * PS + q] = m; }
    });
    ex.par(P, [&](int p) { ws.mu[p] = ws.S[P * PS + p]; });
    ex.one([&]() { ws.scal[1] = ws.S[P * PS + P]; });
    ex.mark(14);
    const double n = ws.scal[1], inv_n = 1.0 / n;
    double fac = inv_n;
    if (md.scaled) {
        // g = std1(all N*P raw values) * sqrt((N-1)/N)   (config.py:302), evaluated around the grand mean
        const double tot = ex.sum(P, [&](int p) { return ws.mu[p] + n * md.shift[p]; });
        const double np_ = n * (double)P, grand = tot / np_;
        const double ss = ex.sum(P, [&](int p) {
            const double d = md.shift[p] - grand;
            return ws.S[p * PS + p] + 2.0 * d * ws.mu[p] + n * d * d;
        });
        const double g2 = ss / (np_ - 1.0) * ((n - 1.0) / n);
        fac = 1.0 / (n * g2);
    }
    ex.one([&]() { ws.scal[2] = fac; ws.scal[3] = (double)ST_OK; });
    ex.mark(15);
    // 2. centre and scale in place: S <- (M - mu mu' / n) * fac   (thread p owns column p: conflict-free LDS walk)
    ex.par(P, [&](int p) {
        const double mp = ws.mu[p];
        ws.sd[p] = treated_sd(ws.S[p * PS + p], mp, inv_n, fac);      // (from the RAW diagonal, before the walk below rewrites it)
        int q = 0;
        for (; q + 3 < P; q += 4) {
            const double a0 = ws.S[q * PS + p], a1 = ws.S[(q + 1) * PS + p], a2 = ws.S[(q + 2) * PS + p], a3 = ws.S[(q + 3) * PS + p];
            const double m0 = ws.mu[q], m1 = ws.mu[q + 1], m2 = ws.mu[q + 2], m3 = ws.mu[q + 3];
            ws.S[q * PS + p] = (a0 - (mp * m0) * inv_n) * fac;           // (mu_p mu_q) first: bitwise symmetric in (p, q)
            ws.S[(q + 1) * PS + p] = (a1 - (mp * m1) * inv_n) * fac;
            ws.S[(q + 2) * PS + p] = (a2 - (mp * m2) * inv_n) * fac;
            ws.S[(q + 3) * PS + p] = (a3 - (mp * m3) * inv_n) * fac;
        }
        for (; q < P; ++q) ws.S[q * PS + p] = (ws.S[q * PS + p] - (mp * ws.mu[q]) * inv_n) * fac;
    });
    ex.par(P, [&](int p) { ws.cs[p] = sqrt(fac * n); });   // cs = 1/g (scaled) or 1
    // a zero-variance column (treated_sd above): its row and column of S are rounding residue -- exact zeros instead, so that a Mode-B block that holds it is rank deficient by
    // the pivot rule and takes the minimum-norm answer (weight 0 for that column, like the reference's gelsd) instead of dividing by the residue
    if (ex.any(P, [&](int p) { return ws.sd[p] == 0.0; }))
        ex.par(P, [&](int p) {
            const bool mine0 = ws.sd[p] == 0.0;
            for (int q = 0; q < P; ++q) if (mine0 || ws.sd[q] == 0.0) ws.S[q * PS + p] = 0.0;
        });
}

// ---------------------------------------------------------------------------------------------
// Stage 0 (metric data with missing values): mean imputation on the moments.
// The reference imputes every NaN with its column's mean over the present values -- of the data set handed to treat(), i.e.
// per bootstrap replicate the mean of the *resampled* column (util.py:61-68 via config.py:300, bootstrap.py:57).  The device
// matrix carries, after the P data columns, one 0/1 "missing" indicator column d_p for every column p that has NaNs
// (ind_of[p] = its aug column, -1 if p is complete); the NaN cells of a data column hold one constant e_p (the caller's
// full-data mean, minus the upload shift).  With g_p = (replicate mean of the present cells) - e_p the imputed column is
// x_p + g_p d_p, hence
//     M~(p,q) = M(p,q) + g_q M(p,d_q) + g_p M(d_p,q) + g_p g_q M(d_p,d_q),      M~(p,1) = M(p,1) + g_p M(d_p,1),
// all entries of the SAME weighted Gram of the aug columns.  Reads the Ta-tile packed aug matrix, writes the Ts-tile packed
// matrix of the P logical columns (+ ones) that moments_to_cov expects.  `gam`: P doubles of scratch.
// A column that lost all its present cells (replicate mean undefined -> NaN in the reference) poisons its moments with NaN.
template <class Ex>
PLSPM_HD void impute_collapse(Ex& ex, int P, int Qa, int Ta, int Ts, const int* ind_of, const double* Min, double* Mout, double* gam) {
    const double n = Min[packed_index(Ta, Qa, Qa)];
    ex.par(P, [&](int p) {
        const int c = ind_of[p];
        double g = 0.0;
        if (c >= 0) {
            const double nd = Min[packed_index(Ta, c, Qa)], sp1 = Min[packed_index(Ta, p, Qa)], spd = Min[packed_index(Ta, p, c)];
            const double present = n - nd;
            if (nd > 0.0) g = (present > 0.0) ? (sp1 - spd) / present - spd / nd : NAN;
        }
        gam[p] = g;
    });
    const int ntile = Ts * (Ts + 1) / 2;
    ex.par(ntile * 256, [&](int e) {
        const int tile = e >> 8, r = (e >> 6) & 3, lane = e & 63;
        int t = 0, rem = tile;
        while (rem >= Ts - t) { rem -= Ts - t; ++t; }
        int p, q;
        packed_coords(Ts, t, t + rem, r, lane, p, q);
        double v = 0.0;
        if (p <= P && q <= P) {
            const int ap = p < P ? p : Qa, aq = q < P ? q : Qa;
            const int cp = p < P ? ind_of[p] : -1, cq = q < P ? ind_of[q] : -1;
            const double gp = cp >= 0 ? gam[p] : 0.0, gq = cq >= 0 ? gam[q] : 0.0;
            v = Min[packed_index(Ta, ap, aq)];
            if (cq >= 0 && gq != 0.0) v += gq * Min[packed_index(Ta, ap, cq)];
            if (cp >= 0 && gp != 0.0) v += gp * Min[packed_index(Ta, cp, aq)];
            if (cp >= 0 && cq >= 0 && gp != 0.0 && gq != 0.0) v += (gp * gq) * Min[packed_index(Ta, cp, cq)];
        }
        Mout[e] = v;
    });
}

// Where the treated covariance S lives.  CovLds: the (P+1) x PS array ws.S (LDS or global scratch), any thread count.
// CovRows<PMAX>: thread p keeps row p of S in registers (P <= PMAX, at least P threads: one wave per problem) -- the P x L product
// S W, the only O(P^2) work of an iteration, then runs out of registers with w broadcast from the workspace.
struct CovLds {
    template <class Ex>
    PLSPM_HD void block_products(Ex& ex, const ModelDesc& md, Workspace& ws) const {
        const int L = md.L, PS = ws.PS;
        ex.par2(md.P, L, [&](int p, int m) { ws.V[p * L + m] = dot_col(ws.S, PS, p, ws.w, md.boff[m], md.boff[m + 1]); });
    }
    PLSPM_HD void cov_row(const Workspace& ws, int p, int P, double* dst) const { for (int q = 0; q < P; ++q) dst[q] = ws.S[q * ws.PS + p]; }
};
template <int PMAX>
struct CovRows {
    double s[PMAX];             // s[j] = S[q0 + j][p] of the calling thread's column p (symmetric: its row as well)
    unsigned long long ends;    // bit j set: column q0 + j is the last one of its LV block (wave-uniform; PMAX <= 64)
    // The thread's window of columns: the whole matrix (q0 = 0, nq = P, m0 = 0: one thread per MV, P <= PMAX), or -- split form, two threads
    // per MV, P <= 2 PMAX -- the columns of the LV blocks [m0, ...) on one side of a block boundary (rows_split_block): no block is shared
    // between the two windows, so each thread closes its own blocks and nothing has to be combined.
    int p, q0, nq, m0;
    // V[p, m] = sum over the columns q of block m of s[q] w[q]: the executor's segmented product (device: kernels_solver.h seg_products).
    // Every thread stores (idle ones into a sink -- ws.wn is dead while V is formed and holds P >= L doubles), so closing a block is
    // the same straight-line code on all lanes.
    template <class Ex>
    PLSPM_HD void block_products(Ex& ex, const ModelDesc& md, Workspace& ws) const {
        const int P = md.P, L = md.L;
        ex.template seg_products<PMAX>(s, ws.w + q0, nq, ends, (p < P) ? ws.V + p * L + m0 : ex.sink(ws.wn));
        ex.sync();
    }
    PLSPM_HD void cov_row(const Workspace&, int, int, double* dst) const {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) if (q < nq) dst[q0 + q] = s[q];
    }
};
// Split form of the rows solver: the block boundary that divides the MVs into two windows of at most PMAX columns each (the last
// boundary at or below PMAX); 0 when there is none -- the host asks before it launches (rows_split_covers).
PLSPM_HD int rows_split_block(const int* boff, int L, int PMAX) {
    int ms = 0;
    for (int l = 1; l < L; ++l) if (boff[l] <= PMAX) ms = l;
    return (ms > 0 && boff[L] - boff[ms] <= PMAX) ? ms : 0;
}

// V[p,m] = sum_{q in block m} S[p,q] w[q];   Q[l,m] = sum_{p in block l} w[p] V[p,m]
template <class Ex, class Cov>
PLSPM_HD void apply_cov(Ex& ex, const ModelDesc& md, Workspace& ws, const Cov& cov) {
    const int L = md.L;
    ex.mark(16);
    cov.block_products(ex, md, ws);
    ex.mark(17);
    ex.par(L * L, [&](int e) {
        const int l = e / L, m = e - l * L;
        // four columns per trip, two chains: the LDS reads of a trip are in flight together and a trip pays one loop branch
        // (~20 clocks on the device, taken or not) instead of four
        double s0 = 0.0, s1 = 0.0;
        int p = md.boff[l];
        const int pe = md.boff[l + 1];
        for (; p + 3 < pe; p += 4) {
            s0 += ws.w[p] * ws.V[p * L + m]; s1 += ws.w[p + 1] * ws.V[(p + 1) * L + m];
            s0 += ws.w[p + 2] * ws.V[(p + 2) * L + m]; s1 += ws.w[p + 3] * ws.V[(p + 3) * L + m];
        }
        for (; p < pe; ++p) s0 += ws.w[p] * ws.V[p * L + m];
        ws.Q[e] = s0 + s1;
    });
    ex.mark(18);
}
template <class Ex>
PLSPM_HD void apply_cov(Ex& ex, const ModelDesc& md, Workspace& ws) { apply_cov(ex, md, ws, CovLds{}); }

// Inner weights E from G = cov0(Yhat) (scheme.py:27-28 / 36-37 / 45-54)
// `Graw`: raw (uncentred) second moments of the scores for the PATH scheme's no-intercept OLS (scheme.py:50); null when the
// scores are centred and the covariance ws.G serves both purposes.
template <class Ex>
PLSPM_HD void inner_weights(Ex& ex, const ModelDesc& md, Workspace& ws, double corr2, const double* Graw = nullptr) {
    const int L = md.L;
    const double* Gols = Graw ? Graw : ws.G;
    if (md.scheme == SCHEME_PATH) {
        // column i of E: regression coefficients on the predecessors of i ("follow", scheme.py:47-50), correlations with its
        // successors ("predec", scheme.py:51-53), zero elsewhere.  The L^2 entries first (one thread each: a sqrt + a divide),
        // then the L regressions -- inside one per-LV loop the successor entries added up to three sqrt / divide chains per thread.
        ex.par(L * L, [&](int e) {
            const int s2 = e / L, i = e - s2 * L;
            ws.E[e] = md.C[s2 * L + i] ? ws.G[e] / sqrt(ws.G[s2 * L + s2] * ws.G[i * L + i]) : 0.0;
        });
        ex.par(L, [&](int i) {
            const int km = md.kmax;
            double* scratch = ws.scr + (long)i * regression_scratch_doubles(km);
            const int* f = md.pred_idx + md.pred_off[i];                    // predecessors of i
            const int k = md.pred_off[i + 1] - md.pred_off[i];
            if (k > 0) {
                double* x = scratch + 2 * km * km;
                if (!spd_solve<Ex::kcap>(Gols, L, f, k, i, x, scratch)) ws.scal[3] = (double)ST_SINGULAR;
                for (int r = 0; r < k; ++r) ws.E[f[r] * L + i] = x[r];
            }
        });
    } else {
        ex.par(L * L, [&](int e) {
            const int l = e / L, m = e - l * L;
            const int d = (int)md.C[l * L + m] + (int)md.C[m * L + l];
            double v = 0.0;
            if (d) {
                if (md.scheme == SCHEME_CENTROID) {
                    const double g = ws.G[e];
                    v = (g > 0.0) ? 1.0 : ((g < 0.0) ? -1.0 : 0.0);
                } else {
                    v = ws.G[e] * corr2 * (double)d;                           // cov1 = cov0 * N/(N-1)
                }
            }
            ws.E[e] = v;
        });
    }
}

// One PLS iteration (weights.py:41-54).  Returns the convergence measure (identical on every thread).
template <class Ex, class Cov>
PLSPM_HD double iterate(Ex& ex, const ModelDesc& md, Workspace& ws, double corr2, const Cov& cov) {
    const int P = md.P, L = md.L;
    ex.mark(8);
    apply_cov(ex, md, ws, cov);
    ex.mark(9);
    ex.par(L, [&](int l) { ws.a[l] = 1.0 / (corr2 * sqrt(ws.Q[l * L + l])); });   // Yhat_l = Y_l / std1 / corr
    ex.par(L * L, [&](int e) { ws.G[e] = ws.a[e / L] * ws.a[e % L] * ws.Q[e]; });
    ex.mark(10);
    inner_weights(ex, md, ws, corr2);
    ex.mark(11);
    ex.par(P, [&](int p) {                                                         // (S Wn E)[p, lv(p)]  == X'Z/N  (mode.py:29)
        const int l = md.lvof[p];
        double s0 = 0.0, s1 = 0.0;                            // two LVs per trip, two chains
        int m = 0;
        for (; m + 1 < L; m += 2) { s0 += ws.a[m] * ws.V[p * L + m] * ws.E[m * L + l]; s1 += ws.a[m + 1] * ws.V[p * L + m + 1] * ws.E[(m + 1) * L + l]; }
        if (m < L) s0 += ws.a[m] * ws.V[p * L + m] * ws.E[m * L + l];
        const double s = s0 + s1;
        ws.cv[p] = s;
        ws.wn[p] = s;
    });
    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {                                                     // Mode B: S_bb w = c_b  (mode.py:51)
            if (md.mode[l] == MODE_B) {
                const int b0 = md.boff[l], k = md.boff[l + 1] - b0;
                psd_solve(ws.chol + md.chol_off[l], k, ws.wn + b0);
            }
        });
    }
    ex.mark(12);
    const double conv = ex.sum(P, [&](int p) { const double d = fabs(ws.w[p]) - fabs(ws.wn[p]); return d * d; });
    ex.par(P, [&](int p) { ws.w[p] = ws.wn[p]; });
    ex.mark(13);
    return conv;
}

struct FitOutputs {             // any pointer may be null
    double* row;                // [2P + L + 2 n_eff + 2]  weights | r2 | total | direct | loadings | status | iterations
                                //                         (device column order; the last two as doubles, for the one-collective gather)
    double* weights;            // [P]
    double* loadings;           // [P]
    double* crossloadings;      // [P*L]
    double* path_coef;          // [L*L]
    double* r2;                 // [L]
    double* lv_cov;             // [L*L] population covariance of the (sign-corrected) scores
    double* indirect;           // [n_eff]
    double* score_w;            // [P]  sgn_l * w_p / g-free factor: scores = (x' - mu') * score_w summed per block
    double* score_c;            // [L]  constant term of the score map
    double* cov;                // [P*P] treated covariance S (row-major)
    double* mean;               // [P]   column means of the raw data of this problem
    int8_t* sign;               // [L]
    int* iters;
    int* status;
};

template <class Ex, class Cov>
PLSPM_HD void finish_problem(Ex& ex, const ModelDesc& md, Workspace& ws, const FitOutputs& out, int iteration, bool sign_rule, const Cov& cov);
template <class Ex>
PLSPM_HD void finish_problem(Ex& ex, const ModelDesc& md, Workspace& ws, const FitOutputs& out, int iteration, bool sign_rule) {
    finish_problem(ex, md, ws, out, iteration, sign_rule, CovLds{});
}

template <class Ex>
PLSPM_HD void solve_problem(Ex& ex, const ModelDesc& md, Workspace& ws, const double* Mp, const FitOutputs& out) {
    const int P = md.P, L = md.L, PS = ws.PS;
    ex.mark(0);
    moments_to_cov(ex, md, ws, Mp);
    ex.mark(1);
    const double n = ws.scal[1];
    const double corr2 = n / (n - 1.0);

    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {
            if (md.mode[l] == MODE_B) {
                const int b0 = md.boff[l], k = md.boff[l + 1] - b0;
                const bool ok = psd_factor(ws.chol + md.chol_off[l], k, [&](double* R) {
                    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) R[r * k + c] = ws.S[(b0 + r) * PS + b0 + c];
                });
                if (!ok) ws.scal[3] = (double)ST_SINGULAR;
            }
        });
    }
    // init (weights.py:28-39): w_p = corr / std1(sum of the block's MVs) = 1 / sqrt(sum(S_bb))
    ex.par(P, [&](int p) { const int l = md.lvof[p]; ws.w[p] = 1.0; ws.dv[p] = 0.0; (void)l; });
    ex.par(P, [&](int p) { const int l = md.lvof[p]; ws.dv[p] = dot_col(ws.S, PS, p, ws.w, md.boff[l], md.boff[l + 1]); });   // block row sums
    ex.par(L, [&](int l) {
        double s = 0.0;
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += ws.dv[p];
        ws.wf[l] = 1.0 / sqrt(s);
    });
    ex.par(P, [&](int p) { ws.w[p] = ws.wf[md.lvof[p]]; });
    ex.mark(2);

    // weights.py:179-186: iteration counter, stop on conv < tol or counter > max_iter, fail if counter > max_iter
    int iteration = 0;
    while (true) {
        ++iteration;
        const double conv = iterate(ex, md, ws, corr2, CovLds{});
        if (conv < md.tol || iteration > md.max_iter) break;
    }
    ex.one([&]() { if (iteration > md.max_iter && ws.scal[3] == (double)ST_OK) ws.scal[3] = (double)ST_NOT_CONVERGED; });
    ex.mark(3);
    finish_problem(ex, md, ws, out, iteration, true);
}

// -------------------------------------------------------------------------------------------------------------
// Rows variant of solve_problem for narrow models (P <= PMAX, at least P threads -- one 64-lane wave per problem on the device):
// thread p keeps column p of the treated covariance in registers (CovRows), so the workspace holds the small arrays only
// (~10 KB instead of 40 KB at P = 60: twice the resident problems per CU, no cross-wave barriers) and the P x L product of every
// iteration needs no loads of S.  `Md`: the DENSE moment matrix [(P+1) x PS] of the mean-shifted columns + ones (row / column P),
// UPPER triangle only (entry (r, c), r <= c, at r * PS + c), as the int8 digit-plane Gram writes it (kernels_gram_i8.h, dense
// slots: one store per element, 16 consecutive columns of a row per store group).  Same arithmetic as moments_to_cov /
// solve_problem; sums over a block run in a different (fixed) order, so results agree to rounding, not bitwise.
// SPLIT (64 < P <= 2 PMAX, 4 PMAX threads = two per MV; round 4): thread t serves MV p = t mod 2 PMAX with the columns on side t / (2 PMAX) of
// the block boundary rows_split_block -- the covariance of a 128-MV model sits in the registers of four waves instead of 115 KB of LDS
// (one problem per CU), and everything outside the three places that touch `cov` is the same code on more threads.
template <int PMAX, bool SPLIT = false, class Ex>
PLSPM_HD void solve_problem_rows(Ex& ex, const ModelDesc& md, Workspace& ws, const double* Md, const FitOutputs& out) {
    const int P = md.P, L = md.L, PS = cov_ld(P), p = SPLIT ? (ex.tid & (2 * PMAX - 1)) : ex.tid;
    const int side = SPLIT ? ex.tid / (2 * PMAX) : 0;
    const bool mine = p < P && side == 0;                 // the thread that owns MV p's entries of the per-MV arrays
    CovRows<PMAX> cov;
    cov.p = p;
    {
        const int ms = SPLIT ? rows_split_block(md.boff, L, PMAX) : 0, l0 = side ? ms : 0, l1 = (SPLIT && !side) ? ms : L;
        cov.m0 = l0; cov.q0 = md.boff[l0]; cov.nq = md.boff[l1] - md.boff[l0];
        unsigned long long e = 0ull;
        for (int l = l0; l < l1; ++l) e |= 1ull << (md.boff[l + 1] - 1 - cov.q0);
        cov.ends = ex.uniform(e);                         // the same value on every thread of a wave: block boundaries become scalar tests
    }
    const int q0 = cov.q0, nq = cov.nq;
    ex.mark(0);
    // 1. moments -> treated covariance (config.py:299-305, util.py:33-39).  Column p = entries (q, p) above the diagonal (one row of
    //    Md across the threads: coalesced) + the thread's own row (p, q) behind it (a contiguous run per thread).
    //    One load per column with a selected address (a select of two loads became two exec-masked branches per column); idle threads
    //    and columns past P re-read valid entries, zeroed below.
    const int pc = (p < P) ? p : P - 1;
    if constexpr (SPLIT) {
        // (the executor's block loader: on the device rows of the matrix are read along the lanes and turned through LDS -- the lane-per-row
        //  walk below touches 64 cache lines per wave instruction: 42-48 k -> 34 k clocks of a problem's first-round load at 120 MVs)
        ex.template load_cov_block<PMAX>(Md, PS, P, pc, q0, nq, cov.s);
    } else {
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const int qc = q0 + ((q < nq) ? q : nq - 1);
        const unsigned off = (unsigned)((qc <= pc) ? qc * PS + pc : pc * PS + qc) * 8u;      // 32-bit byte offset from one base: one
        cov.s[q] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Md) + off); // address register per load, all in flight
    }
    }
    double dpp = 0.0, mup = 0.0;
    if (p < P) { mup = Md[(long)p * PS + P]; dpp = Md[(long)p * PS + p]; }
    if (mine) { ws.mu[p] = mup; ws.dv[p] = dpp; }
    if (ex.tid == 0) ws.scal[1] = Md[(long)P * PS + P];
    ex.sync();
    ex.mark(14);
    const double n = ws.scal[1], inv_n = 1.0 / n;
    double fac = inv_n;
    if (md.scaled) {
        const double tot = ex.sum(P, [&](int i) { return ws.mu[i] + n * md.shift[i]; });
        const double np_ = n * (double)P, grand = tot / np_;
        const double ss = ex.sum(P, [&](int i) {
            const double d = md.shift[i] - grand;
            return ws.dv[i] + 2.0 * d * ws.mu[i] + n * d * d;
        });
        const double g2 = ss / (np_ - 1.0) * ((n - 1.0) / n);
        fac = 1.0 / (n * g2);
    }
    ex.one([&]() { ws.scal[2] = fac; ws.scal[3] = (double)ST_OK; });
    ex.mark(15);
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const double v = (cov.s[q] - (mup * ws.mu[q0 + ((q < nq) ? q : nq - 1)]) * inv_n) * fac;          // (mu_p mu_q) first: bitwise symmetric in (p, q)
        cov.s[q] = (q < nq) ? v : 0.0;
    }
    if (mine) {
        ws.sd[p] = treated_sd(dpp, mup, inv_n, fac);
        ws.cs[p] = sqrt(fac * n);
    }
    ex.sync();
    if (ex.any(P, [&](int e) { return ws.sd[e] == 0.0; })) {      // zero-variance columns: exact zeros in their rows and columns (moments_to_cov)
        const bool mine0 = p < P && ws.sd[p] == 0.0;
#pragma unroll
        for (int q = 0; q < PMAX; ++q) if (q < nq && (mine0 || ws.sd[q0 + q] == 0.0)) cov.s[q] = 0.0;
    }
    ex.mark(1);
    const double corr2 = n / (n - 1.0);

    if (md.n_chol > 0) {
        // every thread of a Mode-B block writes its row of S_bb into the block's factor slot and into the scratch half behind it
        // (psd_factor restores from there for the minimum-norm fallback)
        if (p < P) {                                      // (split form: the side whose window holds the block finds its columns)
            const int l = md.lvof[p];
            if (md.mode[l] == MODE_B) {
                const int b0 = md.boff[l] - q0, b1 = md.boff[l + 1] - q0, k = b1 - b0, pr = p - md.boff[l];
                double* F = ws.chol + md.chol_off[l];
#pragma unroll
                for (int q = 0; q < PMAX; ++q)
                    if (q >= b0 && q < b1 && q < nq) { F[pr * k + (q - b0)] = cov.s[q]; F[(long)k * k + pr * k + (q - b0)] = cov.s[q]; }
            }
        }
        ex.sync();
        ex.par(L, [&](int l) {
            if (md.mode[l] == MODE_B) {
                const int k = md.boff[l + 1] - md.boff[l];
                const bool ok = psd_factor(ws.chol + md.chol_off[l], k, [&](double* R) { for (int e = 0; e < k * k; ++e) R[e] = R[(long)k * k + e]; });
                if (!ok) ws.scal[3] = (double)ST_SINGULAR;
            }
        });
    }
    // init (weights.py:28-39): w_p = 1 / sqrt(sum(S_bb)) of the MV's own block -- the block row sums are the products with w = 1
    // (a per-thread `q in [b0, b1)` select chain over the 64 registers cost three instructions per column)
    ex.par(P, [&](int i) { ws.w[i] = 1.0; });
    cov.block_products(ex, md, ws);
    ex.par(P, [&](int i) { ws.dv[i] = ws.V[i * L + md.lvof[i]]; });
    ex.par(L, [&](int l) {
        double s0 = 0.0, s1 = 0.0;                            // (four reads per trip: one loop branch instead of four, the reads in flight together)
        int q = md.boff[l];
        const int qe = md.boff[l + 1];
        for (; q + 3 < qe; q += 4) { s0 += ws.dv[q]; s1 += ws.dv[q + 1]; s0 += ws.dv[q + 2]; s1 += ws.dv[q + 3]; }
        for (; q < qe; ++q) s0 += ws.dv[q];
        ws.wf[l] = 1.0 / sqrt(s0 + s1);
    });
    ex.par(P, [&](int i) { ws.w[i] = ws.wf[md.lvof[i]]; });
    ex.mark(2);

    int iteration = 0;
    while (true) {
        ++iteration;
        const double conv = iterate(ex, md, ws, corr2, cov);
        if (conv < md.tol || iteration > md.max_iter) break;
    }
    ex.one([&]() { if (iteration > md.max_iter && ws.scal[3] == (double)ST_OK) ws.scal[3] = (double)ST_NOT_CONVERGED; });
    ex.mark(3);
    finish_problem(ex, md, ws, out, iteration, true, cov);
}

// Inner model (inner_model.py:58-75: OLS with intercept == centred normal equations on the score covariance ws.Cs) and the
// effects (inner_model.py:33-53): indirect = sum_{k>=2} B^k, total = B + indirect.  -> ws.Bm, ws.r2, ws.Ind
template <class Ex>
PLSPM_HD void inner_model_effects(Ex& ex, const ModelDesc& md, Workspace& ws) {
    const int L = md.L;
    ex.par(L, [&](int i) {
        for (int j = 0; j < L; ++j) ws.Bm[i * L + j] = 0.0;
        ws.r2[i] = 0.0;
        const int km = md.kmax;
        double* scratch = ws.scr + (long)i * regression_scratch_doubles(km);
        const int* f = md.pred_idx + md.pred_off[i];
        const int k = md.pred_off[i + 1] - md.pred_off[i];
        if (k > 0) {
            double* x = scratch + 2 * km * km;
            if (!spd_solve<Ex::kcap>(ws.Cs, L, f, k, i, x, scratch)) ws.scal[3] = (double)ST_SINGULAR;
            double expl = 0.0;
            for (int r = 0; r < k; ++r) { ws.Bm[i * L + f[r]] = x[r]; expl += x[r] * ws.Cs[f[r] * L + i]; }
            ws.r2[i] = expl / ws.Cs[i * L + i];
        }
    });
    ex.mark(5);
    // indirect = B^2 + B^3 + ... (inner_model.py:37-44 sums the matrix powers) = (I - B)^-1 - I - B.  B is strictly lower triangular in
    // path order, so column j of (I - B)^-1 follows by forward substitution: x_j = 1, x_i = B[i][j] + sum_{j < k < i} B[i][k] x_k.  One thread
    // per column; ws.Pw holds the columns of (I - B)^-1 (row-major like B).
    ex.par(L, [&](int j) {
        for (int i = 0; i < L; ++i) {
            double ind = 0.0;                                    // paths j -> ... -> i of length >= 2 (no cancellation: summed on its own)
            for (int k = j + 1; k < i; ++k) ind += ws.Bm[i * L + k] * ws.Pw[k * L + j];
            ws.Pw[i * L + j] = (i == j) ? 1.0 : ((i > j) ? ws.Bm[i * L + j] + ind : 0.0);
            ws.Ind[i * L + j] = ind;
        }
    });
}

// Everything after the iteration (shared by the metric and the non-metric solver): final normalisation, the metric sign
// rule, inner model, effects, loadings, outputs.  Expects the last weights in ws.w and the treated covariance in ws.S.
template <class Ex, class Cov>
PLSPM_HD void finish_problem(Ex& ex, const ModelDesc& md, Workspace& ws, const FitOutputs& out, int iteration, bool sign_rule, const Cov& cov) {
    const int P = md.P, L = md.L;
    // finalize (weights.py:56-70; non-metric: weights.py:130-132, no sign rule)
    apply_cov(ex, md, ws, cov);
    ex.par(L, [&](int l) { ws.wf[l] = 1.0 / sqrt(ws.Q[l * L + l]); });        // 1 / (std1(X w_l) / corr)
    ex.par(P, [&](int p) { ws.w[p] *= ws.wf[md.lvof[p]]; });                  // returned weights: never sign-flipped
    // sign rule: EVERY MV votes (weights.py:62-64); sign(cor[p,l]) == sign(V[p,l]): cor = V * wf / sd with wf, sd > 0.
    // (round 5: the P votes of an LV counted by up to eight threads -- partial counts in ws.Pw2, free until the effects -- instead of one thread walking
    //  all P entries of a column: 11 k -> ~3 k clocks at 120 MVs; integer counts, the same result)
    const int NC = L < 8 ? L : 8;
    ex.par(L * NC, [&](int e) {
        const int l = e / NC, c = e - l * NC;
        const int p0 = (int)((long)P * c / NC), p1 = (int)((long)P * (c + 1) / NC);
        // (a zero-variance column -- treated_sd -- votes +1 in EVERY LV: pandas' corr() returns np.nan for it, sign bit clear, and copysign(1.0, nan) of weights.py:63 is +1.
        //  Its entries of V are rounding residue of either sign: sd, exactly 0 for such a column and for no other, keeps them out of the negative count.)
        int vote = 0;
        int p = p0;
        for (; p + 3 < p1; p += 4) {
            const double v0 = ws.V[p * L + l], v1 = ws.V[(p + 1) * L + l], v2 = ws.V[(p + 2) * L + l], v3 = ws.V[(p + 3) * L + l];
            const bool f0 = ws.sd[p] == 0.0, f1 = ws.sd[p + 1] == 0.0, f2 = ws.sd[p + 2] == 0.0, f3 = ws.sd[p + 3] == 0.0;
            vote += ((v0 < 0.0 && !f0) ? -1 : 1) + ((v1 < 0.0 && !f1) ? -1 : 1) + ((v2 < 0.0 && !f2) ? -1 : 1) + ((v3 < 0.0 && !f3) ? -1 : 1);
        }
        for (; p < p1; ++p) vote += (ws.V[p * L + l] < 0.0 && ws.sd[p] != 0.0) ? -1 : 1;
        ws.Pw2[e] = (double)vote;
    });
    ex.par(L, [&](int l) {
        int vote = 0;
        for (int c = 0; c < NC; ++c) vote += (int)ws.Pw2[l * NC + c];
        ws.sgn[l] = (sign_rule && vote < 0) ? -1.0 : 1.0;
    });
    ex.par(L * L, [&](int e) { const int l = e / L, m = e - l * L; ws.Cs[e] = ws.sgn[l] * ws.sgn[m] * ws.wf[l] * ws.wf[m] * ws.Q[e]; });

    ex.mark(4);
    inner_model_effects(ex, md, ws);

    ex.mark(6);
    // outputs
    ex.par(P, [&](int p) {
        const int l = md.lvof[p];
        const bool flat = ws.sd[p] == 0.0;                                 // a zero-variance column (treated_sd): loading 0, cross-loadings NaN, like the reference
        const double ld = flat ? 0.0 : ws.sgn[l] * ws.V[p * L + l] * ws.wf[l] / ws.sd[p];
        if (out.row) { out.row[p] = ws.w[p]; out.row[P + L + 2 * md.n_eff + p] = ld; }
        if (out.weights) out.weights[p] = ws.w[p];
        if (out.loadings) out.loadings[p] = ld;
        if (out.crossloadings) for (int m = 0; m < L; ++m) out.crossloadings[p * L + m] = flat ? sqrt(-1.0) : ws.sgn[m] * ws.V[p * L + m] * ws.wf[m] / ws.sd[p];
        // scores_l = sgn_l * sum_p (x'_p - mu'_p) * cs_p * w_p
        const double n_ = ws.scal[1];
        if (out.score_w) out.score_w[p] = ws.sgn[l] * ws.w[p] * ws.cs[p];
        if (out.mean) out.mean[p] = ws.mu[p] / n_ + md.shift[p];
        if (out.cov) cov.cov_row(ws, p, P, out.cov + (long)p * P);
    });
    ex.par(L, [&](int l) {
        if (out.row) out.row[P + l] = ws.r2[l];
        if (out.r2) out.r2[l] = ws.r2[l];
        if (out.sign) out.sign[l] = (int8_t)ws.sgn[l];
        if (out.score_c) {
            const double n_ = ws.scal[1];
            double s = 0.0;
            for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += (ws.mu[p] / n_) * ws.sgn[l] * ws.w[p] * ws.cs[p];
            out.score_c[l] = -s;
        }
    });
    ex.par(L * L, [&](int e) {
        if (out.path_coef) out.path_coef[e] = ws.Bm[e];
        if (out.lv_cov) out.lv_cov[e] = ws.Cs[e];
    });
    ex.par(md.n_eff, [&](int e) {
        const int idx = md.eff_to[e] * L + md.eff_from[e];
        if (out.row) { out.row[P + L + e] = ws.Bm[idx] + ws.Ind[idx]; out.row[P + L + md.n_eff + e] = ws.Bm[idx]; }
        if (out.indirect) out.indirect[e] = ws.Ind[idx];
    });
    const bool bad = ex.any(P + L, [&](int e) {
        if (e < P) return !(isfinite(ws.w[e]) && isfinite(ws.sd[e]) && ws.sd[e] >= 0.0);
        return !isfinite(ws.r2[e - P]);
    });
    ex.one([&]() {
        int st = (int)ws.scal[3];
        if (st == ST_OK && bad) st = ST_NONFINITE;
        if (out.status) *out.status = st;
        if (out.iters) *out.iters = iteration;
        if (out.row) { out.row[2 * P + L + 2 * md.n_eff] = (double)st; out.row[2 * P + L + 2 * md.n_eff + 1] = (double)iteration; }
    });
    ex.mark(7);
}


// =============================================================================================================
// Non-metric data with Scale.NUM / Scale.RAW (reference _NonmetricWeights, plspm/weights.py:73-133; mode.py:31-42, 54-61;
// scale.py:22-39; Config.treat config.py:306-318).  Every MV is population-standardised, so the iteration lives on the
// correlation matrix R; scores are y_l = Xs a_l with a_l supported on block l.  One step:
//     Cy = A' R A;  E = scheme(Cy);  u_l = (R A E)_l;  Mode A  w_l = u_l / (E' Cy E)_ll,  Mode B  w_l = R_bb^-1 u_l;
//     a_l <- w_l / sqrt(w_l' R_bb w_l)                                   (treat_numpy(X w) * correction, mode.py:41,60)
// The stop rule is on the SCORES, sum_il (|y_old| - |y_new|)^2 (weights.py:120): the absolute values do not reduce to
// second moments, so after every step a streaming pass over the observations (nm_conv kernels) evaluates it exactly
// from the two coefficient sets; nm_step() of the next launch reads that sum and decides.
// Persistent per-problem state (global memory, nm_state_doubles()):
//     [0..8) scal: 0 n, 1 status, 2 iteration, 3 active, 4 last convergence value
//     a_old[P] a_new[P]   coefficients on the standardised MVs        c_old[P] c_new[P]  the same per uploaded column (a / sigma)
//     k_old[L] k_new[L]   constant terms of the score maps             sd[P] sigma_p      mu[P] column sums      chol[n_chol]
struct NmState {
    double *scal, *a_old, *a_new, *c_old, *c_new, *k_old, *k_new, *sd, *mu, *chol;
};
PLSPM_HD long nm_state_doubles(int P, int L, int n_chol) { return 8 + 6L * P + 2L * L + n_chol; }
PLSPM_HD void nm_carve(NmState& st, double* base, int P, int L) {
    double* p = base;
    st.scal = p; p += 8; st.a_old = p; p += P; st.a_new = p; p += P; st.c_old = p; p += P; st.c_new = p; p += P;
    st.k_old = p; p += L; st.k_new = p; p += L; st.sd = p; p += P; st.mu = p; p += P; st.chol = p;
}

// score-map coefficients of a coefficient vector a on the standardised MVs: y = sum_p x'_p c_p + k_l
template <class Ex>
PLSPM_HD void nm_score_map(Ex& ex, const ModelDesc& md, const NmState& st, const double* a, double* c, double* k) {
    const double n = st.scal[0];
    ex.par(md.P, [&](int p) { c[p] = a[p] / st.sd[p]; });
    ex.par(md.L, [&](int l) {
        double s = 0.0;
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += (st.mu[p] / n) * c[p];
        k[l] = -s;
    });
}

// packed raw scatter -> correlation matrix R (ws.S, global memory), sigma, mu; initial coefficients 1/sqrt(k_l) (weights.py:82-98)
template <class Ex>
PLSPM_HD void nm_prepare(Ex& ex, const ModelDesc& md, Workspace& ws, NmState& st, const double* Mp) {
    const int P = md.P, L = md.L, PS = ws.PS, T = md.T;
    const int ntile = T * (T + 1) / 2;
    // n, the column sums and the diagonal first (2 P + 1 loads straight from the packed matrix): the scatter then writes the CORRELATIONS in the
    // same pass (a separate normalisation pass re-read and re-wrote the whole square).  Same expressions as before, entry by entry.
    const double n = Mp[packed_index(T, P, P)], inv_n = 1.0 / n;
    ex.par(P, [&](int p) {
        const double mu = Mp[packed_index(T, p, P)];
        st.mu[p] = mu;
        st.sd[p] = nm_column_sd(Mp[packed_index(T, p, p)] * inv_n, mu * inv_n);                  // population std (config.py:314); NaN for a constant column
    });
    ex.one([&]() { st.scal[0] = n; st.scal[1] = (double)ST_OK; st.scal[2] = 0.0; st.scal[3] = 1.0; st.scal[4] = 0.0; });
    ex.par_chunks64(ntile * 4, Mp, [&](int chunk, int lane, double m) {
        const int tile = chunk >> 2, r = chunk & 3;
        int t, u;
        if (md.tile_tu) { const int tu = md.tile_tu[tile]; t = tu & 255; u = tu >> 8; }
        else { t = 0; int rem = tile; while (rem >= T - t) { rem -= T - t; ++t; } u = t + rem; }
        const int p = packed_col_of(T, t, 4 * r + (lane >> 4));
        const int q = packed_col_of(T, u, lane & 15);
        if ((t != u || p <= q) && p <= P && q <= P) {
            double v = m;                                   // row / column P: raw column sums and n
            if (p < P && q < P) v = ((m - (st.mu[p] * st.mu[q]) * inv_n) * inv_n) / (st.sd[p] * st.sd[q]);
            ws.S[q * PS + p] = v; ws.S[p * PS + q] = v;
        }
    });
    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {
            if (md.mode[l] == MODE_B) {
                const int b0 = md.boff[l], k = md.boff[l + 1] - b0;
                const bool ok = psd_factor(st.chol + md.chol_off[l], k, [&](double* R) {
                    for (int r = 0; r < k; ++r) for (int c = 0; c < k; ++c) R[r * k + c] = ws.S[(b0 + r) * PS + b0 + c];
                });
                if (!ok) st.scal[1] = (double)ST_SINGULAR;
            }
        });
    }
    ex.par(P, [&](int p) { const int l = md.lvof[p]; st.a_old[p] = 1.0 / sqrt((double)(md.boff[l + 1] - md.boff[l])); st.a_new[p] = st.a_old[p]; });
    nm_score_map(ex, md, st, st.a_old, st.c_old, st.k_old);
    nm_score_map(ex, md, st, st.a_new, st.c_new, st.k_new);
}

// Decide on the previous step's convergence value (sum of `nparts` partial sums, fixed order), then -- if the problem is
// still active -- run one more iteration.  Returns true when the problem is still active after this call.
template <class Ex>
PLSPM_HD bool nm_step(Ex& ex, const ModelDesc& md, Workspace& ws, NmState& st, const double* partial, int nparts) {
    const int P = md.P, L = md.L, PS = ws.PS;
    if (st.scal[3] == 0.0) return false;
    const int iteration = (int)st.scal[2];
    if (iteration > 0) {
        const double conv = ex.sum(nparts, [&](int c) { return partial[c]; });
        // (a NaN criterion is absorbing -- solver_nmg.h nmg_step --: the problem leaves with the record its max_iter + 1 trips would end in)
        const bool never = conv != conv;
        const bool stop = (conv < md.tol) || (iteration > md.max_iter) || never;            // weights.py:183
        ex.one([&]() {
            st.scal[4] = conv;
            if (stop) { st.scal[3] = 0.0; if ((iteration > md.max_iter || never) && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED; }
            if (never) st.scal[2] = (double)(md.max_iter + 1);
        });
        if (stop) return false;
        ex.par(P, [&](int p) { st.a_old[p] = st.a_new[p]; st.c_old[p] = st.c_new[p]; });
        ex.par(L, [&](int l) { st.k_old[l] = st.k_new[l]; });
    }
    const double n = st.scal[0], corr2 = n / (n - 1.0);
    ex.one([&]() { ws.scal[3] = (double)ST_OK; });                                // the small workspace does not survive between launches
    ex.par(P, [&](int p) { ws.w[p] = st.a_old[p]; });
    apply_cov(ex, md, ws);                                                        // V = R A, Q = Cy = A' R A
    ex.par(L * L, [&](int e) { ws.G[e] = ws.Q[e]; });
    inner_weights(ex, md, ws, corr2);                                             // scheme on the (un-normalised) score covariance
    ex.par(L, [&](int l) {                                                        // zeta_l = (E' Cy E)_ll = sum z_l^2 / n
        double s = 0.0;
        for (int m = 0; m < L; ++m) {
            const double em = ws.E[m * L + l];
            if (em == 0.0) continue;
            double t = 0.0;
            for (int m2 = 0; m2 < L; ++m2) t += ws.Q[m * L + m2] * ws.E[m2 * L + l];
            s += em * t;
        }
        ws.a[l] = s;
    });
    ex.par(P, [&](int p) {                                                        // u_l = (R A E)[p, lv(p)] = X' z_l / n
        const int l = md.lvof[p];
        double s = 0.0;
        for (int m = 0; m < L; ++m) s += ws.V[p * L + m] * ws.E[m * L + l];
        ws.wn[p] = (md.mode[l] == MODE_A) ? s / ws.a[l] : s;                      // Mode A: X_b' z / sum z^2 (mode.py:38)
    });
    if (md.n_chol > 0) {
        ex.par(L, [&](int l) {                                                    // Mode B: lstsq(X_b, z) = R_bb^-1 u_b (mode.py:58)
            if (md.mode[l] == MODE_B) { const int b0 = md.boff[l]; psd_solve(st.chol + md.chol_off[l], md.boff[l + 1] - b0, ws.wn + b0); }
        });
    }
    ex.par(P, [&](int p) { const int l = md.lvof[p]; ws.dv[p] = ws.wn[p] * dot_col(ws.S, PS, p, ws.wn, md.boff[l], md.boff[l + 1]); });
    ex.par(L, [&](int l) {                                                        // std0(X_b w_l)
        double s = 0.0;
        for (int p = md.boff[l]; p < md.boff[l + 1]; ++p) s += ws.dv[p];
        ws.wf[l] = 1.0 / sqrt(s);
    });
    ex.par(P, [&](int p) { st.a_new[p] = ws.wn[p] * ws.wf[md.lvof[p]]; });
    nm_score_map(ex, md, st, st.a_new, st.c_new, st.k_new);
    ex.one([&]() { st.scal[2] = (double)(iteration + 1); if (ws.scal[3] != (double)ST_OK && st.scal[1] == (double)ST_OK) st.scal[1] = ws.scal[3]; });
    // The stop-rule value of THIS step, sum_il c_i (|y_old| - |y_new|)^2 (weights.py:120), needs a pass over the observations -- but
    // ||a| - |b|| <= |a - b|, so it is bounded from above by  sum_il c_i (y_old - y_new)^2 = n sum_l d_l' R_bb d_l,  d = a_new - a_old,
    // which lives on the correlation matrix.  The two differ only in the observations whose score changes sign between the steps
    // (both scores near zero there), so when the iteration has converged the bound says so too: the problem stops HERE, with the same
    // iteration count as the reference, and the pass of the last iteration -- a third of the stop-rule work at three iterations -- is
    // not needed (its replicate groups find themselves inactive).  Never for the first step: the launch that prepares a problem does
    // not finish it.  The margin covers the rounding of the bound itself.
    if (iteration >= 1) {
        ex.par(P, [&](int p) { ws.cv[p] = st.a_new[p] - st.a_old[p]; });
        ex.par(P, [&](int p) { const int l = md.lvof[p]; ws.dv[p] = ws.cv[p] * dot_col(ws.S, PS, p, ws.cv, md.boff[l], md.boff[l + 1]); });
        const double ub = n * ex.sum(P, [&](int p) { return ws.dv[p]; });
        if (ub < md.tol * (1.0 - 1e-9)) {
            ex.one([&]() {
                st.scal[4] = ub; st.scal[3] = 0.0;
                if (iteration + 1 > md.max_iter && st.scal[1] == (double)ST_OK) st.scal[1] = (double)ST_NOT_CONVERGED;      // (weights.py:183-186)
            });
            return false;
        }
    }
    return true;
}

// After the loop: weights = a_new (already normalised, weights.py:130-132), scores = X a_new (no sign rule), the rest as metric.
template <class Ex>
PLSPM_HD void nm_finish(Ex& ex, const ModelDesc& md, Workspace& ws, NmState& st, const FitOutputs& out) {
    const int P = md.P, PS = ws.PS;
    ex.par(P, [&](int p) { ws.w[p] = st.a_new[p]; ws.mu[p] = st.mu[p]; ws.cs[p] = 1.0 / st.sd[p]; ws.sd[p] = sqrt(ws.S[p * PS + p]); });
    ex.one([&]() { ws.scal[1] = st.scal[0]; ws.scal[2] = 1.0 / st.scal[0]; ws.scal[3] = st.scal[1]; });
    finish_problem(ex, md, ws, out, (int)st.scal[2], false);
}

}  // namespace plspm
