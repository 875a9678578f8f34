// solver_ops.h -- the reference's operator plug-ins on second moments.
//
//   Scheme.X.value.calculate(path, y) -> E                       plspm/scheme.py:27-28 (centroid), 36-37 (factorial), 45-54 (path)
//   Mode.X.value.outer_weights_metric(data, Z, lv, mvs) -> w     plspm/mode.py:28-29 (A), 50-52 (B)
//
// Both consume an N x cols matrix only through its moment matrix, which the MFMA Gram kernels produce for the SHIFTED columns
// x' = x - mu (mu = the upload's column means, md.shift) with a ones column: M = sum_i [x'_i, 1][x'_i, 1]^T in the tile-packed
// layout.  With s_p = M(p, ones) and n = M(ones, ones):
//     sum_i x_p x_q = M_pq + mu_q s_p + mu_p s_q + n mu_p mu_q            (raw second moment)
//     cov0(x_p, x_q) = (M_pq - s_p s_q / n) / n                            (population covariance)
// Same execution model as solver_core.h (one cooperating group; host/device portable so that the CPU emulation can run it).
#pragma once
#include "solver_core.h"

namespace plspm {

PLSPM_HD double op_raw_moment(const double* Mp, int T, int ones, const double* mu, double n, int p, int q) {
    const double sp = Mp[packed_index(T, p, ones)], sq = Mp[packed_index(T, q, ones)];
    return Mp[packed_index(T, p, q)] + mu[q] * sp + mu[p] * sq + n * mu[p] * mu[q];
}

// Inner weights of a scheme for the score matrix y (N x L; here: the L uploaded columns of a handle whose "blocks" are the single
// score columns, so that its descriptors carry the path matrix and the predecessor / successor lists).  E_out [L*L] row-major:
// exactly the array the reference operator returns.
template <class Ex>
PLSPM_HD void op_inner_weights(Ex& ex, const ModelDesc& md, Workspace& ws, const double* Mp, double* E_out) {
    const int L = md.L, T = md.T;
    const double n = Mp[packed_index(T, L, L)];
    ex.par(L * L, [&](int e) {
        const int i = e / L, j = e - i * L;
        const double si = Mp[packed_index(T, i, L)], sj = Mp[packed_index(T, j, L)];
        ws.G[e] = (Mp[packed_index(T, i, j)] - si * sj / n) / n;                      // what corrcoef / cov see (scheme.py:28,37,53)
        ws.Q[e] = op_raw_moment(Mp, T, L, md.shift, n, i, j) / n;                      // what the no-intercept OLS sees (scheme.py:50)
    });
    ex.one([&]() { ws.scal[3] = (double)ST_OK; });
    inner_weights(ex, md, ws, n / (n - 1.0), ws.Q);
    ex.par(L * L, [&](int e) { E_out[e] = ws.E[e]; });
}

// Outer weights of one block: uploaded columns 0 .. k-1 = the block's MVs, column k = the inner estimate z.
//   Mode A  w = X' z / N                               (mode.py:29)
//   Mode B  w = argmin |X w - z|, minimum norm         (mode.py:51, scipy.linalg.lstsq)
// A, F, V: k x k scratch each (Mode B only).  Returns false when the Mode-B solve did not converge.
template <class Ex>
PLSPM_HD bool op_outer_weights(Ex& ex, int mode, int k, int T, const double* Mp, const double* mu, double* A, double* F, double* V, double* w_out, double* flag) {
    const int ones = k + 1;
    const double n = Mp[packed_index(T, ones, ones)];
    ex.par(k, [&](int j) { w_out[j] = op_raw_moment(Mp, T, ones, mu, n, j, k) / (mode == MODE_A ? n : 1.0); });
    ex.one([&]() { *flag = 1.0; });
    if (mode == MODE_A) return true;
    ex.par(k * k, [&](int e) { const int r = e / k, c = e - r * k; A[e] = op_raw_moment(Mp, T, ones, mu, n, r, c); });
    ex.one([&]() { if (!psd_solve_once(A, k, F, V, w_out)) *flag = 0.0; });
    return *flag != 0.0;
}

}  // namespace plspm
