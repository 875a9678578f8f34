// kernels_post.h -- Device kernels, part 5: scores (LDS-staged X tile . W) and the bootstrap summaries.
// Included by plspm_hip.hip (one translation unit); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------ scores kernel
// scores[i][l] = sum_{p in block l} xa[i][p] * score_w[p] + score_c[l]   (weights.py:60, sign rule folded into score_w)
// A 16-row tile of Xa (16*PA*8 contiguous bytes) is staged in LDS with coalesced 16-byte loads (row stride PA+1 doubles:
// conflict-free column walks); thread (row, l-group) forms the short per-block dot products; the tile's scores leave
// through LDS as one contiguous 16*L block.  Small tiles keep several workgroups per CU resident so that one
// workgroup's HBM loads overlap another's LDS phase (HBM-bound: 8*N*(PA+L) bytes).
__global__ void __launch_bounds__(256) scores_kernel(const double* __restrict__ Xa, long N, int PA, int P, int L, const int* __restrict__ boff,
                                                      const double* __restrict__ score_w, const double* __restrict__ score_c,
                                                      double* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);     // [16][PA+1]
    double* wsh = tile + SCORE_ROWS * (PA + 1);             // [P]
    double* osh = wsh + P;                                  // [16*L]
    int* bsh = reinterpret_cast<int*>(osh + SCORE_ROWS * L); // [L+1]
    const int tid = threadIdx.x;
    for (int p = tid; p < P; p += 256) wsh[p] = score_w[p];
    for (int l = tid; l <= L; l += 256) bsh[l] = boff[l];
    const long ntiles = (N + SCORE_ROWS - 1) / SCORE_ROWS;
    const int half = PA >> 1;
    const int r_c = tid & 15, lg = tid >> 4;
    for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long i0 = tl * SCORE_ROWS;
        const int rows = (int)lmin(SCORE_ROWS, N - i0);
        __syncthreads();
        const double2* src = reinterpret_cast<const double2*>(Xa + i0 * PA);
        const int n2 = rows * half;
        for (int e = tid; e < n2; e += 256) {
            const double2 v = src[e];
            const int r = e / half, c = 2 * (e - r * half);
            tile[r * (PA + 1) + c] = v.x;
            tile[r * (PA + 1) + c + 1] = v.y;
        }
        __syncthreads();
        for (int l = lg; l < L; l += 16) {
            const double* row = tile + r_c * (PA + 1);
            double s0 = 0.0, s1 = 0.0;
            int p = bsh[l];
            const int pe = bsh[l + 1];
            for (; p + 1 < pe; p += 2) { s0 += row[p] * wsh[p]; s1 += row[p + 1] * wsh[p + 1]; }
            if (p < pe) s0 += row[p] * wsh[p];
            osh[r_c * L + l] = (s0 + s1) + score_c[l];
        }
        __syncthreads();
        double* dst = scores + i0 * L;
        for (int e = tid; e < rows * L; e += 256) dst[e] = osh[e];
    }
}

// ------------------------------------------------------------------------------------------------ bootstrap summaries
// reference _create_summary (plspm/bootstrap.py:24-32): per result column mean, std (ddof 1), 2.5 % / 97.5 % quantiles with
// linear interpolation, t = original / std -- over the replicates whose status is OK.  One workgroup per column: gather the
// column into `buf` (LDS when it fits, else a global scratch slice), bitonic sort, tree reductions.
// out[c*6 + {0..5}] = original, mean, std.error, perc.025, perc.975, t stat.
__device__ __forceinline__ double quantile_linear(const double* sorted, int m, double q) {
    const double pos = q * (double)(m - 1);
    const int lo = (int)floor(pos);
    const int hi = (lo + 1 < m) ? lo + 1 : lo;
    const double t = pos - (double)lo, a = sorted[lo], b = sorted[hi], d = b - a;
    return (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;             // numpy's _lerp (monotone form)
}
template <bool IN_LDS>
__global__ void __launch_bounds__(256) summary_kernel(const double* __restrict__ rows, long B, int stride, int R, const double* __restrict__ original,
                                                       double* __restrict__ gbuf, int npad, double* __restrict__ out, int* __restrict__ n_used) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[256];
    __shared__ int cnt_s;
    const int c = blockIdx.x, tid = threadIdx.x;
    double* buf = IN_LDS ? reinterpret_cast<double*>(smem_raw) : gbuf + (long)c * npad;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    // gather the OK replicates' values (order is irrelevant: they get sorted)
    for (long b0 = 0; b0 < B; b0 += 256) {
        const long b = b0 + tid;
        const bool ok = (b < B) && rows[b * stride + R] == 0.0;
        const unsigned long long bal = __ballot(ok);
        __shared__ int wbase[4];
        if ((tid & 63) == 0) wbase[tid >> 6] = atomicAdd(&cnt_s, __popcll(bal));
        __syncthreads();
        if (ok) buf[wbase[tid >> 6] + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = rows[b * stride + c];
        __syncthreads();
    }
    const int m = cnt_s;
    if (tid == 0 && c == 0) *n_used = m;
    int n2 = 1;
    while (n2 < m) n2 <<= 1;
    for (int i = m + tid; i < n2; i += 256) buf[i] = 1.0e308 * 10.0;            // +inf padding sorts to the end
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double a = buf[i], b = buf[ixj];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { buf[i] = b; buf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    double s = 0.0;
    for (int i = tid; i < m; i += 256) s += buf[i];
    red[tid] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) red[tid] += red[tid + h]; __syncthreads(); }
    const double mean = (m > 0) ? red[0] / (double)m : 0.0;
    __syncthreads();
    double v = 0.0;
    for (int i = tid; i < m; i += 256) { const double d = buf[i] - mean; v += d * d; }
    red[tid] = v;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) red[tid] += red[tid + h]; __syncthreads(); }
    if (tid == 0) {
        const double nan = __builtin_nan("");
        const double sd = (m > 1) ? sqrt(red[0] / (double)(m - 1)) : nan;
        double* o = out + (long)c * 6;
        o[0] = original[c];
        o[1] = (m > 0) ? mean : nan;
        o[2] = sd;
        o[3] = (m > 0) ? quantile_linear(buf, m, 0.025) : nan;
        o[4] = (m > 0) ? quantile_linear(buf, m, 0.975) : nan;
        o[5] = original[c] / sd;
    }
}
