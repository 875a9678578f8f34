// kernels_scores.h -- Device kernels, part 5a: scores (LDS-staged X tile . W).  Included by plspm_fit.hip only; not a stand-alone header.
#pragma once


// ------------------------------------------------------------------------------------------------ scores kernel
// scores[i][l] = sum_{p in block l} xa[i][p] * score_w[p] + score_c[l]   (weights.py:60, sign rule folded into score_w)
// HBM-bound stream: 8*N*(PA+L) bytes.  A workgroup takes TR-row tiles of Xa (TR*PA*8 contiguous bytes):
//   load    thread (row_sub = tid>>4, c = tid&15) issues ALL its 16-byte loads of the tile back to back -- rows row_sub, +16, ...,
//           chunks c, c+16, ... of the row -- so a wave reads four 256-byte row segments per instruction and every thread has
//           TR*PA/512 independent loads in flight; no index division anywhere
//   stage   the values go to LDS with row stride PA+1 doubles (odd: the 32 lanes of a ds_read_b64 group walk one column of 32
//           different rows conflict-free)
//   dot     thread (r = tid % TR, g = tid / TR) forms the per-block dot products of row r for LVs g, g + 256/TR, ...
//   store   the tile's TR*L scores are contiguous in memory: they leave through LDS as coalesced 8-byte stores
// Two workgroups per CU (LDS <= 64 KiB each) overlap one's loads with the other's LDS phase.
// Software pipeline: the loads of the workgroup's NEXT tile are issued (into registers) right after the current tile went to LDS,
// so they are in flight during the dot and store phases; NCH = 16-byte chunks per row per thread (compile time: the
// prefetch registers are a fixed array), 0 = any width without the prefetch.
template <int TR, int NCH>
__global__ void __launch_bounds__(256) scores_kernel(const double* __restrict__ Xa, long N, int PA, int P, int L, const int* __restrict__ boff,
                                                      const double* __restrict__ score_w, const double* __restrict__ score_c,
                                                      double* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int S = PA + 1;
    double* tile = reinterpret_cast<double*>(smem_raw);     // [TR][PA+1]
    double* wsh = tile + TR * S;                            // [P]
    double* csh = wsh + P;                                  // [L]
    double* osh = csh + L;                                  // [TR*L]
    int* bsh = reinterpret_cast<int*>(osh + TR * L);        // [L+1]
    const int tid = threadIdx.x;
    for (int p = tid; p < P; p += 256) wsh[p] = score_w[p];
    for (int l = tid; l < L; l += 256) csh[l] = score_c[l];
    for (int l = tid; l <= L; l += 256) bsh[l] = boff[l];
    const long ntiles = (N + TR - 1) / TR;
    const int half = PA >> 1;                               // 16-byte chunks per row
    const int row_sub = tid >> 4, c16 = tid & 15;
    constexpr int RPT = TR / 16;                            // rows per thread
    constexpr int LG = 256 / TR;                            // LV groups in the dot phase
    constexpr int NV = NCH > 0 ? NCH : 1;
    const int r_c = tid % TR, lg = tid / TR;
    double2 v[RPT][NV];
    auto issue = [&](long tl) {
        const long i0 = tl * TR;
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const long row = i0 + row_sub + 16 * rr;
            const double2* src = reinterpret_cast<const double2*>(Xa + (row < N ? row : N) * PA);      // row N: the all-zero pad row
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = c16 + 16 * j;
                if (c < half) { const dv2 t = __builtin_nontemporal_load(reinterpret_cast<const dv2*>(src) + c); v[rr][j] = double2{t.x, t.y}; }     // streamed once
                else v[rr][j] = double2{0.0, 0.0};
            }
        }
    };
    long tl = blockIdx.x;
    if (NCH > 0 && tl < ntiles) issue(tl);
    for (; tl < ntiles; tl += gridDim.x) {
        const long i0 = tl * TR;
        const int rows = (int)lmin(TR, N - i0);
        __syncthreads();                                    // previous tile's LDS reads are done (and the weights are staged)
        if (NCH > 0) {
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr) {
                double* dst = tile + (row_sub + 16 * rr) * S;
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int c = c16 + 16 * j;
                    if (c < half) { dst[2 * c] = v[rr][j].x; dst[2 * c + 1] = v[rr][j].y; }
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < RPT; ++rr) {
                const int r = row_sub + 16 * rr;
                if (r < rows) {
                    const double2* src = reinterpret_cast<const double2*>(Xa + (i0 + r) * PA);
                    double* dst = tile + r * S;
                    for (int c = c16; c < half; c += 16) { const double2 x = src[c]; dst[2 * c] = x.x; dst[2 * c + 1] = x.y; }
                }
            }
        }
        __syncthreads();
        if (NCH > 0 && tl + gridDim.x < ntiles) issue(tl + gridDim.x);
        if (r_c < rows) {
            const double* row = tile + r_c * S;
            for (int l = lg; l < L; l += LG) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int p = bsh[l];
                const int pe = bsh[l + 1];
                for (; p + 3 < pe; p += 4) {
                    s0 += row[p] * wsh[p]; s1 += row[p + 1] * wsh[p + 1];
                    s2 += row[p + 2] * wsh[p + 2]; s3 += row[p + 3] * wsh[p + 3];
                }
                for (; p < pe; ++p) s0 += row[p] * wsh[p];
                osh[r_c * L + l] = ((s0 + s1) + (s2 + s3)) + csh[l];
            }
        }
        __syncthreads();
        double* dst = scores + i0 * L;
        for (int e = tid; e < rows * L; e += 256) __builtin_nontemporal_store(osh[e], dst + e);
    }
}

// scores of the incomplete rows come from the solver state, not from the score map (plspm_fit)
__global__ void __launch_bounds__(64) patch_scores_kernel(double* __restrict__ scores, int L, const int* __restrict__ rowid, const double* __restrict__ Yn) {
    const long j = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += blockDim.x) scores[(long)rowid[j] * L + l] = Yn[j * L + l];
}
