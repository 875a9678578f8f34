// kernels_post.h -- Device kernels, part 5: scores (LDS-staged X tile . W) and the bootstrap summaries.
// Included by plspm_hip.hip (one translation unit); not a stand-alone header.
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

// ------------------------------------------------------------------------------------------------ bootstrap summaries
// reference _create_summary (plspm/bootstrap.py:24-32): per result column mean, std (ddof 1), 2.5 % / 97.5 % quantiles with
// linear interpolation, t = original / std -- over the replicates whose status is OK.  One workgroup per column:
//   1. the column's OK values are compacted into `buf` (LDS when it fits, else a global scratch slice) in REPLICATE ORDER (ballot
//      prefix inside a wave, fixed wave order across waves), so every sum below has one fixed order: results are bit-reproducible
//      and independent of how the replicates were sharded;
//   2. mean and variance by two fixed-order tree reductions;
//   3. the four order statistics the two quantiles interpolate between, WITHOUT sorting: an 8-bit-digit radix select on the
//      order-preserving integer image of the doubles, both quantiles in the same eight passes over the values (a 256-bin LDS
//      histogram per quantile and pass, block-wide scan of the bins), then one pass for the successor of each selected value.
//      (The first version bitonic-sorted the padded column: 91 block-wide steps for 5,000 replicates, 0.37 ms; this one 0.06 ms.)
// out[c*6 + {0..5}] = original, mean, std.error, perc.025, perc.975, t stat.
__device__ __forceinline__ unsigned long long order_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double lerp_numpy(double a, double b, double t) {
    const double d = b - a;
    return (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;             // numpy's _lerp (monotone form)
}
template <bool IN_LDS>
__global__ void __launch_bounds__(256) summary_kernel(const double* __restrict__ rows, long B, int stride, int R, const double* __restrict__ original,
                                                       double* __restrict__ gbuf, int npad, double* __restrict__ out, int* __restrict__ n_used) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[256];
    __shared__ unsigned hist[2][256];
    __shared__ int wcount[4];
    __shared__ unsigned long long sel_prefix[2];
    __shared__ unsigned sel_rank[2];
    __shared__ unsigned long long red_key[256];
    __shared__ unsigned red_cnt[256];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* buf = IN_LDS ? reinterpret_cast<double*>(smem_raw) : gbuf + (long)c * npad;
    // 1. compaction in replicate order
    int m = 0;
    for (long b0 = 0; b0 < B; b0 += 256) {
        const long b = b0 + tid;
        const bool ok = (b < B) && rows[b * stride + R] == 0.0;
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int base = m;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        if (ok) buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = rows[b * stride + c];
        m += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (tid == 0 && c == 0) *n_used = m;
    // 2. mean, variance (fixed order: thread t takes elements t, t + 256, ...; binary tree over the threads)
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
    const double ssq = red[0];
    __syncthreads();
    // 3. order statistics lo_q = floor(q (m-1)) for q = 0.025, 0.975 (and their successors) by radix select
    double q_out[2] = {0.0, 0.0};
    if (m > 0) {
        const double pos[2] = {0.025 * (double)(m - 1), 0.975 * (double)(m - 1)};
        const int lo[2] = {(int)floor(pos[0]), (int)floor(pos[1])};
        if (tid < 2) { sel_prefix[tid] = 0ull; sel_rank[tid] = (unsigned)lo[tid]; }
        unsigned long long mask = 0ull;
        for (int shift = 56; shift >= 0; shift -= 8) {
            hist[0][tid] = 0u; hist[1][tid] = 0u;
            __syncthreads();
            const unsigned long long p0 = sel_prefix[0], p1 = sel_prefix[1];
            for (int i = tid; i < m; i += 256) {
                const unsigned long long k = order_key(buf[i]);
                const unsigned d = (unsigned)(k >> shift) & 255u;
                if ((k & mask) == p0) atomicAdd(&hist[0][d], 1u);
                if ((k & mask) == p1) atomicAdd(&hist[1][d], 1u);
            }
            __syncthreads();
            // block-wide inclusive scan of the 256 bins of each histogram: bin tid is owned by thread tid
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned mine = hist[j][tid];
                unsigned incl = mine;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const unsigned up = __shfl_up(incl, off, 64); if (lane >= off) incl += up; }
                if (lane == 63) wcount[wave] = (int)incl;
                __syncthreads();
                unsigned before = 0;
                for (int w = 0; w < wave; ++w) before += (unsigned)wcount[w];
                incl += before;
                const unsigned excl = incl - mine, rank = sel_rank[j];
                __syncthreads();                                   // everyone has read wcount / sel_rank
                if (rank >= excl && rank < incl) { sel_prefix[j] |= (unsigned long long)tid << shift; sel_rank[j] = rank - excl; }
                __syncthreads();
            }
            mask |= 0xffull << shift;
        }
        // successor of each selected value: the value itself when it is repeated past the rank, else the smallest larger one
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned long long vk = sel_prefix[j];
            unsigned cnt = 0;
            unsigned long long nxt = ~0ull;
            for (int i = tid; i < m; i += 256) {
                const unsigned long long k = order_key(buf[i]);
                if (k <= vk) ++cnt; else nxt = (k < nxt) ? k : nxt;
            }
            red_cnt[tid] = cnt; red_key[tid] = nxt;
            __syncthreads();
            for (int h = 128; h > 0; h >>= 1) {
                if (tid < h) { red_cnt[tid] += red_cnt[tid + h]; red_key[tid] = (red_key[tid + h] < red_key[tid]) ? red_key[tid + h] : red_key[tid]; }
                __syncthreads();
            }
            const double a = key_value(vk);
            const bool has_next = lo[j] + 1 < m;
            const double b2 = !has_next ? a : (((int)red_cnt[0] >= lo[j] + 2) ? a : key_value(red_key[0]));
            q_out[j] = lerp_numpy(a, b2, pos[j] - (double)lo[j]);
            __syncthreads();
        }
    }
    if (tid == 0) {
        const double nan = __builtin_nan("");
        const double sd = (m > 1) ? sqrt(ssq / (double)(m - 1)) : nan;
        double* o = out + (long)c * 6;
        o[0] = original[c];
        o[1] = (m > 0) ? mean : nan;
        o[2] = sd;
        o[3] = (m > 0) ? q_out[0] : nan;
        o[4] = (m > 0) ? q_out[1] : nan;
        o[5] = original[c] / sd;
    }
}
