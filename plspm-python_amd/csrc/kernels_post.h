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
//      (The first version bitonic-sorted the padded column: 91 block-wide steps for 5,000 replicates, 0.37 ms per call; radix select
//      with one chunk of 256 replicates per barrier pair in the compaction and one LDS atomic per value 0.14 ms per call, the kernel
//      itself 117 us; batched loads, run-length aggregated atomics, shared scans and the early exit: 68 us, 0.106 ms per call.)
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
// block-wide reductions of the summary kernel: 64-lane shuffle tree, then the four wave results through LDS in wave order (one fixed
// order -> bit-reproducible; two barriers instead of the eight of a 256-slot LDS tree)
__device__ __forceinline__ double sum256(double v, double* red4, int lane, int wave) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red4[wave] = v;
    __syncthreads();
    const double t = (red4[0] + red4[1]) + (red4[2] + red4[3]);
    __syncthreads();
    return t;
}
template <bool IN_LDS>
__global__ void __launch_bounds__(256) summary_kernel(const double* __restrict__ rows, long B, int stride, int R, const double* __restrict__ original,
                                                       double* __restrict__ gbuf, int npad, double* __restrict__ out, int* __restrict__ n_used) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NB = 16;                        // chunks of 256 replicates whose loads are in flight together
    __shared__ double red4[4];
    __shared__ unsigned hist[2][256];
    __shared__ int wcount[NB][4];
    __shared__ unsigned wscan[2][4];
    __shared__ unsigned long long sel_prefix[2];
    __shared__ unsigned sel_rank[2], sel_cnt[2];
    __shared__ unsigned long long red_key[2][4];
    __shared__ unsigned red_cnt[2][4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* buf = IN_LDS ? reinterpret_cast<double*>(smem_raw) : gbuf + (long)c * npad;
    // 1. compaction in replicate order.  The status and the value of NB x 256 replicates are loaded before anything is consumed
    //    (both are strided reads; one chunk of 256 per barrier pair made the pass a chain of 20 memory round trips at B = 5,000)
    int m = 0;
    for (long s0 = 0; s0 < B; s0 += 256L * NB) {
        double val[NB], st[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const long b = s0 + 256L * k + tid;
            const long bc = (b < B) ? b : B - 1;
            st[k] = rows[bc * stride + R];
            val[k] = rows[bc * stride + c];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const bool ok = (s0 + 256L * k + tid < B) && st[k] == 0.0;
            const unsigned long long bal = __ballot(ok);
            if (lane == 0) wcount[k][wave] = __popcll(bal);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const bool ok = (s0 + 256L * k + tid < B) && st[k] == 0.0;
            const unsigned long long bal = __ballot(ok);
            int base = m;
            for (int w = 0; w < wave; ++w) base += wcount[k][w];
            if (ok) buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = val[k];
            m += wcount[k][0] + wcount[k][1] + wcount[k][2] + wcount[k][3];
        }
        __syncthreads();
    }
    if (tid == 0 && c == 0) *n_used = m;
    // 2. mean, variance (fixed order: thread t takes elements t, t + 256, ...; shuffle tree + wave order)
    double s = 0.0;
    for (int i = tid; i < m; i += 256) s += buf[i];
    const double tot = sum256(s, red4, lane, wave);
    const double mean = (m > 0) ? tot / (double)m : 0.0;
    double v = 0.0;
    for (int i = tid; i < m; i += 256) { const double d = buf[i] - mean; v += d * d; }
    const double ssq = sum256(v, red4, lane, wave);
    // 3. order statistics lo_q = floor(q (m-1)) for q = 0.025, 0.975 (and their successors) by radix select: both quantiles in the same
    //    passes, the two bin scans share their barriers, and the passes stop as soon as both selected bins hold a single value (5,000
    //    distinct doubles separate after about four of the eight digits)
    double q_out[2] = {0.0, 0.0};
    if (m > 0) {
        const double pos[2] = {0.025 * (double)(m - 1), 0.975 * (double)(m - 1)};
        const int lo[2] = {(int)floor(pos[0]), (int)floor(pos[1])};
        if (tid < 2) { sel_prefix[tid] = 0ull; sel_rank[tid] = (unsigned)lo[tid]; sel_cnt[tid] = (unsigned)m; }
        unsigned long long mask = 0ull;
        for (int shift = 56; shift >= 0; shift -= 8) {
            hist[0][tid] = 0u; hist[1][tid] = 0u;
            __syncthreads();
            if (sel_cnt[0] <= 1u && sel_cnt[1] <= 1u) break;       // (uniform: read behind the barrier)
            const unsigned long long p0 = sel_prefix[0], p1 = sel_prefix[1];
            // run-length aggregation per thread: in the leading passes every value of a column has the same digit (sign, exponent) --
            // 2 x 5,000 increments of ONE bin are 64-way same-address LDS atomics, 12 k clocks per pass; a thread now adds a run of
            // equal digits with one atomic
            unsigned run_d[2] = {0u, 0u}, run_n[2] = {0u, 0u};
            for (int i = tid; i < m; i += 256) {
                const unsigned long long k = order_key(buf[i]);
                const unsigned d = (unsigned)(k >> shift) & 255u;
                const bool hit[2] = {(k & mask) == p0, (k & mask) == p1};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (hit[j]) {
                        if (run_n[j] && run_d[j] != d) { atomicAdd(&hist[j][run_d[j]], run_n[j]); run_n[j] = 0u; }
                        run_d[j] = d; ++run_n[j];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) if (run_n[j]) atomicAdd(&hist[j][run_d[j]], run_n[j]);
            __syncthreads();
            // inclusive scan of the 256 bins of both histograms: bin tid is owned by thread tid
            unsigned mine[2], incl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                mine[j] = hist[j][tid];
                incl[j] = mine[j];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const unsigned up = __shfl_up(incl[j], off, 64); if (lane >= off) incl[j] += up; }
                if (lane == 63) wscan[j][wave] = incl[j];
            }
            __syncthreads();
            unsigned rank[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                for (int w = 0; w < wave; ++w) incl[j] += wscan[j][w];
                rank[j] = sel_rank[j];
            }
            __syncthreads();                                       // everyone has read wscan / sel_rank
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned excl = incl[j] - mine[j];
                if (rank[j] >= excl && rank[j] < incl[j]) { sel_prefix[j] |= (unsigned long long)tid << shift; sel_rank[j] = rank[j] - excl; sel_cnt[j] = mine[j]; }
            }
            mask |= 0xffull << shift;
            __syncthreads();
        }
        // a bin with a single value: the remaining digits are that value's (found by the prefix); then the successor of each selected
        // value: the value itself when it is repeated past the rank, else the smallest larger one
        __syncthreads();
        if (mask != ~0ull) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned long long pj = sel_prefix[j];
                for (int i = tid; i < m; i += 256) {
                    const unsigned long long k = order_key(buf[i]);
                    if ((k & mask) == pj) red_key[j][0] = k;       // (one writer, or equal values)
                }
            }
            __syncthreads();
            if (tid < 2) sel_prefix[tid] = red_key[tid][0];
            __syncthreads();
        }
        unsigned cnt[2] = {0u, 0u};
        unsigned long long nxt[2] = {~0ull, ~0ull};
        const unsigned long long vk[2] = {sel_prefix[0], sel_prefix[1]};
        for (int i = tid; i < m; i += 256) {
            const unsigned long long k = order_key(buf[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) { if (k <= vk[j]) ++cnt[j]; else nxt[j] = (k < nxt[j]) ? k : nxt[j]; }
        }
        __syncthreads();                                           // (red_key was read above)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                cnt[j] += __shfl_down(cnt[j], off, 64);
                const unsigned long long o = __shfl_down(nxt[j], off, 64);
                nxt[j] = (o < nxt[j]) ? o : nxt[j];
            }
            if (lane == 0) { red_cnt[j][wave] = cnt[j]; red_key[j][wave] = nxt[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned total = red_cnt[j][0] + red_cnt[j][1] + red_cnt[j][2] + red_cnt[j][3];
            unsigned long long nk = red_key[j][0];
            for (int w = 1; w < 4; ++w) nk = (red_key[j][w] < nk) ? red_key[j][w] : nk;
            const double a = key_value(vk[j]);
            const bool has_next = lo[j] + 1 < m;
            const double b2 = !has_next ? a : (((int)total >= lo[j] + 2) ? a : key_value(nk));
            q_out[j] = lerp_numpy(a, b2, pos[j] - (double)lo[j]);
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
