// kernels_summary.h -- Device kernels, part 5b: bootstrap summaries (record transpose, per-column statistics).
// Included by plspm_bootstrap.hip only (wave_ops.h in front); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------ bootstrap summaries
// reference _create_summary (plspm/bootstrap.py:24-32): per result column mean, std (ddof 1), 2.5 % / 97.5 % quantiles with
// linear interpolation, t = original / std -- over the replicates whose status is OK.  One workgroup of 1,024 threads per column:
//   1. the column's OK values are compacted into `buf` (LDS when it fits, else a global scratch slice) in REPLICATE ORDER (ballot
//      prefix inside a wave, scan over the (chunk, wave) counts across waves), so every sum below has one fixed order: results are
//      bit-reproducible and independent of how the replicates were sharded.  Up to 8,192 values then live in REGISTERS (eight per
//      thread): every later pass runs over registers, not over LDS;
//   2. mean and variance by two fixed-order tree reductions;
//   3. the four order statistics the two quantiles interpolate between, WITHOUT sorting: an 8-bit-digit radix select on the
//      order-preserving integer image of the doubles, both quantiles in the same passes (a 256-bin LDS histogram per quantile and
//      pass, block-wide scan of the bins).  The leading bytes every value of the column shares (sign, exponent: the block-wide minimum
//      and maximum key agree on them) need no pass, a constant column none at all; the passes stop as soon as both selected bins hold
//      a single value.  Then one pass for the successor of each selected value.
//      (History, 5,000 replicates x 158 columns: bitonic sort of the padded column 0.37 ms per call; radix select with 256 threads over
//      LDS 117 us, with batched loads / run-length atomics / early exit 51 us (phase clocks: compaction 39 k, select 48 k, successors
//      16 k of 108 k -- every loop one LDS round trip per value, every value a cache line); this version 47 k clocks = 22 us behind a 3 us
//      transpose (compaction 10 k, mean + variance 4 k, select 25 k, successors 8 k: profiles/r03_solver_marks.txt).)
// `original` and `out` may live in pinned host memory (the handle's staging area): one read and six writes per column.
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
// records [B x stride] (row-major) -> cols [ncol x ld]: column c of the first ncol columns contiguous over the replicates.  Tiles of 64 x 64
// through LDS (pitch 65): 512-byte runs on both sides.
__global__ void __launch_bounds__(256) records_transpose_kernel(const double* __restrict__ rows, long B, int stride, int ncol, double* __restrict__ cols, long ld) {
    __shared__ double tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long b0 = (long)blockIdx.x * 64;
    const int c0 = (int)blockIdx.y * 64;
#pragma unroll 4
    for (int r = w; r < 64; r += 4) {
        const long b = b0 + r;
        if (b < B && c0 + lane < ncol) tile[r][lane] = rows[b * stride + c0 + lane];
    }
    __syncthreads();
#pragma unroll 4
    for (int cc = w; cc < 64; cc += 4) {
        if (c0 + cc < ncol && b0 + lane < B) cols[(long)(c0 + cc) * ld + b0 + lane] = tile[lane][cc];
    }
}
constexpr int SUM_NT = 1024, SUM_NW = SUM_NT / 64;      // threads / waves of a summary workgroup
constexpr int SUM_NB = 8;                               // chunks of 1,024 replicates whose loads are in flight together
constexpr int SUM_RI = 8;                               // values a thread keeps in registers
// block-wide sum in one fixed order: 64-lane butterfly (wave_ops.h), then the sixteen wave results pairwise in wave order
__device__ __forceinline__ double sum_block(double v, double* redw, int lane, int wave) {
    v = wv::allsum(v);
    if (lane == 0) redw[wave] = v;
    __syncthreads();
    double t[SUM_NW];
#pragma unroll
    for (int w = 0; w < SUM_NW; ++w) t[w] = redw[w];
#pragma unroll
    for (int n = SUM_NW; n > 1; n >>= 1)
#pragma unroll
        for (int i = 0; i < n / 2; ++i) t[i] = t[2 * i] + t[2 * i + 1];
    __syncthreads();
    return t[0];
}
#ifdef PLSPM_DEBUG_MARKS       // phase clocks of column 0 (the marks build only)
__device__ long long g_summary_marks[16];
#define PLSPM_SMARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_summary_marks[i] = clock64(); } while (0)
#else
#define PLSPM_SMARK(i) do { } while (0)
#endif
template <bool IN_LDS>
__global__ void __launch_bounds__(SUM_NT) summary_kernel(const double* __restrict__ cols, long cols_ld, long B, int R, const double* __restrict__ original,
                                                          double* __restrict__ gbuf, int npad, double* __restrict__ out, int* __restrict__ n_used) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    static_assert(SUM_NB * SUM_NW == 128, "the (chunk, wave) counts are scanned by exactly two waves");
    __shared__ double redw[SUM_NW];
    __shared__ unsigned hist[2][256];
    __shared__ int wcount[SUM_NB * SUM_NW];
    __shared__ int wtot[2];
    __shared__ unsigned wscan[2][4];
    __shared__ unsigned long long sel_prefix[2];
    __shared__ unsigned sel_rank[2], sel_cnt[2];
    __shared__ unsigned long long red_key[2][SUM_NW];
    __shared__ unsigned red_cnt[2][SUM_NW];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double orig = original[c];                    // (possibly a read across the host link: issued first, consumed last)
    double* buf = IN_LDS ? reinterpret_cast<double*>(smem_raw) : gbuf + (long)c * npad;
    PLSPM_SMARK(0);
    // 1. compaction in replicate order.  The status and the value of up to SUM_NB x 1,024 replicates are loaded before anything is
    //    consumed (`cols`: the records column-major, records_transpose_kernel -- read from the row-major records, every value is a 128-byte
    //    line of its own and the L1 fills of 2 x B lines per column bound this phase)
    int m = 0;
    for (long s0 = 0; s0 < B; s0 += (long)SUM_NT * SUM_NB) {
        double val[SUM_NB], st[SUM_NB];
#pragma unroll
        for (int k = 0; k < SUM_NB; ++k) {
            st[k] = 1.0; val[k] = 0.0;
            if (s0 + (long)SUM_NT * k < B) {                        // (uniform: chunks beyond the last replicate load nothing)
                const long b = s0 + (long)SUM_NT * k + tid;
                const long bc = (b < B) ? b : B - 1;
                st[k] = cols[(long)R * cols_ld + bc];
                val[k] = cols[(long)c * cols_ld + bc];
            }
        }
        unsigned long long bal[SUM_NB];
#pragma unroll
        for (int k = 0; k < SUM_NB; ++k) {
            const bool ok = (s0 + (long)SUM_NT * k + tid < B) && st[k] == 0.0;
            bal[k] = __ballot(ok);
            if (lane == 0) wcount[k * SUM_NW + wave] = __popcll(bal[k]);
        }
        __syncthreads();
        int excl = 0;
        if (tid < SUM_NB * SUM_NW) {                                // waves 0 and 1: exclusive scan of the 128 counts in (chunk, wave) order
            const int v = wcount[tid];
            const int incl = wv::inclusive_scan(v);
            excl = incl - v;
            if (lane == 63) wtot[wave] = incl;
        }
        __syncthreads();
        if (tid < SUM_NB * SUM_NW) wcount[tid] = m + excl + (wave == 1 ? wtot[0] : 0);
        const int total = wtot[0] + wtot[1];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SUM_NB; ++k) {
            const bool ok = (s0 + (long)SUM_NT * k + tid < B) && st[k] == 0.0;
            if (ok) buf[wcount[k * SUM_NW + wave] + __popcll(bal[k] & ((1ull << lane) - 1ull))] = val[k];
        }
        m += total;
        __syncthreads();
    }
    if (tid == 0 && c == 0) *n_used = m;
    // the values this thread owns in every later pass: t, t + 1,024, ... -- in registers when the column fits
    const bool cached = m <= SUM_NT * SUM_RI;
    double item[SUM_RI];
#pragma unroll
    for (int j = 0; j < SUM_RI; ++j) item[j] = (cached && tid + SUM_NT * j < m) ? buf[tid + SUM_NT * j] : 0.0;
    auto each = [&](auto f) {
        if (cached) {
#pragma unroll
            for (int j = 0; j < SUM_RI; ++j) if (tid + SUM_NT * j < m) f(item[j]);
        } else {
            for (int i = tid; i < m; i += SUM_NT) f(buf[i]);
        }
    };
    PLSPM_SMARK(1);
    // 2. mean, variance (fixed order: thread t takes elements t, t + 1,024, ...; shuffle tree + wave order)
    double s = 0.0;
    each([&](double x) { s += x; });
    const double tot = sum_block(s, redw, lane, wave);
    const double mean = (m > 0) ? tot / (double)m : 0.0;
    double v = 0.0;
    each([&](double x) { const double d = x - mean; v += d * d; });
    const double ssq = sum_block(v, redw, lane, wave);
    PLSPM_SMARK(2);
    // 3. order statistics lo_q = floor(q (m-1)) for q = 0.025, 0.975 (and their successors) by radix select
    double q_out[2] = {0.0, 0.0};
    if (m > 0) {
        const double pos[2] = {0.025 * (double)(m - 1), 0.975 * (double)(m - 1)};
        const int lo[2] = {(int)floor(pos[0]), (int)floor(pos[1])};
        // the leading bytes all keys share: minimum and maximum key of the column
        unsigned long long kmin = ~0ull, kmax = 0ull;
        each([&](double x) { const unsigned long long k = order_key(x); kmin = (k < kmin) ? k : kmin; kmax = (k > kmax) ? k : kmax; });
        kmin = wv::allmin(kmin); kmax = wv::allmax(kmax);
        if (lane == 0) { red_key[0][wave] = kmin; red_key[1][wave] = kmax; }
        __syncthreads();
        kmin = red_key[0][0]; kmax = red_key[1][0];
#pragma unroll
        for (int w = 1; w < SUM_NW; ++w) { const unsigned long long a = red_key[0][w], b = red_key[1][w]; kmin = (a < kmin) ? a : kmin; kmax = (b > kmax) ? b : kmax; }
        const unsigned long long diff = kmin ^ kmax;
        const int shared_bytes = diff ? (__clzll((long long)diff) >> 3) : 8;
        unsigned long long mask = shared_bytes ? (shared_bytes == 8 ? ~0ull : ~0ull << (64 - 8 * shared_bytes)) : 0ull;
        __syncthreads();                                           // (red_key is written again below)
        if (tid < 2) { sel_prefix[tid] = kmin & mask; sel_rank[tid] = (unsigned)lo[tid]; sel_cnt[tid] = (unsigned)m; }
        for (int shift = 56 - 8 * shared_bytes; shift >= 0; shift -= 8) {
            if (tid < 512) (&hist[0][0])[tid] = 0u;
            __syncthreads();
            if (sel_cnt[0] <= 1u && sel_cnt[1] <= 1u) break;       // (uniform: read behind the barrier)
            const unsigned long long p0 = sel_prefix[0], p1 = sel_prefix[1];
            // run-length aggregation per thread and per wave: values whose digits agree (heavy ties, a narrow column) would be 64-way
            // same-address LDS atomics; a thread adds a run of equal digits with one atomic, a wave that agrees on the bin adds once
            unsigned run_d[2] = {0u, 0u}, run_n[2] = {0u, 0u};
            each([&](double x) {
                const unsigned long long k = order_key(x);
                const unsigned d = (unsigned)(k >> shift) & 255u;
                const bool hit[2] = {(k & mask) == p0, (k & mask) == p1};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (hit[j]) {
                        if (run_n[j] && run_d[j] != d) { atomicAdd(&hist[j][run_d[j]], run_n[j]); run_n[j] = 0u; }
                        run_d[j] = d; ++run_n[j];
                    }
                }
            });
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned long long have = __ballot(run_n[j] != 0u);
                if (have) {                                        // (uniform)
                    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)run_d[j], __ffsll((long long)have) - 1);
                    if (__all(run_n[j] == 0u || run_d[j] == d0)) {
                        const unsigned n = wv::allsum(run_n[j]);
                        if (lane == 0) atomicAdd(&hist[j][d0], n);
                    } else if (run_n[j]) atomicAdd(&hist[j][run_d[j]], run_n[j]);
                }
            }
            __syncthreads();
            // inclusive scan of the 256 bins of both histograms: bin t is owned by thread t (waves 0 .. 3)
            unsigned mine[2] = {0u, 0u}, incl[2] = {0u, 0u}, rank[2] = {0u, 0u};
            if (tid < 256) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    mine[j] = hist[j][tid];
                    incl[j] = wv::inclusive_scan(mine[j]);
                    if (lane == 63) wscan[j][wave] = incl[j];
                }
            }
            __syncthreads();
            if (tid < 256) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    for (int w = 0; w < wave; ++w) incl[j] += wscan[j][w];
                    rank[j] = sel_rank[j];
                }
            }
            __syncthreads();                                       // everyone has read wscan / sel_rank
            if (tid < 256) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned excl = incl[j] - mine[j];
                    if (rank[j] >= excl && rank[j] < incl[j]) { sel_prefix[j] |= (unsigned long long)tid << shift; sel_rank[j] = rank[j] - excl; sel_cnt[j] = mine[j]; }
                }
            }
            mask |= 0xffull << shift;
            __syncthreads();
        }
        PLSPM_SMARK(3);
        // a bin with a single value: the remaining digits are that value's (found by the prefix); then the successor of each selected
        // value: the value itself when it is repeated past the rank, else the smallest larger one
        __syncthreads();
        if (mask != ~0ull) {
            const unsigned long long pj[2] = {sel_prefix[0], sel_prefix[1]};
            each([&](double x) {
                const unsigned long long k = order_key(x);
                if ((k & mask) == pj[0]) red_key[0][0] = k;        // (one writer, or equal values)
                if ((k & mask) == pj[1]) red_key[1][0] = k;
            });
            __syncthreads();
            if (tid < 2) sel_prefix[tid] = red_key[tid][0];
            __syncthreads();
        }
        unsigned cnt[2] = {0u, 0u};
        unsigned long long nxt[2] = {~0ull, ~0ull};
        const unsigned long long vk[2] = {sel_prefix[0], sel_prefix[1]};
        each([&](double x) {
            const unsigned long long k = order_key(x);
#pragma unroll
            for (int j = 0; j < 2; ++j) { if (k <= vk[j]) ++cnt[j]; else nxt[j] = (k < nxt[j]) ? k : nxt[j]; }
        });
        __syncthreads();                                           // (red_key was read above)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cnt[j] = wv::allsum(cnt[j]);
            nxt[j] = wv::allmin(nxt[j]);
            if (lane == 0) { red_cnt[j][wave] = cnt[j]; red_key[j][wave] = nxt[j]; }
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned total = 0u;
                unsigned long long nk = ~0ull;
                for (int w = 0; w < SUM_NW; ++w) { total += red_cnt[j][w]; nk = (red_key[j][w] < nk) ? red_key[j][w] : nk; }
                const double a = key_value(vk[j]);
                const bool has_next = lo[j] + 1 < m;
                const double b2 = !has_next ? a : (((int)total >= lo[j] + 2) ? a : key_value(nk));
                q_out[j] = lerp_numpy(a, b2, pos[j] - (double)lo[j]);
            }
        }
    }
    PLSPM_SMARK(4);
    if (tid == 0) {
        const double nan = __builtin_nan("");
        const double sd = (m > 1) ? sqrt(ssq / (double)(m - 1)) : nan;
        double* o = out + (long)c * 6;
        o[0] = orig;
        o[1] = (m > 0) ? mean : nan;
        o[2] = sd;
        o[3] = (m > 0) ? q_out[0] : nan;
        o[4] = (m > 0) ? q_out[1] : nan;
        o[5] = orig / sd;
    }
}

