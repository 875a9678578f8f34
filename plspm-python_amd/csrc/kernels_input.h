// kernels_input.h -- Device kernels, part 1: Philox4x32-10 resampling streams, upload (column means, pack) and the resample + compaction kernels.
// Included by plspm_hip.hip (one translation unit); not a stand-alone header.
#pragma once

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct u32x4 { uint32_t v[4]; };
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
// Resample index i (0 <= i < N) of replicate `rep`: word (i & 3) of Philox(counter = (i >> 2, 0, rep), key = seed),
// mapped to [0, N) by the 32x32 -> high-word multiply (bias <= N / 2^32).
__host__ __device__ __forceinline__ u32x4 resample_quad(uint64_t seed, uint64_t rep, uint32_t q) {
    return philox4x32_10(q, 0u, (uint32_t)rep, (uint32_t)(rep >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
}
__host__ __device__ __forceinline__ int32_t to_index(uint32_t u, uint32_t n) { return (int32_t)mulhi32(u, n); }

// ------------------------------------------------------------------------------------------------ upload kernels
// Column sums, row-major source: block = 64 columns x 4 row lanes; partial[blockIdx.x][p].
__global__ void __launch_bounds__(256) colsum_rowmajor_kernel(const double* __restrict__ X, long N, int src_cols, const int* __restrict__ colidx,
                                                               int P, double* __restrict__ partial) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = lmin(N, r0 + rows_per_block);
    for (int pbase = 0; pbase < P; pbase += 64) {
        const int p = pbase + tx;
        double s = 0.0;
        if (p < P) {
            const int c = colidx[p];
            for (long i = r0 + ty; i < r1; i += 4) s += X[i * src_cols + c];
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && p < P) partial[(long)blockIdx.x * P + p] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        __syncthreads();
    }
}
// Column sums, column-major source: grid (chunks, P); threads run along the rows.
__global__ void __launch_bounds__(256) colsum_colmajor_kernel(const double* __restrict__ X, long N, const int* __restrict__ colidx, int P,
                                                               double* __restrict__ partial) {
    __shared__ double red[256];
    const int p = blockIdx.y;
    const double* col = X + (long)colidx[p] * N;
    const long rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = lmin(N, r0 + rows_per_block);
    double s = 0.0;
    for (long i = r0 + threadIdx.x; i < r1; i += 256) s += col[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h]; __syncthreads(); }
    if (threadIdx.x == 0) partial[(long)blockIdx.x * P + p] = red[0];
}
__global__ void colmean_kernel(const double* __restrict__ partial, int nblk, int P, long N, double* __restrict__ shift) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(long)b * P + p];
    shift[p] = s / (double)N;
}
// Xa[i][p] = X[i][colidx[p]] - shift[p] (p < P), 1 (p == P), 0 (p > P).  Row-major source: one thread per output element.
__global__ void __launch_bounds__(256) pack_rowmajor_kernel(const double* __restrict__ X, long N, int src_cols, const int* __restrict__ colidx,
                                                             int P, int PA, const double* __restrict__ shift, double* __restrict__ Xa) {
    const long total = N * PA;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e / PA;
        const int p = (int)(e - i * PA);
        double v = 0.0;
        if (p < P) v = X[i * src_cols + colidx[p]] - shift[p];
        else if (p == P) v = 1.0;
        Xa[e] = v;
    }
}
// Column-major source: 64-row x 32-column LDS transpose tile (reads run along rows, writes along columns).
__global__ void __launch_bounds__(256) pack_colmajor_kernel(const double* __restrict__ X, long N, const int* __restrict__ colidx, int P, int PA,
                                                             const double* __restrict__ shift, double* __restrict__ Xa) {
    __shared__ double tile[32][65];
    const long i0 = (long)blockIdx.x * 64;
    const int p0 = blockIdx.y * 32;
    {
        const int r = threadIdx.x & 63;
        for (int c = threadIdx.x >> 6; c < 32; c += 4) {
            const int p = p0 + c;
            const long i = i0 + r;
            double v = 0.0;
            if (i < N) {
                if (p < P) v = X[(long)colidx[p] * N + i] - shift[p];
                else if (p == P) v = 1.0;
            }
            tile[c][r] = v;
        }
    }
    __syncthreads();
    {
        const int c = threadIdx.x & 31;
        for (int r = threadIdx.x >> 5; r < 64; r += 8) {
            const long i = i0 + r;
            if (i < N && p0 + c < PA) Xa[i * PA + p0 + c] = tile[c][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ resample + compact
// One workgroup per replicate: LDS histogram of the N drawn row indices, then an ordered compaction into
// (row, multiplicity) pairs -- ~63 % of the rows survive, so the Gram kernel issues 37 % fewer MFMAs than a
// gather of all N draws.  The list is zero-padded to a multiple of 4 entries (one MFMA k-group).
// `dcnt` (optional): the histogram itself as [replicate][dcnt_stride] uint16, zero-padded -- the dense stop-rule pass of the
// non-metric solvers reads it (nm_conv_dense_kernel); N <= 65535 here, so a count always fits.
__global__ void __launch_bounds__(256) resample_kernel(int N, const int* __restrict__ idx, uint64_t seed, int64_t rep0, int2* __restrict__ ent,
                                                        int* __restrict__ nent, long ent_stride, int* __restrict__ err, unsigned short* __restrict__ dcnt,
                                                        long dcnt_stride) {
    // 16-bit counters, two per LDS word (a count never exceeds N <= 65535 on this path): half the LDS of 32-bit counters, i.e.
    // twice the resident workgroups.  Row r lives in the low (even r) or high (odd r) half of word r / 2.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    const int nwords = (N + 1) >> 1;
    for (int i = tid; i < nwords; i += 256) hist[i] = 0u;
    __syncthreads();
    if (idx) {
        const int* my = idx + b * (long)N;
        for (int i = tid; i < N; i += 256) {
            const int r = my[i];
            if ((unsigned)r < (unsigned)N) atomicAdd(&hist[r >> 1], (r & 1) ? 0x10000u : 1u);
            else atomicOr(err, 1);
        }
    } else {
        const uint64_t rep = (uint64_t)(rep0 + b);
        const int nq = (N + 3) >> 2;
        for (int q = tid; q < nq; q += 256) {
            const u32x4 u = resample_quad(seed, rep, (uint32_t)q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < N) { const unsigned r = to_index(u.v[j], (uint32_t)N); atomicAdd(&hist[r >> 1], (r & 1u) ? 0x10000u : 1u); }
        }
    }
    __syncthreads();
    if (dcnt) {                                                  // the packed words ARE the little-endian uint16 row of the dense histogram
        unsigned* mine_cnt = reinterpret_cast<unsigned*>(dcnt + b * dcnt_stride);
        for (int i = tid; i < (int)(dcnt_stride >> 1); i += 256) mine_cnt[i] = (i < nwords) ? hist[i] : 0u;
    }
    auto count_of = [&](int row) -> int { return (int)((hist[row >> 1] >> ((row & 1) << 4)) & 0xffffu); };
    // ordered compaction with two barriers: wave w owns the contiguous row range [w*Q, (w+1)*Q); pass 1 counts its
    // non-empty rows, pass 2 writes them behind the preceding waves' totals (ballot + popcount prefix inside a wave).
    int2* my_ent = ent + b * ent_stride;
    const int Q = (((N + 3) >> 2) + 63) & ~63;
    const int r0 = wave * Q, r1 = min(N, r0 + Q);
    int mine = 0;
    for (int c0 = r0; c0 < r1; c0 += 64) {
        const int row = c0 + lane;
        const int cnt = (row < r1) ? count_of(row) : 0;
        mine += __popcll(__ballot(cnt > 0));
    }
    if (lane == 0) wave_tot[wave] = mine;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    for (int c0 = r0; c0 < r1; c0 += 64) {
        const int row = c0 + lane;
        const int cnt = (row < r1) ? count_of(row) : 0;
        const unsigned long long bal = __ballot(cnt > 0);
        if (cnt > 0) my_ent[off + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(row, cnt);
        off += __popcll(bal);
    }
    const int padded = (total + 3) & ~3;
    if (tid < padded - total) my_ent[total + tid] = make_int2(0, 0);
    if (tid == 0) nent[b] = total;
}

// Large-N variant (N * 4 bytes no longer fits LDS): the histogram lives in a per-replicate slice of a global scratch
// buffer.  Counting uses L2 atomics; the compaction passes read the slice with agent-scope relaxed loads, which bypass
// this CU's L1 (the zero-fill went through L1, the atomics did not).
__global__ void __launch_bounds__(256) resample_global_kernel(int N, const int* __restrict__ idx, uint64_t seed, int64_t rep0, unsigned* __restrict__ ghist,
                                                               int2* __restrict__ ent, int* __restrict__ nent, long ent_stride, int* __restrict__ err) {
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long b = blockIdx.x;
    unsigned* hist = ghist + b * (long)N;
    for (int i = tid; i < N; i += 256) __hip_atomic_store(&hist[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (idx) {
        const int* my = idx + b * (long)N;
        for (int i = tid; i < N; i += 256) {
            const int r = my[i];
            if ((unsigned)r < (unsigned)N) __hip_atomic_fetch_add(&hist[r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else atomicOr(err, 1);
        }
    } else {
        const uint64_t rep = (uint64_t)(rep0 + b);
        const int nq = (N + 3) >> 2;
        for (int q = tid; q < nq; q += 256) {
            const u32x4 u = resample_quad(seed, rep, (uint32_t)q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < N) __hip_atomic_fetch_add(&hist[to_index(u.v[j], (uint32_t)N)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    int2* my_ent = ent + b * ent_stride;
    const int Q = (((N + 3) >> 2) + 63) & ~63;
    const int r0 = wave * Q, r1 = min(N, r0 + Q);
    int mine = 0;
    for (int c0 = r0; c0 < r1; c0 += 64) {
        const int row = c0 + lane;
        const int cnt = (row < r1) ? (int)__hip_atomic_load(&hist[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        mine += __popcll(__ballot(cnt > 0));
    }
    if (lane == 0) wave_tot[wave] = mine;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    for (int c0 = r0; c0 < r1; c0 += 64) {
        const int row = c0 + lane;
        const int cnt = (row < r1) ? (int)__hip_atomic_load(&hist[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const unsigned long long bal = __ballot(cnt > 0);
        if (cnt > 0) my_ent[off + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(row, cnt);
        off += __popcll(bal);
    }
    const int padded = (total + 3) & ~3;
    if (tid < padded - total) my_ent[total + tid] = make_int2(0, 0);
    if (tid == 0) nent[b] = total;
}
