// kernels_resample.h -- Device kernels, part 1b: resample + ordered compaction into (row, count) lists (fp64 Gram route, stop-rule passes).
// Included by plspm_bootstrap.hip only (philox.h in front); not a stand-alone header.
#pragma once

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

