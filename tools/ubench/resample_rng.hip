// What does the counter-based generator cost inside the resample kernel?  A copy of resample_i8_kernel<false>'s three phases (zero the 16-bit
// LDS histogram, N draws -> LDS atomics, histogram -> int8 pieces in the fragment-major layout) at the headline shape (N = 10,000, 5,000
// replicates, 256 threads, 20 KB of LDS), with the generator as a template parameter:
//   0  Philox4x32-10 (csrc/philox.h)           1  Philox4x32-7
//   2  Threefry4x32-20 (Salmon et al. 2011)    3  Threefry4x32-16        4  Threefry4x32-12
//   5  no generator (the counter itself: the floor of everything else in the kernel)
// Philox is 32-bit multiplies (quarter rate on CDNA), Threefry is add / rotate / xor only (full rate, v_alignbit_b32 = one rotate).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/resample_rng.hip -o tools/ubench/resample_rng
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
struct u32x4 { uint32_t v[4]; };
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
template <int R>
__host__ __device__ __forceinline__ u32x4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}
// Philox4x32-10 with BOTH halves of a product from one v_mad_u64_u32 (the compiler's choice is v_mul_hi_u32 + v_mul_lo_u32) and the three-way xors as v_xor3_b32: the same bits
__device__ __forceinline__ u32x4 philox10_mad(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0, p1;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "v"(0xD2511F53u) : "vcc");
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "v"(0xCD9E8D57u) : "vcc");
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
// Threefry4x32-R: key (k0..k3), counter (c0..c3); rotation constants and key schedule of the Random123 definition (Skein's 4x32 variant)
template <int R>
__host__ __device__ __forceinline__ u32x4 threefry(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3) {
    constexpr int ROT[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
    const uint32_t ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1], x2 = c2 + ks[2], x3 = c3 + ks[3];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if ((r & 1) == 0) {
            x0 += x1; x1 = rotl32(x1, ROT[r & 7][0]); x1 ^= x0;
            x2 += x3; x3 = rotl32(x3, ROT[r & 7][1]); x3 ^= x2;
        } else {
            x0 += x3; x3 = rotl32(x3, ROT[r & 7][0]); x3 ^= x0;
            x2 += x1; x1 = rotl32(x1, ROT[r & 7][1]); x1 ^= x2;
        }
        if ((r & 3) == 3) {
            const int s = r / 4 + 1;
            x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + (uint32_t)s;
        }
    }
    return u32x4{{x0, x1, x2, x3}};
}
template <int GEN>
__host__ __device__ __forceinline__ u32x4 quad(uint64_t seed, uint64_t rep, uint32_t q) {
    if (GEN == 0) return philox<10>(q, 0u, (uint32_t)rep, (uint32_t)(rep >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    if (GEN == 1) return philox<7>(q, 0u, (uint32_t)rep, (uint32_t)(rep >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    if (GEN == 2) return threefry<20>(q, 0u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)rep, (uint32_t)(rep >> 32));
    if (GEN == 3) return threefry<16>(q, 0u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)rep, (uint32_t)(rep >> 32));
    if (GEN == 4) return threefry<12>(q, 0u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)rep, (uint32_t)(rep >> 32));
#if defined(__HIP_DEVICE_COMPILE__)
    if (GEN == 6) return philox10_mad(q, 0u, (uint32_t)rep, (uint32_t)(rep >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
    return u32x4{{q * 0x9E3779B9u, q * 0x85EBCA6Bu + (uint32_t)rep, q * 0xC2B2AE35u, (q + (uint32_t)rep) * 0x27D4EB2Fu}};
}

template <int GEN>
__global__ void __launch_bounds__(1024) resample_kernel(int N, int KB, int MT, uint64_t seed, int64_t rep0, uint4* __restrict__ Cd, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long b = blockIdx.x;
    const int nwords = KB * 32;
    for (int i = tid; i < nwords; i += nthr) hist[i] = 0u;
    __syncthreads();
    const uint64_t rep = (uint64_t)(rep0 + b);
    const int nq = (N + 3) >> 2;
    for (int q = tid; q < nq; q += nthr) {
        const u32x4 u = quad<GEN>(seed, rep, (uint32_t)q);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * q + j < N) { const unsigned w = mulhi32(u.v[j], (uint32_t)N); atomicAdd(&hist[w >> 1], (w & 1u) ? 0x10000u : 1u); }
    }
    __syncthreads();
    const int mt = (int)(b >> 4), r = (int)(b & 15);
    const uint4* h4 = reinterpret_cast<const uint4*>(hist);
    bool over = false;
    for (int c = tid; c < KB * 4; c += nthr) {
        const uint4 lo = h4[2 * c], hi = h4[2 * c + 1];
        const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned a = w[2 * k], bb = w[2 * k + 1];
            over |= ((a | bb) & 0xff80ff80u) != 0u;
            o[k] = (a & 0xffu) | ((a >> 8) & 0xff00u) | ((bb & 0xffu) << 16) | ((bb << 8) & 0xff000000u);
        }
        Cd[((long)(c >> 2) * MT + mt) * 64 + (c & 3) * 16 + r] = uint4{o[0], o[1], o[2], o[3]};
    }
    if (over) atomicOr(err, 2);
}

template <int GEN> void run(const char* name, int N, int B, int threads, uint4* cd, int* err) {
    const int KB = (N + 63) / 64, MT = (B + 15) / 16;
    const size_t lds = (size_t)KB * 32 * 4;
    CK(hipFuncSetAttribute((const void*)resample_kernel<GEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 30; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(resample_kernel<GEN>, dim3(B), dim3(threads), lds, 0, N, KB, MT, 1234ull, (int64_t)it * B, cd, err);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 5) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    // total of the counts of replicate 0: N whatever the generator
    std::vector<unsigned char> h((size_t)KB * MT * 1024);
    CK(hipMemcpy(h.data(), cd, h.size(), hipMemcpyDeviceToHost));
    long tot = 0, mx = 0;
    for (int kb = 0; kb < KB; ++kb) for (int g = 0; g < 4; ++g) for (int j = 0; j < 16; ++j) { const int v = h[(((size_t)kb * MT) * 64 + g * 16 + 0) * 16 + j]; tot += v; mx = std::max<long>(mx, v); }
    printf("{\"generator\": \"%s\", \"N\": %d, \"replicates\": %d, \"threads\": %d, \"us_min\": %.2f, \"us_median\": %.2f, \"draws_of_replicate_0\": %ld, \"max_count\": %ld}\n", name, N, B, threads,
           ms.front() * 1e3, ms[ms.size() / 2] * 1e3, tot, mx);
}

// ---- second question: what else does the kernel wait for?  RPW replicates per workgroup (the 16-byte pieces of RPW consecutive replicates are
// contiguous in the fragment-major block: RPW = 8 stores whole 128-byte lines), 8-bit counters (BYTES: the histogram already is the output
// layout; a count of 128 has P < 1e-200 on generated draws), SKIP bit 0: no stores, bit 1: no LDS atomics.
template <int GEN, int RPW, bool BYTES, int SKIP>
__global__ void __launch_bounds__(1024) resample_multi_kernel(int N, int KB, int MT, uint64_t seed, int64_t rep0, uint4* __restrict__ Cd, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    long b0 = (long)blockIdx.x * RPW;
    if (SKIP & 4) {      // the 16 replicates of a count tile on ONE XCD (workgroup id mod 8), consecutive there: their 16-byte pieces meet in that XCD's L2
        const long i = blockIdx.x, xcd = i & 7, j = i >> 3;
        constexpr int WPT = 16 / RPW;      // workgroups per tile
        b0 = (((j / WPT) * 8 + xcd) * WPT + (j % WPT)) * RPW;
        if (b0 >= (long)gridDim.x * RPW) return;
    }
    const int hstride = KB * (BYTES ? 16 : 32) + (RPW > 1 ? 4 : 0);      // words per histogram (+16 B: neighbours start in different banks)
    for (int i = tid; i < hstride * RPW; i += nthr) hist[i] = 0u;
    __syncthreads();
    const int nq = (N + 3) >> 2;
    unsigned keep = 0;
    for (int e = tid; e < nq * RPW; e += nthr) {
        const int j = e / nq, q = e - j * nq;
        const u32x4 u = quad<GEN>(seed, (uint64_t)(rep0 + b0 + j), (uint32_t)q);
        unsigned* h = hist + j * hstride;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * q + t < N) {
                const unsigned w = mulhi32(u.v[t], (uint32_t)N);
                if (SKIP & 2) keep ^= w;
                else if (BYTES) atomicAdd(&h[w >> 2], 1u << (8u * (w & 3u)));
                else atomicAdd(&h[w >> 1], (w & 1u) ? 0x10000u : 1u);
            }
    }
    __syncthreads();
    const int mt = (int)(b0 >> 4), r0 = (int)(b0 & 15);
    bool over = false;
    for (int e = tid; e < KB * 4 * RPW; e += nthr) {
        const int c = e / RPW, j = e - c * RPW;
        const uint4* h4 = reinterpret_cast<const uint4*>(hist + j * hstride);
        uint4 out;
        if (BYTES) {
            out = h4[c];
            over |= ((out.x | out.y | out.z | out.w) & 0x80808080u) != 0u;
        } else {
            const uint4 lo = h4[2 * c], hi = h4[2 * c + 1];
            const unsigned w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            unsigned o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned a = w[2 * k], bb = w[2 * k + 1];
                over |= ((a | bb) & 0xff80ff80u) != 0u;
                o[k] = (a & 0xffu) | ((a >> 8) & 0xff00u) | ((bb & 0xffu) << 16) | ((bb << 8) & 0xff000000u);
            }
            out = uint4{o[0], o[1], o[2], o[3]};
        }
        if (!(SKIP & 1) || (out.x == 0xdeadbeefu && keep == 77u)) {
            uint4* dst = &Cd[((long)(c >> 2) * MT + mt) * 64 + (c & 3) * 16 + r0 + j];
            if (SKIP & 8) { __builtin_nontemporal_store(out.x, &dst->x); __builtin_nontemporal_store(out.y, &dst->y); __builtin_nontemporal_store(out.z, &dst->z); __builtin_nontemporal_store(out.w, &dst->w); }
            else *dst = out;
        }
    }
    if (over || ((SKIP & 2) && keep == 0x12345u)) atomicOr(err, 2);
}
template <int GEN, int RPW, bool BYTES, int SKIP> void run_multi(int N, int B, int threads, uint4* cd, int* err) {
    const int KB = (N + 63) / 64, MT = (B + 15) / 16;
    const size_t lds = (size_t)(KB * (BYTES ? 16 : 32) + (RPW > 1 ? 4 : 0)) * RPW * 4;
    if (lds > 160 * 1024) return;
    CK(hipFuncSetAttribute((const void*)resample_multi_kernel<GEN, RPW, BYTES, SKIP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    CK(hipMemset(cd, 0, (size_t)KB * MT * 1024));
    for (int it = 0; it < 30; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((resample_multi_kernel<GEN, RPW, BYTES, SKIP>), dim3(B / RPW), dim3(threads), lds, 0, N, KB, MT, 1234ull, (int64_t)it * B, cd, err);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 5) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<unsigned char> h((size_t)KB * MT * 1024);
    CK(hipMemcpy(h.data(), cd, h.size(), hipMemcpyDeviceToHost));
    long tot = 0;
    for (int kb = 0; kb < KB; ++kb) for (int g = 0; g < 4; ++g) for (int j = 0; j < 16; ++j) tot += h[(((size_t)kb * MT) * 64 + g * 16 + 3) * 16 + j];
    printf("{\"generator\": %d, \"replicates_per_workgroup\": %d, \"byte_counters\": %s, \"skip\": %d, \"N\": %d, \"replicates\": %d, \"threads\": %d, \"lds_kb\": %.1f, \"us_min\": %.2f, \"us_median\": %.2f, \"draws_of_replicate_3\": %ld}\n",
           GEN, RPW, BYTES ? "true" : "false", SKIP, N, B, threads, lds / 1024.0, ms.front() * 1e3, ms[ms.size() / 2] * 1e3, tot);
}
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000, B = argc > 2 ? atoi(argv[2]) : 5000, threads = argc > 3 ? atoi(argv[3]) : 256;
    const int KB = (N + 63) / 64, MT = (B + 15) / 16;
    uint4* cd; int* err;
    CK(hipMalloc(&cd, (size_t)KB * MT * 1024)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    CK(hipMemset(cd, 0, (size_t)KB * MT * 1024));
    // known-answer check of the Threefry restatement: Random123's kat_vectors entry "threefry4x32 20", counter = key = 0
    const u32x4 kat = threefry<20>(0, 0, 0, 0, 0, 0, 0, 0);
    printf("{\"threefry4x32_20_zero_kat\": \"%08x %08x %08x %08x\", \"expected\": \"9c6ca96a e17eae66 fc10ecd4 5256a7d8\"}\n", kat.v[0], kat.v[1], kat.v[2], kat.v[3]);
    if (N <= 65536) {
    run<0>("philox4x32-10", N, B, threads, cd, err);
    run<1>("philox4x32-7", N, B, threads, cd, err);
    run<2>("threefry4x32-20", N, B, threads, cd, err);
    run<3>("threefry4x32-16", N, B, threads, cd, err);
    run<4>("threefry4x32-12", N, B, threads, cd, err);
    run<5>("none", N, B, threads, cd, err);
    run<6>("philox4x32-10 (v_mad_u64_u32)", N, B, threads, cd, err);
    }
    if (N > 20000) {      // round 5: data sets beyond one 16-bit window (byte histogram, one workgroup of 1,024 threads per CU): the generator's share there
        for (int rnd = 0; rnd < 2; ++rnd) {
            run_multi<0, 1, true, 0>(N, B, 1024, cd, err); run_multi<6, 1, true, 0>(N, B, 1024, cd, err); run_multi<1, 1, true, 0>(N, B, 1024, cd, err); run_multi<5, 1, true, 0>(N, B, 1024, cd, err);
            run_multi<0, 1, true, 2>(N, B, 1024, cd, err); run_multi<6, 1, true, 2>(N, B, 1024, cd, err); run_multi<5, 1, true, 2>(N, B, 1024, cd, err); run_multi<0, 1, true, 1>(N, B, 1024, cd, err);
        }
        return 0;
    }
    for (int rnd = 0; rnd < 3; ++rnd) {
        run_multi<0, 1, false, 0>(N, B, 256, cd, err); run_multi<0, 1, false, 4>(N, B, 256, cd, err); run_multi<0, 1, false, 8>(N, B, 256, cd, err); run_multi<0, 1, false, 12>(N, B, 256, cd, err);
        run_multi<0, 1, true, 0>(N, B, 256, cd, err); run_multi<0, 1, true, 4>(N, B, 256, cd, err); run_multi<0, 1, true, 12>(N, B, 256, cd, err);
        run_multi<0, 2, true, 4>(N, B, 256, cd, err); run_multi<0, 4, true, 4>(N, B, 512, cd, err); run_multi<0, 4, true, 4>(N, B, 1024, cd, err);
        run_multi<1, 1, true, 4>(N, B, 256, cd, err); run_multi<5, 1, true, 4>(N, B, 256, cd, err); run_multi<5, 1, true, 5>(N, B, 256, cd, err);
    }
    for (int th : {64, 128}) { run_multi<0, 1, true, 0>(N, B, th, cd, err); run_multi<0, 1, false, 0>(N, B, th, cd, err); run_multi<5, 1, true, 0>(N, B, th, cd, err); run_multi<1, 1, true, 0>(N, B, th, cd, err); }
    if (argc > 4) return 0;
    for (int th : {256, 512, 1024}) {
        run_multi<0, 1, false, 0>(N, B, th, cd, err); run_multi<0, 1, true, 0>(N, B, th, cd, err);
        run_multi<0, 2, false, 0>(N, B, th, cd, err); run_multi<0, 2, true, 0>(N, B, th, cd, err);
        run_multi<0, 4, false, 0>(N, B, th, cd, err); run_multi<0, 4, true, 0>(N, B, th, cd, err);
        run_multi<0, 8, false, 0>(N, B, th, cd, err); run_multi<0, 8, true, 0>(N, B, th, cd, err);
    }
    run_multi<0, 1, false, 1>(N, B, 256, cd, err); run_multi<0, 1, false, 2>(N, B, 256, cd, err); run_multi<0, 1, false, 3>(N, B, 256, cd, err);
    run_multi<5, 1, false, 1>(N, B, 256, cd, err); run_multi<5, 1, false, 2>(N, B, 256, cd, err); run_multi<5, 1, false, 3>(N, B, 256, cd, err);
    run_multi<0, 8, true, 1>(N, B, 512, cd, err); run_multi<0, 8, true, 2>(N, B, 512, cd, err); run_multi<0, 8, true, 3>(N, B, 512, cd, err);
    return 0;
}
