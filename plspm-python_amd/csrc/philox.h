// philox.h -- Philox4x32-10 resampling streams (host + device inline functions; every translation unit may include it).
#pragma once
#include <cstdint>

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct u32x4 { uint32_t v[4]; };
__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
// (device, round 5: both halves of a product from ONE v_mad_u64_u32 -- hipcc emits v_mul_hi_u32 + v_mul_lo_u32 for the two expressions below, 38 multiplies
//  per four draws where 19 do; same bits.  tools/ubench/resample_rng.hip: resample kernel 52.1 -> 46.7 us at 10,000 rows, 706 -> 625 us at 100,000)
__host__ __device__ __forceinline__ void mul_hilo32(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long p;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "v"(b) : "vcc");
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
#else
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
#endif
}
__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mul_hilo32(0xD2511F53u, c0, hi0, lo0);
        mul_hilo32(0xCD9E8D57u, c2, hi1, lo1);
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

