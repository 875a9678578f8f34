// philox.h -- Philox4x32-10 resampling streams (host + device inline functions; every translation unit may include it).
#pragma once
#include <cstdint>

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

