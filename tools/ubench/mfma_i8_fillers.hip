// What does one wave per SIMD pay for the instructions it has to issue between its int8 MFMAs?  (round 4: the count operand of the int8
// Gram out of LDS -- DESIGN.md 7b.)  One workgroup of four waves per CU, every wave 60 accumulator tiles (240 AGPRs) and 60
// v_mfma_i32_16x16x64_i8 per loop trip = one k-step of the 320-replicate x 32-pair x 6-plane tile; MODE adds fillers at fixed places:
//   0  none (matrix-pipe floor)
//   1  12 ds_read_b128                          (digit-plane fragments of four waves sharing 12 blocks)
//   2  5 global_load_dwordx4 -> VGPR            (private count fragments, 1 KB per wave instruction, walking a 32 MB buffer)
//   3  3 global_load_lds_dwordx4                (LDS-DMA of the wave's share of the 12 digit blocks)
//   4  3 global_load_dwordx4 + 3 ds_write_b128  (the same blocks through registers)
//   5  1 + 2 + 3                                (the proposed k-step, no barrier)
//   6  5 + s_barrier                            (with the workgroup barrier of a three-stage ring)
//   7  4 global_load_lds_dwordx4 + 11 ds_read_b128 + s_barrier, 30 MFMAs per wave, 8 waves   (the round-3 kernel's k-step per wave, two waves per SIMD)
//   8  1 + 4                                    (proposed k-step with the digit blocks through registers, no barrier)
//   9  8 + s_barrier
// Output: ns per trip and per CU-k-step, and the int8 rate on the executed MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE, int SH = 16>
__global__ void __launch_bounds__(MODE == 7 ? 512 : 256) fill_kernel(const uint4* __restrict__ src, long span_bytes, int iters, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    constexpr int NM = MODE == 7 ? 30 : 60;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned voff = (unsigned)lane * 16u;
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    i32x4 acc[SH == 16 ? NM : 1];
    i32x16 acc32[SH == 32 ? 15 : 1];
#pragma unroll
    for (int i = 0; i < (SH == 16 ? NM : 1); ++i) acc[i] = (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < (SH == 32 ? 15 : 1); ++i) acc32[i] = i32x16{};
    i32x4 fa[5], fb[12], ga[5], gb[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) { fa[i] = (i32x4){lane, 1, 2, 3}; ga[i] = fa[i]; }
#pragma unroll
    for (int i = 0; i < 12; ++i) fb[i] = (i32x4){lane, 3, 2, 1};
#pragma unroll
    for (int i = 0; i < 3; ++i) gb[i] = (i32x4){0, 0, 0, 0};
    // every wave walks its own stripe of the source buffer (wraps inside span_bytes)
    const long stride = 8 * 1024;
    long pos = ((long)blockIdx.x * 8 + wave) * stride % span_bytes;
    const long step = (long)gridDim.x * 8 * stride;
    constexpr bool RD = MODE == 1 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9 || MODE >= 10, GA = MODE == 2 || MODE == 5 || MODE == 6 || MODE == 8 || MODE == 9 || MODE >= 10,
                   DMA = MODE == 3 || MODE == 5 || MODE == 6 || MODE >= 10, GW = MODE == 4 || MODE == 8 || MODE == 9, BAR = MODE == 6 || MODE == 7 || MODE == 9 || MODE == 11 || MODE == 13 || MODE == 15;
    constexpr int STG = MODE >= 14 ? 5 : MODE >= 12 ? 3 : 1;      // MFMAs of stagger between consecutive waves
    unsigned stage = 0;
    auto body = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;      // the wave's fillers ride PH MFMAs later than wave 0's
    for (int it = 0; it < iters; ++it) {
        const char* base = (const char*)src + pos + 4096;
        const unsigned long long ub = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)base >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)base);
        // loads of the previous trip have had a whole trip to land
        if (MODE == 7) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        if (BAR) asm volatile("s_barrier" ::: "memory");
        if (GW) {
#pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(lds0 + 65536u + voff + (unsigned)wave * 3072u), "v"(gb[i]), "i"(i * 1024) : "memory");
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SH == 32) { if (m % 2 == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc32[(m / 2) % 15]) : "v"(fa[(m / 2) % 5]), "v"(fb[(m / 2) % 12])); }      // 30 per trip, 32 clocks each
            else if (MODE == 7) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(fa[m % 5]), "v"(fb[m % 6]));
            else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(fa[m / 12]), "v"(fb[m % 12]));
            if (MODE == 7) {
                if (m % 2 == 1 && m / 2 < 11) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fb[m / 2]) : "v"(lds0 + voff + stage), "i"((m / 2) * 1024));
                if (m % 8 == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0" : : "v"(voff + (unsigned)(m / 8) * 1024u), "s"(ub), "s"(lds0 + 65536u + (unsigned)wave * 4096u + (unsigned)(m / 8) * 1024u) : "memory");
                continue;
            }
            if (RD && (m + NM - PH) % NM % 5 == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fb[((m + NM - PH) % NM) / 5]) : "v"(lds0 + voff + stage), "i"((((m + NM - PH) % NM) / 5) * 1024));
            if (GA && (m + NM - PH) % NM % 12 == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(ga[((m + NM - PH) % NM) / 12]) : "v"(voff), "s"(ub), "i"((((m + NM - PH) % NM) / 12) * 1024 - 2048) : "memory");
            if (DMA && (m + NM - PH) % NM % 20 == 7) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0" : : "v"(voff + 5120u + (unsigned)(((m + NM - PH) % NM) / 20) * 1024u), "s"(ub), "s"(lds0 + 65536u + (unsigned)wave * 3072u + (unsigned)(((m + NM - PH) % NM) / 20) * 1024u) : "memory");
            if (GW && (m + NM - PH) % NM % 20 == 7) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(gb[((m + NM - PH) % NM) / 20]) : "v"(voff), "s"(ub), "i"((((m + NM - PH) % NM) / 20) * 1024 + 1024) : "memory");
        }
        pos += step;
        if (pos >= span_bytes - 16 * 1024) pos -= span_bytes - 16 * 1024;
        stage = stage == 24576u ? 0u : stage + 12288u;
    }
    };
    if (MODE >= 10) { if (wave == 0) body(std::integral_constant<int, 0>{}); else if (wave == 1) body(std::integral_constant<int, STG>{}); else if (wave == 2) body(std::integral_constant<int, 2 * STG>{}); else body(std::integral_constant<int, 3 * STG>{}); }
    else body(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    int s = 0;
#pragma unroll
    for (int i = 0; i < (SH == 16 ? NM : 1); ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < (SH == 32 ? 15 : 1); ++i) s += acc32[i][0] + acc32[i][15];
#pragma unroll
    for (int i = 0; i < 5; ++i) s += ga[i][0];
#pragma unroll
    for (int i = 0; i < 3; ++i) s += gb[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int SH = 16>
static void run(const char* what, const uint4* d_src, long span, int* d_out, int cus) {
    const int iters = 4000, threads = MODE == 7 ? 512 : 256;
    const size_t lds = 128 * 1024;
    CK(hipFuncSetAttribute((const void*)fill_kernel<MODE, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((fill_kernel<MODE, SH>), dim3(cus), dim3(threads), lds, 0, d_src, span, rep == 0 ? 400 : iters, d_out);
        if (rep == 0) { CK(hipDeviceSynchronize()); continue; }
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fill_kernel<MODE, SH>), dim3(cus), dim3(threads), lds, 0, d_src, span, iters, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double ns_trip = best * 1e6 / iters;
    const double mfma_per_cu_trip = (MODE == 7 ? 30.0 * 8 : 60.0 * 4);
    const double tops = mfma_per_cu_trip * 32768.0 * cus / (ns_trip * 1e-9) / 1e12;
    printf("{\"mfma\": %d, \"mode\": %d, \"what\": \"%s\", \"ns_per_kstep\": %.1f, \"cycles_at_2.4GHz\": %.0f, \"executed_TOPs\": %.0f}\n", SH, MODE, what, ns_trip, ns_trip * 2.4, tops);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const long span = 64L << 20;
    uint4* d_src; int* d_out;
    CK(hipMalloc(&d_src, span)); CK(hipMemset(d_src, 1, span));
    CK(hipMalloc(&d_out, (size_t)cus * 512 * sizeof(int)));
    for (int round = 0; round < 2; ++round) {
        run<0>("60 MFMA only", d_src, span, d_out, cus);
        run<1>("+12 ds_read_b128", d_src, span, d_out, cus);
        run<2>("+5 global_load_dwordx4", d_src, span, d_out, cus);
        run<3>("+3 global_load_lds_dwordx4", d_src, span, d_out, cus);
        run<4>("+3 global_load_dwordx4 +3 ds_write_b128", d_src, span, d_out, cus);
        run<5>("+12 ds_read +5 gload +3 lds-dma", d_src, span, d_out, cus);
        run<6>("+12 ds_read +5 gload +3 lds-dma +barrier", d_src, span, d_out, cus);
        run<8>("+12 ds_read +5 gload +3 gload/ds_write", d_src, span, d_out, cus);
        run<9>("+12 ds_read +5 gload +3 gload/ds_write +barrier", d_src, span, d_out, cus);
        run<7>("8 waves: 30 MFMA +11 ds_read +4 lds-dma per wave +barrier", d_src, span, d_out, cus);
        run<10>("mode 5, the waves' fillers staggered by 1 MFMA", d_src, span, d_out, cus);
        run<11>("mode 6 (+barrier), staggered by 1 MFMA", d_src, span, d_out, cus);
        run<12>("mode 5, staggered by 3 MFMAs", d_src, span, d_out, cus);
        run<13>("mode 6 (+barrier), staggered by 3 MFMAs", d_src, span, d_out, cus);
        run<14>("mode 5, staggered by 5 MFMAs", d_src, span, d_out, cus);
        run<15>("mode 6 (+barrier), staggered by 5 MFMAs", d_src, span, d_out, cus);
        if (round == 0) {
        run<0, 32>("30 MFMA 32x32x32 only", d_src, span, d_out, cus);
        run<1, 32>("+12 ds_read_b128", d_src, span, d_out, cus);
        run<2, 32>("+5 global_load_dwordx4", d_src, span, d_out, cus);
        run<3, 32>("+3 global_load_lds_dwordx4", d_src, span, d_out, cus);
        run<4, 32>("+3 global_load_dwordx4 +3 ds_write_b128", d_src, span, d_out, cus);
        run<5, 32>("+12 ds_read +5 gload +3 lds-dma", d_src, span, d_out, cus);
        run<6, 32>("+12 ds_read +5 gload +3 lds-dma +barrier", d_src, span, d_out, cus);
        run<8, 32>("+12 ds_read +5 gload +3 gload/ds_write", d_src, span, d_out, cus);
        run<9, 32>("+12 ds_read +5 gload +3 gload/ds_write +barrier", d_src, span, d_out, cus);
        }
    }
    return 0;
}
