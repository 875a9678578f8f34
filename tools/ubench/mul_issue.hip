// Micro-benchmark: issue cost of the 32-bit integer multiplies Philox4x32 is made of, one wave, 64 independent instructions between two
// s_memtime reads: v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32 (both halves of the product in one instruction), v_xor_b32 for scale; round 5: the conversions and integer
// combines of the digit-plane epilogue (kernels_nmp.h): v_cvt_f64_i32 / _u32, v_lshl_add_u32, v_mad_i32_i24, v_add_f64, v_fma_f64.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mul_issue.hip -o tools/ubench/mul_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned* out, long long* clk, unsigned seed) {
    unsigned a[8], r[8];
    unsigned long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 3) + threadIdx.x; r[i] = 0; w[i] = 0; }
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            else if (MODE == 1) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            else if (MODE == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]) : "vcc");
            else if (MODE == 4) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(w[j]) : "v"(a[j]));
            else if (MODE == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(w[j]) : "v"(a[j]));
            else if (MODE == 6) asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(r[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            else if (MODE == 7) asm volatile("v_add_f64 %0, %1, %1" : "=v"(w[j]) : "v"(w[(j + 1) & 7]));
            else if (MODE == 8) asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(w[j]) : "v"(w[(j + 1) & 7]));
            else if (MODE == 9) asm volatile("v_mad_i32_i24 %0, %1, %2, %1" : "=v"(r[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
            else if (MODE == 10) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(r[j]) : "v"(a[j]));
            else if (MODE == 11) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(w[j]) : "v"(a[j]));
            else if (MODE == 12) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(w[j]) : "v"(w[(j + 1) & 7]), "v"(a[j]));
            else asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r[j]) : "v"(a[j]), "v"(a[(j + 1) & 7]));
        }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32);
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    unsigned* out; long long* clk; CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&clk, 8));
    long long h = 0, best = 1 << 30;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, out, clk, 12345u);
        CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("%-16s %5lld cycles per 64 instructions = %.1f each\n", name, best, best / 64.0);
    CK(hipFree(out)); CK(hipFree(clk));
}
int main() { run<4>("v_cvt_f64_i32"); run<5>("v_cvt_f64_u32"); run<6>("v_lshl_add_u32"); run<7>("v_add_f64"); run<8>("v_fma_f64"); run<9>("v_mad_i32_i24"); run<10>("v_cvt_f32_i32"); run<11>("v_cvt_f64_f32"); run<12>("v_ldexp_f64"); run<3>("v_xor_b32"); run<0>("v_mul_lo_u32"); run<1>("v_mul_hi_u32"); run<2>("v_mad_u64_u32"); return 0; }
