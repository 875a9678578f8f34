// Micro-benchmark: what one wave pays per instruction in dependent fp64 multiply-add streams on gfx950 -- the rows solver's
// segmented product is such a stream (kernels_solver.h seg_products).  One wave, s_memtime around 64 multiply-adds:
//   mode 0: v_fmac_f64 with a scalar multiplicand, CH independent chains
//   mode 1: v_fmac_f64_dpp row_newbcast (the multiplicand broadcast from a lane), CH chains
//   mode 2: mode 1 + s_bitcmp1_b32 + s_cbranch_scc1 (never taken) behind every multiply-add
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE, int CH>
__global__ void __launch_bounds__(64) k(double* out, long long* clk, double seed, unsigned mask) {
    double s[16], r[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) s[i] = seed + i + threadIdx.x;
    double W = seed * threadIdx.x;
    unsigned long long sw = 0x3ff0000000000000ull;
    asm volatile("" : "+v"(W), "+s"(sw));
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double& acc = r[j % CH];
            if (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "s"(sw), "v"(s[j]));
            else if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(W), "v"(s[j]));
            else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_bitcmp1_b32 %3, 5\n\ts_cbranch_scc1 .Lx%=\n.Lx%=:" : "+v"(acc) : "v"(W), "v"(s[j]), "s"(mask) : "scc");
        }
    asm volatile("s_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int MODE, int CH> void run(const char* name) {
    double* out; long long* clk; CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&clk, 8));
    long long h = 0, best = 1 << 30;
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((k<MODE, CH>), dim3(1), dim3(64), 0, 0, out, clk, 1.0, 0u);
        CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("%-44s chains %d: %lld cycles per 64 multiply-adds = %.1f per instruction\n", name, CH, best, best / 64.0);
    CK(hipFree(out)); CK(hipFree(clk));
}
int main() {
    run<0, 1>("v_fmac_f64 scalar operand"); run<0, 2>("v_fmac_f64 scalar operand"); run<0, 4>("v_fmac_f64 scalar operand");
    run<1, 1>("v_fmac_f64_dpp row_newbcast"); run<1, 2>("v_fmac_f64_dpp row_newbcast"); run<1, 4>("v_fmac_f64_dpp row_newbcast");
    run<2, 2>("v_fmac_f64_dpp + s_bitcmp1 + s_cbranch"); run<2, 4>("v_fmac_f64_dpp + s_bitcmp1 + s_cbranch");
    return 0;
}
