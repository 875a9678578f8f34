// Micro-benchmarks that decide the Gram-kernel design on gfx950:
//  (1) v_mfma_f64_16x16x4_f64 fragment layout check (A=asymmetric, B=asymmetric)
//  (2) f64 MFMA issue rate per SIMD
//  (3) f64 VALU FMA rate per SIMD
//  (4) do (2) and (3) overlap when co-resident on one SIMD / one CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* D) {
    // A: 16x4 row-major (i,k), B: 4x16 row-major (k,j), D: 16x16 row-major
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int MODE>  // 0 = mfma only, 1 = valu only, 2 = even waves mfma / odd waves valu, 3 = both in every wave
__global__ void __launch_bounds__(512) rate_kernel(double* out, int iters, double seed) {
    int l = threadIdx.x & 63;
    int w = threadIdx.x >> 6;
    double a = seed + l * 1e-3, b = seed - l * 1e-3;
    d4 acc[10];
    for (int t = 0; t < 10; ++t) acc[t] = (d4){0, 0, 0, 0};
    double v[16];
    for (int t = 0; t < 16; ++t) v[t] = seed * t;
    bool do_mfma = (MODE == 0) || (MODE == 3) || (MODE == 2 && (w & 1) == 0);
    bool do_valu = (MODE == 1) || (MODE == 3) || (MODE == 2 && (w & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int t = 0; t < 10; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 10; ++r) {
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = __builtin_fma(v[t], a, b);
            }
        }
    }
    double s = 0;
    for (int t = 0; t < 10; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int t = 0; t < 16; ++t) s += v[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run_rate(const char* name, int blocks, int threads, int iters, double* d_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, 10, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int waves = blocks * threads / 64;
    double mfma_waves = (MODE == 0 || MODE == 3) ? waves : (MODE == 2 ? waves / 2 : 0);
    double valu_waves = (MODE == 1 || MODE == 3) ? waves : (MODE == 2 ? waves / 2 : 0);
    double mfma_flop = mfma_waves * (double)iters * 10 * 2.0 * 16 * 16 * 4;
    double valu_flop = valu_waves * (double)iters * 160 * 64 * 2.0;
    printf("%-34s blocks=%4d thr=%3d  %8.3f ms  mfma %7.2f TF  valu %7.2f TF  total %7.2f TF\n", name, blocks, threads, ms,
           mfma_flop / ms * 1e-9, valu_flop / ms * 1e-9, (mfma_flop + valu_flop) / ms * 1e-9);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    // (1) layout
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1.0 + i * 0.37 + (i % 5) * 0.11; B[i] = -2.0 + i * 0.23 + (i % 7) * 0.31; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double maxerr = 0; for (int i = 0; i < 256; ++i) maxerr = fmax(maxerr, fabs(D[i] - R[i]));
    printf("layout check max abs err = %.3e  (%s)\n", maxerr, maxerr < 1e-9 ? "OK" : "MISMATCH");
    // (2..4) rates
    double* d_out; CK(hipMalloc(&d_out, (size_t)4096 * 512 * 8));
    int cus = prop.multiProcessorCount;
    for (int thr : {256, 512}) {
        for (int bpc : {1, 2}) {
            int blocks = cus * bpc;
            run_rate<0>("mfma_f64 only", blocks, thr, 20000, d_out);
            run_rate<1>("valu_fma_f64 only", blocks, thr, 2000, d_out);
            run_rate<2>("even waves mfma / odd waves valu", blocks, thr, 2000, d_out);
            run_rate<3>("mfma+valu in every wave", blocks, thr, 2000, d_out);
        }
    }
    return 0;
}
