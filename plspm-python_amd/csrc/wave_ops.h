// wave_ops.h -- 64-lane wave reductions and scans on the DPP and permlane-swap data paths of gfx950.  `__shfl_*` compiles to
// ds_bpermute_b32: every step of a shuffle tree is an LDS-crossbar round trip (~100 clocks for a double); a DPP move costs one VALU
// issue, v_permlane16_swap / v_permlane32_swap (new with gfx950) exchange rows / wave halves in one instruction.
// Included by the .hip translation units (device code only).
#pragma once

namespace wv {
typedef unsigned u2 __attribute__((ext_vector_type(2)));
// DPP controls (GFX9 encoding)
constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E;            // quad_perm:[1,0,3,2], [2,3,0,1]
constexpr int ROW_HALF_MIRROR = 0x141, ROW_MIRROR = 0x140;
constexpr int ROW_SHR = 0x110, ROW_BCAST15 = 0x142, ROW_BCAST31 = 0x143;

template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ unsigned dpp(unsigned v) { return (unsigned)dpp<CTRL>((int)v); }
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
    return __hiloint2double(dpp<CTRL>(__double2hiint(v)), dpp<CTRL>(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp(unsigned long long v) {
    return ((unsigned long long)dpp<CTRL>((unsigned)(v >> 32)) << 32) | dpp<CTRL>((unsigned)v);
}
// a = my value or my partner's, b = the other one -- the same (a, b) on both lanes of a pair: rows r, r ^ 1 (swap16), halves (swap32)
__device__ __forceinline__ void swap16(unsigned v, unsigned& a, unsigned& b) { const u2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap32(unsigned v, unsigned& a, unsigned& b) { const u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap16(int v, int& a, int& b) { unsigned x, y; swap16((unsigned)v, x, y); a = (int)x; b = (int)y; }
__device__ __forceinline__ void swap32(int v, int& a, int& b) { unsigned x, y; swap32((unsigned)v, x, y); a = (int)x; b = (int)y; }
#define PLSPM_WV_SWAP64(NAME)                                                                                       \
    __device__ __forceinline__ void NAME(unsigned long long v, unsigned long long& a, unsigned long long& b) {      \
        unsigned al, bl, ah, bh;                                                                                    \
        NAME((unsigned)v, al, bl); NAME((unsigned)(v >> 32), ah, bh);                                               \
        a = ((unsigned long long)ah << 32) | al; b = ((unsigned long long)bh << 32) | bl;                           \
    }                                                                                                               \
    __device__ __forceinline__ void NAME(double v, double& a, double& b) {                                          \
        unsigned long long x, y;                                                                                    \
        NAME((unsigned long long)__double_as_longlong(v), x, y);                                                    \
        a = __longlong_as_double((long long)x); b = __longlong_as_double((long long)y);                             \
    }
PLSPM_WV_SWAP64(swap16)
PLSPM_WV_SWAP64(swap32)
#undef PLSPM_WV_SWAP64

// Butterfly all-reduce with a COMMUTATIVE op (lane ^ 1, ^ 2, the other quad of the half row, the other half of the row, the other row
// of the pair, the other half of the wave): after every level the lanes of a group hold bitwise the same value, so every lane ends
// with the value of ONE fixed tree -- the tree of v = op(v, shfl_xor(v, k)), k = 1, 2, 4, ... 32.
template <class T, class Op> __device__ __forceinline__ T allreduce(T v, Op op) {
    v = op(v, dpp<QP_XOR1>(v));
    v = op(v, dpp<QP_XOR2>(v));
    v = op(v, dpp<ROW_HALF_MIRROR>(v));
    v = op(v, dpp<ROW_MIRROR>(v));
    T a, b;
    swap16(v, a, b); v = op(a, b);
    swap32(v, a, b); v = op(a, b);
    return v;
}
__device__ __forceinline__ double allsum(double v) { return allreduce(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ unsigned allsum(unsigned v) { return allreduce(v, [](unsigned a, unsigned b) { return a + b; }); }
__device__ __forceinline__ unsigned long long allmin(unsigned long long v) { return allreduce(v, [](unsigned long long a, unsigned long long b) { return a < b ? a : b; }); }
__device__ __forceinline__ unsigned long long allmax(unsigned long long v) { return allreduce(v, [](unsigned long long a, unsigned long long b) { return a > b ? a : b; }); }

// inclusive prefix sum over the lanes: Hillis-Steele inside the rows of 16 (row_shr, lanes without a source add 0), then the row
// totals across (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, ROW_SHR + 1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ROW_SHR + 2, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ROW_SHR + 4, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ROW_SHR + 8, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ROW_BCAST15, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, ROW_BCAST31, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ unsigned inclusive_scan(unsigned v) { return (unsigned)inclusive_scan((int)v); }
}  // namespace wv
