// Shared device/host helpers for the gfx950 kernels of libmstg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "mstg_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace mstg {

// ---- error plumbing --------------------------------------------------------------------------------------
extern thread_local char g_last_error[256];
inline int fail_launch(hipError_t e, const char* what) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
    return MSTG_E_LAUNCH;
}
inline int fail_arg(int code, const char* what) {
    snprintf(g_last_error, sizeof(g_last_error), "%s", what);
    return code;
}
#define MSTG_CHECK_LAUNCH(what)                                   \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return mstg::fail_launch(e__, what); \
    } while (0)

constexpr int WAVE = 64;

// fp32-in / fp32-accumulate MFMA, 16x16 tile, K = 4 per instruction (v_mfma_f32_16x16x4_f32).
//   lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15];
//   the accumulator register r of lane l is D[row m = 4 * (l >> 4) + r][col n = l & 15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case MSTG_ACT_RELU: return v > 0.f ? v : 0.f;
        case MSTG_ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        case MSTG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative of act at pre-activation value v (for tanh the caller passes y = tanh(v) instead)
__device__ __forceinline__ float act_grad(float v, int act) {
    switch (act) {
        case MSTG_ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case MSTG_ACT_LEAKY02: return v > 0.f ? 1.f : 0.2f;
        case MSTG_ACT_TANH: return 1.f - v * v;
        default: return 1.f;
    }
}

// Blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous run of tiles so neighbouring tiles,
// which share halo pixels, hit the same L2.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace mstg
