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

// ---- per-launch profiler (mstg_prof_*, include/mstg_hip.h) -----------------------------------------------------------
// Every kernel of the library is launched through MSTG_LAUNCH.  With profiling on (bench.py's roofline leg) the launch is
// bracketed by two HIP events on ITS stream and recorded under the symbol hipKernelNameRefByPtr reports (demangled: the name
// rocprofv3 prints), so per-kernel durations exist even where one C-ABI call launches several kernels.
extern bool g_prof_on;
void prof_begin(const void* host_fn, hipStream_t st);
void prof_end(hipStream_t st);
#define MSTG_LAUNCH(kern, grid, block, lds, st, ...)                               \
    do {                                                                           \
        const bool prof__ = mstg::g_prof_on;                                       \
        if (prof__) mstg::prof_begin((const void*)(kern), (hipStream_t)(st));      \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);               \
        if (prof__) mstg::prof_end((hipStream_t)(st));                             \
    } while (0)

// ---- caller-cached filter packs ----------------------------------------------------------------------------------------------
// Every convolution launch re-packs its filter into the caller's workspace first (a ~5 us kernel).  The *_cached entry points let
// a caller that KNOWS the workspace still holds the pack of the same weights (same descriptor, same pass) skip that launch:
// they raise this thread-local flag around the plain entry point, and every pack site tests it.  Not library state: it is false
// outside a *_cached call.
extern thread_local bool t_ws_packed;
#define MSTG_PACK_LAUNCH(...)                      \
    do {                                           \
        if (!mstg::t_ws_packed) MSTG_LAUNCH(__VA_ARGS__); \
    } while (0)

// ---- runtime switches ------------------------------------------------------------------------------------------------
// The MSTG_* environment switches are read ONCE, when the library is loaded, and again on
// mstg_env_refresh(): a train step makes ~1400 launches and each planner used to call getenv() several times per launch.
enum EnvKnob {
    ENV_ATTN_BLK4, ENV_ATTN_BLK64, ENV_WGRAD_1X1, ENV_WGRAD_TS_MAXCH, ENV_WGRAD_PLAIN, ENV_WGRAD_OLD, ENV_NO_DPACK, ENV_IGEMM,
    ENV_STREAM, ENV_PF, ENV_WGLOB, ENV_HEAVY_PER_CU, ENV_DBG, ENV_DBG_LDS_KB, ENV_MS_WGRAD_PACKED, ENV_MS_FWD4, ENV_NO_PACK_CACHE,
    ENV_P32, ENV_P32_TH, ENV_P32_WLDS, ENV_P32_DBG, ENV_P32_OCC, ENV_ATTN_REG, ENV_NFW_ADAPT, ENV_BSUMS_ALL, ENV_ATTN_BIG32, ENV_NORM_WGS,
    ENV_CONV_IMG, ENV_COUNT
};
const char* env_get(EnvKnob k);  // value as of the last refresh, nullptr when unset (runtime.hip)

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
        case MSTG_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // nn.GELU() (exact, erf form)
        default: return v;
    }
}
// derivative of act at pre-activation value v (for tanh the caller passes y = tanh(v) instead)
__device__ __forceinline__ float act_grad(float v, int act) {
    switch (act) {
        case MSTG_ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case MSTG_ACT_LEAKY02: return v > 0.f ? 1.f : 0.2f;
        case MSTG_ACT_TANH: return 1.f - v * v;
        case MSTG_ACT_GELU: return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
        default: return 1.f;
    }
}

// Blocks b and b+8 share an XCD (and its L2): give each XCD a contiguous run of tiles so neighbouring tiles,
// which share halo pixels, hit the same L2.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- batched, branch-free staging of pixel windows into LDS -------------------------------------------------------------
// A staging loop written as "for each element: load, store to LDS" makes every wave wait out one global-memory round trip
// per iteration (6-10 per tile for the windows of this library).  The helpers below issue four independent, UNCONDITIONAL
// 16-byte loads per thread (an element outside the image reads offset 0 of the image and is zeroed at the LDS write), so a
// window costs ceil(elements / 1024) round trips, and index arithmetic is 32-bit with multiply-high divisions.

// n / d for n < 65536, 1 < d < 65536
__host__ __device__ __forceinline__ unsigned magic_u32(unsigned d) { return 0xFFFFFFFFu / d + 1u; }

// Window of an NHWC tensor: rows x cols pixels from (y0, x0), nq channel quads per pixel (nq_valid of them exist),
// element (r, c, q) -> lds[(r * cols + c) * ckp + 4 * q].  img = first channel of pixel (0, 0) of the image; the image's
// H * W * ctot must stay below 2^30 floats (checked by the host).  NB = loads in flight per thread.
template <int NB>
__device__ __forceinline__ void stage_window_batch(const float* __restrict__ img, float* __restrict__ lds, int total, int e0, int cols,
                                                   int nq, unsigned m_cols, unsigned m_nq, int y0, int x0, int H, int W, int ctot,
                                                   int nq_valid, int ckp, int tid, int rowpad) {
    f32x4 v[NB];
    int dst[NB];
    bool ok[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int e = e0 + 256 * k + tid;
        const int p = nq == 1 ? e : (int)__umulhi((unsigned)e, m_nq);
        const int q = e - p * nq;
        const int r = (int)__umulhi((unsigned)p, m_cols), c = p - r * cols;
        const int iy = y0 + r, ix = x0 + c;
        ok[k] = e < total && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && q < nq_valid;
        const unsigned off = ok[k] ? (unsigned)(iy * W + ix) * (unsigned)ctot + 4u * q : 0u;
        v[k] = *reinterpret_cast<const f32x4*>(img + off);
        dst[k] = e < total ? p * ckp + 4 * q + r * rowpad : -1;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (dst[k] >= 0) *reinterpret_cast<f32x4*>(&lds[dst[k]]) = ok[k] ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void stage_window(const float* __restrict__ img, float* __restrict__ lds, int rows, int cols, int nq,
                                             unsigned m_cols, unsigned m_nq, int y0, int x0, int H, int W, int ctot, int nq_valid,
                                             int ckp, int tid, int rowpad = 0) {  // rowpad: extra floats between the rows of the LDS image
    const int total = rows * cols * nq;
    int e0 = 0;
    for (; e0 + 1024 < total; e0 += 2048)  // more than four elements per thread left: eight loads in flight
        stage_window_batch<8>(img, lds, total, e0, cols, nq, m_cols, m_nq, y0, x0, H, W, ctot, nq_valid, ckp, tid, rowpad);
    for (; e0 < total; e0 += 1024)
        stage_window_batch<4>(img, lds, total, e0, cols, nq, m_cols, m_nq, y0, x0, H, W, ctot, nq_valid, ckp, tid, rowpad);
}

// Window of a tensor with at most 4 channels (any layout: sy / sx / sc = row / pixel / channel stride in floats),
// pixel (r, c) -> the quad lds[(r * cols + c) * 4 ..], channels beyond nch zero.
__device__ __forceinline__ void stage_window_c4(const float* __restrict__ img, float* __restrict__ lds, int rows, int cols,
                                                unsigned m_cols, int y0, int x0, int H, int W, unsigned sy, unsigned sx, unsigned sc,
                                                int nch, int tid) {
    const int total = rows * cols;
    for (int e0 = 0; e0 < total; e0 += 1024) {
        f32x4 v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + 256 * k + tid;
            const int r = (int)__umulhi((unsigned)e, m_cols), c = e - r * cols;
            const int iy = y0 + r, ix = x0 + c;
            ok[k] = e < total && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned off = ok[k] ? (unsigned)iy * sy + (unsigned)ix * sx : 0u;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) v[k][ch] = img[off + (ch < nch ? ch * sc : 0u)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = e0 + 256 * k + tid;
            if (e < total) {
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ok[k] && ch < nch) w[ch] = v[k][ch];
                *reinterpret_cast<f32x4*>(&lds[e * 4]) = w;
            }
        }
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace mstg
