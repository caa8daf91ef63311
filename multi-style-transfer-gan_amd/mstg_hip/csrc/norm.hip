// InstanceNorm2d(affine=False) / BatchNorm2d fused with the following activation (and an optional residual add),
// forward and backward, on NHWC fp32.  HBM-bound: the forward reads x twice (statistics, then normalise -- the
// second read of a per-image slab normally hits L2 / Infinity Cache) and writes y once.
//
// Numerics: sums are taken around a per-channel pivot (the first pixel) so that E[(x-K)^2] - E[x-K]^2 does not
// cancel when |mean| >> std; partial sums per (image, split) are combined in a fixed order (bit-reproducible).
//
// Reference sites replaced: nn.InstanceNorm2d + nn.ReLU/nn.LeakyReLU(0.2) (enhanced_generator.py:54-75,93-94,
// 100-101,107-108,122-123,129-130,242-251,263-264), the residual `+ x` (:84), nn.BatchNorm2d (pretrain.py:69-89).
#include "common.h"
#include <stdlib.h>

namespace mstg {

constexpr float NORM_EPS = 1e-5f;

struct NormGeom {
    int N, HW, C, split, chunk;  // chunk = pixels per split
};

static NormGeom norm_geom(int N, int HW, int C) {
    NormGeom g;
    g.N = N; g.HW = HW; g.C = C;
    int wgs = 1024;  // workgroups the streaming passes aim for
    { const char* e = env_get(ENV_NORM_WGS); if (e && atoi(e) >= 64) wgs = atoi(e); }
    int split = wgs / (N > 0 ? N : 1);
    if (split < 1) split = 1;
    const int maxsplit = cdiv(HW, 256);  // >= 256 pixels per workgroup (batch 1-2 at 256 x 256: 256 rows per image, not 1024)
    if (split > maxsplit) split = maxsplit;
    g.chunk = cdiv(HW, split);
    g.split = cdiv(HW, g.chunk);
    return g;
}

// partial[(n*split + sp)*2*C + {0,1}*C + c]
template <bool BWD>
__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ partial,
                                                           NormGeom g, int act, int batch) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rows][2][C]
    const int C = g.C, C4 = C >> 2, rows = 256 / C4;
    const int sp = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int q = tid % C4, row = tid / C4;
    const int p0 = sp * g.chunk, p1 = min(g.HW, p0 + g.chunk);
    const float* xb = x + (size_t)n * g.HW * C;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
        if (!BWD) {
            const f32x4 K = *reinterpret_cast<const f32x4*>((batch ? x : xb) + 4 * q);
            int p = p0 + row;
            constexpr int PX = 8;  // loads in flight per thread; the additions keep the order p, p + rows, ...
            for (; p + (PX - 1) * rows < p1; p += PX * rows) {
                f32x4 v[PX];
#pragma unroll
                for (int k = 0; k < PX; ++k) v[k] = *reinterpret_cast<const f32x4*>(xb + (size_t)(p + k * rows) * C + 4 * q);
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const f32x4 w = v[k] - K;
                    s1 += w;
                    s2 += w * w;
                }
            }
            for (; p < p1; p += rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)p * C + 4 * q) - K;
                s1 += v;
                s2 += v * v;
            }
        } else {
            const float* st = stats + (size_t)(batch ? 0 : n) * C * 2;
            f32x4 mu, rs, ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) { mu[e] = st[(4 * q + e) * 2]; rs[e] = st[(4 * q + e) * 2 + 1]; }
            if (batch) { ga = *reinterpret_cast<const f32x4*>(gamma + 4 * q); be = *reinterpret_cast<const f32x4*>(beta + 4 * q); }
            const float* dyb = dy + (size_t)n * g.HW * C;
            int p = p0 + row;
            constexpr int PX = 8;
            for (; p + (PX - 1) * rows < p1; p += PX * rows) {
                f32x4 xv[PX], gv[PX];
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    xv[k] = *reinterpret_cast<const f32x4*>(xb + (size_t)(p + k * rows) * C + 4 * q);
                    gv[k] = *reinterpret_cast<const f32x4*>(dyb + (size_t)(p + k * rows) * C + 4 * q);
                }
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const f32x4 xh = (xv[k] - mu) * rs;
                    f32x4 gg = gv[k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) gg[e] *= act_grad(xh[e] * ga[e] + be[e], act);
                    s1 += gg;
                    s2 += gg * xh;
                }
            }
            for (; p < p1; p += rows) {
                const f32x4 xh = (*reinterpret_cast<const f32x4*>(xb + (size_t)p * C + 4 * q) - mu) * rs;
                f32x4 gg = *reinterpret_cast<const f32x4*>(dyb + (size_t)p * C + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) gg[e] *= act_grad(xh[e] * ga[e] + be[e], act);
                s1 += gg;
                s2 += gg * xh;
            }
        }
        *reinterpret_cast<f32x4*>(&red[(row * 2 + 0) * C + 4 * q]) = s1;
        *reinterpret_cast<f32x4*>(&red[(row * 2 + 1) * C + 4 * q]) = s2;
    }
    __syncthreads();
    for (int e = tid; e < 2 * C; e += 256) {
        float acc = 0.f;
        for (int r = 0; r < rows; ++r) acc += red[r * 2 * C + e];
        partial[((size_t)n * g.split + sp) * 2 * C + e] = acc;
    }
}

template <bool BWD>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ residual, float* __restrict__ out,
                                                         float* __restrict__ stats, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, const float* __restrict__ partial,
                                                         NormGeom g, int act, int batch, int psplit) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][C]: a, b (+ mu, rs for backward)
    const int C = g.C, C4 = C >> 2, rows = 256 / C4;
    const int sp = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const float cnt = batch ? (float)g.N * (float)g.HW : (float)g.HW;
    for (int c = tid; c < C; c += 256) {
        float t1 = 0.f, t2 = 0.f;
        if (batch != 2 && batch != 3) {
            const int nb = batch ? 0 : n, ne = batch ? g.N : n + 1;
            for (int nn = nb; nn < ne; ++nn) {  // psplit = g.split, or what a producer's epilogue left (1 .. 16 rows per image)
                const float* pp = partial + (size_t)nn * psplit * 2 * C + c;
                int s = 0;
                for (; s + 8 <= psplit; s += 8) {  // sixteen loads in flight per trip (this prologue is a chain of L2 latencies
                    float a[8], b[8];              // otherwise); the additions keep the order s = 0, 1, ...
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        a[k] = pp[(size_t)(s + k) * 2 * C];
                        b[k] = pp[(size_t)(s + k) * 2 * C + C];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        t1 += a[k];
                        t2 += b[k];
                    }
                }
                for (; s < psplit; ++s) {
                    t1 += pp[(size_t)s * 2 * C];
                    t2 += pp[(size_t)s * 2 * C + C];
                }
            }
        }
        if (!BWD) {
            float mean, rstd;
            if (batch == 2) {
                mean = running_mean[c];
                rstd = rsqrtf(running_var[c] + NORM_EPS);
            } else if (batch == 3) {  // InstanceNorm with the statistics already known (a producer's epilogue summed them)
                mean = stats[((size_t)n * C + c) * 2];
                rstd = stats[((size_t)n * C + c) * 2 + 1];
            } else {
                const float K = (batch ? x : x + (size_t)n * g.HW * C)[c];
                const float e1 = t1 / cnt;
                float var = t2 / cnt - e1 * e1;
                var = var > 0.f ? var : 0.f;
                mean = K + e1;
                rstd = rsqrtf(var + NORM_EPS);
                if (batch == 1 && sp == 0 && n == 0) {
                    running_mean[c] = 0.9f * running_mean[c] + 0.1f * mean;
                    running_var[c] = 0.9f * running_var[c] + 0.1f * var * (cnt / fmaxf(cnt - 1.f, 1.f));
                }
            }
            if (batch != 3 && sp == 0 && (!batch || n == 0)) {
                stats[((size_t)(batch ? 0 : n) * C + c) * 2] = mean;
                stats[((size_t)(batch ? 0 : n) * C + c) * 2 + 1] = rstd;
            }
            const bool affine = batch == 1 || batch == 2;
            const float ga = affine ? gamma[c] : 1.f, be = affine ? beta[c] : 0.f;
            sm[c] = rstd * ga;                 // y = (x - mean) * a + b
            sm[C + c] = be;
            sm[2 * C + c] = mean;
        } else {
            const float* st = stats + (size_t)(batch ? 0 : n) * C * 2;
            const float mu = st[c * 2], rs = st[c * 2 + 1];
            const float ga = batch ? gamma[c] : 1.f;
            const float m1 = batch == 2 ? 0.f : t1 / cnt, m2 = batch == 2 ? 0.f : t2 / cnt;
            if (batch == 1 && sp == 0 && n == 0) { dgamma[c] = t2; dbeta[c] = t1; }
            sm[c] = mu;
            sm[C + c] = rs;
            sm[2 * C + c] = m1;
            sm[3 * C + c] = m2;
            (void)ga;
        }
    }
    __syncthreads();
    const int q = tid % C4, row = tid / C4;
    if (row >= rows) return;
    const int p0 = sp * g.chunk, p1 = min(g.HW, p0 + g.chunk);
    const size_t base = (size_t)n * g.HW * C;
    if (!BWD) {
        const f32x4 A = *reinterpret_cast<const f32x4*>(&sm[4 * q]), B = *reinterpret_cast<const f32x4*>(&sm[C + 4 * q]);
        const f32x4 M = *reinterpret_cast<const f32x4*>(&sm[2 * C + 4 * q]);
        int p = p0 + row;
        constexpr int PX = 8;  // independent pixels per trip: 8-16 sixteen-byte loads in flight per thread, read-once (non-temporal)
        for (; p + (PX - 1) * rows < p1; p += PX * rows) {
            f32x4 v[PX], r[PX];
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const size_t o = base + (size_t)(p + k * rows) * C + 4 * q;
                v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + o));
                if (residual) r[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(residual + o));
            }
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const size_t o = base + (size_t)(p + k * rows) * C + 4 * q;
                f32x4 w = (v[k] - M) * A + B;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = apply_act(w[e], act);
                if (residual) w += r[k];
                *reinterpret_cast<f32x4*>(out + o) = w;
            }
        }
        for (; p < p1; p += rows) {
            const size_t o = base + (size_t)p * C + 4 * q;
            f32x4 v = (*reinterpret_cast<const f32x4*>(x + o) - M) * A + B;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
            if (residual) v += *reinterpret_cast<const f32x4*>(residual + o);
            *reinterpret_cast<f32x4*>(out + o) = v;
        }
    } else {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(&sm[4 * q]), rs = *reinterpret_cast<const f32x4*>(&sm[C + 4 * q]);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(&sm[2 * C + 4 * q]), m2 = *reinterpret_cast<const f32x4*>(&sm[3 * C + 4 * q]);
        f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if (batch) { ga = *reinterpret_cast<const f32x4*>(gamma + 4 * q); be = *reinterpret_cast<const f32x4*>(beta + 4 * q); }
        int p = p0 + row;
        constexpr int PX = 8;  // pixels per trip: 16 sixteen-byte loads in flight per thread
        for (; p + (PX - 1) * rows < p1; p += PX * rows) {
            f32x4 xv[PX], gv[PX];
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const size_t o = base + (size_t)(p + k * rows) * C + 4 * q;
                xv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + o));
                gv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + o));
            }
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const size_t o = base + (size_t)(p + k * rows) * C + 4 * q;
                const f32x4 xh = (xv[k] - mu) * rs;
                f32x4 gg = gv[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) gg[e] *= act_grad(xh[e] * ga[e] + be[e], act);
                *reinterpret_cast<f32x4*>(out + o) = rs * ga * (gg - m1 - xh * m2);
            }
        }
        for (; p < p1; p += rows) {
            const size_t o = base + (size_t)p * C + 4 * q;
            const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * rs;
            f32x4 gg = *reinterpret_cast<const f32x4*>(dy + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) gg[e] *= act_grad(xh[e] * ga[e] + be[e], act);
            *reinterpret_cast<f32x4*>(out + o) = rs * ga * (gg - m1 - xh * m2);
        }
    }
}

// Sum of an image's `split` partial rows per channel, t1 / t2, in a fixed order.  groups == 1: s = 0, 1, ... (the order of
// norm_apply_kernel's own prologue, bit for bit); groups > 1 (small batches: up to 1024 rows per image): thread group k takes rows
// k, k + groups, ..., the groups are added 0, 1, ... through LDS.  Loads are batched eight rows at a time.
__device__ __forceinline__ void norm_row_sums(const float* __restrict__ partial, int n, int split, int C, int groups, float* red, float& t1,
                                              float& t2) {
    const int tid = threadIdx.x, c = tid % C, grp = tid / C;
    t1 = t2 = 0.f;
    if (grp < groups) {
        const float* pp = partial + (size_t)n * split * 2 * C + c;
        int s = grp;
        for (; s + 7 * groups < split; s += 8 * groups) {
            float a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a[k] = pp[(size_t)(s + k * groups) * 2 * C];
                b[k] = pp[(size_t)(s + k * groups) * 2 * C + C];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                t1 += a[k];
                t2 += b[k];
            }
        }
        for (; s < split; s += groups) {
            t1 += pp[(size_t)s * 2 * C];
            t2 += pp[(size_t)s * 2 * C + C];
        }
    }
    if (groups > 1) {
        if (grp < groups) {
            red[(grp * 2 + 0) * C + c] = t1;
            red[(grp * 2 + 1) * C + c] = t2;
        }
        __syncthreads();
        if (grp == 0) {
            t1 = t2 = 0.f;
            for (int k = 0; k < groups; ++k) {
                t1 += red[(k * 2 + 0) * C + c];
                t2 += red[(k * 2 + 1) * C + c];
            }
        }
    }
}

// threads per workgroup of the two kernels below: C channels x `groups` row groups (C <= 1024)
static int norm_groups(const NormGeom& g) { return g.split > 32 && g.C <= 128 ? 256 / g.C : 1; }
static int norm_row_threads(const NormGeom& g) { const int t = g.C * norm_groups(g); return t < 64 ? 64 : ((t + 63) / 64) * 64; }

// (mean, rstd) per (image, channel) from the pivoted partial sums: the arithmetic of norm_apply_kernel<false>
__global__ void norm_stats_kernel(const float* __restrict__ x, const float* __restrict__ partial, float* __restrict__ stats, NormGeom g,
                                  int groups) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int n = blockIdx.x, C = g.C;
    const float cnt = (float)g.HW;
    float t1, t2;
    norm_row_sums(partial, n, g.split, C, groups, red, t1, t2);
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        const float K = x[(size_t)n * g.HW * C + c];
        const float e1 = t1 / cnt;
        float var = t2 / cnt - e1 * e1;
        var = var > 0.f ? var : 0.f;
        stats[((size_t)n * C + c) * 2] = K + e1;
        stats[((size_t)n * C + c) * 2 + 1] = rsqrtf(var + NORM_EPS);
    }
}

// backward: sums[n][2][C] = the image's two reductions, for norm_apply_kernel<true> with one row per image
__global__ void norm_sums_kernel(const float* __restrict__ partial, float* __restrict__ sums, NormGeom g, int groups) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int n = blockIdx.x, C = g.C;
    float t1, t2;
    norm_row_sums(partial, n, g.split, C, groups, red, t1, t2);
    if ((int)threadIdx.x < C) {
        sums[((size_t)n * 2 + 0) * C + threadIdx.x] = t1;
        sums[((size_t)n * 2 + 1) * C + threadIdx.x] = t2;
    }
}

static int norm_check(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return fail_arg(MSTG_E_BADARG, "norm: empty tensor");
    if (C % 4 || C > 1024) return fail_arg(MSTG_E_ALIGN, "norm: C must be a multiple of 4 and <= 1024");
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

extern "C" size_t mstg_norm_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    const NormGeom g = norm_geom(N, HW, C);
    return ((size_t)N * g.split * 2 * C + (size_t)N * 2 * C) * sizeof(float);  // partial rows + one summed row per image
}

extern "C" int mstg_norm_act_fwd(const float* x, const float* residual, float* y, float* stats, int N, int HW, int C, int act,
                                 int batch_stats, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = norm_check(N, HW, C)) return rc;
    if (!x || !y || !stats || !workspace) return fail_arg(MSTG_E_BADARG, "norm_fwd: null pointer");
    if (batch_stats && (!gamma || !beta || !running_mean || !running_var)) return fail_arg(MSTG_E_BADARG, "norm_fwd: batch norm needs gamma/beta/running stats");
    if (act == MSTG_ACT_TANH) return fail_arg(MSTG_E_UNSUPPORTED, "norm_fwd: tanh epilogue not supported");
    const NormGeom g = norm_geom(N, HW, C);
    if (workspace_bytes < mstg_norm_workspace_bytes(N, HW, C)) return fail_arg(MSTG_E_WORKSPACE, "norm_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int rows = 256 / (C / 4);
    dim3 grid(g.split, N);
    if (batch_stats != 2) {
        MSTG_LAUNCH((norm_partial_kernel<false>), grid, dim3(256), (size_t)rows * 2 * C * sizeof(float), st, x, nullptr, nullptr,
                           gamma, beta, (float*)workspace, g, act, batch_stats);
        MSTG_CHECK_LAUNCH("norm_partial_kernel");
    }
    if (batch_stats == 0 && g.split > 32) {
        // small batch: an image has up to 1024 partial rows, and EVERY apply workgroup would re-add them in its prologue (at batch 1
        // that prologue was the whole kernel): one statistics launch, then the apply pass with the statistics given
        const int groups = norm_groups(g);
        MSTG_LAUNCH(norm_stats_kernel, dim3(N), dim3(norm_row_threads(g)), (size_t)groups * 2 * C * sizeof(float), st, x,
                    (const float*)workspace, stats, g, groups);
        MSTG_CHECK_LAUNCH("norm_stats_kernel");
        MSTG_LAUNCH((norm_apply_kernel<false>), grid, dim3(256), (size_t)4 * C * sizeof(float), st, x, nullptr, residual, y, stats, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, g, act, 3, 0);
        MSTG_CHECK_LAUNCH("norm_apply_kernel");
        return MSTG_OK;
    }
    MSTG_LAUNCH((norm_apply_kernel<false>), grid, dim3(256), (size_t)4 * C * sizeof(float), st, x, nullptr, residual, y, stats,
                       gamma, beta, running_mean, running_var, nullptr, nullptr, (const float*)workspace, g, act, batch_stats, g.split);
    MSTG_CHECK_LAUNCH("norm_apply_kernel");
    return MSTG_OK;
}

extern "C" int mstg_norm_act_bwd(const float* x, const float* stats, const float* dy, float* dx, int N, int HW, int C, int act,
                                 int batch_stats, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = norm_check(N, HW, C)) return rc;
    if (!x || !stats || !dy || !dx || !workspace) return fail_arg(MSTG_E_BADARG, "norm_bwd: null pointer");
    if (batch_stats == 1 && (!gamma || !beta || !dgamma || !dbeta)) return fail_arg(MSTG_E_BADARG, "norm_bwd: batch norm needs gamma/beta/dgamma/dbeta");
    if (batch_stats == 2) return fail_arg(MSTG_E_UNSUPPORTED, "norm_bwd: eval-mode batch norm backward not implemented");
    const NormGeom g = norm_geom(N, HW, C);
    if (workspace_bytes < mstg_norm_workspace_bytes(N, HW, C)) return fail_arg(MSTG_E_WORKSPACE, "norm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int rows = 256 / (C / 4);
    dim3 grid(g.split, N);
    MSTG_LAUNCH((norm_partial_kernel<true>), grid, dim3(256), (size_t)rows * 2 * C * sizeof(float), st, x, dy, stats, gamma, beta,
                       (float*)workspace, g, act, batch_stats);
    MSTG_CHECK_LAUNCH("norm_partial_kernel<bwd>");
    if (batch_stats == 0 && g.split > 32) {  // as in the forward: the rows are added once, not by every apply workgroup
        const int groups = norm_groups(g);
        float* sums = (float*)workspace + (size_t)N * g.split * 2 * C;
        MSTG_LAUNCH(norm_sums_kernel, dim3(N), dim3(norm_row_threads(g)), (size_t)groups * 2 * C * sizeof(float), st, (const float*)workspace,
                    sums, g, groups);
        MSTG_CHECK_LAUNCH("norm_sums_kernel");
        MSTG_LAUNCH((norm_apply_kernel<true>), grid, dim3(256), (size_t)4 * C * sizeof(float), st, x, dy, nullptr, dx, const_cast<float*>(stats),
                    gamma, beta, nullptr, nullptr, dgamma, dbeta, (const float*)sums, g, act, batch_stats, 1);
        MSTG_CHECK_LAUNCH("norm_apply_kernel<bwd>");
        return MSTG_OK;
    }
    MSTG_LAUNCH((norm_apply_kernel<true>), grid, dim3(256), (size_t)4 * C * sizeof(float), st, x, dy, nullptr, dx,
                       const_cast<float*>(stats), gamma, beta, nullptr, nullptr, dgamma, dbeta, (const float*)workspace, g, act,
                       batch_stats, g.split);
    MSTG_CHECK_LAUNCH("norm_apply_kernel<bwd>");
    return MSTG_OK;
}

// InstanceNorm statistics only: stats[n][c] = (mean, rstd).  For a norm whose consumer normalises while it loads
// (mstg_window_attn_norm_fwd): the normalised tensor is never written.
extern "C" int mstg_norm_stats(const float* x, float* stats, int N, int HW, int C, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (int rc = norm_check(N, HW, C)) return rc;
    if (!x || !stats || !workspace) return fail_arg(MSTG_E_BADARG, "norm_stats: null pointer");
    const NormGeom g = norm_geom(N, HW, C);
    if (workspace_bytes < mstg_norm_workspace_bytes(N, HW, C)) return fail_arg(MSTG_E_WORKSPACE, "norm_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int rows = 256 / (C / 4);
    MSTG_LAUNCH((norm_partial_kernel<false>), dim3(g.split, N), dim3(256), (size_t)rows * 2 * C * sizeof(float), st, x, nullptr,
                       nullptr, nullptr, nullptr, (float*)workspace, g, MSTG_ACT_NONE, 0);
    MSTG_CHECK_LAUNCH("norm_partial_kernel");
    const int groups = norm_groups(g);
    MSTG_LAUNCH(norm_stats_kernel, dim3(N), dim3(norm_row_threads(g)), (size_t)groups * 2 * C * sizeof(float), st, x, (const float*)workspace,
                stats, g, groups);
    MSTG_CHECK_LAUNCH("norm_stats_kernel");
    return MSTG_OK;
}

// Second half of the InstanceNorm backward when the producer of dy already summed it: sums[n][2][C] = per (image, channel)
// sum(dy * act'(.)) and sum(dy * act'(.) * x^) (mstg_window_attn_norm_bwd) -> dx = rstd * (dy * act' - mean1 - x^ * mean2).
extern "C" int mstg_norm_bwd_apply(const float* x, const float* stats, const float* dy, const float* sums, int sums_split, float* dx,
                                   int N, int HW, int C, int act, void* stream) {
    if (int rc = norm_check(N, HW, C)) return rc;
    if (!x || !stats || !dy || !sums || !dx) return fail_arg(MSTG_E_BADARG, "norm_bwd_apply: null pointer");
    if (sums_split < 1) return fail_arg(MSTG_E_BADARG, "norm_bwd_apply: sums_split must be >= 1");
    const NormGeom g = norm_geom(N, HW, C);
    MSTG_LAUNCH((norm_apply_kernel<true>), dim3(g.split, N), dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream, x, dy,
                       nullptr, dx, const_cast<float*>(stats), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sums, g, act, 0, sums_split);
    MSTG_CHECK_LAUNCH("norm_apply_kernel<bwd>");
    return MSTG_OK;
}

// Forward apply pass of InstanceNorm + activation (+ residual) with the statistics given (mstg_conv2d_fwd_norm's out_stats):
// y = act((x - mean) * rstd) [+ residual]; one read (two with the residual), one write.
extern "C" int mstg_norm_apply_fwd(const float* x, const float* stats, const float* residual, float* y, int N, int HW, int C, int act,
                                   void* stream) {
    if (int rc = norm_check(N, HW, C)) return rc;
    if (!x || !stats || !y) return fail_arg(MSTG_E_BADARG, "norm_apply_fwd: null pointer");
    if (act == MSTG_ACT_TANH) return fail_arg(MSTG_E_UNSUPPORTED, "norm_apply_fwd: tanh epilogue not supported");
    const NormGeom g = norm_geom(N, HW, C);
    MSTG_LAUNCH((norm_apply_kernel<false>), dim3(g.split, N), dim3(256), (size_t)4 * C * sizeof(float), (hipStream_t)stream, x, nullptr,
                       residual, y, const_cast<float*>(stats), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, g, act, 3, 0);
    MSTG_CHECK_LAUNCH("norm_apply_kernel");
    return MSTG_OK;
}
