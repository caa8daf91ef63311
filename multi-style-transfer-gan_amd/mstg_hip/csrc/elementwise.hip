// HBM-bound element-wise and reduction kernels of the training step: activations, mean losses and their
// gradients, per-channel sums (bias gradients / global average pool), flat Adam.
// All reductions are two-stage with a fixed order (bit-reproducible, no float atomics).
//
// Reference sites replaced: nn.LeakyReLU(0.2) of the discriminator stem (enhanced_generator.py:238), nn.Tanh
// backward (:138), nn.MSELoss / nn.L1Loss (enhanced_train.py:49-52,72-115), nn.AdaptiveAvgPool2d(1)
// (enhanced_generator.py:257), conv bias gradients, torch.optim.Adam (enhanced_train.py:36-43).
#include "common.h"

namespace mstg {

constexpr int EW_BLOCK = 256;
constexpr int EW_MAX_BLOCKS = 2048;

static int ew_grid(size_t n4) {
    size_t b = cdivz(n4, EW_BLOCK);
    if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        y[i] = apply_act(x[i], act);
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ xy, const float* __restrict__ dy, float* __restrict__ dx, size_t n, int act) {
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(xy)[i];
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= act_grad(v[e], act);
        reinterpret_cast<f32x4*>(dx)[i] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        dx[i] = dy[i] * act_grad(xy[i], act);
    }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;  // valid in thread 0
}

__global__ void loss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float bconst, size_t n, int kind,
                                    float* __restrict__ partial) {
    __shared__ float sh[8];
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - (b ? b[i] : bconst);
        acc += kind == 0 ? fabsf(d) : d * d;
    }
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void loss_final_kernel(const float* __restrict__ partial, int nb, float inv_n, float* __restrict__ out) {
    __shared__ float sh[8];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = r * inv_n;
}

__global__ void loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float bconst, size_t n, int kind,
                                const float* __restrict__ gscale, float scale, float* __restrict__ da, float* __restrict__ db) {
    const float gs = (gscale ? gscale[0] : 1.f) * scale / (float)n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - (b ? b[i] : bconst);
        float g;
        if (kind == 0) g = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);  // torch: sign(0) = 0
        else g = 2.f * d * gs;
        da[i] = g;
        if (db) db[i] = -g;
    }
}


// y = a + b (residual connections of the build-defined transformer block)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = a[(n4 << 2) + threadIdx.x] + b[(n4 << 2) + threadIdx.x];
}

// masked-image pre-training loss (pretrain.py:160-162): mean(|a * (1 - m) - b * (1 - m)|), products formed first as the reference does
__global__ void masked_l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m, size_t n,
                                         float* __restrict__ partial) {
    __shared__ float sh[8];
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float w = 1.f - m[i];
        acc += fabsf(a[i] * w - b[i] * w);
    }
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ void masked_l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m, size_t n,
                                     const float* __restrict__ gscale, float* __restrict__ da) {
    const float gs = (gscale ? gscale[0] : 1.f) / (float)n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float w = 1.f - m[i];
        const float d = a[i] * w - b[i] * w;
        da[i] = (d > 0.f ? gs : (d < 0.f ? -gs : 0.f)) * w;
    }
}
// torch.nn.utils.clip_grad_norm_(max_norm, norm_type=2): g *= min(1, max_norm / (||g|| + 1e-6)); every block re-adds the partial
// sums of squares in the same fixed order
__global__ void clip_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ partial, int nb, float max_norm,
                                  float* __restrict__ norm_out) {
    __shared__ float sh[8];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
    __shared__ float coef_s;
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        const float total = sqrtf(r);
        float c = max_norm / (total + 1e-6f);
        coef_s = c > 1.f ? 1.f : c;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
    }
    __syncthreads();
    const float c = coef_s;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= c;
}

// out[c] = scale * sum_p x[p*ctot + coff + c], two stages: [nb][C] partials then a column sum
__global__ void channel_sum_partial_kernel(const float* __restrict__ x, size_t P, int ctot, int coff, int C, float* __restrict__ partial) {
    // thread t handles channel t % C for pixels (t / C) + k * (blockDim / C) of this block's pixel range
    extern __shared__ float sh[];
    const int rows = blockDim.x / C;
    const int c = threadIdx.x % C, row = threadIdx.x / C;
    const size_t per = (P + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
    float acc = 0.f;
    if (row < rows)
        for (size_t p = p0 + row; p < p1; p += rows) acc += x[p * ctot + coff + c];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += sh[r * C + threadIdx.x];
        partial[(size_t)blockIdx.x * C + threadIdx.x] = s;
    }
}

// 16-byte aligned channel slices: four independent 16-byte loads in flight per thread (the scalar kernel above keeps one
// 4-byte load in flight and is latency-bound at ~1.4 TB/s; this one streams like the norm statistics kernel)
__global__ void channel_sum_partial_v4_kernel(const float* __restrict__ x, size_t P, int ctot, int coff, int C, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [rows][C]
    const int C4 = C >> 2, rows = blockDim.x / C4;
    const int q = threadIdx.x % C4, row = threadIdx.x / C4;
    const size_t per = (P + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < P ? p0 + per : P;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (row < rows) {
        const float* base = x + coff + 4 * q;
        size_t p = p0 + row;
        for (; p + 3 * (size_t)rows < p1; p += 4 * (size_t)rows) {
            a0 += *reinterpret_cast<const f32x4*>(base + p * ctot);
            a1 += *reinterpret_cast<const f32x4*>(base + (p + rows) * ctot);
            a2 += *reinterpret_cast<const f32x4*>(base + (p + 2 * (size_t)rows) * ctot);
            a3 += *reinterpret_cast<const f32x4*>(base + (p + 3 * (size_t)rows) * ctot);
        }
        for (; p < p1; p += rows) a0 += *reinterpret_cast<const f32x4*>(base + p * ctot);
        *reinterpret_cast<f32x4*>(&sh[row * C + 4 * q]) = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += sh[r * C + threadIdx.x];
        partial[(size_t)blockIdx.x * C + threadIdx.x] = s;
    }
}

// one workgroup per channel: 256 threads stride over the partial rows, fixed-order combine
__global__ void channel_sum_final_kernel(const float* __restrict__ partial, int nb, int C, float scale, float* __restrict__ out) {
    __shared__ float sh[8];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) s += partial[(size_t)b * C + c];
    const float r = block_sum(s, sh);
    if (threadIdx.x == 0) out[c] = r * scale;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, const unsigned char* __restrict__ mask) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i]) continue;
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        // torch.optim.Adam: denom = sqrt(v)/sqrt(bias_correction2) + eps ; p -= (lr / bias_correction1) * m / denom
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// planar (NCHW) per-channel sum, stage 1: partial[b][c] = sum over a slice of the N*HW elements of channel c
__global__ void plane_sum_partial_kernel(const float* __restrict__ x, int N, int C, size_t HW, float* __restrict__ partial) {
    __shared__ float sh[8];
    const int c = blockIdx.y;
    const size_t total = (size_t)N * HW, stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t n = e / HW, i = e - n * HW;
        acc += x[(n * C + c) * HW + i];
    }
    const float r = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * C + c] = r;
}

// out[s][c] = scale * sum_{p < P} x[(s*P + p)*C + c]   (global average pool: one workgroup per segment)
__global__ void segment_sum_kernel(const float* __restrict__ x, size_t P, int C, float scale, float* __restrict__ out) {
    extern __shared__ float sh[];
    const int rows = blockDim.x / C;
    const int c = threadIdx.x % C, row = threadIdx.x / C;
    const float* xs = x + (size_t)blockIdx.x * P * C;
    float acc = 0.f;
    if (row < rows)
        for (size_t p = row; p < P; p += rows) acc += xs[p * C + c];
    sh[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += sh[r * C + threadIdx.x];
        out[(size_t)blockIdx.x * C + threadIdx.x] = s * scale;
    }
}

// dx[s][p][c] = scale * dy[s][c]
__global__ void segment_broadcast_kernel(const float* __restrict__ dy, size_t P, int C, float scale, float* __restrict__ dx, size_t total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, per = P * C;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t s_ = e / per, c = e % C;
        dx[e] = dy[s_ * C + c] * scale;
    }
}

static int channel_sum_blocks(size_t P) {
    size_t nb = cdivz(P, 1024);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_act_fwd(const float* x, float* y, size_t n, int act, void* stream) {
    if (!x || !y) return fail_arg(MSTG_E_BADARG, "act_fwd: null pointer");
    if (n == 0) return MSTG_OK;
    MSTG_LAUNCH(act_fwd_kernel, dim3(ew_grid(n >> 2)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, y, n, act);
    MSTG_CHECK_LAUNCH("act_fwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_act_bwd(const float* x_or_y, const float* dy, float* dx, size_t n, int act, void* stream) {
    if (!x_or_y || !dy || !dx) return fail_arg(MSTG_E_BADARG, "act_bwd: null pointer");
    if (n == 0) return MSTG_OK;
    MSTG_LAUNCH(act_bwd_kernel, dim3(ew_grid(n >> 2)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x_or_y, dy, dx, n, act);
    MSTG_CHECK_LAUNCH("act_bwd_kernel");
    return MSTG_OK;
}

extern "C" size_t mstg_loss_workspace_bytes(size_t n) {
    (void)n;
    return (size_t)EW_MAX_BLOCKS * sizeof(float);
}

extern "C" int mstg_loss_mean_fwd(const float* a, const float* b, float bconst, size_t n, int kind, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!a || !out || !workspace) return fail_arg(MSTG_E_BADARG, "loss_fwd: null pointer");
    if (n == 0) return fail_arg(MSTG_E_BADARG, "loss_fwd: empty tensor");
    if (kind != 0 && kind != 1) return fail_arg(MSTG_E_BADARG, "loss_fwd: kind must be 0 (L1) or 1 (MSE)");
    if (workspace_bytes < mstg_loss_workspace_bytes(n)) return fail_arg(MSTG_E_WORKSPACE, "loss_fwd: workspace too small");
    const int nb = ew_grid(n);
    hipStream_t st = (hipStream_t)stream;
    MSTG_LAUNCH(loss_partial_kernel, dim3(nb), dim3(EW_BLOCK), 0, st, a, b, bconst, n, kind, (float*)workspace);
    MSTG_CHECK_LAUNCH("loss_partial_kernel");
    MSTG_LAUNCH(loss_final_kernel, dim3(1), dim3(EW_BLOCK), 0, st, (const float*)workspace, nb, 1.f / (float)n, out);
    MSTG_CHECK_LAUNCH("loss_final_kernel");
    return MSTG_OK;
}


// out[j] = sum_i w[j][i] * term_i for up to 8 scalar (0-dim) loss terms and up to 6 outputs (the total and the reported components
// of enhanced_train.py:72-81, 95-131 in ONE launch instead of one torch kernel per '+' and '*'), and the backward of output 0:
// d term_i = g * w[0][i].  Terms are weighted individually and summed left to right ((a + b) * w == a * w + b * w only up to rounding;
// the parity bar for losses is 1e-4 relative).
struct ScalarTerms {
    const float* t[8];
    float w[6][8];
    int n, nout;
};
__global__ void weighted_sum_kernel(ScalarTerms a, float* __restrict__ out) {
    const int j = threadIdx.x;
    if (blockIdx.x == 0 && j < a.nout) {
        float acc = 0.f;
        for (int i = 0; i < a.n; ++i) acc += a.w[j][i] * a.t[i][0];
        out[j] = acc;
    }
}
__global__ void weighted_sum_bwd_kernel(const float* __restrict__ g, ScalarTerms a, float* __restrict__ out) {
    const int i = threadIdx.x;
    if (blockIdx.x == 0 && i < a.n) out[i] = g[0] * a.w[0][i];
}

extern "C" int mstg_weighted_sum_fwd(const float* const* terms, const float* weights, int n, int nout, float* out, void* stream) {
    if (!terms || !weights || !out || n < 1 || n > 8 || nout < 1 || nout > 6) return fail_arg(MSTG_E_BADARG, "weighted_sum_fwd: 1..8 terms, 1..6 outputs");
    ScalarTerms a{};
    a.n = n;
    a.nout = nout;
    for (int i = 0; i < n; ++i) {
        if (!terms[i]) return fail_arg(MSTG_E_BADARG, "weighted_sum_fwd: null term");
        a.t[i] = terms[i];
        for (int j = 0; j < nout; ++j) a.w[j][i] = weights[j * n + i];
    }
    MSTG_LAUNCH(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out);
    MSTG_CHECK_LAUNCH("weighted_sum_kernel");
    return MSTG_OK;
}

extern "C" int mstg_weighted_sum_bwd(const float* g, const float* weights, int n, float* dterms, void* stream) {
    if (!g || !weights || !dterms || n < 1 || n > 8) return fail_arg(MSTG_E_BADARG, "weighted_sum_bwd: 1..8 terms");
    ScalarTerms a{};
    a.n = n;
    a.nout = 1;
    for (int i = 0; i < n; ++i) a.w[0][i] = weights[i];
    MSTG_LAUNCH(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, a, dterms);
    MSTG_CHECK_LAUNCH("weighted_sum_bwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_add(const float* a, const float* b, float* y, size_t n, void* stream) {
    if (!a || !b || !y) return fail_arg(MSTG_E_BADARG, "add: null pointer");
    if (n == 0) return MSTG_OK;
    MSTG_LAUNCH(add_kernel, dim3(ew_grid((n >> 2) + 1)), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, b, y, n);
    MSTG_CHECK_LAUNCH("add_kernel");
    return MSTG_OK;
}

extern "C" int mstg_masked_l1_mean_fwd(const float* a, const float* b, const float* m, size_t n, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!a || !b || !m || !out || !workspace) return fail_arg(MSTG_E_BADARG, "masked_l1_fwd: null pointer");
    if (n == 0) return fail_arg(MSTG_E_BADARG, "masked_l1_fwd: empty tensor");
    if (workspace_bytes < mstg_loss_workspace_bytes(n)) return fail_arg(MSTG_E_WORKSPACE, "masked_l1_fwd: workspace too small");
    const int nb = ew_grid(n);
    hipStream_t st = (hipStream_t)stream;
    MSTG_LAUNCH(masked_l1_partial_kernel, dim3(nb), dim3(EW_BLOCK), 0, st, a, b, m, n, (float*)workspace);
    MSTG_CHECK_LAUNCH("masked_l1_partial_kernel");
    MSTG_LAUNCH(loss_final_kernel, dim3(1), dim3(EW_BLOCK), 0, st, (const float*)workspace, nb, 1.f / (float)n, out);
    MSTG_CHECK_LAUNCH("loss_final_kernel");
    return MSTG_OK;
}

extern "C" int mstg_masked_l1_mean_bwd(const float* a, const float* b, const float* m, size_t n, const float* gscale, float* da,
                                       void* stream) {
    if (!a || !b || !m || !da) return fail_arg(MSTG_E_BADARG, "masked_l1_bwd: null pointer");
    if (n == 0) return fail_arg(MSTG_E_BADARG, "masked_l1_bwd: empty tensor");
    MSTG_LAUNCH(masked_l1_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, b, m, n, gscale, da);
    MSTG_CHECK_LAUNCH("masked_l1_bwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_clip_grad_norm(float* g, size_t n, float max_norm, float* norm_out, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!g || !workspace) return fail_arg(MSTG_E_BADARG, "clip_grad_norm: null pointer");
    if (n == 0) return fail_arg(MSTG_E_BADARG, "clip_grad_norm: empty buffer");
    if (!(max_norm > 0.f)) return fail_arg(MSTG_E_BADARG, "clip_grad_norm: max_norm must be positive");
    if (workspace_bytes < mstg_loss_workspace_bytes(n)) return fail_arg(MSTG_E_WORKSPACE, "clip_grad_norm: workspace too small");
    const int nb = ew_grid(n);
    hipStream_t st = (hipStream_t)stream;
    MSTG_LAUNCH(loss_partial_kernel, dim3(nb), dim3(EW_BLOCK), 0, st, (const float*)g, (const float*)nullptr, 0.f, n, 1, (float*)workspace);
    MSTG_CHECK_LAUNCH("loss_partial_kernel");
    MSTG_LAUNCH(clip_scale_kernel, dim3(nb), dim3(EW_BLOCK), 0, st, g, n, (const float*)workspace, nb, max_norm, norm_out);
    MSTG_CHECK_LAUNCH("clip_scale_kernel");
    return MSTG_OK;
}

extern "C" int mstg_loss_mean_bwd(const float* a, const float* b, float bconst, size_t n, int kind, const float* gscale, float scale,
                                  float* da, float* db, void* stream) {
    if (!a || !da) return fail_arg(MSTG_E_BADARG, "loss_bwd: null pointer");
    if (n == 0) return fail_arg(MSTG_E_BADARG, "loss_bwd: empty tensor");
    if (kind != 0 && kind != 1) return fail_arg(MSTG_E_BADARG, "loss_bwd: kind must be 0 (L1) or 1 (MSE)");
    MSTG_LAUNCH(loss_bwd_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, b, bconst, n, kind, gscale, scale, da, db);
    MSTG_CHECK_LAUNCH("loss_bwd_kernel");
    return MSTG_OK;
}

extern "C" size_t mstg_channel_sum_workspace_bytes(size_t P, int C) {
    if (C <= 0) return 0;
    return (size_t)channel_sum_blocks(P) * C * sizeof(float);
}

extern "C" int mstg_channel_sum(const float* x, size_t P, int ctot, int coff, int C, float scale, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!x || !out || !workspace) return fail_arg(MSTG_E_BADARG, "channel_sum: null pointer");
    if (P == 0 || C <= 0 || C > 1024 || ctot < coff + C) return fail_arg(MSTG_E_BADARG, "channel_sum: bad shape");
    if (workspace_bytes < mstg_channel_sum_workspace_bytes(P, C)) return fail_arg(MSTG_E_WORKSPACE, "channel_sum: workspace too small");
    const int nb = channel_sum_blocks(P);
    const int threads = C <= 256 ? 256 : 1024;
    hipStream_t st = (hipStream_t)stream;
    if (((ctot | coff | C) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        MSTG_LAUNCH(channel_sum_partial_v4_kernel, dim3(nb), dim3(threads), (size_t)(threads / (C / 4)) * C * sizeof(float), st, x, P, ctot,
                           coff, C, (float*)workspace);
    else
        MSTG_LAUNCH(channel_sum_partial_kernel, dim3(nb), dim3(threads), threads * sizeof(float), st, x, P, ctot, coff, C, (float*)workspace);
    MSTG_CHECK_LAUNCH("channel_sum_partial_kernel");
    MSTG_LAUNCH(channel_sum_final_kernel, dim3(C), dim3(256), 0, st, (const float*)workspace, nb, C, scale, out);
    MSTG_CHECK_LAUNCH("channel_sum_final_kernel");
    return MSTG_OK;
}

extern "C" size_t mstg_plane_sum_workspace_bytes(int N, int C, size_t HW) {
    if (N <= 0 || C <= 0) return 0;
    return (size_t)channel_sum_blocks((size_t)N * HW) * C * sizeof(float);
}

extern "C" int mstg_plane_sum(const float* x, int N, int C, size_t HW, float scale, float* out, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (!x || !out || !workspace) return fail_arg(MSTG_E_BADARG, "plane_sum: null pointer");
    if (N <= 0 || C <= 0 || C > 1024 || HW == 0) return fail_arg(MSTG_E_BADARG, "plane_sum: bad shape");
    if (workspace_bytes < mstg_plane_sum_workspace_bytes(N, C, HW)) return fail_arg(MSTG_E_WORKSPACE, "plane_sum: workspace too small");
    const int nb = channel_sum_blocks((size_t)N * HW);
    hipStream_t st = (hipStream_t)stream;
    MSTG_LAUNCH(plane_sum_partial_kernel, dim3(nb, C), dim3(256), 0, st, x, N, C, HW, (float*)workspace);
    MSTG_CHECK_LAUNCH("plane_sum_partial_kernel");
    MSTG_LAUNCH(channel_sum_final_kernel, dim3(C), dim3(256), 0, st, (const float*)workspace, nb, C, scale, out);
    MSTG_CHECK_LAUNCH("channel_sum_final_kernel");
    return MSTG_OK;
}

extern "C" int mstg_segment_mean_fwd(const float* x, int S, size_t P, int C, float* out, void* stream) {
    if (!x || !out) return fail_arg(MSTG_E_BADARG, "segment_mean: null pointer");
    if (S <= 0 || P == 0 || C <= 0 || C > 1024) return fail_arg(MSTG_E_BADARG, "segment_mean: bad shape");
    const int threads = C <= 256 ? 256 : 1024;
    MSTG_LAUNCH(segment_sum_kernel, dim3(S), dim3(threads), threads * sizeof(float), (hipStream_t)stream, x, P, C, 1.f / (float)P, out);
    MSTG_CHECK_LAUNCH("segment_sum_kernel");
    return MSTG_OK;
}

extern "C" int mstg_segment_mean_bwd(const float* dy, int S, size_t P, int C, float* dx, void* stream) {
    if (!dy || !dx) return fail_arg(MSTG_E_BADARG, "segment_mean_bwd: null pointer");
    if (S <= 0 || P == 0 || C <= 0) return fail_arg(MSTG_E_BADARG, "segment_mean_bwd: bad shape");
    const size_t total = (size_t)S * P * C;
    MSTG_LAUNCH(segment_broadcast_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, (hipStream_t)stream, dy, P, C, 1.f / (float)P, dx, total);
    MSTG_CHECK_LAUNCH("segment_broadcast_kernel");
    return MSTG_OK;
}

extern "C" int mstg_adam_step_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                                   float eps, int step, const unsigned char* mask, void* stream) {
    if (!p || !g || !m || !v) return fail_arg(MSTG_E_BADARG, "adam: null pointer");
    if (step < 1) return fail_arg(MSTG_E_BADARG, "adam: step counts from 1");
    if (n == 0) return MSTG_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    MSTG_LAUNCH(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), mask);
    MSTG_CHECK_LAUNCH("adam_kernel");
    return MSTG_OK;
}
