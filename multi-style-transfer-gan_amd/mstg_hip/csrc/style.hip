// Kernels of the BUILD-DEFINED multi-style perceptual loss (no reference implementation exists: SURVEY.md F2 -- the
// north star names a "VGG-feature Gram-matrix / perceptual style loss", the reference only has a README bullet).
//   * 2x2 max-pool with the arg-max slot kept as a byte (exact index semantics: first maximum in row-major window order,
//     what torch.nn.functional.max_pool2d returns), forward and backward, NHWC;
//   * Gram matrix G[n] = scale * F[n]^T F[n] (F = NHWC features viewed (HW, C)) as a C x (HW) . (HW) x C contraction on the
//     fp32 MFMA, and its backward dF[n] = scale * F[n] (dG[n] + dG[n]^T).
// The 3x3 convolutions of the VGG-topology feature stack run through conv_igemm.hip (ReLU as the epilogue).
#include "common.h"

namespace mstg {

// ---- 2x2 / stride 2 max-pool ----------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int N, int H,
                                   int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * C4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int q = e % C4;
        size_t r = e / C4;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int n = r / Ho;
        const f32x4* src = reinterpret_cast<const f32x4*>(x) + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C4 + q;
        const f32x4 v0 = src[0], v1 = src[C4], v2 = src[(size_t)W * C4], v3 = src[(size_t)W * C4 + C4];
        f32x4 m;
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float best = v0[k];
            int bi = 0;
            if (v1[k] > best || v1[k] != v1[k]) { best = v1[k]; bi = 1; }
            if (v2[k] > best || v2[k] != v2[k]) { best = v2[k]; bi = 2; }
            if (v3[k] > best || v3[k] != v3[k]) { best = v3[k]; bi = 3; }
            m[k] = best;
            code |= (unsigned)bi << (8 * k);
        }
        reinterpret_cast<f32x4*>(y)[e] = m;
        reinterpret_cast<unsigned*>(idx)[e] = code;
    }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx, int N,
                                   int H, int W, int C4) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * C4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int q = e % C4;
        size_t r = e / C4;
        const int ox = r % Wo; r /= Wo;
        const int oy = r % Ho;
        const int n = r / Ho;
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[e];
        const unsigned code = reinterpret_cast<const unsigned*>(idx)[e];
        f32x4 o[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[s][k] = ((code >> (8 * k)) & 255u) == (unsigned)s ? g[k] : 0.f;
        f32x4* dst = reinterpret_cast<f32x4*>(dx) + (((size_t)n * H + 2 * oy) * W + 2 * ox) * C4 + q;
        dst[0] = o[0];
        dst[C4] = o[1];
        dst[(size_t)W * C4] = o[2];
        dst[(size_t)W * C4 + C4] = o[3];
    }
}

// ---- Gram forward: partial[n][split][c1][c2] = sum over the split's pixels of F[p][c1] * F[p][c2] -------------------------
// workgroup = (c1 fragment of 16, c2 group of 64, pixel split, image); four waves stride over 64-pixel LDS tiles, each wave
// takes 16 pixels of a tile (4 MFMA k-steps); accumulators are summed through LDS at the end.
constexpr int GP = 64;  // pixels per LDS tile
__global__ __launch_bounds__(256) void gram_fwd_kernel(const float* __restrict__ f, float* __restrict__ partial, int HW, int C, int S,
                                                       int pix_per_split) {
    __shared__ __attribute__((aligned(16))) float fa[GP][20];
    __shared__ __attribute__((aligned(16))) float fb[GP][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int n1 = C / 16, n2 = (C + 63) / 64;
    const int c1 = (blockIdx.x % n1) * 16, c2 = (blockIdx.x / n1) * 64;
    const int sp = blockIdx.y, n = blockIdx.z;
    const int p0 = sp * pix_per_split, p1 = min(HW, p0 + pix_per_split);
    const float* fn = f + (size_t)n * HW * C;
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int pt = p0; pt < p1; pt += GP) {
        __syncthreads();
        {   // stage 64 pixels x (16 + 64) channels
            const int p = tid >> 2, q = tid & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pt + p < p1) v = *reinterpret_cast<const f32x4*>(fn + (size_t)(pt + p) * C + c1 + 4 * q);
            *reinterpret_cast<f32x4*>(&fa[p][4 * q]) = v;
            for (int e = tid; e < GP * 16; e += 256) {
                const int pp = e >> 4, qq = e & 15;
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (pt + pp < p1 && c2 + 4 * qq < C) w = *reinterpret_cast<const f32x4*>(fn + (size_t)(pt + pp) * C + c2 + 4 * qq);
                *reinterpret_cast<f32x4*>(&fb[pp][4 * qq]) = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int p = wave * 16 + 4 * ks + g;
            const float a = fa[p][i];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = mfma16(a, fb[p][16 * k + i], acc[k]);
        }
    }
    __syncthreads();
    float* red = &fb[0][0];  // 64*68 floats >= 4 fragments * 256
    for (int wv = 1; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(&red[(k * 64 + lane) * 4]) = acc[k];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += *reinterpret_cast<const f32x4*>(&red[(k * 64 + lane) * 4]);
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* out = partial + ((size_t)n * S + sp) * C * C;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = c1 + 4 * g + e, c = c2 + 16 * k + i;
                if (c < C) out[(size_t)r * C + c] = acc[k][e];
            }
    }
}

// G[n][e] = scale * sum_s partial[n][s][e]
__global__ void gram_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gm, int S, int CC, float scale) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (e >= CC) return;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += partial[((size_t)n * S + k) * CC + e];
    gm[(size_t)n * CC + e] = s * scale;
}

// ---- Gram backward: dF[n][p][c] = scale * sum_c2 (dG[n][c][c2] + dG[n][c2][c]) F[n][p][c2] ------------------------------------
// workgroup = 64 pixels x 64 output channels of one image; K = C in chunks of 16 staged in LDS; the symmetrised dG rows are read
// straight from global memory (C x C per image, L2 resident).
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ f, const float* __restrict__ dg, float* __restrict__ df,
                                                       int HW, int C, float scale) {
    __shared__ __attribute__((aligned(16))) float ft[GP][20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int n = blockIdx.z, co0 = blockIdx.y * 64, pt = blockIdx.x * GP;
    const float* fn = f + (size_t)n * HW * C;
    const float* dgn = dg + (size_t)n * C * C;
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < C; k0 += 16) {
        __syncthreads();
        {
            const int p = tid >> 2, q = tid & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pt + p < HW) v = *reinterpret_cast<const f32x4*>(fn + (size_t)(pt + p) * C + k0 + 4 * q);
            *reinterpret_cast<f32x4*>(&ft[p][4 * q]) = v;
        }
        __syncthreads();
        // B(k = c2, n = pixel): lane (pixel i of this wave's 16, k-slot g) reads 4 consecutive c2
        const f32x4 b = *reinterpret_cast<const f32x4*>(&ft[wave * 16 + i][4 * g]);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int c = co0 + 16 * mf + i;  // A(m = c, k = c2) = dG[c][c2] + dG[c2][c]
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (c < C) {
                a = *reinterpret_cast<const f32x4*>(dgn + (size_t)c * C + k0 + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] += dgn[(size_t)(k0 + 4 * g + j) * C + c];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mf] = mfma16(a[j], b[j], acc[mf]);
        }
    }
    const int p = pt + wave * 16 + i;
    if (p < HW) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int c = co0 + 16 * mf + 4 * g;
            if (c < C) *reinterpret_cast<f32x4*>(df + ((size_t)n * HW + p) * C + c) = acc[mf] * scale;
        }
    }
}

static int gram_splits(int HW) {
    int s = HW / 4096;
    return s < 1 ? 1 : (s > 16 ? 16 : s);
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_maxpool2x2_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, void* stream) {
    if (!x || !y || !idx) return fail_arg(MSTG_E_BADARG, "maxpool_fwd: null pointer");
    if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) return fail_arg(MSTG_E_BADARG, "maxpool_fwd: H and W must be even and >= 2");
    if (C & 3) return fail_arg(MSTG_E_ALIGN, "maxpool_fwd: C must be a multiple of 4");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    const int nb = (int)(cdivz(total, 256) > 4096 ? 4096 : cdivz(total, 256));
    MSTG_LAUNCH(maxpool_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H, W, C / 4);
    MSTG_CHECK_LAUNCH("maxpool_fwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_maxpool2x2_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C, void* stream) {
    if (!dy || !dx || !idx) return fail_arg(MSTG_E_BADARG, "maxpool_bwd: null pointer");
    if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C <= 0) return fail_arg(MSTG_E_BADARG, "maxpool_bwd: H and W must be even and >= 2");
    if (C & 3) return fail_arg(MSTG_E_ALIGN, "maxpool_bwd: C must be a multiple of 4");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    const int nb = (int)(cdivz(total, 256) > 4096 ? 4096 : cdivz(total, 256));
    MSTG_LAUNCH(maxpool_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, N, H, W, C / 4);
    MSTG_CHECK_LAUNCH("maxpool_bwd_kernel");
    return MSTG_OK;
}

extern "C" size_t mstg_gram_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return (size_t)N * gram_splits(HW) * C * C * sizeof(float);
}

extern "C" int mstg_gram_fwd(const float* f, float* g, int N, int HW, int C, float scale, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!f || !g || !workspace) return fail_arg(MSTG_E_BADARG, "gram_fwd: null pointer");
    if (N <= 0 || HW <= 0 || C <= 0) return fail_arg(MSTG_E_BADARG, "gram_fwd: empty tensor");
    if (C % 16) return fail_arg(MSTG_E_ALIGN, "gram_fwd: C must be a multiple of 16");
    if (workspace_bytes < mstg_gram_workspace_bytes(N, HW, C)) return fail_arg(MSTG_E_WORKSPACE, "gram_fwd: workspace too small");
    const int S = gram_splits(HW);
    int pps = cdiv(HW, S);
    pps = cdiv(pps, GP) * GP;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((C / 16) * cdiv(C, 64), S, N);
    MSTG_LAUNCH(gram_fwd_kernel, grid, dim3(256), 0, st, f, (float*)workspace, HW, C, S, pps);
    MSTG_CHECK_LAUNCH("gram_fwd_kernel");
    MSTG_LAUNCH(gram_reduce_kernel, dim3(cdiv(C * C, 256), N), dim3(256), 0, st, (const float*)workspace, g, S, C * C, scale);
    MSTG_CHECK_LAUNCH("gram_reduce_kernel");
    return MSTG_OK;
}

extern "C" int mstg_gram_bwd(const float* f, const float* dg, float* df, int N, int HW, int C, float scale, void* stream) {
    if (!f || !dg || !df) return fail_arg(MSTG_E_BADARG, "gram_bwd: null pointer");
    if (N <= 0 || HW <= 0 || C <= 0) return fail_arg(MSTG_E_BADARG, "gram_bwd: empty tensor");
    if (C % 16) return fail_arg(MSTG_E_ALIGN, "gram_bwd: C must be a multiple of 16");
    dim3 grid(cdiv(HW, GP), cdiv(C, 64), N);
    MSTG_LAUNCH(gram_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, f, dg, df, HW, C, scale);
    MSTG_CHECK_LAUNCH("gram_bwd_kernel");
    return MSTG_OK;
}
