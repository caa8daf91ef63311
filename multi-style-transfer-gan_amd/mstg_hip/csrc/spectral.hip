// Spectral normalisation of a convolution weight (torch.nn.utils.spectral_norm, one power iteration per training-mode
// forward), forward and backward, one launch each.
//
//   training:  v <- normalize(W^T u),  u <- normalize(W v)        (in place: the module's weight_u / weight_v buffers)
//   always:    sigma = u . (W v),      W_sn = W / sigma
//   backward:  dW = dW_sn / sigma - (sum(dW_sn * W) / sigma^2) * u v^T      (u, v constants, as in torch)
//
// The reference applies it to every Conv2d of EnhancedDiscriminator (enhanced_generator.py:269-271).  Through torch's own
// hook that is ~14 tiny launches per convolution and forward (two GEMVs, two norms, clamps, divisions, a dot, a clone pair)
// plus their autograd mirror: ~700 launches and 2.5 ms per training step.  The matrices are small (<= 128 x 1152 at the
// bench width), so one workgroup per weight does the whole thing out of L2.
#include "common.h"

namespace mstg {

__device__ __forceinline__ float block_sum_1024(float v, float* red) {  // all threads get the sum; red: >= 17 floats of LDS
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];  // fixed order
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}

// W: (M, K) row-major (= weight.view(Cout, -1)); LDS: su[M] | sv[K] | red[32]
__global__ __launch_bounds__(1024) void spectral_norm_fwd_kernel(const float* __restrict__ W, float* __restrict__ u, float* __restrict__ v,
                                                                 float* __restrict__ Wn, float* __restrict__ sigma_out, int M, int K,
                                                                 float eps, int training) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* su = sm;
    float* sv = sm + M;
    float* red = sv + K;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int m = tid; m < M; m += nt) su[m] = u[m];
    if (!training)
        for (int k = tid; k < K; k += nt) sv[k] = v[k];
    __syncthreads();
    if (training) {
        // t = W^T u (thread per column: coalesced across k), v = t / max(||t||, eps)
        float ss = 0.f;
        for (int k = tid; k < K; k += nt) {
            float t = 0.f;
            for (int m = 0; m < M; ++m) t = fmaf(W[(size_t)m * K + k], su[m], t);
            sv[k] = t;
            ss = fmaf(t, t, ss);
        }
        const float nv = fmaxf(sqrtf(block_sum_1024(ss, red)), eps);
        for (int k = tid; k < K; k += nt) {
            const float t = sv[k] / nv;
            sv[k] = t;
            v[k] = t;
        }
        __syncthreads();
    }
    // s = W v (one wave per row)
    __shared__ float s_rows[4096];
    for (int m = wave; m < M; m += nw) {
        float p = 0.f;
        for (int k = lane; k < K; k += 64) p = fmaf(W[(size_t)m * K + k], sv[k], p);
        p = wave_sum(p);
        if (lane == 0) s_rows[m] = p;
    }
    __syncthreads();
    float sig_part = 0.f;
    if (training) {
        float ss = 0.f;
        for (int m = tid; m < M; m += nt) ss = fmaf(s_rows[m], s_rows[m], ss);
        const float nu = fmaxf(sqrtf(block_sum_1024(ss, red)), eps);
        for (int m = tid; m < M; m += nt) {
            const float un = s_rows[m] / nu;
            u[m] = un;
            sig_part = fmaf(un, s_rows[m], sig_part);
        }
    } else {
        for (int m = tid; m < M; m += nt) sig_part = fmaf(su[m], s_rows[m], sig_part);
    }
    const float sigma = block_sum_1024(sig_part, red);
    if (tid == 0) *sigma_out = sigma;
    const size_t total = (size_t)M * K;
    for (size_t e = tid; e < total; e += nt) Wn[e] = W[e] / sigma;
}

__global__ __launch_bounds__(1024) void spectral_norm_bwd_kernel(const float* __restrict__ dWn, const float* __restrict__ W,
                                                                 const float* __restrict__ u, const float* __restrict__ v,
                                                                 const float* __restrict__ sigma_p, float* __restrict__ dW, int M, int K) {
    __shared__ float red[32];
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t total = (size_t)M * K;
    float dot = 0.f;
    for (size_t e = tid; e < total; e += nt) dot = fmaf(dWn[e], W[e], dot);
    dot = block_sum_1024(dot, red);
    const float sigma = *sigma_p;
    const float c = dot / (sigma * sigma);
    for (size_t e = tid; e < total; e += nt) {
        const int m = (int)(e / K), k = (int)(e - (size_t)m * K);
        dW[e] = dWn[e] / sigma - c * u[m] * v[k];
    }
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_spectral_norm_fwd(const float* w, float* u, float* v, float* w_out, float* sigma, int M, int K, float eps,
                                      int training, void* stream) {
    if (!w || !u || !v || !w_out || !sigma) return fail_arg(MSTG_E_BADARG, "spectral_norm: null pointer");
    if (M <= 0 || K <= 0 || M > 4096 || K > 16384) return fail_arg(MSTG_E_UNSUPPORTED, "spectral_norm: matrix larger than 4096 x 16384");
    const size_t lds = (size_t)(M + K + 64) * sizeof(float);
    hipLaunchKernelGGL(spectral_norm_fwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, w, u, v, w_out, sigma, M, K, eps, training);
    MSTG_CHECK_LAUNCH("spectral_norm_fwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_spectral_norm_bwd(const float* dwn, const float* w, const float* u, const float* v, const float* sigma, float* dw,
                                      int M, int K, void* stream) {
    if (!dwn || !w || !u || !v || !sigma || !dw) return fail_arg(MSTG_E_BADARG, "spectral_norm_bwd: null pointer");
    if (M <= 0 || K <= 0) return fail_arg(MSTG_E_BADARG, "spectral_norm_bwd: bad shape");
    hipLaunchKernelGGL(spectral_norm_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dwn, w, u, v, sigma, dw, M, K);
    MSTG_CHECK_LAUNCH("spectral_norm_bwd_kernel");
    return MSTG_OK;
}
