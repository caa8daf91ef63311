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
                                                                 float eps, int training, float* __restrict__ u_save,
                                                                 float* __restrict__ v_save) {
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
    if (v_save)
        for (int k = tid; k < K; k += nt) v_save[k] = sv[k];
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
            if (u_save) u_save[m] = un;
            sig_part = fmaf(un, s_rows[m], sig_part);
        }
    } else {
        for (int m = tid; m < M; m += nt) {
            sig_part = fmaf(su[m], s_rows[m], sig_part);
            if (u_save) u_save[m] = su[m];
        }
    }
    const float sigma = block_sum_1024(sig_part, red);
    if (tid == 0) *sigma_out = sigma;
    const size_t total = (size_t)M * K;
    for (size_t e = tid; e < total; e += nt) Wn[e] = W[e] / sigma;
}

__global__ __launch_bounds__(1024) void spectral_norm_bwd_kernel(const float* __restrict__ dWn, const float* __restrict__ W,
                                                                 const float* __restrict__ u, const float* __restrict__ v,
                                                                 const float* __restrict__ sigma_p, float* __restrict__ dW, int M, int K,
                                                                 int accumulate) {
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
        const float r = dWn[e] / sigma - c * u[m] * v[k];
        dW[e] = accumulate ? dW[e] + r : r;
    }
}

// ---- large matrices (>= 32768 elements): the same arithmetic spread over SN_G workgroups in three launches --------------------
// One workgroup streams a 590 KB matrix three times through a single CU's load path (~26-70 us); split by rows the passes take
// a few microseconds each.  Every reduction across workgroups goes through a small scratch buffer and is re-done redundantly,
// in a fixed order, by each consumer workgroup (bit-reproducible, no atomics).  scratch: tpart[SN_G][K] | s[M] | sq[SN_G]
constexpr int SN_G = 16;

// A: workgroup b owns rows [m0, m1): tpart[b][k] = sum_{m in slab} W[m][k] u[m]
__global__ __launch_bounds__(256) void spectral_a_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ scratch,
                                                         int M, int K) {
    const int b = blockIdx.x, rows = (M + SN_G - 1) / SN_G, m0 = b * rows, m1 = min(M, m0 + rows);
    float* tpart = scratch + (size_t)b * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        float t = 0.f;
        for (int m = m0; m < m1; ++m) t = fmaf(W[(size_t)m * K + k], u[m], t);
        tpart[k] = t;
    }
}

// B: every workgroup rebuilds t = sum_b tpart[b] and v = t / max(||t||, eps) (workgroup 0 stores v), then s[m] = W[m] . v for
// its own rows
__global__ __launch_bounds__(256) void spectral_b_kernel(const float* __restrict__ W, float* __restrict__ v, float* __restrict__ scratch,
                                                         int M, int K, float eps, int training, float* __restrict__ v_save) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // sv[K] | red[32]
    float* sv = sm;
    float* red = sm + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float ss = 0.f;
    for (int k = tid; k < K; k += 256) {
        float t;
        if (training) {
            t = 0.f;
            for (int g = 0; g < SN_G; ++g) t += scratch[(size_t)g * K + k];
        } else {
            t = v[k];
        }
        sv[k] = t;
        ss = fmaf(t, t, ss);
    }
    if (training) {
        const float nv = fmaxf(sqrtf(block_sum_1024(ss, red)), eps);
        for (int k = tid; k < K; k += 256) {
            const float t = sv[k] / nv;
            sv[k] = t;
            if (blockIdx.x == 0) v[k] = t;
        }
    }
    __syncthreads();
    if (v_save && blockIdx.x == 0)
        for (int k = tid; k < K; k += 256) v_save[k] = sv[k];
    const int b = blockIdx.x, rows = (M + SN_G - 1) / SN_G, m0 = b * rows, m1 = min(M, m0 + rows);
    float* s_out = scratch + (size_t)SN_G * K;
    for (int m = m0 + wave; m < m1; m += 4) {
        float p = 0.f;
        for (int k = lane; k < K; k += 64) p = fmaf(W[(size_t)m * K + k], sv[k], p);
        p = wave_sum(p);
        if (lane == 0) s_out[m] = p;
    }
}

// C: every workgroup rebuilds ||s||, u = s / max(||s||, eps) and sigma = u . s (workgroup 0 stores u and sigma; eval mode
// keeps u), then writes its rows of W / sigma
__global__ __launch_bounds__(256) void spectral_c_kernel(const float* __restrict__ W, float* __restrict__ u, const float* __restrict__ scratch,
                                                         float* __restrict__ Wn, float* __restrict__ sigma_out, int M, int K, float eps,
                                                         int training, float* __restrict__ u_save) {
    __shared__ float red[32];
    const float* s_in = scratch + (size_t)SN_G * K;
    const int tid = threadIdx.x;
    float sigma;
    if (training) {
        float ss = 0.f;
        for (int m = tid; m < M; m += 256) ss = fmaf(s_in[m], s_in[m], ss);
        const float nu = fmaxf(sqrtf(block_sum_1024(ss, red)), eps);
        float sp = 0.f;
        for (int m = tid; m < M; m += 256) {
            const float un = s_in[m] / nu;
            if (blockIdx.x == 0) {
                u[m] = un;
                if (u_save) u_save[m] = un;
            }
            sp = fmaf(un, s_in[m], sp);
        }
        sigma = block_sum_1024(sp, red);
    } else {
        float sp = 0.f;
        for (int m = tid; m < M; m += 256) {
            sp = fmaf(u[m], s_in[m], sp);
            if (u_save && blockIdx.x == 0) u_save[m] = u[m];
        }
        sigma = block_sum_1024(sp, red);
    }
    if (blockIdx.x == 0 && tid == 0) *sigma_out = sigma;
    const int b = blockIdx.x, rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    for (size_t e = e0 + tid; e < e1; e += 256) Wn[e] = W[e] / sigma;
}

// backward, large matrices: A = per-workgroup partial of sum(dWn * W); B = every workgroup re-sums the partials, writes its rows
__global__ __launch_bounds__(256) void spectral_bwd_a_kernel(const float* __restrict__ dWn, const float* __restrict__ W,
                                                             float* __restrict__ scratch, int M, int K) {
    __shared__ float red[32];
    const int b = blockIdx.x, rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    float dot = 0.f;
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) dot = fmaf(dWn[e], W[e], dot);
    dot = block_sum_1024(dot, red);
    if (threadIdx.x == 0) scratch[b] = dot;
}
__global__ __launch_bounds__(256) void spectral_bwd_b_kernel(const float* __restrict__ dWn, const float* __restrict__ u,
                                                             const float* __restrict__ v, const float* __restrict__ sigma_p,
                                                             const float* __restrict__ scratch, float* __restrict__ dW, int M, int K,
                                                             int accumulate) {
    float dot = 0.f;
    for (int g = 0; g < SN_G; ++g) dot += scratch[g];
    const float sigma = *sigma_p, c = dot / (sigma * sigma);
    const int b = blockIdx.x, rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int m = (int)(e / K), k = (int)(e - (size_t)m * K);
        const float r = dWn[e] / sigma - c * u[m] * v[k];
        dW[e] = accumulate ? dW[e] + r : r;
    }
}

}  // namespace mstg

using namespace mstg;

extern "C" size_t mstg_spectral_norm_workspace_bytes(int M, int K) {
    if (M <= 0 || K <= 0) return 0;
    return ((size_t)SN_G * K + M + SN_G + 16) * sizeof(float);
}

static bool spectral_multi(int M, int K) { return (size_t)M * K >= 32768 && M >= SN_G; }

extern "C" int mstg_spectral_norm_fwd(const float* w, float* u, float* v, float* w_out, float* sigma, float* u_save, float* v_save, int M,
                                      int K, float eps, int training, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !u || !v || !w_out || !sigma) return fail_arg(MSTG_E_BADARG, "spectral_norm: null pointer");
    if (M <= 0 || K <= 0 || M > 4096 || K > 16384) return fail_arg(MSTG_E_UNSUPPORTED, "spectral_norm: matrix larger than 4096 x 16384");
    if (spectral_multi(M, K)) {
        if (!workspace || workspace_bytes < mstg_spectral_norm_workspace_bytes(M, K))
            return fail_arg(MSTG_E_WORKSPACE, "spectral_norm: workspace too small");
        float* scratch = (float*)workspace;
        hipStream_t st = (hipStream_t)stream;
        if (training) {
            MSTG_LAUNCH(spectral_a_kernel, dim3(SN_G), dim3(256), 0, st, w, (const float*)u, scratch, M, K);
            MSTG_CHECK_LAUNCH("spectral_a_kernel");
        }
        MSTG_LAUNCH(spectral_b_kernel, dim3(SN_G), dim3(256), (size_t)(K + 64) * sizeof(float), st, w, v, scratch, M, K, eps, training, v_save);
        MSTG_CHECK_LAUNCH("spectral_b_kernel");
        MSTG_LAUNCH(spectral_c_kernel, dim3(SN_G), dim3(256), 0, st, w, u, (const float*)scratch, w_out, sigma, M, K, eps, training, u_save);
        MSTG_CHECK_LAUNCH("spectral_c_kernel");
        return MSTG_OK;
    }
    const size_t lds = (size_t)(M + K + 64) * sizeof(float);
    MSTG_LAUNCH(spectral_norm_fwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, w, u, v, w_out, sigma, M, K, eps, training, u_save,
                       v_save);
    MSTG_CHECK_LAUNCH("spectral_norm_fwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_spectral_norm_bwd(const float* dwn, const float* w, const float* u, const float* v, const float* sigma, float* dw,
                                      int accumulate, int M, int K, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dwn || !w || !u || !v || !sigma || !dw) return fail_arg(MSTG_E_BADARG, "spectral_norm_bwd: null pointer");
    if (M <= 0 || K <= 0) return fail_arg(MSTG_E_BADARG, "spectral_norm_bwd: bad shape");
    if (spectral_multi(M, K)) {
        if (!workspace || workspace_bytes < mstg_spectral_norm_workspace_bytes(M, K))
            return fail_arg(MSTG_E_WORKSPACE, "spectral_norm_bwd: workspace too small");
        float* scratch = (float*)workspace;
        hipStream_t st = (hipStream_t)stream;
        MSTG_LAUNCH(spectral_bwd_a_kernel, dim3(SN_G), dim3(256), 0, st, dwn, w, scratch, M, K);
        MSTG_CHECK_LAUNCH("spectral_bwd_a_kernel");
        MSTG_LAUNCH(spectral_bwd_b_kernel, dim3(SN_G), dim3(256), 0, st, dwn, u, v, sigma, (const float*)scratch, dw, M, K, accumulate);
        MSTG_CHECK_LAUNCH("spectral_bwd_b_kernel");
        return MSTG_OK;
    }
    MSTG_LAUNCH(spectral_norm_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dwn, w, u, v, sigma, dw, M, K, accumulate);
    MSTG_CHECK_LAUNCH("spectral_norm_bwd_kernel");
    return MSTG_OK;
}
