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
__device__ __forceinline__ void spectral_a_body(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ scratch, int M,
                                                int K, int b) {
    const int rows = (M + SN_G - 1) / SN_G, m0 = b * rows, m1 = min(M, m0 + rows);
    float* tpart = scratch + (size_t)b * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        float t = 0.f;
        for (int m = m0; m < m1; ++m) t = fmaf(W[(size_t)m * K + k], u[m], t);
        tpart[k] = t;
    }
}
__global__ __launch_bounds__(256) void spectral_a_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ scratch,
                                                         int M, int K) {
    spectral_a_body(W, u, scratch, M, K, blockIdx.x);
}

// B: every workgroup rebuilds t = sum_b tpart[b] and v = t / max(||t||, eps) (workgroup 0 stores v), then s[m] = W[m] . v for
// its own rows
__device__ __forceinline__ void spectral_b_body(const float* __restrict__ W, float* __restrict__ v, float* __restrict__ scratch, int M, int K,
                                                float eps, int training, float* __restrict__ v_save, float* sm, int b) {
    float* sv = sm;  // sv[K] | red[32]
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
            if (b == 0) v[k] = t;
        }
    }
    __syncthreads();
    if (v_save && b == 0)
        for (int k = tid; k < K; k += 256) v_save[k] = sv[k];
    const int rows = (M + SN_G - 1) / SN_G, m0 = b * rows, m1 = min(M, m0 + rows);
    float* s_out = scratch + (size_t)SN_G * K;
    for (int m = m0 + wave; m < m1; m += 4) {
        float p = 0.f;
        for (int k = lane; k < K; k += 64) p = fmaf(W[(size_t)m * K + k], sv[k], p);
        p = wave_sum(p);
        if (lane == 0) s_out[m] = p;
    }
}
__global__ __launch_bounds__(256) void spectral_b_kernel(const float* __restrict__ W, float* __restrict__ v, float* __restrict__ scratch,
                                                         int M, int K, float eps, int training, float* __restrict__ v_save) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    spectral_b_body(W, v, scratch, M, K, eps, training, v_save, sm, blockIdx.x);
}

// C: every workgroup rebuilds ||s||, u = s / max(||s||, eps) and sigma = u . s (workgroup 0 stores u and sigma; eval mode
// keeps u), then writes its rows of W / sigma
__device__ __forceinline__ void spectral_c_body(const float* __restrict__ W, float* __restrict__ u, const float* __restrict__ scratch,
                                                float* __restrict__ Wn, float* __restrict__ sigma_out, int M, int K, float eps, int training,
                                                float* __restrict__ u_save, float* red, int b) {
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
            if (b == 0) {
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
            if (u_save && b == 0) u_save[m] = u[m];
        }
        sigma = block_sum_1024(sp, red);
    }
    if (b == 0 && tid == 0) *sigma_out = sigma;
    const int rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    for (size_t e = e0 + tid; e < e1; e += 256) Wn[e] = W[e] / sigma;
}
__global__ __launch_bounds__(256) void spectral_c_kernel(const float* __restrict__ W, float* __restrict__ u, const float* __restrict__ scratch,
                                                         float* __restrict__ Wn, float* __restrict__ sigma_out, int M, int K, float eps,
                                                         int training, float* __restrict__ u_save) {
    __shared__ float red[32];
    spectral_c_body(W, u, scratch, Wn, sigma_out, M, K, eps, training, u_save, red, blockIdx.x);
}

// backward, large matrices: A = per-workgroup partial of sum(dWn * W); B = every workgroup re-sums the partials, writes its rows
__device__ __forceinline__ void spectral_bwd_a_body(const float* __restrict__ dWn, const float* __restrict__ W, float* __restrict__ scratch,
                                                    int M, int K, float* red, int b) {
    const int rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    float dot = 0.f;
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) dot = fmaf(dWn[e], W[e], dot);
    dot = block_sum_1024(dot, red);
    if (threadIdx.x == 0) scratch[b] = dot;
}
__global__ __launch_bounds__(256) void spectral_bwd_a_kernel(const float* __restrict__ dWn, const float* __restrict__ W,
                                                             float* __restrict__ scratch, int M, int K) {
    __shared__ float red[32];
    spectral_bwd_a_body(dWn, W, scratch, M, K, red, blockIdx.x);
}
__device__ __forceinline__ void spectral_bwd_b_body(const float* __restrict__ dWn, const float* __restrict__ u, const float* __restrict__ v,
                                                    const float* __restrict__ sigma_p, const float* __restrict__ scratch,
                                                    float* __restrict__ dW, int M, int K, int accumulate, int b) {
    float dot = 0.f;
    for (int g = 0; g < SN_G; ++g) dot += scratch[g];
    const float sigma = *sigma_p, c = dot / (sigma * sigma);
    const int rows = (M + SN_G - 1) / SN_G;
    const size_t e0 = (size_t)b * rows * K, e1 = min((size_t)M * K, e0 + (size_t)rows * K);
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const int m = (int)(e / K), k = (int)(e - (size_t)m * K);
        const float r = dWn[e] / sigma - c * u[m] * v[k];
        dW[e] = accumulate ? dW[e] + r : r;
    }
}
__global__ __launch_bounds__(256) void spectral_bwd_b_kernel(const float* __restrict__ dWn, const float* __restrict__ u,
                                                             const float* __restrict__ v, const float* __restrict__ sigma_p,
                                                             const float* __restrict__ scratch, float* __restrict__ dW, int M, int K,
                                                             int accumulate) {
    spectral_bwd_b_body(dWn, u, v, sigma_p, scratch, dW, M, K, accumulate, blockIdx.x);
}

// ---- a group of weights (every convolution of one discriminator forward) in the same three / two launches: blockIdx.y = weight --------
// A discriminator forward normalises seven weights; one at a time that is 13 launches of a few microseconds each (and their backward
// mirror), 160 of the train step's launches.  The split kernels above take any shape (a workgroup whose row slab is empty writes
// zeros / nothing), so the whole group runs through them at once.
constexpr int SN_MAX_GROUP = 8;
struct SnGroup {
    const float* w[SN_MAX_GROUP];
    float* u[SN_MAX_GROUP];
    float* v[SN_MAX_GROUP];
    float* wn[SN_MAX_GROUP];
    float* us[SN_MAX_GROUP];
    float* vs[SN_MAX_GROUP];
    float* scratch[SN_MAX_GROUP];
    int M[SN_MAX_GROUP], K[SN_MAX_GROUP];
};
struct SnGroupBwd {
    const float* dwn[SN_MAX_GROUP];
    const float* w[SN_MAX_GROUP];
    const float* u[SN_MAX_GROUP];
    const float* v[SN_MAX_GROUP];
    const float* sigma[SN_MAX_GROUP];
    float* dw[SN_MAX_GROUP];
    float* scratch[SN_MAX_GROUP];
    int M[SN_MAX_GROUP], K[SN_MAX_GROUP], accumulate[SN_MAX_GROUP];
};
__global__ __launch_bounds__(256) void spectral_group_a_kernel(const SnGroup g) {
    const int j = blockIdx.y;
    spectral_a_body(g.w[j], g.u[j], g.scratch[j], g.M[j], g.K[j], blockIdx.x);
}
__global__ __launch_bounds__(256) void spectral_group_b_kernel(const SnGroup g, float eps, int training) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int j = blockIdx.y;
    spectral_b_body(g.w[j], g.v[j], g.scratch[j], g.M[j], g.K[j], eps, training, g.vs[j], sm, blockIdx.x);
}
__global__ __launch_bounds__(256) void spectral_group_c_kernel(const SnGroup g, float* __restrict__ sigma, float eps, int training) {
    __shared__ float red[32];
    const int j = blockIdx.y;
    spectral_c_body(g.w[j], g.u[j], g.scratch[j], g.wn[j], sigma + j, g.M[j], g.K[j], eps, training, g.us[j], red, blockIdx.x);
}
__global__ __launch_bounds__(256) void spectral_group_bwd_a_kernel(const SnGroupBwd g) {
    __shared__ float red[32];
    const int j = blockIdx.y;
    spectral_bwd_a_body(g.dwn[j], g.w[j], g.scratch[j], g.M[j], g.K[j], red, blockIdx.x);
}
__global__ __launch_bounds__(256) void spectral_group_bwd_b_kernel(const SnGroupBwd g) {
    const int j = blockIdx.y;
    spectral_bwd_b_body(g.dwn[j], g.u[j], g.v[j], g.sigma[j], g.scratch[j], g.dw[j], g.M[j], g.K[j], g.accumulate[j], blockIdx.x);
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

// ---- grouped entry points -------------------------------------------------------------------------------------------------------
static size_t sn_group_floats(int M, int K) { return ((size_t)SN_G * K + M + SN_G + 16 + 3) & ~(size_t)3; }

extern "C" int mstg_spectral_norm_group_max(void) { return SN_MAX_GROUP; }

extern "C" size_t mstg_spectral_norm_group_workspace_bytes(int count, const int* M, const int* K) {
    if (count <= 0 || count > SN_MAX_GROUP || !M || !K) return 0;
    size_t n = 0;
    for (int j = 0; j < count; ++j) {
        if (M[j] <= 0 || K[j] <= 0) return 0;
        n += sn_group_floats(M[j], K[j]);
    }
    return n * sizeof(float);
}

// w / u / v / w_out / u_save / v_save: `count` pointers each (u_save / v_save may be null as a whole: no backward follows);
// sigma: `count` floats.  Per weight the semantics of mstg_spectral_norm_fwd.
extern "C" int mstg_spectral_norm_group_fwd(int count, const float* const* w, float* const* u, float* const* v, float* const* w_out,
                                            float* sigma, float* const* u_save, float* const* v_save, const int* M, const int* K, float eps,
                                            int training, void* workspace, size_t workspace_bytes, void* stream) {
    if (count <= 0 || count > SN_MAX_GROUP) return fail_arg(MSTG_E_BADARG, "spectral_norm_group: 1..8 weights per call");
    if (!w || !u || !v || !w_out || !sigma || !M || !K) return fail_arg(MSTG_E_BADARG, "spectral_norm_group: null pointer");
    if (!workspace || workspace_bytes < mstg_spectral_norm_group_workspace_bytes(count, M, K))
        return fail_arg(MSTG_E_WORKSPACE, "spectral_norm_group: workspace too small");
    SnGroup g{};
    float* scratch = (float*)workspace;
    int kmax = 0;
    for (int j = 0; j < count; ++j) {
        if (!w[j] || !u[j] || !v[j] || !w_out[j]) return fail_arg(MSTG_E_BADARG, "spectral_norm_group: null pointer");
        if (M[j] <= 0 || K[j] <= 0 || M[j] > 4096 || K[j] > 16384) return fail_arg(MSTG_E_UNSUPPORTED, "spectral_norm_group: matrix larger than 4096 x 16384");
        g.w[j] = w[j]; g.u[j] = u[j]; g.v[j] = v[j]; g.wn[j] = w_out[j];
        g.us[j] = u_save ? u_save[j] : nullptr;
        g.vs[j] = v_save ? v_save[j] : nullptr;
        g.M[j] = M[j]; g.K[j] = K[j];
        g.scratch[j] = scratch;
        scratch += sn_group_floats(M[j], K[j]);
        if (K[j] > kmax) kmax = K[j];
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(SN_G, count);
    if (training) {
        MSTG_LAUNCH(spectral_group_a_kernel, grid, dim3(256), 0, st, g);
        MSTG_CHECK_LAUNCH("spectral_group_a_kernel");
    }
    MSTG_LAUNCH(spectral_group_b_kernel, grid, dim3(256), (size_t)(kmax + 64) * sizeof(float), st, g, eps, training);
    MSTG_CHECK_LAUNCH("spectral_group_b_kernel");
    MSTG_LAUNCH(spectral_group_c_kernel, grid, dim3(256), 0, st, g, sigma, eps, training);
    MSTG_CHECK_LAUNCH("spectral_group_c_kernel");
    return MSTG_OK;
}

// dwn / w / u / v / sigma / dw: `count` pointers each (sigma[j] points at that weight's sigma); accumulate[j] != 0 adds into dw[j].
extern "C" int mstg_spectral_norm_group_bwd(int count, const float* const* dwn, const float* const* w, const float* const* u,
                                            const float* const* v, const float* const* sigma, float* const* dw, const int* accumulate,
                                            const int* M, const int* K, void* workspace, size_t workspace_bytes, void* stream) {
    if (count <= 0 || count > SN_MAX_GROUP) return fail_arg(MSTG_E_BADARG, "spectral_norm_group_bwd: 1..8 weights per call");
    if (!dwn || !w || !u || !v || !sigma || !dw || !accumulate || !M || !K) return fail_arg(MSTG_E_BADARG, "spectral_norm_group_bwd: null pointer");
    if (!workspace || workspace_bytes < mstg_spectral_norm_group_workspace_bytes(count, M, K))
        return fail_arg(MSTG_E_WORKSPACE, "spectral_norm_group_bwd: workspace too small");
    SnGroupBwd g{};
    float* scratch = (float*)workspace;
    for (int j = 0; j < count; ++j) {
        if (!dwn[j] || !w[j] || !u[j] || !v[j] || !sigma[j] || !dw[j]) return fail_arg(MSTG_E_BADARG, "spectral_norm_group_bwd: null pointer");
        if (M[j] <= 0 || K[j] <= 0) return fail_arg(MSTG_E_BADARG, "spectral_norm_group_bwd: bad shape");
        g.dwn[j] = dwn[j]; g.w[j] = w[j]; g.u[j] = u[j]; g.v[j] = v[j]; g.sigma[j] = sigma[j]; g.dw[j] = dw[j];
        g.M[j] = M[j]; g.K[j] = K[j]; g.accumulate[j] = accumulate[j];
        g.scratch[j] = scratch;
        scratch += sn_group_floats(M[j], K[j]);
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(SN_G, count);
    MSTG_LAUNCH(spectral_group_bwd_a_kernel, grid, dim3(256), 0, st, g);
    MSTG_CHECK_LAUNCH("spectral_group_bwd_a_kernel");
    MSTG_LAUNCH(spectral_group_bwd_b_kernel, grid, dim3(256), 0, st, g);
    MSTG_CHECK_LAUNCH("spectral_group_bwd_b_kernel");
    return MSTG_OK;
}
