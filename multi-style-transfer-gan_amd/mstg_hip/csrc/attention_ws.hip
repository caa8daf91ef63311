// LocalAttention core for window sizes other than 4 (enhanced_generator.py:7: the constructor default is window_size=8; every
// caller in the reference passes 4, which csrc/attention*.hip serve).  Same arithmetic as the 4x4 kernels -- per ws x ws window
// (P = ws^2 pixels): q^, k^ = F.normalize over channels per pixel, S[c1][c2] = sum_p q^[p][c1] k^[p][c2], softmax over c2,
// O[p][c1] = sum_c2 P[c1][c2] v[p][c2] -- and its backward, written for coverage rather than speed: one 256-thread workgroup per
// window, the window's q | k | v (and dO) staged in LDS, plain fp32 FMA loops in a fixed order (bit-reproducible).  qkv NHWC
// (N, H, W, 3C) -> o NHWC (N, H, W, C); supported while the tiles fit the CU's LDS.
#include "common.h"

namespace mstg {

namespace {

struct WsGeom {
    int P, C, LDT, LDS_, T, S, INV, DO, DS, DQK, end_fwd, end_bwd;  // offsets in floats
};
__host__ __device__ inline WsGeom ws_geom(int ws, int C) {
    WsGeom g;
    g.P = ws * ws;
    g.C = C;
    g.LDT = 3 * C + 1;   // q^ | k^ | v per pixel
    g.LDS_ = C + 1;      // C x C matrices and [P][C] tiles
    g.T = 0;
    g.S = g.T + g.P * g.LDT;
    g.INV = g.S + C * g.LDS_;
    g.end_fwd = g.INV + 2 * g.P;
    g.DO = g.end_fwd;
    g.DS = g.DO + g.P * g.LDS_;
    g.DQK = g.DS + C * g.LDS_;       // dq^ | dk^ [P][2C + 1]
    g.end_bwd = g.DQK + g.P * (2 * C + 1);
    return g;
}

// stage a window of an (N, H, W, ctot) tensor into tile[p][ld] (channels 0 .. ctot-1), p = row-major pixel of the window
__device__ void ws_load(const float* __restrict__ src, float* tile, int ld, int ctot, int H, int W, int ws, int n, int wy, int wx) {
    const int P = ws * ws;
    for (int e = threadIdx.x; e < P * ctot; e += blockDim.x) {
        const int c = e % ctot, p = e / ctot;
        const int y = ws * wy + p / ws, x = ws * wx + p % ws;
        tile[p * ld + c] = src[(((size_t)n * H + y) * W + x) * ctot + c];
    }
}

// q^, k^ in place, 1 / max(|q|, eps) and 1 / max(|k|, eps) per pixel, P = softmax_c2(S) in sm + g.S
__device__ void ws_forward_tiles(float* sm, const WsGeom& g) {
    const int C = g.C, P = g.P;
    float* T = sm + g.T;
    float* S = sm + g.S;
    float* inv = sm + g.INV;
    for (int e = threadIdx.x; e < 2 * P; e += blockDim.x) {
        const int p = e % P, which = e / P;
        float* v = T + p * g.LDT + which * C;
        float ss = 0.f;
        for (int c = 0; c < C; ++c) ss = fmaf(v[c], v[c], ss);
        const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        for (int c = 0; c < C; ++c) v[c] *= r;
        inv[which * P + p] = r;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int c1 = e / C, c2 = e % C;
        float acc = 0.f;
        for (int p = 0; p < P; ++p) acc = fmaf(T[p * g.LDT + c1], T[p * g.LDT + C + c2], acc);
        S[c1 * g.LDS_ + c2] = acc;
    }
    __syncthreads();
    for (int c1 = threadIdx.x; c1 < C; c1 += blockDim.x) {
        float* row = S + c1 * g.LDS_;
        float mx = row[0];
        for (int c2 = 1; c2 < C; ++c2) mx = fmaxf(mx, row[c2]);
        float sum = 0.f;
        for (int c2 = 0; c2 < C; ++c2) { row[c2] = expf(row[c2] - mx); sum += row[c2]; }
        const float r = 1.f / sum;
        for (int c2 = 0; c2 < C; ++c2) row[c2] *= r;
    }
    __syncthreads();
}

}  // namespace

__global__ __launch_bounds__(256) void attn_ws_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o, int N, int H, int W, int C,
                                                          int ws) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const WsGeom g = ws_geom(ws, C);
    const int nwx = W / ws, nwy = H / ws;
    const int w = blockIdx.x, wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
    ws_load(qkv, sm + g.T, g.LDT, 3 * C, H, W, ws, n, wy, wx);
    __syncthreads();
    ws_forward_tiles(sm, g);
    const float* T = sm + g.T;
    const float* Pm = sm + g.S;
    for (int e = threadIdx.x; e < g.P * C; e += blockDim.x) {
        const int c1 = e % C, p = e / C;
        float acc = 0.f;
        for (int c2 = 0; c2 < C; ++c2) acc = fmaf(Pm[c1 * g.LDS_ + c2], T[p * g.LDT + 2 * C + c2], acc);
        const int y = ws * wy + p / ws, x = ws * wx + p % ws;
        o[(((size_t)n * H + y) * W + x) * C + c1] = acc;
    }
}

__global__ __launch_bounds__(256) void attn_ws_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                          float* __restrict__ dqkv, int N, int H, int W, int C, int ws) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const WsGeom g = ws_geom(ws, C);
    const int P = g.P;
    const int nwx = W / ws, nwy = H / ws;
    const int w = blockIdx.x, wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
    float* T = sm + g.T;
    float* Pm = sm + g.S;
    float* inv = sm + g.INV;
    float* dO = sm + g.DO;
    float* dS = sm + g.DS;
    float* dQK = sm + g.DQK;
    ws_load(qkv, T, g.LDT, 3 * C, H, W, ws, n, wy, wx);
    ws_load(d_o, dO, g.LDS_, C, H, W, ws, n, wy, wx);
    __syncthreads();
    ws_forward_tiles(sm, g);
    // dP[c1][c2] = sum_p dO[p][c1] v[p][c2]
    for (int e = threadIdx.x; e < C * C; e += blockDim.x) {
        const int c1 = e / C, c2 = e % C;
        float acc = 0.f;
        for (int p = 0; p < P; ++p) acc = fmaf(dO[p * g.LDS_ + c1], T[p * g.LDT + 2 * C + c2], acc);
        dS[c1 * g.LDS_ + c2] = acc;
    }
    // dV[p][c2] = sum_c1 P[c1][c2] dO[p][c1]  (straight to global: channel block 2)
    for (int e = threadIdx.x; e < P * C; e += blockDim.x) {
        const int c2 = e % C, p = e / C;
        float acc = 0.f;
        for (int c1 = 0; c1 < C; ++c1) acc = fmaf(Pm[c1 * g.LDS_ + c2], dO[p * g.LDS_ + c1], acc);
        const int y = ws * wy + p / ws, x = ws * wx + p % ws;
        dqkv[(((size_t)n * H + y) * W + x) * 3 * C + 2 * C + c2] = acc;
    }
    __syncthreads();
    // dS = P (dP - rowsum(dP P))
    for (int c1 = threadIdx.x; c1 < C; c1 += blockDim.x) {
        float dot = 0.f;
        for (int c2 = 0; c2 < C; ++c2) dot = fmaf(dS[c1 * g.LDS_ + c2], Pm[c1 * g.LDS_ + c2], dot);
        for (int c2 = 0; c2 < C; ++c2) dS[c1 * g.LDS_ + c2] = Pm[c1 * g.LDS_ + c2] * (dS[c1 * g.LDS_ + c2] - dot);
    }
    __syncthreads();
    // dq^[p][c1] = sum_c2 dS[c1][c2] k^[p][c2] ; dk^[p][c2] = sum_c1 dS[c1][c2] q^[p][c1]
    for (int e = threadIdx.x; e < 2 * P * C; e += blockDim.x) {
        const int c = e % C, p = (e / C) % P, which = e / (C * P);
        float acc = 0.f;
        if (which == 0) {
            for (int c2 = 0; c2 < C; ++c2) acc = fmaf(dS[c * g.LDS_ + c2], T[p * g.LDT + C + c2], acc);
        } else {
            for (int c1 = 0; c1 < C; ++c1) acc = fmaf(dS[c1 * g.LDS_ + c], T[p * g.LDT + c1], acc);
        }
        dQK[p * (2 * C + 1) + which * C + c] = acc;
    }
    __syncthreads();
    // backward of F.normalize: dq = (dq^ - q^ (q^ . dq^)) / max(|q|, eps)
    for (int e = threadIdx.x; e < 2 * P; e += blockDim.x) {
        const int p = e % P, which = e / P;
        const float* hat = T + p * g.LDT + which * C;
        float* d = dQK + p * (2 * C + 1) + which * C;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(hat[c], d[c], dot);
        const float r = inv[which * P + p];
        const int y = ws * wy + p / ws, x = ws * wx + p % ws;
        float* dst = dqkv + (((size_t)n * H + y) * W + x) * 3 * C + which * C;
        for (int c = 0; c < C; ++c) dst[c] = (d[c] - hat[c] * dot) * r;
    }
}

}  // namespace mstg

using namespace mstg;

static int ws_check(int N, int H, int W, int C, int ws, bool bwd, size_t* lds) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || ws <= 0) return fail_arg(MSTG_E_BADARG, "window_attn_ws: empty tensor");
    if (H % ws || W % ws) return fail_arg(MSTG_E_BADARG, "window_attn_ws: H and W must be multiples of the window size");
    const WsGeom g = ws_geom(ws, C);
    *lds = (size_t)(bwd ? g.end_bwd : g.end_fwd) * sizeof(float);
    if (*lds > 160 * 1024 || ws > 16)
        return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_ws: window x channels does not fit one CU's LDS (this kernel keeps a whole window on chip)");
    if ((size_t)N * (H / ws) * (W / ws) > 0x7fffffffu) return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_ws: too many windows");
    return MSTG_OK;
}

extern "C" int mstg_window_attn_ws_supported(int C, int ws) {
    size_t lds;
    return ws > 0 && ws_check(1, ws, ws, C, ws, true, &lds) == MSTG_OK;
}

extern "C" int mstg_window_attn_ws_fwd(const float* qkv, float* o, int N, int H, int W, int C, int ws, void* stream) {
    size_t lds;
    if (int rc = ws_check(N, H, W, C, ws, false, &lds)) return rc;
    if (!qkv || !o) return fail_arg(MSTG_E_BADARG, "window_attn_ws_fwd: null pointer");
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_ws_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    MSTG_LAUNCH(attn_ws_fwd_kernel, dim3(N * (H / ws) * (W / ws)), dim3(256), lds, (hipStream_t)stream, qkv, o, N, H, W, C, ws);
    MSTG_CHECK_LAUNCH("attn_ws_fwd_kernel");
    return MSTG_OK;
}

extern "C" int mstg_window_attn_ws_bwd(const float* qkv, const float* d_o, float* dqkv, int N, int H, int W, int C, int ws, void* stream) {
    size_t lds;
    if (int rc = ws_check(N, H, W, C, ws, true, &lds)) return rc;
    if (!qkv || !d_o || !dqkv) return fail_arg(MSTG_E_BADARG, "window_attn_ws_bwd: null pointer");
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_ws_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    MSTG_LAUNCH(attn_ws_bwd_kernel, dim3(N * (H / ws) * (W / ws)), dim3(256), lds, (hipStream_t)stream, qkv, d_o, dqkv, N, H, W, C, ws);
    MSTG_CHECK_LAUNCH("attn_ws_bwd_kernel");
    return MSTG_OK;
}
