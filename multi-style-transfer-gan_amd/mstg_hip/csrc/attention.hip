// LocalAttention core (window 4x4 channel attention), forward and backward, fp32 MFMA, one wave per window.
//
// Per window (16 pixels, C channels):  S[c1][c2] = sum_p q^[p][c1] k^[p][c2],  P = softmax_c2(S),
// O[p][c1] = sum_c2 P[c1][c2] V[p][c2].  All five small products of the backward are expressed through one
// helper (tile_mma) that takes its operands from LDS tiles with arbitrary strides, so a transposed operand is a
// stride swap, not a data movement.  The C x C attention matrix never leaves the CU; the backward recomputes it
// from qkv.  Workgroup = one wave (64 threads): tiles are wave-private and need no cross-wave barriers.
//
// Reference site replaced: enhanced_generator.py:22-35,39-42 (view/permute/contiguous window partition, two
// F.normalize, two batched matmuls, softmax, inverse permute).
#include "common.h"

namespace mstg {

// acc[mf][nf] += A (16*MF x K) * B (K x 16*NF);  A(m,k) = Ap[m*a_sm + k*a_sk],  B(k,n) = Bp[k*b_sk + n*b_sn]
template <int MF, int NF>
__device__ __forceinline__ void tile_mma(f32x4 (&acc)[MF][NF], const float* Ap, int a_sm, int a_sk, const float* Bp, int b_sk,
                                         int b_sn, int K, int lane) {
    const int i = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + g;
        float a[MF], b[NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a[mf] = Ap[(16 * mf + i) * a_sm + k * a_sk];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) b[nf] = Bp[k * b_sk + (16 * nf + i) * b_sn];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = mfma16(a[mf], b[nf], acc[mf][nf]);
    }
}

template <int MF, int NF>
__device__ __forceinline__ void tile_zero(f32x4 (&acc)[MF][NF]) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// D(m = 16mf + 4g + r, n = 16nf + i)  ->  Dp[m*d_sm + n*d_sn]
template <int MF, int NF>
__device__ __forceinline__ void tile_store(const f32x4 (&acc)[MF][NF], float* Dp, int d_sm, int d_sn, int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 4; ++r) Dp[(16 * mf + 4 * g + r) * d_sm + (16 * nf + i) * d_sn] = acc[mf][nf][r];
}

// Reductions over the 16 lanes that share lane >> 4 (one accumulator row lives in 16 lanes).  Written with DPP row operations
// (quad_perm, row_half_mirror, row_mirror: VALU-speed cross-lane moves inside a row of 16): __shfl_xor compiles to
// ds_bpermute_b32 here, a ~100-cycle LDS round trip per step, and the softmax / normalisation chains are four dependent
// steps deep -- time stamps showed them to be 60 % of a 32-channel window's forward cycles.  Every lane ends with the result.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);  // row_half_mirror
    v += dpp_move<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return v;
}

#ifdef MSTG_STAMPS
__device__ unsigned long long g_att_stamps[64 * 8];
#define ATT_STAMP(k) if (threadIdx.x == 0 && wcount == 2 && (blockIdx.x % 31) == 0 && blockIdx.x / 31 < 64) g_att_stamps[(blockIdx.x / 31) * 8 + (k)] = __builtin_amdgcn_s_memtime();
// inside the forward helpers (no window counter there): every window stamps, the last one stays
__device__ unsigned long long g_att_fstamps[64 * 8];
#define ATT_FSTAMP(k) if (threadIdx.x == 0 && (blockIdx.x % 31) == 0 && blockIdx.x / 31 < 64) g_att_fstamps[(blockIdx.x / 31) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define ATT_STAMP(k)
#define ATT_FSTAMP(k)
#endif
#define WAVE_SYNC() __syncthreads() /* one-wave workgroup: orders this wave's LDS writes before its reads */

// CP = C rounded up to a multiple of 16 (16, 32 or 64).  LDS tiles (floats):
//   qkv [16][3*CP + 4]   q^ | k^ | v blocks, zero-padded beyond C
//   P   [CP][CP + 4]
//   invn[2][16]          1/max(||q||,eps), 1/max(||k||,eps) per pixel
template <int CP>
struct AttnTiles {
    static constexpr int LDQ = 3 * CP + 4, LDP = CP + 4, NF = CP / 16;
    static constexpr int QKV = 0, P = 16 * LDQ, INVN = P + CP * LDP, END_FWD = INVN + 32;
    // backward extras: dO [16][CP+4], dQK [16][2*CP+4]
    static constexpr int LDO = CP + 4, LDK = 2 * CP + 4;
    static constexpr int DO = END_FWD, DQK = DO + 16 * LDO, END_BWD = DQK + 16 * LDK;
};

// load one window of an (N,H,W,ctot) tensor into a [16][ld] LDS tile, `nblk` channel blocks of C -> CP padding
template <int CP>
__device__ __forceinline__ void load_window(const float* __restrict__ src, float* tile, int ld, int nblk, int C, int H, int W,
                                            int n, int wy, int wx, int lane, int nthr = 64) {
    const int qpb = CP / 4;  // float4 slots per block (padded)
    const int ctot = nblk * C;
    for (int e = lane; e < 16 * nblk * qpb; e += nthr) {
        const int q = e % qpb, rest = e / qpb;
        const int blk = rest % nblk, p = rest / nblk;
        const int y = 4 * wy + (p >> 2), x = 4 * wx + (p & 3);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (4 * q < C) v = *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + y) * W + x) * ctot + blk * C + 4 * q);
        *reinterpret_cast<f32x4*>(&tile[p * ld + blk * CP + 4 * q]) = v;
    }
}

// Split form of load_window for a C-channel tensor (nblk == 1, C == CP): fetch into registers now, write the LDS tile later.
// A persistent wave issues the fetch of its NEXT window before it starts on the current one, so the global round trip
// (2-3k cycles under load, a fifth of a window's whole processing time) runs behind ~10k cycles of compute.
template <int C>
__device__ __forceinline__ void fetch_window(f32x4 (&r)[C / 16], const float* __restrict__ src, int H, int W, int n, int wy, int wx, int lane) {
    constexpr int qpb = C / 4;
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int e = lane + 64 * k, q = e % qpb, p = e / qpb;
        const int y = 4 * wy + (p >> 2), x = 4 * wx + (p & 3);
        r[k] = *reinterpret_cast<const f32x4*>(src + (((size_t)n * H + y) * W + x) * C + 4 * q);
    }
}
template <int C>
__device__ __forceinline__ void put_window(const f32x4 (&r)[C / 16], float* tile, int ld, int lane) {
    constexpr int qpb = C / 4;
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int e = lane + 64 * k, q = e % qpb, p = e / qpb;
        *reinterpret_cast<f32x4*>(&tile[p * ld + 4 * q]) = r[k];
    }
}

// InstanceNorm + ReLU folded into the window load (the stage's first norm, whose only consumer is this module): the lane's
// channel quad is the same for every element it stages, so it needs (mean, rstd) of 4 channels of the window's image.
struct QuadStats { f32x4 a, b; };  // a = (mean0, rstd0, mean1, rstd1), b = (mean2, rstd2, mean3, rstd3): stats[n][c][2] as stored
template <int C>
__device__ __forceinline__ void fetch_stats(QuadStats& qs, const float* __restrict__ stats, int n, int lane) {
    const float* st = stats + ((size_t)n * C + 4 * (lane % (C / 4))) * 2;
    qs.a = *reinterpret_cast<const f32x4*>(st);
    qs.b = *reinterpret_cast<const f32x4*>(st + 4);
}
template <int C>
__device__ __forceinline__ void put_window_norm(const f32x4 (&r)[C / 16], const QuadStats& qs, float* tile, int ld, int lane) {
    constexpr int qpb = C / 4;
#pragma unroll
    for (int k = 0; k < C / 16; ++k) {
        const int e = lane + 64 * k, q = e % qpb, p = e / qpb;
        f32x4 v = r[k];  // same arithmetic as norm_apply_kernel: (x - mean) * rstd, then ReLU
        v[0] = fmaxf((v[0] - qs.a[0]) * qs.a[1], 0.f);
        v[1] = fmaxf((v[1] - qs.a[2]) * qs.a[3], 0.f);
        v[2] = fmaxf((v[2] - qs.b[0]) * qs.b[1], 0.f);
        v[3] = fmaxf((v[3] - qs.b[2]) * qs.b[3], 0.f);
        *reinterpret_cast<f32x4*>(&tile[p * ld + 4 * q]) = v;
    }
}

// q^ , k^ , P into LDS; returns with tiles ready
template <int CP>
__device__ __forceinline__ void attn_forward_tiles(float* sm, int C, int lane) {
    typedef AttnTiles<CP> T;
    constexpr int NF = T::NF;
    const int i = lane & 15, g = lane >> 4;
    float* qkv = sm + T::QKV;
    float* Ps = sm + T::P;
    float* invn = sm + T::INVN;
    ATT_FSTAMP(2)
    // ---- L2 normalise q and k per pixel over channels (F.normalize: v / max(||v||, 1e-12)) ----------------------
    // lane = (pixel lane >> 2, channel quarter lane & 3): 16-byte LDS accesses and a reduction that stays inside a quad (two DPP
    // quad_perm steps) -- the previous (pixel i, quarter g) split needed cross-row ds_bpermute shuffles and scalar LDS traffic
    {
        constexpr int QW = CP / 4, NV = QW / 4;
        const int p = lane >> 2, cq = lane & 3;
        float* qp = &qkv[p * T::LDQ + cq * QW];
        f32x4 qa[NV], ka[NV];
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            qa[v] = *reinterpret_cast<const f32x4*>(qp + 4 * v);
            ka[v] = *reinterpret_cast<const f32x4*>(qp + CP + 4 * v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sq = fmaf(qa[v][e], qa[v][e], sq);
                sk = fmaf(ka[v][e], ka[v][e], sk);
            }
        }
        sq += dpp_move<0xB1>(sq); sq += dpp_move<0x4E>(sq);
        sk += dpp_move<0xB1>(sk); sk += dpp_move<0x4E>(sk);
        const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            *reinterpret_cast<f32x4*>(qp + 4 * v) = qa[v] * iq;
            *reinterpret_cast<f32x4*>(qp + CP + 4 * v) = ka[v] * ik;
        }
        if (cq == 0) { invn[p] = iq; invn[16 + p] = ik; }
    }
    WAVE_SYNC();
    ATT_FSTAMP(3)
    // ---- S[c1][c2] = sum_p q^[p][c1] k^[p][c2] ; softmax over c2 (the 16 lanes of a row x NF fragments) ---------
    f32x4 s[NF][NF];
    tile_zero<NF, NF>(s);
    tile_mma<NF, NF>(s, qkv, 1, T::LDQ, qkv + CP, T::LDQ, 1, 16, lane);
#pragma unroll
    for (int mf = 0; mf < NF; ++mf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if (16 * nf + i >= C) s[mf][nf][r] = -INFINITY;
                mx = fmaxf(mx, s[mf][nf][r]);
            }
            mx = row16_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                s[mf][nf][r] = __expf(s[mf][nf][r] - mx);
                sum += s[mf][nf][r];
            }
            sum = row16_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) s[mf][nf][r] *= inv;
        }
    }
    tile_store<NF, NF>(s, Ps, T::LDP, 1, lane);
    WAVE_SYNC();
    ATT_FSTAMP(4)
}

template <int CP>
__global__ __launch_bounds__(64) void attn_core_fwd_kernel(const float* __restrict__ qkv_g, float* __restrict__ o_g, int N, int H,
                                                           int W, int C) {
    typedef AttnTiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        WAVE_SYNC();
        load_window<CP>(qkv_g, sm + T::QKV, T::LDQ, 3, C, H, W, n, wy, wx, lane);
        WAVE_SYNC();
        attn_forward_tiles<CP>(sm, C, lane);
        // ---- O^T[c1][p] = sum_c2 P[c1][c2] V[p][c2] : rows = channels so a lane owns 4 consecutive channels ------
        f32x4 o[NF][1];
        tile_zero<NF, 1>(o);
        tile_mma<NF, 1>(o, sm + T::P, T::LDP, 1, sm + T::QKV + 2 * CP, 1, T::LDQ, CP, lane);
        const int y = 4 * wy + (i >> 2), x = 4 * wx + (i & 3);
        float* dst = o_g + (((size_t)n * H + y) * W + x) * C;
#pragma unroll
        for (int mf = 0; mf < NF; ++mf) {
            const int c = 16 * mf + 4 * g;
            if (c < C) *reinterpret_cast<f32x4*>(dst + c) = o[mf][0];
        }
    }
}

template <int CP>
__global__ __launch_bounds__(64) void attn_core_bwd_kernel(const float* __restrict__ qkv_g, const float* __restrict__ do_g,
                                                           float* __restrict__ dqkv_g, int N, int H, int W, int C) {
    typedef AttnTiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    float* qkv = sm + T::QKV;
    float* Ps = sm + T::P;
    float* invn = sm + T::INVN;
    float* dOs = sm + T::DO;
    float* dQK = sm + T::DQK;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        WAVE_SYNC();
        load_window<CP>(qkv_g, qkv, T::LDQ, 3, C, H, W, n, wy, wx, lane);
        load_window<CP>(do_g, dOs, T::LDO, 1, C, H, W, n, wy, wx, lane);
        WAVE_SYNC();
        attn_forward_tiles<CP>(sm, C, lane);  // q^, k^ (in place), P, inverse norms

        const int y = 4 * wy + (i >> 2), x = 4 * wx + (i & 3);
        float* dst = dqkv_g + (((size_t)n * H + y) * W + x) * 3 * C;

        // ---- dP[c1][c2] = sum_p dO[p][c1] V[p][c2] ; dS = P * (dP - rowsum(dP * P)) (kept in registers) ------------
        f32x4 ds[NF][NF];
        tile_zero<NF, NF>(ds);
        tile_mma<NF, NF>(ds, dOs, 1, T::LDO, qkv + 2 * CP, T::LDQ, 1, 16, lane);
#pragma unroll
        for (int mf = 0; mf < NF; ++mf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr[NF], dot = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    pr[nf] = Ps[(16 * mf + 4 * g + r) * T::LDP + 16 * nf + i];
                    dot += pr[nf] * ds[mf][nf][r];
                }
                dot = row16_sum(dot);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) ds[mf][nf][r] = pr[nf] * (ds[mf][nf][r] - dot);
            }
        }
        // ---- dV^T[c2][p] = sum_c1 P[c1][c2] dO[p][c1]  (needs P: do it before P is overwritten by dS) -------------
        {
            f32x4 dv[NF][1];
            tile_zero<NF, 1>(dv);
            tile_mma<NF, 1>(dv, Ps, 1, T::LDP, dOs, 1, T::LDO, CP, lane);
#pragma unroll
            for (int mf = 0; mf < NF; ++mf) {
                const int c = 16 * mf + 4 * g;
                if (c < C) *reinterpret_cast<f32x4*>(dst + 2 * C + c) = dv[mf][0];
            }
        }
        WAVE_SYNC();
        tile_store<NF, NF>(ds, Ps, T::LDP, 1, lane);  // P <- dS
        WAVE_SYNC();
        // ---- dq^[p][c1] = sum_c2 dS[c1][c2] k^[p][c2] ; dk^[p][c2] = sum_c1 dS[c1][c2] q^[p][c1] -------------------
        // normalisation backward needs a dot over channels per pixel: keep pixels on the accumulator rows.
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            f32x4 d[1][NF];
            tile_zero<1, NF>(d);
            if (which == 0) tile_mma<1, NF>(d, qkv + CP, T::LDQ, 1, Ps, 1, T::LDP, CP, lane);  // A = k^ (p x c2), B = dS^T
            else tile_mma<1, NF>(d, qkv, T::LDQ, 1, Ps, T::LDP, 1, CP, lane);                  // A = q^ (p x c1), B = dS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 4 * g + r;
                float hat[NF], dot = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    hat[nf] = qkv[p * T::LDQ + which * CP + 16 * nf + i];
                    dot += hat[nf] * d[0][nf][r];
                }
                dot = row16_sum(dot);
                const float inv = invn[which * 16 + p];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) dQK[p * T::LDK + which * CP + 16 * nf + i] = (d[0][nf][r] - hat[nf] * dot) * inv;
            }
        }
        WAVE_SYNC();
        // ---- write dq | dk (pixel-major tile -> NHWC) ---------------------------------------------------------------
        for (int e = lane; e < 16 * 2 * (CP / 4); e += 64) {
            const int q = e % (CP / 4), rest = e / (CP / 4);
            const int blk = rest % 2, p = rest / 2;
            if (4 * q < C) {
                const int yy = 4 * wy + (p >> 2), xx = 4 * wx + (p & 3);
                *reinterpret_cast<f32x4*>(dqkv_g + (((size_t)n * H + yy) * W + xx) * 3 * C + blk * C + 4 * q) =
                    *reinterpret_cast<const f32x4*>(&dQK[p * T::LDK + blk * CP + 4 * q]);
            }
        }
    }
}

// =====================================================================================================================
// Fully fused LocalAttention (C = 16 or 32): qkv 1x1 conv + window attention + proj 1x1 conv in ONE kernel per direction.
// x is read once and y written once (2C floats per pixel instead of the 9C of the unfused chain); q/k/v, the C x C
// attention matrix and o never leave the CU.  The 1x1-conv weights live in LDS for the whole (persistent) wave; in the
// backward their gradients are accumulated in MFMA accumulators across all windows a wave processes and reduced over
// waves by a fixed-order second kernel.
// =====================================================================================================================
template <int C>
struct FusedTiles {
    typedef AttnTiles<C> T;  // QKV, P, INVN first: attn_forward_tiles() works on them unchanged
    static constexpr int NF = C / 16, LDX = C + 4;
    static constexpr int XS = T::INVN + 32, OS = XS + 16 * LDX, END_FWD = OS + 16 * LDX;
    // backward: dY and dO are dead before the first element of dQKV is written, so dQKV overlays them
    static constexpr int DYS = END_FWD, DOS = DYS + 16 * LDX, DQKV = END_FWD;
    static constexpr int END_BWD = END_FWD + (16 * T::LDQ > 32 * LDX ? 16 * T::LDQ : 32 * LDX);
    static constexpr int SLAB = 4 * C * C + 4 * C;  // dWqkv | dWp | dbqkv | dbp
};

// x window -> Xs ; qkv = x Wqkv^T + b -> QKV tile ; q^,k^,P (attn_forward_tiles) ; O -> Os.  Ends with the tiles ready.
template <int C>
__device__ __forceinline__ void fused_forward_tiles(float* sm, const float* __restrict__ x, const float* __restrict__ wqkv,
                                                    const float* __restrict__ bqkv, int H, int W, int n, int wy, int wx, int lane) {
    typedef FusedTiles<C> F;
    typedef AttnTiles<C> T;
    constexpr int NF = F::NF;
    const int i = lane & 15, g = lane >> 4;
    ATT_FSTAMP(0)
    if (x) load_window<C>(x, sm + F::XS, F::LDX, 1, C, H, W, n, wy, wx, lane);  // x == nullptr: the caller has filled Xs already
    WAVE_SYNC();
    ATT_FSTAMP(1)
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        f32x4 acc[1][NF];
        tile_zero<1, NF>(acc);
        // A = X (pixel x ci), B(k = ci, n = j) = Wqkv[j][ci]
        tile_mma<1, NF>(acc, sm + F::XS, F::LDX, 1, wqkv + blk * C * C, 1, C, C, lane);  // filter straight from L1/L2
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const float b = bqkv[blk * C + 16 * nf + i];
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[T::QKV + (4 * g + r) * T::LDQ + blk * C + 16 * nf + i] = acc[0][nf][r] + b;
        }
    }
    WAVE_SYNC();
    attn_forward_tiles<C>(sm, C, lane);
    // O^T[c1][p] = sum_c2 P[c1][c2] V[p][c2], stored as Os[p][c1]
    f32x4 o[NF][1];
    tile_zero<NF, 1>(o);
    tile_mma<NF, 1>(o, sm + T::P, T::LDP, 1, sm + T::QKV + 2 * C, 1, T::LDQ, C, lane);
    tile_store<NF, 1>(o, sm + F::OS, 1, F::LDX, lane);
    WAVE_SYNC();
    ATT_FSTAMP(5)
}

template <int C, bool NORM>
__global__ __launch_bounds__(64) void attn_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wqkv,
                                                            const float* __restrict__ bqkv, const float* __restrict__ wp,
                                                            const float* __restrict__ bp, float* __restrict__ y, int N, int H, int W,
                                                            const float* __restrict__ in_stats) {
    typedef FusedTiles<C> F;
    constexpr int NF = F::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    f32x4 xr[C / 16];
    QuadStats qs;
    if ((int)blockIdx.x < nwin) {
        const int w = blockIdx.x;
        fetch_window<C>(xr, x, H, W, w / (nwx * nwy), (w / nwx) % nwy, w % nwx, lane);
        if (NORM) fetch_stats<C>(qs, in_stats, w / (nwx * nwy), lane);
    }
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        WAVE_SYNC();
        if (NORM) put_window_norm<C>(xr, qs, sm + F::XS, F::LDX, lane);
        else put_window<C>(xr, sm + F::XS, F::LDX, lane);
        {
            const int w2 = w + gridDim.x < nwin ? w + gridDim.x : w;  // next window of this wave (re-fetch the same one at the end)
            fetch_window<C>(xr, x, H, W, w2 / (nwx * nwy), (w2 / nwx) % nwy, w2 % nwx, lane);
            if (NORM) fetch_stats<C>(qs, in_stats, w2 / (nwx * nwy), lane);
        }
        fused_forward_tiles<C>(sm, nullptr, wqkv, bqkv, H, W, n, wy, wx, lane);
        // Y^T[co][p] = sum_c Wp[co][c] O[p][c] + b : rows = channels -> one 16-byte store per lane and fragment
        f32x4 yv[NF][1];
        tile_zero<NF, 1>(yv);
        tile_mma<NF, 1>(yv, wp, C, 1, sm + F::OS, 1, F::LDX, C, lane);
        const int py = 4 * wy + (i >> 2), px = 4 * wx + (i & 3);
        float* dst = y + (((size_t)n * H + py) * W + px) * C;
#pragma unroll
        for (int mf = 0; mf < NF; ++mf) {
            const int c = 16 * mf + 4 * g;
            *reinterpret_cast<f32x4*>(dst + c) = yv[mf][0] + *reinterpret_cast<const f32x4*>(bp + c);
        }
    }
}

// NORM: x is the RAW tensor in front of the stage's InstanceNorm + ReLU (normalised while staged), and the kernel also emits what
// that norm's backward needs from a pass over dx: per (image, channel) the sums of dx * [z > 0] and dx * [z > 0] * z.  A wave takes
// its windows in runs of kblk consecutive ones (kblk divides the windows per image, so a run stays inside one image) and writes
// one row per run: nsum[run][2][C], runs of an image being consecutive.  kblk = 1 without NORM: the plain strided order.
template <int C, bool NORM>
__global__ __launch_bounds__(64) void attn_fused_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wqkv,
                                                            const float* __restrict__ bqkv, const float* __restrict__ wp,
                                                            const float* __restrict__ bp, const float* __restrict__ dy,
                                                            float* __restrict__ dx, float* __restrict__ partial, int N, int H, int W,
                                                            const float* __restrict__ in_stats, float* __restrict__ nsum, int kblk) {
    typedef FusedTiles<C> F;
    typedef AttnTiles<C> T;
    constexpr int NF = F::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    float* qkv = sm + T::QKV;
    float* Ps = sm + T::P;
    float* invn = sm + T::INVN;
    float* dYs = sm + F::DYS;
    float* dOs = sm + F::DOS;
    float* dQKV = sm + F::DQKV;

    // weight-gradient accumulators, alive across every window this wave processes
    f32x4 gwq[3][NF][NF], gwp[NF][NF];
    float gbq[3][NF], gbp[NF];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        tile_zero<NF, NF>(gwq[b]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) gbq[b][nf] = 0.f;
    }
    tile_zero<NF, NF>(gwp);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) gbp[nf] = 0.f;

    f32x4 xr[C / 16], dyr[C / 16];
    QuadStats qs;
    f32x4 ns1[NORM ? NF : 1], ns2[NORM ? NF : 1];  // this wave's norm-backward sums over the current run of windows
    if (NORM) {
#pragma unroll
        for (int mf = 0; mf < NF; ++mf) ns1[mf] = ns2[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto flush_nsum = [&](int run) {
        float* row = nsum + (size_t)run * 2 * C;
#pragma unroll
        for (int mf = 0; mf < (NORM ? NF : 1); ++mf) {
            f32x4 a, b;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = row16_sum(ns1[mf][r]); b[r] = row16_sum(ns2[mf][r]); }
            if (i == 0) {
                *reinterpret_cast<f32x4*>(row + 16 * mf + 4 * g) = a;
                *reinterpret_cast<f32x4*>(row + C + 16 * mf + 4 * g) = b;
            }
            ns1[mf] = ns2[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int nrun = nwin / kblk;  // kblk divides nwin
    if ((int)blockIdx.x < nrun) {
        const int w = blockIdx.x * kblk;
        fetch_window<C>(dyr, dy, H, W, w / (nwx * nwy), (w / nwx) % nwy, w % nwx, lane);
        fetch_window<C>(xr, x, H, W, w / (nwx * nwy), (w / nwx) % nwy, w % nwx, lane);
        if (NORM) fetch_stats<C>(qs, in_stats, w / (nwx * nwy), lane);
    }
    int wcount = 0;
    for (int run = blockIdx.x, j = 0; run < nrun; ++wcount) {
        const int w = run * kblk + j;
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        // the window after this one: the run's next, or the first of this wave's next run (at the very end: this one again)
        const bool run_ends = j + 1 == kblk;
        const int w2 = !run_ends ? w + 1 : (run + (int)gridDim.x < nrun ? (run + (int)gridDim.x) * kblk : w);
        WAVE_SYNC();
        ATT_STAMP(0)
        put_window<C>(dyr, dYs, F::LDX, lane);
        if (NORM) put_window_norm<C>(xr, qs, sm + F::XS, F::LDX, lane);
        else put_window<C>(xr, sm + F::XS, F::LDX, lane);
        {   // in flight behind this window's compute
            const int n2 = w2 / (nwx * nwy), wy2 = (w2 / nwx) % nwy, wx2 = w2 % nwx;
            fetch_window<C>(dyr, dy, H, W, n2, wy2, wx2, lane);
            fetch_window<C>(xr, x, H, W, n2, wy2, wx2, lane);
            if (NORM) fetch_stats<C>(qs, in_stats, n2, lane);
        }
        fused_forward_tiles<C>(sm, nullptr, wqkv, bqkv, H, W, n, wy, wx, lane);  // Xs (filled above), q^, k^, v, P, inverse norms, Os
        ATT_STAMP(1)

        // ---- proj backward: dO = dY Wp ; dWp += dY^T O ; dbp += colsum(dY) --------------------------------------------
        {
            f32x4 d[1][NF];
            tile_zero<1, NF>(d);
            tile_mma<1, NF>(d, dYs, F::LDX, 1, wp, C, 1, C, lane);  // B(k = co, n = c) = Wp[co][c]
            tile_store<1, NF>(d, dOs, F::LDX, 1, lane);
            tile_mma<NF, NF>(gwp, dYs, 1, F::LDX, sm + F::OS, F::LDX, 1, 16, lane);  // A = dY^T (co x p), B = O (p x c)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) gbp[nf] += dYs[(4 * g + r) * F::LDX + 16 * nf + i];
        }
        WAVE_SYNC();
        ATT_STAMP(2)
        // ---- attention core backward (same algebra as attn_core_bwd_kernel), results into the dQKV tile -------------------
        f32x4 ds[NF][NF];
        tile_zero<NF, NF>(ds);
        tile_mma<NF, NF>(ds, dOs, 1, F::LDX, qkv + 2 * C, T::LDQ, 1, 16, lane);  // dP[c1][c2] = sum_p dO[p][c1] V[p][c2]
#pragma unroll
        for (int mf = 0; mf < NF; ++mf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr[NF], dot = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    pr[nf] = Ps[(16 * mf + 4 * g + r) * T::LDP + 16 * nf + i];
                    dot += pr[nf] * ds[mf][nf][r];
                }
                dot = row16_sum(dot);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) ds[mf][nf][r] = pr[nf] * (ds[mf][nf][r] - dot);
            }
        }
        {
            f32x4 dv[1][NF];  // dV[p][c2] = sum_c1 dO[p][c1] P[c1][c2]
            tile_zero<1, NF>(dv);
            tile_mma<1, NF>(dv, dOs, F::LDX, 1, Ps, T::LDP, 1, C, lane);
            tile_store<1, NF>(dv, dQKV + 2 * C, T::LDQ, 1, lane);
        }
        WAVE_SYNC();
        ATT_STAMP(3)
        tile_store<NF, NF>(ds, Ps, T::LDP, 1, lane);  // P <- dS
        WAVE_SYNC();
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            f32x4 d[1][NF];
            tile_zero<1, NF>(d);
            if (which == 0) tile_mma<1, NF>(d, qkv + C, T::LDQ, 1, Ps, 1, T::LDP, C, lane);  // dq^ = k^ dS^T
            else tile_mma<1, NF>(d, qkv, T::LDQ, 1, Ps, T::LDP, 1, C, lane);                 // dk^ = q^ dS
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 4 * g + r;
                float hat[NF], dot = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    hat[nf] = qkv[p * T::LDQ + which * C + 16 * nf + i];
                    dot += hat[nf] * d[0][nf][r];
                }
                dot = row16_sum(dot);
                const float inv = invn[which * 16 + p];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) dQKV[p * T::LDQ + which * C + 16 * nf + i] = (d[0][nf][r] - hat[nf] * dot) * inv;
            }
        }
        WAVE_SYNC();
        ATT_STAMP(4)
        // ---- qkv conv backward: dX = dQKV Wqkv ; dWqkv += dQKV^T X ; dbqkv += colsum(dQKV) --------------------------------
        {
            f32x4 d[NF][1];  // dX^T[ci][p] = sum_j Wqkv[j][ci] dQKV[p][j]
            tile_zero<NF, 1>(d);
            tile_mma<NF, 1>(d, wqkv, 1, C, dQKV, 1, T::LDQ, 3 * C, lane);
            const int py = 4 * wy + (i >> 2), px = 4 * wx + (i & 3);
            float* dst = dx + (((size_t)n * H + py) * W + px) * C;
#pragma unroll
            for (int mf = 0; mf < NF; ++mf) *reinterpret_cast<f32x4*>(dst + 16 * mf + 4 * g) = d[mf][0];
            if (NORM) {  // Xs holds z = relu(x^): where z > 0 it IS x^, elsewhere the element contributes nothing
#pragma unroll
                for (int mf = 0; mf < NF; ++mf) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(&sm[F::XS + i * F::LDX + 16 * mf + 4 * g]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gg = z[r] > 0.f ? d[mf][0][r] : 0.f;
                        ns1[mf][r] += gg;
                        ns2[mf][r] += gg * z[r];
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            tile_mma<NF, NF>(gwq[b], dQKV + b * C, 1, T::LDQ, sm + F::XS, F::LDX, 1, 16, lane);  // A = dQKV_b^T (j x p), B = X (p x ci)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) gbq[b][nf] += dQKV[(4 * g + r) * T::LDQ + b * C + 16 * nf + i];
        }
        ATT_STAMP(5)
        ATT_STAMP(6)
        if (run_ends) {
            if (NORM) flush_nsum(run);
            run += gridDim.x;
            j = 0;
        } else {
            ++j;
        }
    }
    // ---- this wave's slab: dWqkv (3C x C) | dWp (C x C) | dbqkv (3C) | dbp (C) ---------------------------------------------
    float* out = partial + (size_t)blockIdx.x * F::SLAB;
#pragma unroll
    for (int b = 0; b < 3; ++b) tile_store<NF, NF>(gwq[b], out + b * C * C, C, 1, lane);
    tile_store<NF, NF>(gwp, out + 3 * C * C, C, 1, lane);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            float v = gbq[b][nf];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) out[4 * C * C + b * C + 16 * nf + i] = v;
        }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        float v = gbp[nf];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) out[4 * C * C + 3 * C + 16 * nf + i] = v;
    }
}

// Where the fused attention's reduced parameter gradients go: the flat vector `out` (dWqkv | dWp | dbqkv | dbp), or -- `dst.p[0]` set --
// the four parameter-gradient tensors themselves, optionally accumulating (the *_direct entry points: autograd then has nothing to
// add, four launches less per attention and backward pass).
struct AttnGradDst {
    float* p[4];  // dWqkv (3C x C), dWp (C x C), dbqkv (3C), dbp (C)
    int accumulate;
};
thread_local AttnGradDst t_attn_dst = {{nullptr, nullptr, nullptr, nullptr}, 0};

// out[e] = sum_s partial[s][e], fixed order; 16 outputs x 16 strided rows per workgroup
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int S, int n,
                                                          AttnGradDst dst, int C) {
    __shared__ float sh[16][17];
    const int e = threadIdx.x & 15, row = threadIdx.x >> 4, idx = blockIdx.x * 16 + e;
    float sum = 0.f;
    if (idx < n) {  // eight loads in flight per trip; the additions keep the order sp = row, row + 16, ...
        int sp = row;
        for (; sp + 7 * 16 < S; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(sp + 16 * k) * n + idx];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; sp < S; sp += 16) sum += partial[(size_t)sp * n + idx];
    }
    sh[row][e] = sum;
    __syncthreads();
    if (row == 0 && idx < n) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) r += sh[k][e];
        if (!dst.p[0]) {
            out[idx] = r;
        } else {
            const int c2 = C * C;
            float* o = idx < 3 * c2 ? dst.p[0] + idx : (idx < 4 * c2 ? dst.p[1] + (idx - 3 * c2) : (idx < 4 * c2 + 3 * C ? dst.p[2] + (idx - 4 * c2)
                                                                                                                      : dst.p[3] + (idx - 4 * c2 - 3 * C)));
            *o = dst.accumulate ? *o + r : r;
        }
    }
}

// sums[n][sp][2][C] = sum of the image's rows sp * R / S ... of nsum[n][R][2][C] (R rows per image, S splits), fixed order
__global__ __launch_bounds__(256) void nsum_reduce_kernel(const float* __restrict__ nsum, float* __restrict__ sums, int R, int S, int C2) {
    __shared__ f32x4 sh[256];
    const int n = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
    const int Q = C2 / 4, q = tid % Q, rg = tid / Q, nrg = 256 / Q;
    const int per = (R + S - 1) / S, r0 = sp * per, r1 = min(R, r0 + per);
    const f32x4* src = reinterpret_cast<const f32x4*>(nsum) + (size_t)n * R * Q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + rg; r < r1; r += nrg) acc += src[(size_t)r * Q + q];
    sh[tid] = acc;
    __syncthreads();
    if (rg == 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < nrg; ++k) t += sh[k * Q + q];
        reinterpret_cast<f32x4*>(sums)[((size_t)n * S + sp) * Q + q] = t;
    }
}

int attn_reg_bwd_blocks(int C);
int attn_reg_bwd_waves(int C);
// windows per run of the NORM backward: the largest divisor of the windows per image that is <= 16 -- and small enough that every
// wave of the backward gets about two runs (a run is the unit of work there: with 16-window runs a 16-image batch at 128 x 128
// gave half the waves nothing to do)
static int norm_run_len(int N, int H, int W, int C) {
    const int wpi = (H / 4) * (W / 4);
    const long waves = (long)attn_reg_bwd_blocks(C) * attn_reg_bwd_waves(C);
    long cap = ((long)N * wpi) / (2 * waves);
    if (cap > 16) cap = 16;
    if (cap < 1) cap = 1;
    for (int k = (int)cap; k > 1; --k)
        if (wpi % k == 0) return k;
    return 1;
}
constexpr int NSUM_SPLIT = 16;  // rows of norm sums per image handed to mstg_norm_bwd_apply

static int fused_blocks(int N, int H, int W) {
    const int nwin = N * (H / 4) * (W / 4);
    return nwin < 2048 ? nwin : 2048;  // one-wave workgroups: 8 per CU
}

// The register-resident kernels of csrc/attention_reg.hip (round 3): the default; MSTG_ATTN_REG=0 selects the LDS-tile kernels above.
int attn_reg_fwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* bp,
                 float* y, int N, int H, int W, hipStream_t st);
int attn_reg_bwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* dy,
                 float* dx, float* partial, float* nsum, int kblk, int nblocks, int N, int H, int W, hipStream_t st);
static bool attn_reg_on() {
    const char* e = env_get(ENV_ATTN_REG);
    return !(e && e[0] == '0');
}

// in_stats != nullptr: x is the raw tensor in front of InstanceNorm + ReLU; backward then also fills norm_sums[N][2][C]
// (workspace layout: [weight-gradient slabs: nb * SLAB][nsum rows: N * nb * 2C])
template <int C>
struct FusedSlab {
    static constexpr int SLAB = 4 * C * C + 4 * C;  // dWqkv | dWp | dbqkv | dbp
};

template <int C>
static int launch_fused(bool bwd, const float* x, const float* wqkv, const float* bqkv, const float* wp, const float* bp,
                        const float* dy, float* out, float* grads, float* partial, int N, int H, int W, hipStream_t st,
                        const float* in_stats = nullptr, float* norm_sums = nullptr) {
    typedef FusedSlab<C> F;
    const int nb = fused_blocks(N, H, W);
    if (attn_reg_on() || C > 32) {  // C = 64 exists only as the register-chain kernel
        if (!bwd) return attn_reg_fwd(C, x, in_stats, wqkv, bqkv, wp, bp, out, N, H, W, st);
        // slabs: one per workgroup of four waves, never more workgroups than runs / 4 (and so never more than the nb slabs the
        // workspace was sized for); the nsum rows sit behind the nb-slab region as before
        const int kblk = in_stats ? norm_run_len(N, H, W, C) : 1, R = (H / 4) * (W / 4) / kblk, nrun = N * R;
        int nbr = attn_reg_bwd_blocks(C);
        if (nbr > cdiv(nrun, attn_reg_bwd_waves(C))) nbr = cdiv(nrun, attn_reg_bwd_waves(C));
        float* nsum = partial + (size_t)nb * F::SLAB;
        if (int rc = attn_reg_bwd(C, x, in_stats, wqkv, bqkv, wp, dy, out, partial, in_stats ? nsum : nullptr, kblk, nbr, N, H, W, st)) return rc;
        if (in_stats) {
            MSTG_LAUNCH(nsum_reduce_kernel, dim3(NSUM_SPLIT, N), dim3(256), 0, st, (const float*)nsum, norm_sums, R, NSUM_SPLIT, 2 * C);
            MSTG_CHECK_LAUNCH("nsum_reduce_kernel");
        }
        MSTG_LAUNCH(slab_reduce_kernel, dim3(cdiv(F::SLAB, 16)), dim3(256), 0, st, (const float*)partial, grads, nbr, F::SLAB, t_attn_dst, C);
        MSTG_CHECK_LAUNCH("slab_reduce_kernel");
        return MSTG_OK;
    }
    if constexpr (C <= 32) {
        typedef FusedTiles<C> T;
        if (!bwd) {
            if (in_stats)
                MSTG_LAUNCH((attn_fused_fwd_kernel<C, true>), dim3(nb), dim3(64), (size_t)T::END_FWD * sizeof(float), st, x, wqkv, bqkv, wp, bp,
                                   out, N, H, W, in_stats);
            else
                MSTG_LAUNCH((attn_fused_fwd_kernel<C, false>), dim3(nb), dim3(64), (size_t)T::END_FWD * sizeof(float), st, x, wqkv, bqkv, wp, bp,
                                   out, N, H, W, in_stats);
            MSTG_CHECK_LAUNCH("attn_fused_fwd_kernel");
            return MSTG_OK;
        }
        if (in_stats) {
            float* nsum = partial + (size_t)nb * F::SLAB;
            const int kblk = norm_run_len(N, H, W, C), R = (H / 4) * (W / 4) / kblk;
            MSTG_LAUNCH((attn_fused_bwd_kernel<C, true>), dim3(nb), dim3(64), (size_t)T::END_BWD * sizeof(float), st, x, wqkv, bqkv, wp, bp, dy,
                               out, partial, N, H, W, in_stats, nsum, kblk);
            MSTG_CHECK_LAUNCH("attn_fused_bwd_kernel<norm>");
            MSTG_LAUNCH(nsum_reduce_kernel, dim3(NSUM_SPLIT, N), dim3(256), 0, st, (const float*)nsum, norm_sums, R, NSUM_SPLIT, 2 * C);
            MSTG_CHECK_LAUNCH("nsum_reduce_kernel");
        } else {
            MSTG_LAUNCH((attn_fused_bwd_kernel<C, false>), dim3(nb), dim3(64), (size_t)T::END_BWD * sizeof(float), st, x, wqkv, bqkv, wp, bp, dy,
                               out, partial, N, H, W, in_stats, (float*)nullptr, 1);
            MSTG_CHECK_LAUNCH("attn_fused_bwd_kernel");
        }
        MSTG_LAUNCH(slab_reduce_kernel, dim3(cdiv(F::SLAB, 16)), dim3(256), 0, st, (const float*)partial, grads, nb, F::SLAB, t_attn_dst, C);
        MSTG_CHECK_LAUNCH("slab_reduce_kernel");
    }
    return MSTG_OK;
}

// =====================================================================================================================
// Row-blocked core for wide layers (64 < C <= 256: the class-default channels=64 generator has C_l = 128 and 256).  The
// C x C attention matrix no longer fits on chip at once (256 KiB at C = 256), so it is produced 16 rows (c1) at a time:
// softmax needs complete rows only, and everything that sums over c1 (dV, dk^) is accumulated across the row blocks in
// registers.  One wave per window as before; LDS holds q^|k^|v, one 16 x C block of P / dS, dO and dq^.
// =====================================================================================================================
template <int CP>
struct BlkTiles {
    static constexpr int NF = CP / 16, LDQ = 3 * CP + 4, LDP = CP + 4;
    static constexpr int QKV = 0, P = 16 * LDQ, INVN = P + 16 * LDP, END_FWD = INVN + 32;
    static constexpr int DO = END_FWD, DQ = DO + 16 * LDP, DOTQ = DQ + 16 * LDP, END_BWD = DOTQ + 16;
};

// normalise q, k in place (tile [16][LDQ]) and leave 1/||.|| in invn[2][16]
template <int CP>
__device__ __forceinline__ void blk_normalise(float* qkv, float* invn, int lane) {
    typedef BlkTiles<CP> T;
    const int i = lane & 15, g = lane >> 4;
    float sq = 0.f, sk = 0.f;
    const int c0 = g * (CP / 4);
    for (int c = c0; c < c0 + CP / 4; ++c) {
        const float a = qkv[i * T::LDQ + c], b = qkv[i * T::LDQ + CP + c];
        sq += a * a;
        sk += b * b;
    }
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    sk += __shfl_xor(sk, 16, 64); sk += __shfl_xor(sk, 32, 64);
    const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
    for (int c = c0; c < c0 + CP / 4; ++c) {
        qkv[i * T::LDQ + c] *= iq;
        qkv[i * T::LDQ + CP + c] *= ik;
    }
    if (g == 0) { invn[i] = iq; invn[16 + i] = ik; }
}

// P block (rows c1 in [16b, 16b+16)) -> registers (softmaxed) and LDS tile Ps[16][LDP]
template <int CP>
__device__ __forceinline__ void blk_softmax_rows(f32x4 (&s)[1][CP / 16], float* sm, int b, int C, int lane) {
    typedef BlkTiles<CP> T;
    constexpr int NF = T::NF;
    const int i = lane & 15;
    tile_zero<1, NF>(s);
    tile_mma<1, NF>(s, sm + T::QKV + 16 * b, 1, T::LDQ, sm + T::QKV + CP, T::LDQ, 1, 16, lane);  // q^_blk^T k^
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            if (16 * nf + i >= C) s[0][nf][r] = -INFINITY;
            mx = fmaxf(mx, s[0][nf][r]);
        }
        mx = row16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            s[0][nf][r] = __expf(s[0][nf][r] - mx);
            sum += s[0][nf][r];
        }
        sum = row16_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) s[0][nf][r] *= inv;
    }
    tile_store<1, NF>(s, sm + T::P, T::LDP, 1, lane);
}

template <int CP>
__global__ __launch_bounds__(64) void attn_core_fwd_blk_kernel(const float* __restrict__ qkv_g, float* __restrict__ o_g, int N, int H,
                                                               int W, int C) {
    typedef BlkTiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        WAVE_SYNC();
        load_window<CP>(qkv_g, sm + T::QKV, T::LDQ, 3, C, H, W, n, wy, wx, lane);
        WAVE_SYNC();
        blk_normalise<CP>(sm + T::QKV, sm + T::INVN, lane);
        WAVE_SYNC();
        const int y = 4 * wy + (i >> 2), x = 4 * wx + (i & 3);
        float* dst = o_g + (((size_t)n * H + y) * W + x) * C;
        for (int b = 0; 16 * b < C; ++b) {
            f32x4 s[1][NF];
            blk_softmax_rows<CP>(s, sm, b, C, lane);
            WAVE_SYNC();
            f32x4 o[1][1];  // O^T[c1 in block][p] = sum_c2 P[c1][c2] V[p][c2]
            tile_zero<1, 1>(o);
            tile_mma<1, 1>(o, sm + T::P, T::LDP, 1, sm + T::QKV + 2 * CP, 1, T::LDQ, CP, lane);
            const int c = 16 * b + 4 * g;
            if (c < C) *reinterpret_cast<f32x4*>(dst + c) = o[0][0];
            WAVE_SYNC();
        }
    }
}

template <int CP>
__global__ __launch_bounds__(64) void attn_core_bwd_blk_kernel(const float* __restrict__ qkv_g, const float* __restrict__ do_g,
                                                               float* __restrict__ dqkv_g, int N, int H, int W, int C) {
    typedef BlkTiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    float* qkv = sm + T::QKV;
    float* Ps = sm + T::P;
    float* invn = sm + T::INVN;
    float* dOs = sm + T::DO;
    float* dQ = sm + T::DQ;     // raw dq^ [16][LDP]
    float* dotq = sm + T::DOTQ; // per pixel sum_c1 q^ dq^
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        WAVE_SYNC();
        load_window<CP>(qkv_g, qkv, T::LDQ, 3, C, H, W, n, wy, wx, lane);
        load_window<CP>(do_g, dOs, T::LDP, 1, C, H, W, n, wy, wx, lane);
        WAVE_SYNC();
        blk_normalise<CP>(qkv, invn, lane);
        WAVE_SYNC();
        f32x4 dv[1][NF], dk[1][NF];  // accumulated over the row blocks: rows = pixels, columns = c2
        tile_zero<1, NF>(dv);
        tile_zero<1, NF>(dk);
        float qdot[4] = {0.f, 0.f, 0.f, 0.f};  // lane (i, g): partial of sum_c1 q^[p][c1] dq^[p][c1] for p = 4g + r
        for (int b = 0; 16 * b < C; ++b) {
            f32x4 s[1][NF];
            blk_softmax_rows<CP>(s, sm, b, C, lane);  // P rows of this block -> Ps
            WAVE_SYNC();
            f32x4 ds[1][NF];  // dP[c1 blk][c2] = sum_p dO[p][c1] V[p][c2]
            tile_zero<1, NF>(ds);
            tile_mma<1, NF>(ds, dOs + 16 * b, 1, T::LDP, qkv + 2 * CP, T::LDQ, 1, 16, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dot = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) dot += s[0][nf][r] * ds[0][nf][r];
                dot = row16_sum(dot);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) ds[0][nf][r] = s[0][nf][r] * (ds[0][nf][r] - dot);
            }
            // dV[p][c2] += sum_{c1 in blk} dO[p][c1] P[c1][c2]   (P still in Ps)
            tile_mma<1, NF>(dv, dOs + 16 * b, T::LDP, 1, Ps, T::LDP, 1, 16, lane);
            WAVE_SYNC();
            tile_store<1, NF>(ds, Ps, T::LDP, 1, lane);  // Ps <- dS block
            WAVE_SYNC();
            // dq^[p][c1 in blk] = sum_c2 dS[c1][c2] k^[p][c2]
            f32x4 d[1][1];
            tile_zero<1, 1>(d);
            tile_mma<1, 1>(d, qkv + CP, T::LDQ, 1, Ps, 1, T::LDP, CP, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 4 * g + r;
                dQ[p * T::LDP + 16 * b + i] = d[0][0][r];
                qdot[r] += qkv[p * T::LDQ + 16 * b + i] * d[0][0][r];
            }
            // dk^[p][c2] += sum_{c1 in blk} dS[c1][c2] q^[p][c1]
            tile_mma<1, NF>(dk, qkv + 16 * b, T::LDQ, 1, Ps, T::LDP, 1, 16, lane);
            WAVE_SYNC();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = row16_sum(qdot[r]);
            if (i == 0) dotq[4 * g + r] = t;
        }
        // dk = (dk^ - k^ (k^ . dk^)) / ||k||, written straight to global; dV likewise
        const int y = 4 * wy + (i >> 2), x = 4 * wx + (i & 3);
        (void)y; (void)x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 4 * g + r;
            float dot = 0.f;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) dot += qkv[p * T::LDQ + CP + 16 * nf + i] * dk[0][nf][r];
            dot = row16_sum(dot);
            const float inv = invn[16 + p];
            const int py = 4 * wy + (p >> 2), px = 4 * wx + (p & 3);
            float* dst = dqkv_g + (((size_t)n * H + py) * W + px) * 3 * C;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int c = 16 * nf + i;
                if (c < C) {
                    dst[C + c] = (dk[0][nf][r] - qkv[p * T::LDQ + CP + c] * dot) * inv;
                    dst[2 * C + c] = dv[0][nf][r];
                }
            }
        }
        WAVE_SYNC();
        // dq = (dq^ - q^ (q^ . dq^)) / ||q||
        for (int e = lane; e < 16 * (CP / 4); e += 64) {
            const int q4 = e % (CP / 4), p = e / (CP / 4);
            if (4 * q4 < C) {
                const float dot = dotq[p], inv = invn[p];
                f32x4 v = *reinterpret_cast<const f32x4*>(&dQ[p * T::LDP + 4 * q4]);
                const f32x4 h = *reinterpret_cast<const f32x4*>(&qkv[p * T::LDQ + 4 * q4]);
                v = (v - h * dot) * inv;
                const int py = 4 * wy + (p >> 2), px = 4 * wx + (p & 3);
                *reinterpret_cast<f32x4*>(dqkv_g + (((size_t)n * H + py) * W + px) * 3 * C + 4 * q4) = v;
            }
        }
    }
}

// =====================================================================================================================
// Four waves per window for 64 < C <= 256.  The one-wave kernels above hold 50 KB (C = 128) / 100 KB (C = 256) of LDS per
// wave, i.e. three waves / one wave per CU, and every MFMA dependency stalls the SIMD.  Here a 256-thread workgroup shares
// the q^|k^|v, dO and dq^ tiles of ONE window and the four waves take the 16-row blocks of the attention matrix round-robin
// (each with its own P / dS tile); what sums over the row blocks (dk^, dV) is added across the waves through those tiles
// at the end.  LDS per workgroup 78 KB / 151 KB -> eight / four waves per CU.
// =====================================================================================================================
template <int CP>
struct Blk4Tiles {
    static constexpr int NF = CP / 16, LDQ = 3 * CP + 4, LDP = CP + 4;
    static constexpr int QKV = 0, INVN = 16 * LDQ, RED = INVN + 32, PW = RED + 512, END_FWD = PW + 4 * 16 * LDP;
    static constexpr int DO = END_FWD, DQ = DO + 16 * LDP, END_BWD = DQ + 16 * LDP;
};

// F.normalize of q and k in place with 256 threads: thread (pixel t & 15, part t >> 4) owns CP/16 channels of each
template <int CP>
__device__ __forceinline__ void blk4_normalise(float* sm, int tid) {
    typedef Blk4Tiles<CP> T;
    float* qkv = sm + T::QKV;
    float* red = sm + T::RED;  // [2][16 parts][16 pixels]
    const int p = tid & 15, part = tid >> 4, c0 = part * (CP / 16);
    float sq = 0.f, sk = 0.f;
#pragma unroll
    for (int c = 0; c < CP / 16; ++c) {
        const float a = qkv[p * T::LDQ + c0 + c], b = qkv[p * T::LDQ + CP + c0 + c];
        sq = fmaf(a, a, sq);
        sk = fmaf(b, b, sk);
    }
    red[part * 16 + p] = sq;
    red[256 + part * 16 + p] = sk;
    __syncthreads();
    sq = 0.f;
    sk = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {  // same order in every thread: all parts of a pixel get the identical norm
        sq += red[k * 16 + p];
        sk += red[256 + k * 16 + p];
    }
    const float iq = 1.f / fmaxf(sqrtf(sq), 1e-12f), ik = 1.f / fmaxf(sqrtf(sk), 1e-12f);
#pragma unroll
    for (int c = 0; c < CP / 16; ++c) {
        qkv[p * T::LDQ + c0 + c] *= iq;
        qkv[p * T::LDQ + CP + c0 + c] *= ik;
    }
    if (part == 0) {
        sm[T::INVN + p] = iq;
        sm[T::INVN + 16 + p] = ik;
    }
}

// softmaxed rows [16b, 16b + 16) of the attention matrix -> registers and this wave's tile Ps[16][LDP]
template <int CP>
__device__ __forceinline__ void blk4_softmax_rows(f32x4 (&s)[1][CP / 16], const float* qkv, float* Ps, int b, int C, int lane) {
    typedef Blk4Tiles<CP> T;
    constexpr int NF = T::NF;
    const int i = lane & 15;
    tile_zero<1, NF>(s);
    tile_mma<1, NF>(s, qkv + 16 * b, 1, T::LDQ, qkv + CP, T::LDQ, 1, 16, lane);  // q^_blk^T k^
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            if (16 * nf + i >= C) s[0][nf][r] = -INFINITY;
            mx = fmaxf(mx, s[0][nf][r]);
        }
        mx = row16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            s[0][nf][r] = __expf(s[0][nf][r] - mx);
            sum += s[0][nf][r];
        }
        sum = row16_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) s[0][nf][r] *= inv;
    }
    tile_store<1, NF>(s, Ps, T::LDP, 1, lane);
}

template <int CP>
__global__ __launch_bounds__(256) void attn_core_fwd_blk4_kernel(const float* __restrict__ qkv_g, float* __restrict__ o_g, int N, int H,
                                                                 int W, int C) {
    typedef Blk4Tiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    float* qkv = sm + T::QKV;
    float* Ps = sm + T::PW + wave * 16 * T::LDP;
    const int nrb = (C + 15) / 16;  // row blocks; every wave runs the same number of rounds (block barriers inside)
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        __syncthreads();
        load_window<CP>(qkv_g, qkv, T::LDQ, 3, C, H, W, n, wy, wx, tid, 256);
        __syncthreads();
        blk4_normalise<CP>(sm, tid);
        __syncthreads();
        const int y = 4 * wy + (i >> 2), x = 4 * wx + (i & 3);
        float* dst = o_g + (((size_t)n * H + y) * W + x) * C;
        for (int b0 = 0; b0 < nrb; b0 += 4) {
            const int b = b0 + wave;
            if (b < nrb) {  // wave-private tile: the wave's own program order is all the synchronisation this needs ...
                f32x4 s[1][NF];
                blk4_softmax_rows<CP>(s, qkv, Ps, b, C, lane);
            }
            __syncthreads();  // ... but LDS visibility between lanes is only guaranteed across a barrier
            if (b < nrb) {
                f32x4 o[1][1];  // O^T[c1 in block][p] = sum_c2 P[c1][c2] V[p][c2]
                tile_zero<1, 1>(o);
                tile_mma<1, 1>(o, Ps, T::LDP, 1, qkv + 2 * CP, 1, T::LDQ, CP, lane);
                const int c = 16 * b + 4 * g;
                if (c < C) *reinterpret_cast<f32x4*>(dst + c) = o[0][0];
            }
            __syncthreads();
        }
    }
}

template <int CP>
__global__ __launch_bounds__(256) void attn_core_bwd_blk4_kernel(const float* __restrict__ qkv_g, const float* __restrict__ do_g,
                                                                 float* __restrict__ dqkv_g, int N, int H, int W, int C) {
    typedef Blk4Tiles<CP> T;
    constexpr int NF = T::NF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwy * nwx;
    float* qkv = sm + T::QKV;
    float* invn = sm + T::INVN;
    float* PW = sm + T::PW;
    float* Ps = PW + wave * 16 * T::LDP;
    float* dOs = sm + T::DO;
    float* dQ = sm + T::DQ;  // raw dq^ [16][LDP]; wave w fills the columns of its row blocks
    const int nrb = (C + 15) / 16;
    for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int wx = w % nwx, wy = (w / nwx) % nwy, n = w / (nwx * nwy);
        __syncthreads();
        load_window<CP>(qkv_g, qkv, T::LDQ, 3, C, H, W, n, wy, wx, tid, 256);
        load_window<CP>(do_g, dOs, T::LDP, 1, C, H, W, n, wy, wx, tid, 256);
        __syncthreads();
        blk4_normalise<CP>(sm, tid);
        __syncthreads();
        f32x4 dv[1][NF], dk[1][NF];  // this wave's share of the sums over row blocks: rows = pixels, columns = c2
        tile_zero<1, NF>(dv);
        tile_zero<1, NF>(dk);
        for (int b0 = 0; b0 < nrb; b0 += 4) {
            const int b = b0 + wave;
            const bool on = b < nrb;
            f32x4 s[1][NF], ds[1][NF];
            if (on) blk4_softmax_rows<CP>(s, qkv, Ps, b, C, lane);  // P rows of this block -> Ps
            __syncthreads();
            if (on) {
                tile_zero<1, NF>(ds);  // dP[c1 blk][c2] = sum_p dO[p][c1] V[p][c2]
                tile_mma<1, NF>(ds, dOs + 16 * b, 1, T::LDP, qkv + 2 * CP, T::LDQ, 1, 16, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dot = 0.f;
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) dot += s[0][nf][r] * ds[0][nf][r];
                    dot = row16_sum(dot);
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) ds[0][nf][r] = s[0][nf][r] * (ds[0][nf][r] - dot);
                }
                // dV[p][c2] += sum_{c1 in blk} dO[p][c1] P[c1][c2]   (P still in Ps)
                tile_mma<1, NF>(dv, dOs + 16 * b, T::LDP, 1, Ps, T::LDP, 1, 16, lane);
            }
            __syncthreads();
            if (on) tile_store<1, NF>(ds, Ps, T::LDP, 1, lane);  // Ps <- dS block
            __syncthreads();
            if (on) {
                // dq^[p][c1 in blk] = sum_c2 dS[c1][c2] k^[p][c2]
                f32x4 d[1][1];
                tile_zero<1, 1>(d);
                tile_mma<1, 1>(d, qkv + CP, T::LDQ, 1, Ps, 1, T::LDP, CP, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) dQ[(4 * g + r) * T::LDP + 16 * b + i] = d[0][0][r];
                // dk^[p][c2] += sum_{c1 in blk} dS[c1][c2] q^[p][c1]
                tile_mma<1, NF>(dk, qkv + 16 * b, T::LDQ, 1, Ps, T::LDP, 1, 16, lane);
            }
            __syncthreads();
        }
        // ---- combine the four waves: thread (pixel p = tid >> 4, lane-in-row i) owns channels i + 16 j ------------------------------
        const int p = tid >> 4, pi = tid & 15;
        const int py = 4 * wy + (p >> 2), px = 4 * wx + (p & 3);
        float* dst = dqkv_g + (((size_t)n * H + py) * W + px) * 3 * C;
        tile_store<1, NF>(dk, Ps, T::LDP, 1, lane);
        __syncthreads();
        {
            float v[NF], h[NF], dot = 0.f;
            // dk = (dk^ - k^ (k^ . dk^)) / ||k||
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int c = 16 * j + pi;
                v[j] = (PW[p * T::LDP + c] + PW[(16 + p) * T::LDP + c]) + (PW[(32 + p) * T::LDP + c] + PW[(48 + p) * T::LDP + c]);
                h[j] = qkv[p * T::LDQ + CP + c];
                dot = fmaf(h[j], v[j], dot);
            }
            dot = row16_sum(dot);
            const float ik = invn[16 + p];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int c = 16 * j + pi;
                if (c < C) dst[C + c] = (v[j] - h[j] * dot) * ik;
            }
            // dq = (dq^ - q^ (q^ . dq^)) / ||q||
            dot = 0.f;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int c = 16 * j + pi;
                v[j] = c < C ? dQ[p * T::LDP + c] : 0.f;  // columns beyond the last row block were never written
                h[j] = qkv[p * T::LDQ + c];
                dot = fmaf(h[j], v[j], dot);
            }
            dot = row16_sum(dot);
            const float iq = invn[p];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int c = 16 * j + pi;
                if (c < C) dst[c] = (v[j] - h[j] * dot) * iq;
            }
        }
        __syncthreads();
        tile_store<1, NF>(dv, Ps, T::LDP, 1, lane);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int c = 16 * j + pi;
            if (c < C)
                dst[2 * C + c] = (PW[p * T::LDP + c] + PW[(16 + p) * T::LDP + c]) + (PW[(32 + p) * T::LDP + c] + PW[(48 + p) * T::LDP + c]);
        }
    }
}

template <int CP>
static int launch_attn_blk4(bool bwd, const float* qkv, const float* d_o, float* out, int N, int H, int W, int C, hipStream_t st) {
    typedef Blk4Tiles<CP> T;
    const size_t lds = (size_t)(bwd ? T::END_BWD : T::END_FWD) * sizeof(float);
    const int nwin = N * (H / 4) * (W / 4);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int grid = 256 * per_cu * 2;  // two rounds of resident workgroups: evens out windows of unequal cost at no extra LDS
    if (grid > nwin) grid = nwin;
    static bool set_f = false, set_b = false;
    bool& set = bwd ? set_b : set_f;
    if (!set) {
        hipError_t e = bwd ? hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_bwd_blk4_kernel<CP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_fwd_blk4_kernel<CP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(attn_blk4)");
        set = true;
    }
    if (bwd) MSTG_LAUNCH((attn_core_bwd_blk4_kernel<CP>), dim3(grid), dim3(256), lds, st, qkv, d_o, out, N, H, W, C);
    else MSTG_LAUNCH((attn_core_fwd_blk4_kernel<CP>), dim3(grid), dim3(256), lds, st, qkv, out, N, H, W, C);
    MSTG_CHECK_LAUNCH("attn_core_blk4_kernel");
    return MSTG_OK;
}

template <int CP>
static int launch_attn_blk(bool bwd, const float* qkv, const float* d_o, float* out, int N, int H, int W, int C, hipStream_t st) {
    typedef BlkTiles<CP> T;
    const size_t lds = (size_t)(bwd ? T::END_BWD : T::END_FWD) * sizeof(float);
    const int nwin = N * (H / 4) * (W / 4);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 4 ? 4 : (per_cu > 8 ? 8 : per_cu);
    int grid = 256 * per_cu;
    if (grid > nwin) grid = nwin;
    static bool set_f = false, set_b = false;
    bool& set = bwd ? set_b : set_f;
    if (!set) {
        hipError_t e = bwd ? hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_bwd_blk_kernel<CP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_fwd_blk_kernel<CP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(attn_blk)");
        set = true;
    }
    if (bwd) MSTG_LAUNCH((attn_core_bwd_blk_kernel<CP>), dim3(grid), dim3(64), lds, st, qkv, d_o, out, N, H, W, C);
    else MSTG_LAUNCH((attn_core_fwd_blk_kernel<CP>), dim3(grid), dim3(64), lds, st, qkv, out, N, H, W, C);
    MSTG_CHECK_LAUNCH("attn_core_blk_kernel");
    return MSTG_OK;
}

static bool attn_blk4() {  // MSTG_ATTN_BLK4=0: one wave per window also above 64 channels
    const char* e = env_get(ENV_ATTN_BLK4);
    return !(e && e[0] == '0');
}
static bool attn_blk64() {
    // 32 < C <= 64: the row-blocked kernel needs 26 KB of LDS per wave instead of 43 KB (6 waves per CU instead of 3) and was
    // 1.4x (forward) / 1.65x (backward) faster at 256x256, C = 64; MSTG_ATTN_BLK64=0 selects the whole-matrix kernel
    const char* e = env_get(ENV_ATTN_BLK64);
    return !(e && e[0] == '0');
}

static int attn_check(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return fail_arg(MSTG_E_BADARG, "window_attn: empty tensor");
    if (H % 4 || W % 4) return fail_arg(MSTG_E_BADARG, "window_attn: H and W must be multiples of the 4x4 window");
    if (C % 4) return fail_arg(MSTG_E_ALIGN, "window_attn: C must be a multiple of 4");
    if (C > 256) return fail_arg(MSTG_E_UNSUPPORTED, "window_attn: C > 256 not implemented in this build");
    return MSTG_OK;
}

template <int CP>
static int launch_attn(bool bwd, const float* qkv, const float* d_o, float* out, int N, int H, int W, int C, hipStream_t st) {
    typedef AttnTiles<CP> T;
    const size_t lds = (size_t)(bwd ? T::END_BWD : T::END_FWD) * sizeof(float);
    const int nwin = N * (H / 4) * (W / 4);
    int grid = 256 * 8;
    if (grid > nwin) grid = nwin;
    if (bwd) {
        static bool set = false;
        if (!set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_core_bwd_kernel<CP>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(attn_bwd)");
            set = true;
        }
        MSTG_LAUNCH((attn_core_bwd_kernel<CP>), dim3(grid), dim3(64), lds, st, qkv, d_o, out, N, H, W, C);
    } else {
        MSTG_LAUNCH((attn_core_fwd_kernel<CP>), dim3(grid), dim3(64), lds, st, qkv, out, N, H, W, C);
    }
    MSTG_CHECK_LAUNCH("attn_core_kernel");
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_window_attn_core_fwd(const float* qkv, float* o, int N, int H, int W, int C, void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!qkv || !o) return fail_arg(MSTG_E_BADARG, "window_attn_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (C <= 16) return launch_attn<16>(false, qkv, nullptr, o, N, H, W, C, st);
    if (C <= 32) return launch_attn<32>(false, qkv, nullptr, o, N, H, W, C, st);
    if (C <= 64) return attn_blk64() ? launch_attn_blk<64>(false, qkv, nullptr, o, N, H, W, C, st) : launch_attn<64>(false, qkv, nullptr, o, N, H, W, C, st);
    if (C <= 128) return attn_blk4() ? launch_attn_blk4<128>(false, qkv, nullptr, o, N, H, W, C, st) : launch_attn_blk<128>(false, qkv, nullptr, o, N, H, W, C, st);
    return attn_blk4() ? launch_attn_blk4<256>(false, qkv, nullptr, o, N, H, W, C, st) : launch_attn_blk<256>(false, qkv, nullptr, o, N, H, W, C, st);
}

extern "C" int mstg_window_attn_core_bwd(const float* qkv, const float* d_o, float* dqkv, int N, int H, int W, int C,
                                         void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!qkv || !d_o || !dqkv) return fail_arg(MSTG_E_BADARG, "window_attn_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (C <= 16) return launch_attn<16>(true, qkv, d_o, dqkv, N, H, W, C, st);
    if (C <= 32) return launch_attn<32>(true, qkv, d_o, dqkv, N, H, W, C, st);
    if (C <= 64) return attn_blk64() ? launch_attn_blk<64>(true, qkv, d_o, dqkv, N, H, W, C, st) : launch_attn<64>(true, qkv, d_o, dqkv, N, H, W, C, st);
    if (C <= 128) return attn_blk4() ? launch_attn_blk4<128>(true, qkv, d_o, dqkv, N, H, W, C, st) : launch_attn_blk<128>(true, qkv, d_o, dqkv, N, H, W, C, st);
    return attn_blk4() ? launch_attn_blk4<256>(true, qkv, d_o, dqkv, N, H, W, C, st) : launch_attn_blk<256>(true, qkv, d_o, dqkv, N, H, W, C, st);
}

#ifdef MSTG_STAMPS
extern "C" int mstg_debug_stamps_attn(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_stamps), sizeof(unsigned long long) * 64 * 8);
}
extern "C" int mstg_debug_stamps_attn_fwd(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_fstamps), sizeof(unsigned long long) * 64 * 8);
}
#endif
extern "C" int mstg_window_attn_fused_supported(int C) { return C == 16 || C == 32 || C == 64; }

extern "C" int mstg_window_attn_fwd(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj,
                                    float* y, int N, int H, int W, int C, void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!x || !wqkv || !bqkv || !wproj || !bproj || !y) return fail_arg(MSTG_E_BADARG, "window_attn_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) return launch_fused<16>(false, x, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st);
    if (C == 32) return launch_fused<32>(false, x, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st);
    if (C == 64) return launch_fused<64>(false, x, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st);
    return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_fwd: fused kernel exists for C = 16, 32 and 64 (use the qkv/core/proj chain otherwise)");
}

extern "C" int mstg_window_attn_norm_fwd(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv,
                                         const float* wproj, const float* bproj, float* y, int N, int H, int W, int C, void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!x_raw || !in_stats || !wqkv || !bqkv || !wproj || !bproj || !y) return fail_arg(MSTG_E_BADARG, "window_attn_norm_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) return launch_fused<16>(false, x_raw, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st, in_stats);
    if (C == 32) return launch_fused<32>(false, x_raw, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st, in_stats);
    if (C == 64) return launch_fused<64>(false, x_raw, wqkv, bqkv, wproj, bproj, nullptr, y, nullptr, nullptr, N, H, W, st, in_stats);
    return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_norm_fwd: fused kernel exists for C = 16, 32 and 64");
}

extern "C" int mstg_window_attn_norm_sums_split(void) { return NSUM_SPLIT; }

extern "C" size_t mstg_window_attn_norm_bwd_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || !(C == 16 || C == 32 || C == 64)) return 0;
    const size_t nb = (size_t)fused_blocks(N, H, W);
    return (nb * (4 * C * C + 4 * C) + (size_t)N * ((H / 4) * (W / 4) / norm_run_len(N, H, W, C)) * 2 * C) * sizeof(float);
}

extern "C" int mstg_window_attn_norm_bwd(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv,
                                         const float* wproj, const float* bproj, const float* dy, float* dz, float* dparams,
                                         float* norm_sums, int N, int H, int W, int C, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!x_raw || !in_stats || !wqkv || !bqkv || !wproj || !bproj || !dy || !dz || (!dparams && !t_attn_dst.p[0]) || !norm_sums || !workspace)
        return fail_arg(MSTG_E_BADARG, "window_attn_norm_bwd: null pointer");
    if (!(C == 16 || C == 32 || C == 64)) return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_norm_bwd: fused kernel exists for C = 16, 32 and 64");
    if (workspace_bytes < mstg_window_attn_norm_bwd_workspace_bytes(N, H, W, C)) return fail_arg(MSTG_E_WORKSPACE, "window_attn_norm_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) return launch_fused<16>(true, x_raw, wqkv, bqkv, wproj, bproj, dy, dz, dparams, (float*)workspace, N, H, W, st, in_stats, norm_sums);
    if (C == 64) return launch_fused<64>(true, x_raw, wqkv, bqkv, wproj, bproj, dy, dz, dparams, (float*)workspace, N, H, W, st, in_stats, norm_sums);
    return launch_fused<32>(true, x_raw, wqkv, bqkv, wproj, bproj, dy, dz, dparams, (float*)workspace, N, H, W, st, in_stats, norm_sums);
}

extern "C" size_t mstg_window_attn_bwd_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || !(C == 16 || C == 32 || C == 64)) return 0;
    return (size_t)fused_blocks(N, H, W) * (4 * C * C + 4 * C) * sizeof(float);
}

extern "C" int mstg_window_attn_bwd(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj,
                                    const float* dy, float* dx, float* dparams, int N, int H, int W, int C, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (int rc = attn_check(N, H, W, C)) return rc;
    if (!x || !wqkv || !bqkv || !wproj || !bproj || !dy || !dx || (!dparams && !t_attn_dst.p[0]) || !workspace)
        return fail_arg(MSTG_E_BADARG, "window_attn_bwd: null pointer");
    if (!(C == 16 || C == 32 || C == 64)) return fail_arg(MSTG_E_UNSUPPORTED, "window_attn_bwd: fused kernel exists for C = 16, 32 and 64");
    if (workspace_bytes < mstg_window_attn_bwd_workspace_bytes(N, H, W, C)) return fail_arg(MSTG_E_WORKSPACE, "window_attn_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (C == 16) return launch_fused<16>(true, x, wqkv, bqkv, wproj, bproj, dy, dx, dparams, (float*)workspace, N, H, W, st);
    if (C == 64) return launch_fused<64>(true, x, wqkv, bqkv, wproj, bproj, dy, dx, dparams, (float*)workspace, N, H, W, st);
    return launch_fused<32>(true, x, wqkv, bqkv, wproj, bproj, dy, dx, dparams, (float*)workspace, N, H, W, st);
}

// ---- the same two backward entry points with the reduced parameter gradients written (or added) straight into four tensors ----------
namespace {
struct AttnDstScope {
    AttnDstScope(float* a, float* b, float* c, float* d, int acc) { t_attn_dst = AttnGradDst{{a, b, c, d}, acc}; }
    ~AttnDstScope() { t_attn_dst = AttnGradDst{{nullptr, nullptr, nullptr, nullptr}, 0}; }
};
}  // namespace
extern "C" int mstg_window_attn_bwd_direct(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj,
                                           const float* dy, float* dx, float* dwqkv, float* dbqkv, float* dwproj, float* dbproj,
                                           int accumulate, int N, int H, int W, int C, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dwqkv || !dbqkv || !dwproj || !dbproj) return fail_arg(MSTG_E_BADARG, "window_attn_bwd_direct: null gradient pointer");
    AttnDstScope scope(dwqkv, dwproj, dbqkv, dbproj, accumulate);
    return mstg_window_attn_bwd(x, wqkv, bqkv, wproj, bproj, dy, dx, nullptr, N, H, W, C, workspace, workspace_bytes, stream);
}
extern "C" int mstg_window_attn_norm_bwd_direct(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv,
                                                const float* wproj, const float* bproj, const float* dy, float* dz, float* dwqkv,
                                                float* dbqkv, float* dwproj, float* dbproj, int accumulate, float* norm_sums, int N, int H,
                                                int W, int C, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dwqkv || !dbqkv || !dwproj || !dbproj) return fail_arg(MSTG_E_BADARG, "window_attn_norm_bwd_direct: null gradient pointer");
    AttnDstScope scope(dwqkv, dwproj, dbqkv, dbproj, accumulate);
    return mstg_window_attn_norm_bwd(x_raw, in_stats, wqkv, bqkv, wproj, bproj, dy, dz, nullptr, norm_sums, N, H, W, C, workspace,
                                     workspace_bytes, stream);
}
