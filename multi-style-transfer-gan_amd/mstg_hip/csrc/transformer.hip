// BUILD-DEFINED StructuralTransformerBlock kernels -- parity unpinned: the reference imports the class from a file that is not in
// its snapshot (enhanced_generator.py:4), only the call shape is known (:115, :218-225).  Definition: see
// multi-style-transfer-gan_amd/structural_transformer.py; CPU restatement: oracle/restatement.py::structural_transformer_block.
//
//   structure_map      orig image (N,3,H,W) -> (N, H/4, W/4, 4): per 4x4 cell mean R, G, B and mean |dx| + |dy| of the luminance
//   ln_mod             y = (LayerNorm(x) * gamma + beta) * (1 + g[n]) + b[n] over tokens (style modulation), forward / backward
//   flash attention    softmax(q k^T / sqrt(D)) v over ALL L tokens of an image (L = HW/16: 4096 at 256x256, 65536 at 1024x1024),
//                      per head, fp32 MFMA (16x16x4), online softmax, nothing of size L x L ever stored.  Orientation: the
//                      score tile is computed TRANSPOSED (keys on the accumulator rows, queries on the lane column), so that the
//                      probabilities sit in the registers exactly as the B operand of the next product needs them (k-slot g <-> key
//                      4g + r): P never goes through LDS.  Backward = two kernels without atomics: dQ per query tile (loop over
//                      keys) and dK / dV per key tile (loop over queries), each recomputing P from q, k and the saved log-sum-exp.
#include "common.h"

namespace mstg {

template <int CTRL>
__device__ __forceinline__ float tdpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float trow16_sum(float v) {
    v += tdpp<0xB1>(v);
    v += tdpp<0x4E>(v);
    v += tdpp<0x141>(v);
    v += tdpp<0x140>(v);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ void structure_map_kernel(const float* __restrict__ img, float* __restrict__ out, int N, int H, int W) {
    const int H4 = H >> 2, W4 = W >> 2;
    const size_t total = (size_t)N * H4 * W4;
    const size_t plane = (size_t)H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int cx = (int)(e % W4), cy = (int)((e / W4) % H4), n = (int)(e / ((size_t)W4 * H4));
        const float* p = img + (size_t)n * 3 * plane;
        float sr = 0.f, sg = 0.f, sb = 0.f, se = 0.f;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) {
                const int y = 4 * cy + dy, x = 4 * cx + dx;
                const size_t o = (size_t)y * W + x;
                const float r = p[o], g = p[plane + o], b = p[2 * plane + o];
                sr += r; sg += g; sb += b;
                const float lum = 0.299f * r + 0.587f * g + 0.114f * b;
                float gx = 0.f, gy = 0.f;
                if (x + 1 < W) gx = (0.299f * p[o + 1] + 0.587f * p[plane + o + 1] + 0.114f * p[2 * plane + o + 1]) - lum;
                if (y + 1 < H) gy = (0.299f * p[o + W] + 0.587f * p[plane + o + W] + 0.114f * p[2 * plane + o + W]) - lum;
                se += fabsf(gx) + fabsf(gy);
            }
        f32x4 v = {sr * 0.0625f, sg * 0.0625f, sb * 0.0625f, se * 0.0625f};
        *reinterpret_cast<f32x4*>(out + e * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Modulated LayerNorm.  A group of 16 lanes owns a token; lane j holds channels 4j + 64c.  dim % 4 == 0, dim <= 256.
constexpr int LN_MAXC = 4;  // 64-channel chunks per token
__global__ __launch_bounds__(256) void ln_mod_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ gmod,
                                                         const float* __restrict__ bmod, float* __restrict__ y, float* __restrict__ stats,
                                                         int N, int L, int dim, float eps) {
    const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t T = (size_t)N * L;
    const size_t tok = (size_t)blockIdx.x * 16 + grp;
    if (tok >= T) return;
    const int n = (int)(tok / L);
    const float* xp = x + tok * dim;
    f32x4 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (64 * c + 4 * j < dim) v[c] = *reinterpret_cast<const f32x4*>(xp + 64 * c + 4 * j);
        s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = trow16_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c)
        if (64 * c + 4 * j < dim)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
    const float rstd = rsqrtf(trow16_sum(q) / (float)dim + eps);
    if (j == 0) { stats[tok * 2] = mean; stats[tok * 2 + 1] = rstd; }
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = 64 * c + 4 * j;
        if (ch < dim) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch), be = *reinterpret_cast<const f32x4*>(beta + ch);
            f32x4 gm = {0.f, 0.f, 0.f, 0.f}, bm = {0.f, 0.f, 0.f, 0.f};
            if (gmod) { gm = *reinterpret_cast<const f32x4*>(gmod + (size_t)n * dim + ch); bm = *reinterpret_cast<const f32x4*>(bmod + (size_t)n * dim + ch); }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((v[c][e] - mean) * rstd * ga[e] + be[e]) * (1.f + gm[e]) + bm[e];
            *reinterpret_cast<f32x4*>(y + tok * dim + ch) = o;
        }
    }
}

// backward: dx, and per-workgroup partial sums of dgamma, dbeta (all tokens) and dgmod, dbmod (per image).  A workgroup takes
// `chunk` consecutive tokens of ONE image; partial layout [N][nchunk][4][dim] = (dgamma, dbeta, dgmod, dbmod).
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ gmod, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ partial, int N, int L, int dim,
                                                         int chunk, int nchunk) {
    extern __shared__ float red[];  // [16 groups][4][dim]
    const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int n = blockIdx.y, ck = blockIdx.x;
    const int t0 = ck * chunk, t1 = min(L, t0 + chunk);
    f32x4 a_dga[LN_MAXC], a_dbe[LN_MAXC], a_dgm[LN_MAXC], a_dbm[LN_MAXC], ga[LN_MAXC], be[LN_MAXC], gm1[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        a_dga[c] = a_dbe[c] = a_dgm[c] = a_dbm[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        ga[c] = be[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        gm1[c] = f32x4{1.f, 1.f, 1.f, 1.f};
        const int ch = 64 * c + 4 * j;
        if (ch < dim) {
            ga[c] = *reinterpret_cast<const f32x4*>(gamma + ch);
            be[c] = *reinterpret_cast<const f32x4*>(beta + ch);
            if (gmod) gm1[c] = *reinterpret_cast<const f32x4*>(gmod + (size_t)n * dim + ch) + f32x4{1.f, 1.f, 1.f, 1.f};
        }
    }
    for (int t = t0 + grp; t < t1; t += 16) {
        const size_t tok = (size_t)n * L + t;
        const float mean = stats[tok * 2], rstd = stats[tok * 2 + 1];
        f32x4 xh[LN_MAXC], dxh[LN_MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = 64 * c + 4 * j;
            xh[c] = dxh[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < dim) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + tok * dim + ch), g = *reinterpret_cast<const f32x4*>(dy + tok * dim + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[c][e] = (xv[e] - mean) * rstd;
                    const float z = xh[c][e] * ga[c][e] + be[c][e];
                    const float dz = g[e] * gm1[c][e];
                    a_dgm[c][e] += g[e] * z;
                    a_dbm[c][e] += g[e];
                    a_dga[c][e] += dz * xh[c][e];
                    a_dbe[c][e] += dz;
                    dxh[c][e] = dz * ga[c][e];
                    s1 += dxh[c][e];
                    s2 += dxh[c][e] * xh[c][e];
                }
            }
        }
        const float m1 = trow16_sum(s1) / (float)dim, m2 = trow16_sum(s2) / (float)dim;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = 64 * c + 4 * j;
            if (ch < dim) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (dxh[c][e] - m1 - xh[c][e] * m2);
                *reinterpret_cast<f32x4*>(dx + tok * dim + ch) = o;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = 64 * c + 4 * j;
        if (ch < dim) {
            *reinterpret_cast<f32x4*>(&red[(grp * 4 + 0) * dim + ch]) = a_dga[c];
            *reinterpret_cast<f32x4*>(&red[(grp * 4 + 1) * dim + ch]) = a_dbe[c];
            *reinterpret_cast<f32x4*>(&red[(grp * 4 + 2) * dim + ch]) = a_dgm[c];
            *reinterpret_cast<f32x4*>(&red[(grp * 4 + 3) * dim + ch]) = a_dbm[c];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * dim; e += 256) {
        float acc = 0.f;
        for (int gq = 0; gq < 16; ++gq) acc += red[gq * 4 * dim + e];
        partial[((size_t)n * nchunk + ck) * 4 * dim + e] = acc;
    }
}

// dgamma / dbeta = sum over images and chunks; dgmod / dbmod [N][dim] = sum over the image's chunks; fixed order
__global__ void ln_mod_reduce_kernel(const float* __restrict__ partial, int N, int nchunk, int dim, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, float* __restrict__ dgmod, float* __restrict__ dbmod, int accumulate) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= dim) return;
    float tg = 0.f, tb = 0.f;
    for (int n = 0; n < N; ++n) {
        float sg = 0.f, sb = 0.f, mg = 0.f, mb = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* p = partial + ((size_t)n * nchunk + c) * 4 * dim;
            sg += p[e]; sb += p[dim + e]; mg += p[2 * dim + e]; mb += p[3 * dim + e];
        }
        tg += sg; tb += sb;
        if (dgmod) { dgmod[(size_t)n * dim + e] = mg; dbmod[(size_t)n * dim + e] = mb; }
    }
    if (accumulate) { dgamma[e] += tg; dbeta[e] += tb; }
    else { dgamma[e] = tg; dbeta[e] = tb; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Flash attention, fp32.  qkv (N, L, 3 * dim) token-major, q | k | v blocks, head h = channels [h*D, (h+1)*D).  D % 4 == 0.
// DP = D rounded up to 16 (the accumulator rows of the transposed output are the head's channels).
constexpr int FA_TQ = 64, FA_TK = 64;  // queries per workgroup (16 per wave), keys per staged tile

template <int D>
struct FaCfg {
    static constexpr int DP = (D + 15) / 16 * 16, NDF = DP / 16, KS = D / 4;
    static constexpr int SK = D + 1;    // row stride (floats) of tiles read as A[row = token i][k = 4ks + g]
    static constexpr int SV = DP + 4;   // row stride of tiles read as A[row = channel i][k-slot g <-> token 4g + r]
};

// stage `rows` tokens (from token t0 of image n; zero beyond L) of channel block `blk` (0 q, 1 k, 2 v) of head h into LDS twice:
// `a` with stride SK (D columns), `b` with stride SV (DP columns, zero-padded); either may be null
template <int D>
__device__ __forceinline__ void fa_stage(const float* __restrict__ src, int ld, int coff, int n, int L, int t0, int rows,
                                         float* a, float* b, int tid) {
    typedef FaCfg<D> C;
    for (int e = tid; e < rows * (C::DP / 4); e += 256) {
        const int t = e / (C::DP / 4), q = e - t * (C::DP / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t0 + t < L && 4 * q < D) v = *reinterpret_cast<const f32x4*>(src + ((size_t)n * L + t0 + t) * ld + coff + 4 * q);
        if (a && 4 * q < D) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a[t * C::SK + 4 * q + k] = v[k];
        }
        if (b) *reinterpret_cast<f32x4*>(&b[t * C::SV + 4 * q]) = v;
    }
}

template <int D>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ lse,
                                                        int N, int L, int heads, float scale) {
    typedef FaCfg<D> C;
    extern __shared__ float sm[];
    float* Ks = sm;                       // [FA_TK][SK]
    float* Vs = Ks + FA_TK * C::SK;       // [FA_TK][SV]
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, i = l & 15, g = l >> 4;
    const int h = blockIdx.y, n = blockIdx.z, dim = heads * D, ld = 3 * dim;
    const int q0 = blockIdx.x * FA_TQ + 16 * wv;
    const int qi = q0 + i;  // this lane's query (column of the transposed tiles)
    float qr[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qr[ks] = qi < L ? qkv[((size_t)n * L + qi) * ld + h * D + 4 * ks + g] * scale : 0.f;
    f32x4 o[C::NDF];
#pragma unroll
    for (int df = 0; df < C::NDF; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    for (int k0 = 0; k0 < L; k0 += FA_TK) {
        __syncthreads();
        fa_stage<D>(qkv, ld, dim + h * D, n, L, k0, FA_TK, Ks, nullptr, tid);
        fa_stage<D>(qkv, ld, 2 * dim + h * D, n, L, k0, FA_TK, nullptr, Vs, tid);
        __syncthreads();
        f32x4 s[FA_TK / 16];
        float tmax = -INFINITY;
#pragma unroll
        for (int c = 0; c < FA_TK / 16; ++c) {
            s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) s[c] = mfma16(Ks[(16 * c + i) * C::SK + 4 * ks + g], qr[ks], s[c]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 16 * c + 4 * g + r >= L) s[c][r] = -INFINITY;
                tmax = fmaxf(tmax, s[c][r]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = __expf(m - mnew);  // m = -inf on the first tile: exp(-inf) = 0
        lsum *= alpha;
#pragma unroll
        for (int df = 0; df < C::NDF; ++df) o[df] *= alpha;
        m = mnew;
#pragma unroll
        for (int c = 0; c < FA_TK / 16; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[c][r] - m);
                lsum += p;
#pragma unroll
                for (int df = 0; df < C::NDF; ++df) o[df] = mfma16(Vs[(16 * c + 4 * g + r) * C::SV + 16 * df + i], p, o[df]);
            }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < L) {
        const float inv = 1.f / lsum;
#pragma unroll
        for (int df = 0; df < C::NDF; ++df)
            if (16 * df + 4 * g < D) {
                f32x4 v = o[df] * inv;
                *reinterpret_cast<f32x4*>(out + ((size_t)n * L + qi) * dim + h * D + 16 * df + 4 * g) = v;
            }
        if (g == 0) lse[((size_t)n * heads + h) * L + qi] = m + __logf(lsum);
    }
}

// delta[n][h][q] = sum_d dO[q][h*D + d] * O[q][h*D + d]
__global__ void flash_delta_kernel(const float* __restrict__ o, const float* __restrict__ d_o, float* __restrict__ delta, int N, int L,
                                   int heads, int D) {
    const size_t total = (size_t)N * heads * L;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % L), h = (int)((e / L) % heads), n = (int)(e / ((size_t)L * heads));
        const size_t base = ((size_t)n * L + q) * heads * D + (size_t)h * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += o[base + d] * d_o[base + d];
        delta[e] = acc;
    }
}

// dQ: one workgroup per 64 queries, loop over key tiles
template <int D>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int N, int L, int heads, float scale) {
    typedef FaCfg<D> C;
    extern __shared__ float sm[];
    float* Ks = sm;                        // [FA_TK][SK]  rows = key, for S^T
    float* Kb = Ks + FA_TK * C::SK;        // [FA_TK][SV]  rows = key, read as A[row = channel][k-slot <-> key]
    float* Vs = Kb + FA_TK * C::SV;        // [FA_TK][SK]  rows = key, for dP^T
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, i = l & 15, g = l >> 4;
    const int h = blockIdx.y, n = blockIdx.z, dim = heads * D, ld = 3 * dim;
    const int qi = blockIdx.x * FA_TQ + 16 * wv + i;
    float qr[C::KS], dor[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        qr[ks] = qi < L ? qkv[((size_t)n * L + qi) * ld + h * D + 4 * ks + g] * scale : 0.f;
        dor[ks] = qi < L ? d_o[((size_t)n * L + qi) * dim + h * D + 4 * ks + g] : 0.f;
    }
    const float my_lse = qi < L ? lse[((size_t)n * heads + h) * L + qi] : 0.f;
    const float my_delta = qi < L ? delta[((size_t)n * heads + h) * L + qi] : 0.f;
    f32x4 dq[C::NDF];
#pragma unroll
    for (int df = 0; df < C::NDF; ++df) dq[df] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += FA_TK) {
        __syncthreads();
        fa_stage<D>(qkv, ld, dim + h * D, n, L, k0, FA_TK, Ks, Kb, tid);
        fa_stage<D>(qkv, ld, 2 * dim + h * D, n, L, k0, FA_TK, Vs, nullptr, tid);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FA_TK / 16; ++c) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma16(Ks[(16 * c + i) * C::SK + 4 * ks + g], qr[ks], s);
                dp = mfma16(Vs[(16 * c + i) * C::SK + 4 * ks + g], dor[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = k0 + 16 * c + 4 * g + r < L;
                const float p = valid ? __expf(s[r] - my_lse) : 0.f;
                const float ds = p * (dp[r] - my_delta);
#pragma unroll
                for (int df = 0; df < C::NDF; ++df) dq[df] = mfma16(Kb[(16 * c + 4 * g + r) * C::SV + 16 * df + i], ds, dq[df]);
            }
        }
    }
    if (qi < L) {
#pragma unroll
        for (int df = 0; df < C::NDF; ++df)
            if (16 * df + 4 * g < D) {
                f32x4 v = dq[df] * scale;
                *reinterpret_cast<f32x4*>(dqkv + ((size_t)n * L + qi) * ld + h * D + 16 * df + 4 * g) = v;
            }
    }
}

// dK, dV: one workgroup per 64 keys, loop over query tiles
template <int D>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dqkv, int N, int L, int heads, float scale) {
    typedef FaCfg<D> C;
    extern __shared__ float sm[];
    float* Qs = sm;                        // [FA_TQ][SK]  A[row = query][k = channel] for S
    float* Qb = Qs + FA_TQ * C::SK;        // [FA_TQ][SV]  A[row = channel][k-slot <-> query] for dK^T
    float* Os = Qb + FA_TQ * C::SV;        // [FA_TQ][SK]  dO, A[row = query][k = channel] for dP
    float* Ob = Os + FA_TQ * C::SK;        // [FA_TQ][SV]  dO, A[row = channel][k-slot <-> query] for dV^T
    float* Ls = Ob + FA_TQ * C::SV;        // [FA_TQ] log-sum-exp, then [FA_TQ] delta
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, i = l & 15, g = l >> 4;
    const int h = blockIdx.y, n = blockIdx.z, dim = heads * D, ld = 3 * dim;
    const int ki = blockIdx.x * FA_TK + 16 * wv + i;  // this lane's key (column)
    float kr[C::KS], vr[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        kr[ks] = ki < L ? qkv[((size_t)n * L + ki) * ld + dim + h * D + 4 * ks + g] * scale : 0.f;
        vr[ks] = ki < L ? qkv[((size_t)n * L + ki) * ld + 2 * dim + h * D + 4 * ks + g] : 0.f;
    }
    f32x4 dk[C::NDF], dv[C::NDF];
#pragma unroll
    for (int df = 0; df < C::NDF; ++df) dk[df] = dv[df] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < L; q0 += FA_TQ) {
        __syncthreads();
        fa_stage<D>(qkv, ld, h * D, n, L, q0, FA_TQ, Qs, Qb, tid);
        fa_stage<D>(d_o, dim, h * D, n, L, q0, FA_TQ, Os, Ob, tid);
        if (tid < FA_TQ) {
            const bool ok = q0 + tid < L;
            Ls[tid] = ok ? lse[((size_t)n * heads + h) * L + q0 + tid] : 0.f;
            Ls[FA_TQ + tid] = ok ? delta[((size_t)n * heads + h) * L + q0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FA_TQ / 16; ++c) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                s = mfma16(Qs[(16 * c + i) * C::SK + 4 * ks + g], kr[ks], s);     // S[query 16c + 4g + r][key i] (scaled)
                dp = mfma16(Os[(16 * c + i) * C::SK + 4 * ks + g], vr[ks], dp);  // dP[query][key]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = 16 * c + 4 * g + r;
                const bool valid = q0 + qq < L && ki < L;
                const float p = valid ? __expf(s[r] - Ls[qq]) : 0.f;
                const float ds = p * (dp[r] - Ls[FA_TQ + qq]);
#pragma unroll
                for (int df = 0; df < C::NDF; ++df) {
                    dv[df] = mfma16(Ob[qq * C::SV + 16 * df + i], p, dv[df]);
                    dk[df] = mfma16(Qb[qq * C::SV + 16 * df + i], ds, dk[df]);
                }
            }
        }
    }
    if (ki < L) {
#pragma unroll
        for (int df = 0; df < C::NDF; ++df)
            if (16 * df + 4 * g < D) {
                f32x4 a = dk[df] * scale;
                *reinterpret_cast<f32x4*>(dqkv + ((size_t)n * L + ki) * ld + dim + h * D + 16 * df + 4 * g) = a;
                *reinterpret_cast<f32x4*>(dqkv + ((size_t)n * L + ki) * ld + 2 * dim + h * D + 16 * df + 4 * g) = dv[df];
            }
    }
}

template <int D>
static int launch_flash(int pass, const float* qkv, float* out, float* lse, const float* d_o, const float* delta, float* dqkv, int N,
                        int L, int heads, float scale, hipStream_t st) {
    typedef FaCfg<D> C;
    dim3 grid(cdiv(L, FA_TQ), heads, N);
    if (pass == 0) {
        const size_t lds = (size_t)(FA_TK * C::SK + FA_TK * C::SV) * sizeof(float);
        MSTG_LAUNCH((flash_fwd_kernel<D>), grid, dim3(256), lds, st, qkv, out, lse, N, L, heads, scale);
        MSTG_CHECK_LAUNCH("flash_fwd_kernel");
    } else {
        const size_t lds1 = (size_t)(2 * FA_TK * C::SK + FA_TK * C::SV) * sizeof(float);
        MSTG_LAUNCH((flash_bwd_dq_kernel<D>), grid, dim3(256), lds1, st, qkv, d_o, (const float*)lse, delta, dqkv, N, L, heads, scale);
        MSTG_CHECK_LAUNCH("flash_bwd_dq_kernel");
        const size_t lds2 = (size_t)(2 * FA_TQ * C::SK + 2 * FA_TQ * C::SV + 2 * FA_TQ) * sizeof(float);
        static bool attr_set = false;
        if (lds2 > 64 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(flash_bwd_dkv_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            attr_set = true;
        }
        MSTG_LAUNCH((flash_bwd_dkv_kernel<D>), grid, dim3(256), lds2, st, qkv, d_o, (const float*)lse, delta, dqkv, N, L, heads, scale);
        MSTG_CHECK_LAUNCH("flash_bwd_dkv_kernel");
    }
    return MSTG_OK;
}

static int flash_check(int N, int L, int heads, int D) {
    if (N <= 0 || L <= 0 || heads <= 0) return fail_arg(MSTG_E_BADARG, "flash_attn: empty tensor");
    if (D != 8 && D != 16 && D != 32 && D != 64) return fail_arg(MSTG_E_UNSUPPORTED, "flash_attn: head dimension must be 8, 16, 32 or 64");
    if (N > 65535 || heads > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "flash_attn: grid too large");
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_structure_map(const float* img, float* out, int N, int H, int W, void* stream) {
    if (!img || !out || N <= 0 || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "structure_map: bad argument");
    if (H % 4 || W % 4) return fail_arg(MSTG_E_BADARG, "structure_map: H and W must be multiples of 4");
    const size_t total = (size_t)N * (H / 4) * (W / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    MSTG_LAUNCH(structure_map_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, out, N, H, W);
    MSTG_CHECK_LAUNCH("structure_map_kernel");
    return MSTG_OK;
}

extern "C" int mstg_ln_mod_fwd(const float* x, const float* gamma, const float* beta, const float* gmod, const float* bmod, float* y,
                               float* stats, int N, int L, int dim, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !stats) return fail_arg(MSTG_E_BADARG, "ln_mod_fwd: null pointer");
    if ((gmod == nullptr) != (bmod == nullptr)) return fail_arg(MSTG_E_BADARG, "ln_mod_fwd: gmod and bmod come together");
    if (N <= 0 || L <= 0 || dim <= 0 || dim % 4 || dim > 64 * LN_MAXC) return fail_arg(MSTG_E_ALIGN, "ln_mod: dim must be a multiple of 4, <= 256");
    const size_t T = (size_t)N * L;
    MSTG_LAUNCH(ln_mod_fwd_kernel, dim3((unsigned)((T + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, gmod, bmod, y,
                       stats, N, L, dim, eps);
    MSTG_CHECK_LAUNCH("ln_mod_fwd_kernel");
    return MSTG_OK;
}

static int ln_chunks(int L, int* chunk) {
    int nchunk = cdiv(L, 512);
    if (nchunk > 64) nchunk = 64;
    *chunk = cdiv(L, nchunk);
    return cdiv(L, *chunk);
}

extern "C" size_t mstg_ln_mod_bwd_workspace_bytes(int N, int L, int dim) {
    if (N <= 0 || L <= 0 || dim <= 0) return 0;
    int chunk;
    const int nchunk = ln_chunks(L, &chunk);
    return (size_t)N * nchunk * 4 * dim * sizeof(float);
}

extern "C" int mstg_ln_mod_bwd(const float* x, const float* stats, const float* gamma, const float* beta, const float* gmod,
                               const float* dy, float* dx, float* dgamma, float* dbeta, float* dgmod, float* dbmod, int accumulate,
                               int N, int L, int dim, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !stats || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !workspace) return fail_arg(MSTG_E_BADARG, "ln_mod_bwd: null pointer");
    if ((gmod == nullptr) != (dgmod == nullptr) || (dgmod == nullptr) != (dbmod == nullptr)) return fail_arg(MSTG_E_BADARG, "ln_mod_bwd: gmod, dgmod and dbmod come together");
    if (N <= 0 || L <= 0 || dim <= 0 || dim % 4 || dim > 64 * LN_MAXC) return fail_arg(MSTG_E_ALIGN, "ln_mod: dim must be a multiple of 4, <= 256");
    if (N > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "ln_mod_bwd: batch too large");
    if (workspace_bytes < mstg_ln_mod_bwd_workspace_bytes(N, L, dim)) return fail_arg(MSTG_E_WORKSPACE, "ln_mod_bwd: workspace too small");
    int chunk;
    const int nchunk = ln_chunks(L, &chunk);
    hipStream_t st = (hipStream_t)stream;
    MSTG_LAUNCH(ln_mod_bwd_kernel, dim3(nchunk, N), dim3(256), (size_t)16 * 4 * dim * sizeof(float), st, x, stats, gamma, beta, gmod, dy,
                       dx, (float*)workspace, N, L, dim, chunk, nchunk);
    MSTG_CHECK_LAUNCH("ln_mod_bwd_kernel");
    MSTG_LAUNCH(ln_mod_reduce_kernel, dim3(cdiv(dim, 64)), dim3(64), 0, st, (const float*)workspace, N, nchunk, dim, dgamma, dbeta, dgmod,
                       dbmod, accumulate);
    MSTG_CHECK_LAUNCH("ln_mod_reduce_kernel");
    return MSTG_OK;
}

extern "C" int mstg_flash_attn_fwd(const float* qkv, float* out, float* lse, int N, int L, int heads, int D, void* stream) {
    if (!qkv || !out || !lse) return fail_arg(MSTG_E_BADARG, "flash_attn_fwd: null pointer");
    if (int rc = flash_check(N, L, heads, D)) return rc;
    const float scale = 1.f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 8: return launch_flash<8>(0, qkv, out, lse, nullptr, nullptr, nullptr, N, L, heads, scale, st);
        case 16: return launch_flash<16>(0, qkv, out, lse, nullptr, nullptr, nullptr, N, L, heads, scale, st);
        case 32: return launch_flash<32>(0, qkv, out, lse, nullptr, nullptr, nullptr, N, L, heads, scale, st);
        default: return launch_flash<64>(0, qkv, out, lse, nullptr, nullptr, nullptr, N, L, heads, scale, st);
    }
}

extern "C" int mstg_flash_attn_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, float* dqkv, float* delta_ws,
                                   int N, int L, int heads, int D, void* stream) {
    if (!qkv || !out || !lse || !d_out || !dqkv || !delta_ws) return fail_arg(MSTG_E_BADARG, "flash_attn_bwd: null pointer");
    if (int rc = flash_check(N, L, heads, D)) return rc;
    const float scale = 1.f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)N * heads * L;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    MSTG_LAUNCH(flash_delta_kernel, dim3(grid), dim3(256), 0, st, out, d_out, delta_ws, N, L, heads, D);
    MSTG_CHECK_LAUNCH("flash_delta_kernel");
    switch (D) {
        case 8: return launch_flash<8>(1, qkv, nullptr, const_cast<float*>(lse), d_out, delta_ws, dqkv, N, L, heads, scale, st);
        case 16: return launch_flash<16>(1, qkv, nullptr, const_cast<float*>(lse), d_out, delta_ws, dqkv, N, L, heads, scale, st);
        case 32: return launch_flash<32>(1, qkv, nullptr, const_cast<float*>(lse), d_out, delta_ws, dqkv, N, L, heads, scale, st);
        default: return launch_flash<64>(1, qkv, nullptr, const_cast<float*>(lse), d_out, delta_ws, dqkv, N, L, heads, scale, st);
    }
}
