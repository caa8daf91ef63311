// Register-resident fused LocalAttention (C = 16 / 32), forward and backward: the round-3 replacement of attn_fused_*_kernel.
//
// Reference site: enhanced_generator.py:6-47 (qkv 1x1 conv, 4x4 window partition, two F.normalize, q^ k^T, softmax, attn v,
// un-partition, proj 1x1 conv) and its autograd backward.
//
// One wave owns one 4x4 window at a time (16 pixels = one MFMA tile edge) and is persistent.  Everything is built on one
// property of v_mfma_f32_16x16x4_f32: the accumulator layout (register r of lane (i = l & 15, g = l >> 4) is D[4g + r][i]) is
// exactly the operand layout of a following MFMA that contracts over D's ROW index -- instruction r of that chain takes
// register r of every lane as its A (or B) operand, lane group g supplying k = row 4g + r.  Write L(a|b) for "rows a on
// (fragment, g, r), columns b on (fragment, lane i)"; then
//
//        contract(T in L(a|b), U in L(a|c))  =  sum_a T[a][b] U[a][c]   lands in L(b|c)            (no data movement)
//
// and swapping the operands lands it in L(c|b).  A window of an NHWC tensor fetched with one 16-byte load per lane (lane i =
// pixel, channels 16h + 4g .. +3) IS the tile L(ci|p); a 1x1-conv weight fetched with 16-byte loads of its rows is L(ci|j).
// The forward therefore chains
//        q|k = contract(X, Wqk)            L(p|j)          v^T = contract(Wv, X)         L(c|p)
//        S^T = contract(k^, q^)            L(c2|c1)        softmax over rows (registers + two cross-g lane swaps)
//        O^T = contract(P^T, v^T)          L(c1|p)         Y^T = contract(Wp, O^T)       L(co|p)  -> 16-byte stores
// with the filters held in registers for the life of the wave: no LDS, no operand traffic, 12 C^2 FLOP per pixel = exactly the
// algorithmic MFMA count.  The backward needs every matrix contracted over two different indices, so it moves nine small tiles
// through wave-private LDS images (write in one layout, read in the transposed one; strides chosen conflict-free for the
// b128 / b32 lane grouping of gfx950) and re-reads x and dy in their second layout with 4-byte loads (L1 hits).  Weight
// gradients stay in MFMA accumulators across all windows of a wave; the four waves of a workgroup add their slabs in LDS in a
// fixed order, so the second-stage reduce reads 1/16 of what the one-wave kernels produced.  tools/sim_attn_layout.py is the
// lane-level model this data flow was checked with.
#include "common.h"
#include <stdlib.h>

namespace mstg {

namespace {

template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (= the lanes sharing g); every lane ends with the result
__device__ __forceinline__ float row_sum(float v) {
    v += dppm<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dppm<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dppm<0x141>(v);  // row_half_mirror
    v += dppm<0x140>(v);  // row_mirror
    return v;
}
// sum over the 4 lanes sharing i = lane & 15 (lanes i, i+16, i+32, i+48); every lane ends with the result.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper
// half of the first with the lower half of the second: with both operands holding v, first + second is the xor-16 / xor-32 sum.
// (Inline asm: the builtin's second result is mis-lowered by this hipcc; the s_nop covers the VALU-write -> permlane hazard.)
__device__ __forceinline__ float xg_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
// 1 / max(sqrt(s), 1e-12) (F.normalize's denominator) and 1 / z as single v_rsq_f32 / v_rcp_f32 instructions (1 ulp): the IEEE-exact
// sequences hipcc emits for sqrtf and '/' are ~10 VALU instructions each, a fifth of a 16-channel window's whole instruction stream
__device__ __forceinline__ float inv_norm(float s) { return fminf(__builtin_amdgcn_rsqf(s), 1e12f); }
__device__ __forceinline__ float fast_rcp(float z) { return __builtin_amdgcn_rcpf(z); }
__device__ __forceinline__ float hsum(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ void wave_lds_fence() {  // orders this wave's LDS writes before its later LDS reads (LDS is in-order per wave)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- filters and biases in the layouts the chains consume ------------------------------------------------------------------
template <int C>
struct FwdW {
    static constexpr int NF = C / 16;
    f32x4 wqkv[3 * NF][NF];  // [f][h][e] = Wqkv[16f + i][16h + 4g + e]       L(ci|j)
    float bqk[2 * NF];       // bqkv[16f + i]
    f32x4 bv[NF];            // bqkv[2C + 16f + 4g + r]
    __device__ __forceinline__ void load(const float* __restrict__ wq, const float* __restrict__ bq, int i, int g) {
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f)
#pragma unroll
            for (int h = 0; h < NF; ++h) wqkv[f][h] = *reinterpret_cast<const f32x4*>(wq + (16 * f + i) * C + 16 * h + 4 * g);
#pragma unroll
        for (int f = 0; f < 2 * NF; ++f) bqk[f] = bq[16 * f + i];
#pragma unroll
        for (int f = 0; f < NF; ++f) bv[f] = *reinterpret_cast<const f32x4*>(bq + 2 * C + 16 * f + 4 * g);
    }
};

// (mean, rstd) of the 4 channels a lane holds per fragment of the L(ci|p) window tile
template <int C>
struct NormQ {
    static constexpr int NF = C / 16;
    f32x4 a[NF], b[NF];  // stats[n][16h + 4g + {0,1}] , [.. + {2,3}] as stored (mean, rstd pairs)
    __device__ __forceinline__ void load(const float* __restrict__ stats, int n, int g) {
#pragma unroll
        for (int h = 0; h < NF; ++h) {
            const float* st = stats + ((size_t)n * C + 16 * h + 4 * g) * 2;
            a[h] = *reinterpret_cast<const f32x4*>(st);
            b[h] = *reinterpret_cast<const f32x4*>(st + 4);
        }
    }
    __device__ __forceinline__ f32x4 apply(f32x4 v, int h) const {  // norm_apply_kernel's arithmetic: (x - mean) * rstd, ReLU
        v[0] = fmaxf((v[0] - a[h][0]) * a[h][1], 0.f);
        v[1] = fmaxf((v[1] - a[h][2]) * a[h][3], 0.f);
        v[2] = fmaxf((v[2] - b[h][0]) * b[h][1], 0.f);
        v[3] = fmaxf((v[3] - b[h][2]) * b[h][3], 0.f);
        return v;
    }
};

// window tile L(ci|p): lane (i = pixel, g) <- 16 bytes at channels 16h + 4g of its pixel
template <int C>
__device__ __forceinline__ void fetch_cp(f32x4 (&t)[C / 16], const float* __restrict__ src, int H, int W, int n, int wy, int wx, int i, int g) {
    const float* p = src + (((size_t)n * H + 4 * wy + (i >> 2)) * W + 4 * wx + (i & 3)) * C + 4 * g;
#pragma unroll
    for (int h = 0; h < C / 16; ++h) t[h] = *reinterpret_cast<const f32x4*>(p + 16 * h);
}
// window tile L(p|c): lane (i = channel, g) <- register r = pixel 4g + r (window row g, column r), channel 16f + i
template <int C>
__device__ __forceinline__ void fetch_pc(f32x4 (&t)[C / 16], const float* __restrict__ src, int H, int W, int n, int wy, int wx, int i, int g) {
    const float* p = src + (((size_t)n * H + 4 * wy + g) * W + 4 * wx) * C + i;
#pragma unroll
    for (int f = 0; f < C / 16; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[f][r] = p[r * C + 16 * f];
}

// ---- the shared forward chain: x (normalised) -> q, k (raw, L(p|j)), 1/|q|, 1/|k| per row, q^, k^, v^T (L(c|p)), P^T (L(c2|c1)) ----
template <int C>
struct Chain {
    static constexpr int NF = C / 16;
    f32x4 q[NF], k[NF];    // raw q, k              L(p|c)
    f32x4 qh[NF], kh[NF];  // normalised            L(p|c)
    f32x4 vt[NF];          // v^T                   L(c|p)
    f32x4 pt[NF][NF];      // P^T[c2 frag][c1 frag] L(c2|c1)
};

template <int C>
__device__ __forceinline__ void forward_chain(Chain<C>& ch, const f32x4 (&xn)[C / 16], const FwdW<C>& w) {
    constexpr int NF = C / 16;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        ch.q[f] = splat(w.bqk[f]);
        ch.k[f] = splat(w.bqk[NF + f]);
        ch.vt[f] = w.bv[f];
    }
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                ch.q[f] = mfma16(xn[h][e], w.wqkv[f][h][e], ch.q[f]);
                ch.k[f] = mfma16(xn[h][e], w.wqkv[NF + f][h][e], ch.k[f]);
                ch.vt[f] = mfma16(w.wqkv[2 * NF + f][h][e], xn[h][e], ch.vt[f]);
            }
        }
    // F.normalize(dim = channels): v / max(|v|, 1e-12); a row (g, r) is a pixel, its channels lie across the 16 lanes and NF fragments
    f32x4 sq = splat(0.f), sk = splat(0.f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        sq += ch.q[f] * ch.q[f];
        sk += ch.k[f] * ch.k[f];
    }
    f32x4 iq, ik;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        iq[r] = inv_norm(row_sum(sq[r]));
        ik[r] = inv_norm(row_sum(sk[r]));
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        ch.qh[f] = ch.q[f] * iq;
        ch.kh[f] = ch.k[f] * ik;
    }
    // S^T[c2][c1] = sum_p k^[p][c2] q^[p][c1]; |S| <= 1 (unit vectors), so exp needs no max subtraction
    f32x4 st[NF][NF];
#pragma unroll
    for (int m = 0; m < NF; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) st[m][n] = splat(0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) st[m][n] = mfma16(ch.kh[m][r], ch.qh[n][r], st[m][n]);
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        float z = 0.f;
#pragma unroll
        for (int m = 0; m < NF; ++m) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[m][n][r] = __expf(st[m][n][r]);
            z += hsum(st[m][n]);
        }
        const float inv = fast_rcp(xg_sum(z));
#pragma unroll
        for (int m = 0; m < NF; ++m) ch.pt[m][n] = st[m][n] * inv;
    }
}

template <int C>
constexpr int fwd_waves_per_simd() { return C == 16 ? 4 : 2; }

// A wave walks a contiguous range of windows (row-major over (image, window row, window column)); its coordinates are wave-uniform
// scalars advanced by carries -- the three divisions happen once per wave, not per window.
struct WinWalk {
    int n, wy, wx, nwx, nwy;
    __device__ __forceinline__ void start(int win, int nwx_, int nwy_) {
        nwx = nwx_;
        nwy = nwy_;
        wx = win % nwx;
        const int t = win / nwx;
        wy = t % nwy;
        n = t / nwy;
    }
    __device__ __forceinline__ void next() {
        if (++wx == nwx) {
            wx = 0;
            if (++wy == nwy) {
                wy = 0;
                ++n;
            }
        }
    }
};
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace

// ============================================================================================================================
// forward
// ============================================================================================================================
template <int C, bool NORM>
__global__ __launch_bounds__(256, fwd_waves_per_simd<C>()) void attn_reg_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wqkv, const float* __restrict__ bqkv, const float* __restrict__ wp,
    const float* __restrict__ bp, float* __restrict__ y, int N, int H, int W, const float* __restrict__ in_stats) {
    constexpr int NF = C / 16;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwx * nwy;
    const int wv = uniform(blockIdx.x * 4 + (threadIdx.x >> 6)), nwv = gridDim.x * 4;
    const int per = (nwin + nwv - 1) / nwv, w0 = wv * per, w1 = w0 + per < nwin ? w0 + per : nwin;
    if (w0 >= nwin) return;
    FwdW<C> w;
    w.load(wqkv, bqkv, i, g);
    f32x4 wpr[NF][NF], bpv[NF];  // Wp[16cf + i][16n + 4g + r]: L(c1|co);  bp[16cf + 4g + r]
#pragma unroll
    for (int cf = 0; cf < NF; ++cf) {
#pragma unroll
        for (int n = 0; n < NF; ++n) wpr[cf][n] = *reinterpret_cast<const f32x4*>(wp + (16 * cf + i) * C + 16 * n + 4 * g);
        bpv[cf] = *reinterpret_cast<const f32x4*>(bp + 16 * cf + 4 * g);
    }
    f32x4 xr[NF];
    NormQ<C> nq;
    WinWalk cur, nxt;
    cur.start(w0, nwx, nwy);
    nxt = cur;
    fetch_cp<C>(xr, x, H, W, cur.n, cur.wy, cur.wx, i, g);
    int n_stats = -1;
    for (int win = w0; win < w1; ++win) {
        const int n = cur.n, wy = cur.wy, wx = cur.wx;
        f32x4 xn[NF];
        if (NORM) {
            if (n != n_stats) {
                nq.load(in_stats, n, g);
                n_stats = n;
            }
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = nq.apply(xr[h], h);
        } else {
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = xr[h];
        }
        if (win + 1 < w1) nxt.next();  // next window in flight behind this one's chain (the last iteration re-fetches its own)
        fetch_cp<C>(xr, x, H, W, nxt.n, nxt.wy, nxt.wx, i, g);
        cur = nxt;
        Chain<C> ch;
        forward_chain<C>(ch, xn, w);
        // O^T[c1][p] = sum_c2 P^T[c2][c1] v^T[c2][p]
        f32x4 ot[NF];
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) ot[n1] = splat(0.f);
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) ot[n1] = mfma16(ch.pt[m][n1][r], ch.vt[m][r], ot[n1]);
        // Y^T[co][p] = bp[co] + sum_c1 Wp[co][c1] O^T[c1][p]
        f32x4 yt[NF];
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) yt[cf] = bpv[cf];
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) yt[cf] = mfma16(wpr[cf][n1][r], ot[n1][r], yt[cf]);
        float* dst = y + (((size_t)n * H + 4 * wy + (i >> 2)) * W + 4 * wx + (i & 3)) * C + 4 * g;
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) *reinterpret_cast<f32x4*>(dst + 16 * cf) = yt[cf];
    }
}

// ============================================================================================================================
// backward
// ============================================================================================================================
// wave-private LDS images (floats).  "RC": written L(p|c) with 4-byte stores, read L(c|p) with 16-byte loads (stride C + 8);
// "CR" / square: written with 16-byte stores, read with 4-byte loads (stride = 4 mod 8): both sides conflict-free or hidden.
template <int C, int WAVES>
struct BwdLds {
    static constexpr int LD_RC = C + 8, LD_CR = C + 4, LD_J = 3 * C + 4;
    // One wave's images.  P^T's image is dead once dV has read it, so dS^T reuses it; q, k, v and dO are all consumed before dQKV is
    // written, so dQKV overlays them: 14.6 KB per wave at C = 32 (25.6 without the reuse), 7.2 KB at C = 16.  LDS operations of one
    // wave execute in order, so a later write never overtakes an earlier read of the bytes it reuses.
    static constexpr int Q = 0, K = Q + 16 * LD_RC, V = K + 16 * LD_RC, DO = V + 16 * LD_CR, P = DO + 16 * LD_RC, DS = P,
                         END = P + C * LD_CR, DQKV = 0;
    static_assert(16 * LD_J <= P, "dQKV overlays the q / k / v / dO images");
    static constexpr int SLAB = 4 * C * C + 4 * C;
    // the filters, shared by the workgroup's waves, in the three layouts the chains read with one conflict-free 16-byte load per
    // fragment (strides = 8 mod 16): WQ[j][ci] (forward chain), WT[ci][j] = Wqkv^T (dX), WPT[c][co] = Wp^T (dO)
    static constexpr int LD_WQ = C + 8, LD_WT = 3 * C + 8, LD_WP = C + 8;
    static constexpr int WQ = WAVES * END, WT = WQ + 3 * C * LD_WQ, WPT = WT + C * LD_WT, WG_FLOATS = WPT + C * LD_WP;
    static_assert(WAVES * END >= SLAB, "the workgroup's gradient slab is staged in the transpose images");
};

// L(p|c) (one row fragment, NF column fragments) -> LDS image [p][c]
template <int NF>
__device__ __forceinline__ void put_pc(float* img, int ld, const f32x4 (&t)[NF], int i, int g) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(4 * g + r) * ld + 16 * f + i] = t[f][r];
}
template <int NF>
__device__ __forceinline__ void get_pc(f32x4 (&t)[NF], const float* img, int ld, int i, int g) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[f][r] = img[(4 * g + r) * ld + 16 * f + i];
}
// L(c|p) (NF row fragments, one column fragment) <-> LDS image [p][c]
template <int NF>
__device__ __forceinline__ void put_cp(float* img, int ld, const f32x4 (&t)[NF], int i, int g) {
#pragma unroll
    for (int f = 0; f < NF; ++f) *reinterpret_cast<f32x4*>(&img[i * ld + 16 * f + 4 * g]) = t[f];
}
template <int NF>
__device__ __forceinline__ void get_cp(f32x4 (&t)[NF], const float* img, int ld, int i, int g) {
#pragma unroll
    for (int f = 0; f < NF; ++f) t[f] = *reinterpret_cast<const f32x4*>(&img[i * ld + 16 * f + 4 * g]);
}
// square: t in L(a|b) [fa][fb] -> image [b][a] (16-byte stores) -> read as L(b|a) [fb][fa] (4-byte loads)
template <int NF>
__device__ __forceinline__ void put_sq(float* img, int ld, const f32x4 (&t)[NF][NF], int i, int g) {
#pragma unroll
    for (int fa = 0; fa < NF; ++fa)
#pragma unroll
        for (int fb = 0; fb < NF; ++fb) *reinterpret_cast<f32x4*>(&img[(16 * fb + i) * ld + 16 * fa + 4 * g]) = t[fa][fb];
}
template <int NF>
__device__ __forceinline__ void get_sq(f32x4 (&t)[NF][NF], const float* img, int ld, int i, int g) {
#pragma unroll
    for (int fb = 0; fb < NF; ++fb)
#pragma unroll
        for (int fa = 0; fa < NF; ++fa)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[fb][fa][r] = img[(16 * fb + 4 * g + r) * ld + 16 * fa + i];
}

// NORM: x is the RAW tensor in front of the stage's InstanceNorm + ReLU (normalised while loaded), dx is the gradient w.r.t. the
// normalised, ReLU'd tensor z, and the kernel also emits what that norm's backward needs from a pass over dx: per run of kblk
// consecutive windows (kblk divides the windows per image) one row nsum[run][2][C] of sum dz [z > 0] and sum dz [z > 0] z.
// WAVES waves per workgroup share the filters in LDS: 4 at C = 16 (four workgroups per CU), 8 at C = 32 (one workgroup per CU, two
// waves per SIMD).  Measured: the matrix pipe is 63 % busy either way (one or two waves per SIMD, with or without starting the
// upper four waves half a window late) -- the chain's MFMA -> VALU -> MFMA dependencies, not latency, set the pace.
template <int C, bool NORM, int WAVES>
__global__ __launch_bounds__(64 * WAVES, C == 16 ? 4 : WAVES / 4) void attn_reg_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wqkv, const float* __restrict__ bqkv, const float* __restrict__ wp,
    const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ partial, int N, int H, int W,
    const float* __restrict__ in_stats, float* __restrict__ nsum, int kblk) {
    constexpr int NF = C / 16, NT = 64 * WAVES;
    typedef BwdLds<C, WAVES> S;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    float* my = sm + wave * S::END;
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwx * nwy, nrun = nwin / kblk;
    const int wv = uniform(blockIdx.x * WAVES + wave), nwv = gridDim.x * WAVES;
    // this wave's runs [r0, r1) = windows [r0 * kblk, r1 * kblk): contiguous, so the walker advances by carries
    const int perr = (nrun + nwv - 1) / nwv, r0 = wv * perr < nrun ? wv * perr : nrun, r1 = r0 + perr < nrun ? r0 + perr : nrun;
    const int w0 = r0 * kblk, w1 = r1 * kblk;

    // filters -> LDS once per workgroup (registers are for the gradient accumulators: with the three filter layouts in
    // registers the kernel needed 256 VGPRs + ~170 AGPRs and ~140 accvgpr moves per window)
    for (int e = threadIdx.x; e < 3 * C * C; e += NT) {
        const int j = e / C, ci = e - j * C;
        const float v = wqkv[e];
        sm[S::WQ + j * S::LD_WQ + ci] = v;
        sm[S::WT + ci * S::LD_WT + j] = v;
    }
    for (int e = threadIdx.x; e < C * C; e += NT) {
        const int co = e / C, c = e - co * C;
        sm[S::WPT + c * S::LD_WP + co] = wp[e];
    }
    __syncthreads();
    FwdW<C> w;  // biases stay in registers; w.wqkv is re-read from LDS per window
#pragma unroll
    for (int f = 0; f < 2 * NF; ++f) w.bqk[f] = bqkv[16 * f + i];
#pragma unroll
    for (int f = 0; f < NF; ++f) w.bv[f] = *reinterpret_cast<const f32x4*>(bqkv + 2 * C + 16 * f + 4 * g);

    // parameter-gradient accumulators, alive across every window of this wave
    f32x4 gW[3 * NF][NF], gWp[NF][NF];  // dWqkv L(j|ci), dWp L(co|c)
    float gb[3 * NF], gbp[NF];          // per lane (i = j / co), still to be summed over g
#pragma unroll
    for (int jf = 0; jf < 3 * NF; ++jf) {
        gb[jf] = 0.f;
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) gW[jf][cf] = splat(0.f);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        gbp[f] = 0.f;
#pragma unroll
        for (int n = 0; n < NF; ++n) gWp[f][n] = splat(0.f);
    }
    f32x4 ns1[NORM ? NF : 1], ns2[NORM ? NF : 1];
#pragma unroll
    for (int f = 0; f < (NORM ? NF : 1); ++f) ns1[f] = ns2[f] = splat(0.f);

    f32x4 xr[NF], dyr[NF];
    WinWalk cur, nxt;
    cur.start(w0, nwx, nwy);
    nxt = cur;
    if (w0 < w1) {
        fetch_cp<C>(xr, x, H, W, cur.n, cur.wy, cur.wx, i, g);
        fetch_cp<C>(dyr, dy, H, W, cur.n, cur.wy, cur.wx, i, g);
    }
    for (int win = w0, run = r0, j = 0; win < w1; ++win) {
        const int n = cur.n, wy = cur.wy, wx = cur.wx;
        const bool run_ends = j + 1 == kblk;
        // ---- operands of this window ------------------------------------------------------------------------------------
        f32x4 xn[NF], dyt[NF];
        if (NORM) {
            NormQ<C> nq;  // (mean, rstd) re-read per window (L1 hits): sixteen registers less across the window than keeping them
            nq.load(in_stats, n, g);
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = nq.apply(xr[h], h);
        } else {
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = xr[h];
        }
#pragma unroll
        for (int h = 0; h < NF; ++h) dyt[h] = dyr[h];
        if (win + 1 < w1) nxt.next();  // the next window's x and dy in flight behind this one (the last re-fetches its own)
        fetch_cp<C>(xr, x, H, W, nxt.n, nxt.wy, nxt.wx, i, g);
        fetch_cp<C>(dyr, dy, H, W, nxt.n, nxt.wy, nxt.wx, i, g);
        cur = nxt;
        // ---- recompute the forward ---------------------------------------------------------------------------------------
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f)
#pragma unroll
            for (int h = 0; h < NF; ++h) w.wqkv[f][h] = *reinterpret_cast<const f32x4*>(&sm[S::WQ + (16 * f + i) * S::LD_WQ + 16 * h + 4 * g]);
        Chain<C> ch;
        forward_chain<C>(ch, xn, w);
        put_pc<NF>(my + S::Q, S::LD_RC, ch.q, i, g);   // raw q, k -> L(c|p) later
        put_pc<NF>(my + S::K, S::LD_RC, ch.k, i, g);
        put_cp<NF>(my + S::V, S::LD_CR, ch.vt, i, g);  // v^T -> L(p|c) later
        put_sq<NF>(my + S::P, S::LD_CR, ch.pt, i, g);  // P^T -> P later
        // O[p][c1] = sum_c2 v^T[c2][p] P^T[c2][c1]                                                          L(p|c1)
        f32x4 o_pc[NF];
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) o_pc[n1] = splat(0.f);
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) o_pc[n1] = mfma16(ch.vt[m][r], ch.pt[m][n1][r], o_pc[n1]);
        // ---- proj backward: dO = dY Wp (L(p|c1)); dWp += dY^T O; dbp += colsum dY ------------------------------------------
        f32x4 dy_pc[NF];  // the L(p|co) copy of dY (4-byte loads of the lines the 16-byte loads brought in), in flight behind dO
        fetch_pc<C>(dy_pc, dy, H, W, n, wy, wx, i, g);
        f32x4 dO[NF], wpT[NF][NF];  // wpT[f][n][e] = Wp[16f + 4g + e][16n + i]   L(co|c)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1) wpT[f][n1] = *reinterpret_cast<const f32x4*>(&sm[S::WPT + (16 * n1 + i) * S::LD_WP + 16 * f + 4 * g]);
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) dO[n1] = splat(0.f);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) dO[n1] = mfma16(dyt[f][e], wpT[f][n1][e], dO[n1]);
        put_pc<NF>(my + S::DO, S::LD_RC, dO, i, g);    // dO -> L(c1|p) later
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) gWp[f][n1] = mfma16(dy_pc[f][r], o_pc[n1][r], gWp[f][n1]);
#pragma unroll
        for (int f = 0; f < NF; ++f) gbp[f] += hsum(dy_pc[f]);
        // ---- dP^T[c2][c1] = sum_p v[p][c2] dO[p][c1];  dS^T = P^T (dP^T - sum_c2 dP^T P^T) ------------------------------------
        wave_lds_fence();
        f32x4 v_pc[NF];
        get_pc<NF>(v_pc, my + S::V, S::LD_CR, i, g);
        f32x4 dSt[NF][NF];
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1) dSt[m][n1] = splat(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < NF; ++m)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) dSt[m][n1] = mfma16(v_pc[m][r], dO[n1][r], dSt[m][n1]);
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) {
            float d = 0.f;
#pragma unroll
            for (int m = 0; m < NF; ++m) d += hsum(dSt[m][n1] * ch.pt[m][n1]);
            d = xg_sum(d);
#pragma unroll
            for (int m = 0; m < NF; ++m) dSt[m][n1] = ch.pt[m][n1] * (dSt[m][n1] - splat(d));
        }
        // ---- dV^T[c2][p] = sum_c1 P[c1][c2] dO^T[c1][p] ------------------------------------------------------------------------
        f32x4 dqkv[3 * NF];  // dq | dk | dv in L(j|p)
        {
            f32x4 p12[NF][NF], dOt[NF];
            get_sq<NF>(p12, my + S::P, S::LD_CR, i, g);   // [c1 frag][c2 frag]
            get_cp<NF>(dOt, my + S::DO, S::LD_RC, i, g);  // L(c1|p)
#pragma unroll
            for (int m = 0; m < NF; ++m) dqkv[2 * NF + m] = splat(0.f);
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < NF; ++m) dqkv[2 * NF + m] = mfma16(p12[a][m][r], dOt[a][r], dqkv[2 * NF + m]);
        }
        put_sq<NF>(my + S::DS, S::LD_CR, dSt, i, g);   // dS^T -> dS later; into P^T's image, which dV has just read
        // ---- q, k in L(c|p); norms again in that orientation (registers + cross-g) ------------------------------------------
        f32x4 qh_cp[NF], kh_cp[NF];
        float iq2, ik2;
        {
            get_cp<NF>(qh_cp, my + S::Q, S::LD_RC, i, g);
            get_cp<NF>(kh_cp, my + S::K, S::LD_RC, i, g);
            float sq = 0.f, sk = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                sq += hsum(qh_cp[f] * qh_cp[f]);
                sk += hsum(kh_cp[f] * kh_cp[f]);
            }
            iq2 = inv_norm(xg_sum(sq));
            ik2 = inv_norm(xg_sum(sk));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                qh_cp[f] *= iq2;
                kh_cp[f] *= ik2;
            }
        }
        // dq^[c1][p] = sum_c2 dS^T[c2][c1] k^[c2][p]
        f32x4 dqh[NF], dkh[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) dqh[f] = dkh[f] = splat(0.f);
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) dqh[n1] = mfma16(dSt[m][n1][r], kh_cp[m][r], dqh[n1]);
        // dk^[c2][p] = sum_c1 dS[c1][c2] q^[c1][p]
        wave_lds_fence();
        {
            f32x4 dS12[NF][NF];
            get_sq<NF>(dS12, my + S::DS, S::LD_CR, i, g);  // [c1 frag][c2 frag]
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < NF; ++m) dkh[m] = mfma16(dS12[a][m][r], qh_cp[a][r], dkh[m]);
        }
        {   // backward of F.normalize: dq = (dq^ - q^ (q^ . dq^)) / max(|q|, eps), per pixel (= per lane i)
            float dq_dot = 0.f, dk_dot = 0.f;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                dq_dot += hsum(qh_cp[f] * dqh[f]);
                dk_dot += hsum(kh_cp[f] * dkh[f]);
            }
            dq_dot = xg_sum(dq_dot);
            dk_dot = xg_sum(dk_dot);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                dqkv[f] = (dqh[f] - qh_cp[f] * dq_dot) * iq2;
                dqkv[NF + f] = (dkh[f] - kh_cp[f] * dk_dot) * ik2;
            }
        }
        put_cp<3 * NF>(my + S::DQKV, S::LD_J, dqkv, i, g);  // dQKV -> L(p|j) later
        // ---- qkv conv backward: dX^T[ci][p] = sum_j Wqkv[j][ci] dQKV^T[j][p] ---------------------------------------------------
        f32x4 x_pc[NF];  // the L(p|ci) copy of X for dWqkv, in flight behind the dX chain
        fetch_pc<C>(x_pc, x, H, W, n, wy, wx, i, g);
        f32x4 dXt[NF];
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) dXt[cf] = splat(0.f);
#pragma unroll
        for (int jf = 0; jf < 3 * NF; ++jf) {
            f32x4 w2[NF];  // w2[cf][r] = Wqkv[16jf + 4g + r][16cf + i]   L(j|ci)
#pragma unroll
            for (int cf = 0; cf < NF; ++cf) w2[cf] = *reinterpret_cast<const f32x4*>(&sm[S::WT + (16 * cf + i) * S::LD_WT + 16 * jf + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) dXt[cf] = mfma16(w2[cf][r], dqkv[jf][r], dXt[cf]);
        }
        {
            float* dst = dx + (((size_t)n * H + 4 * wy + (i >> 2)) * W + 4 * wx + (i & 3)) * C + 4 * g;
#pragma unroll
            for (int cf = 0; cf < NF; ++cf) *reinterpret_cast<f32x4*>(dst + 16 * cf) = dXt[cf];
        }
        if (NORM) {  // xn = z = relu(x^): where z > 0 it IS x^, elsewhere the element contributes nothing
#pragma unroll
            for (int cf = 0; cf < NF; ++cf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gg = xn[cf][e] > 0.f ? dXt[cf][e] : 0.f;
                    ns1[cf][e] += gg;
                    ns2[cf][e] += gg * xn[cf][e];
                }
        }
        // ---- dWqkv[j][ci] += sum_p dQKV[p][j] X[p][ci]; dbqkv += colsum dQKV ----------------------------------------------------
        if (NORM) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x2 st2 = *reinterpret_cast<const f32x2*>(in_stats + ((size_t)n * C + 16 * f + i) * 2);  // (mean, rstd) of channel 16f + i
#pragma unroll
                for (int r = 0; r < 4; ++r) x_pc[f][r] = fmaxf((x_pc[f][r] - st2[0]) * st2[1], 0.f);
            }
        }
        wave_lds_fence();
        {
            f32x4 dq_pj[3 * NF];
            get_pc<3 * NF>(dq_pj, my + S::DQKV, S::LD_J, i, g);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jf = 0; jf < 3 * NF; ++jf)
#pragma unroll
                    for (int cf = 0; cf < NF; ++cf) gW[jf][cf] = mfma16(dq_pj[jf][r], x_pc[cf][r], gW[jf][cf]);
#pragma unroll
            for (int jf = 0; jf < 3 * NF; ++jf) gb[jf] += hsum(dq_pj[jf]);
        }
        if (run_ends) {
            if (NORM) {
                float* row = nsum + (size_t)run * 2 * C;
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) {
                    f32x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = row_sum(ns1[cf][e]);
                        b[e] = row_sum(ns2[cf][e]);
                    }
                    if (i == 0) {
                        *reinterpret_cast<f32x4*>(row + 16 * cf + 4 * g) = a;
                        *reinterpret_cast<f32x4*>(row + C + 16 * cf + 4 * g) = b;
                    }
                    ns1[cf] = ns2[cf] = splat(0.f);
                }
            }
            ++run;
            j = 0;
        } else {
            ++j;
        }
    }
    // ---- this workgroup's slab = wave 0 + wave 1 + ... (fixed order): dWqkv (3C x C) | dWp (C x C) | dbqkv | dbp ----
    __syncthreads();  // every wave is done with its transpose images
    for (int src = 0; src < WAVES; ++src) {
        if (wave == src) {
            const bool first = src == 0;
#pragma unroll
            for (int jf = 0; jf < 3 * NF; ++jf)
#pragma unroll
                for (int cf = 0; cf < NF; ++cf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = &sm[(16 * jf + 4 * g + r) * C + 16 * cf + i];
                        *p = first ? gW[jf][cf][r] : *p + gW[jf][cf][r];
                    }
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = &sm[3 * C * C + (16 * f + 4 * g + r) * C + 16 * n1 + i];
                        *p = first ? gWp[f][n1][r] : *p + gWp[f][n1][r];
                    }
#pragma unroll
            for (int jf = 0; jf < 3 * NF; ++jf) {
                const float v = xg_sum(gb[jf]);
                if (g == 0) {
                    float* p = &sm[4 * C * C + 16 * jf + i];
                    *p = first ? v : *p + v;
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float v = xg_sum(gbp[f]);
                if (g == 0) {
                    float* p = &sm[4 * C * C + 3 * C + 16 * f + i];
                    *p = first ? v : *p + v;
                }
            }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * S::SLAB;
    for (int e = threadIdx.x; e < S::SLAB; e += NT) out[e] = sm[e];
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
static int reg_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// waves per workgroup / workgroups per CU of the backward
constexpr int bwd_waves(int C) { return C == 16 ? 4 : 8; }
constexpr int bwd_wgs_per_cu(int C) { return C == 16 ? 4 : 1; }
// most workgroups the backward uses (= slabs in its workspace)
int attn_reg_bwd_blocks(int C) { return reg_cus() * bwd_wgs_per_cu(C); }
int attn_reg_bwd_waves(int C) { return bwd_waves(C); }

template <int C>
static int reg_fwd(const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* bp, float* y,
                   int N, int H, int W, hipStream_t st) {
    const int nwin = N * (H / 4) * (W / 4);
    int nb = reg_cus() * fwd_waves_per_simd<C>();
    if (nb * 4 > nwin) nb = cdiv(nwin, 4);
    if (in_stats)
        MSTG_LAUNCH((attn_reg_fwd_kernel<C, true>), dim3(nb), dim3(256), 0, st, x, wqkv, bqkv, wp, bp, y, N, H, W, in_stats);
    else
        MSTG_LAUNCH((attn_reg_fwd_kernel<C, false>), dim3(nb), dim3(256), 0, st, x, wqkv, bqkv, wp, bp, y, N, H, W, in_stats);
    MSTG_CHECK_LAUNCH("attn_reg_fwd_kernel");
    return MSTG_OK;
}

template <int C>
static int reg_bwd(const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* dy, float* dx,
                   float* partial, float* nsum, int kblk, int nb, int N, int H, int W, hipStream_t st) {
    constexpr int WAVES = bwd_waves(C);
    typedef BwdLds<C, WAVES> S;
    const size_t lds = (size_t)S::WG_FLOATS * sizeof(float);
    if (in_stats) {
        static bool once = ((void)hipFuncSetAttribute((const void*)attn_reg_bwd_kernel<C, true, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        MSTG_LAUNCH((attn_reg_bwd_kernel<C, true, WAVES>), dim3(nb), dim3(64 * WAVES), lds, st, x, wqkv, bqkv, wp, dy, dx, partial, N, H, W, in_stats, nsum, kblk);
    } else {
        static bool once = ((void)hipFuncSetAttribute((const void*)attn_reg_bwd_kernel<C, false, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        MSTG_LAUNCH((attn_reg_bwd_kernel<C, false, WAVES>), dim3(nb), dim3(64 * WAVES), lds, st, x, wqkv, bqkv, wp, dy, dx, partial, N, H, W, in_stats, nsum, 1);
    }
    MSTG_CHECK_LAUNCH("attn_reg_bwd_kernel");
    return MSTG_OK;
}

int attn_reg_fwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* bp,
                 float* y, int N, int H, int W, hipStream_t st) {
    return C == 16 ? reg_fwd<16>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st) : reg_fwd<32>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st);
}
int attn_reg_bwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* dy,
                 float* dx, float* partial, float* nsum, int kblk, int nblocks, int N, int H, int W, hipStream_t st) {
    return C == 16 ? reg_bwd<16>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st)
                   : reg_bwd<32>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st);
}

}  // namespace mstg
