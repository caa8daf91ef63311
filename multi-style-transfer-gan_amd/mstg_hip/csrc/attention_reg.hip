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
    // wave-uniform window offset (scalar arithmetic) + the lane's constant offset inside a window (loop-invariant)
    const float* p = src + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(((i >> 2) * W + (i & 3)) * C + 4 * g);
#pragma unroll
    for (int h = 0; h < C / 16; ++h) t[h] = *reinterpret_cast<const f32x4*>(p + 16 * h);
}
// window tile L(p|c): lane (i = channel, g) <- register r = pixel 4g + r (window row g, column r), channel 16f + i
template <int C>
__device__ __forceinline__ void fetch_pc(f32x4 (&t)[C / 16], const float* __restrict__ src, int H, int W, int n, int wy, int wx, int i, int g) {
    const float* p = src + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(g * W * C + i);
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

// second half of the forward chain: q, k (raw) -> q^, k^ -> S^T -> P^T
template <int C>
__device__ __forceinline__ void softmax_chain(Chain<C>& ch);

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
    softmax_chain<C>(ch);
}

// the same first half with the qkv filter streamed from an LDS image wq[j][ci] (row stride ld = 8 mod 16 floats) and the bias from
// bq[3C] (LDS too): one input-channel fragment's 3 C / 16 filter fragments are in registers at a time (C = 64: the whole filter would
// be 192 registers).  The scheduling barriers keep the compiler from hoisting every fragment's loads to the top.
template <int C>
__device__ __forceinline__ void forward_chain_lds(Chain<C>& ch, const f32x4 (&xn)[C / 16], const float* wq, int ld, const float* bq, int i, int g) {
    constexpr int NF = C / 16;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        ch.q[f] = splat(bq[16 * f + i]);
        ch.k[f] = splat(bq[C + 16 * f + i]);
        ch.vt[f] = *reinterpret_cast<const f32x4*>(&bq[2 * C + 16 * f + 4 * g]);
    }
#pragma unroll
    for (int h = 0; h < NF; ++h) {
        f32x4 wv[3 * NF];
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f) wv[f] = *reinterpret_cast<const f32x4*>(&wq[(16 * f + i) * ld + 16 * h + 4 * g]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                ch.q[f] = mfma16(xn[h][e], wv[f][e], ch.q[f]);
                ch.k[f] = mfma16(xn[h][e], wv[NF + f][e], ch.k[f]);
                ch.vt[f] = mfma16(wv[2 * NF + f][e], xn[h][e], ch.vt[f]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    softmax_chain<C>(ch);
}

template <int C>
__device__ __forceinline__ void softmax_chain(Chain<C>& ch) {
    constexpr int NF = C / 16;
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
        float* dst = y + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(((i >> 2) * W + (i & 3)) * C + 4 * g);
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
            float* dst = dx + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(((i >> 2) * W + (i & 3)) * C + 4 * g);
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

// ============================================================================================================================
// C = 64: the same chains, filters in LDS for both directions, square transposes one fragment row at a time
// ============================================================================================================================
// At C = 64 the qkv filter alone is 192 registers per lane, the weight-gradient accumulators are 256, and a whole P^T / dS^T
// transpose image is 17 KB per wave.  So: (a) the forward streams both filters from LDS images (one input-channel fragment's rows at a
// time); (b) the backward runs one wave per SIMD (512 registers: 256 accumulators + the window's working set), keeps Wqkv[j][ci] and
// Wp^T in LDS and reads the Wqkv^T operand of the dX chain from the SAME image with 4-byte loads; (c) a C x C matrix is transposed
// fragment row by fragment row through one 16 x (C + 4) image -- LDS operations of a wave execute in order, so the next row's stores
// cannot overtake this row's loads -- and that image aliases v's, which is consumed first.  146 KB per workgroup of four waves.
template <int C>
struct BigLds {
    static constexpr int NF = C / 16, LD_RC = C + 8, LD_CR = C + 4, LD_J = 3 * C + 4, LD_W = C + 8, WAVES = 4;
    static constexpr int Q = 0, K = Q + 16 * LD_RC, V = K + 16 * LD_RC, SQ = V, DO = V + 16 * LD_CR, END = DO + 16 * LD_RC, DQKV = 0;
    static_assert(16 * LD_J <= DO, "dQKV overlays the q / k / v images");
    static constexpr int SLAB = 4 * C * C + 4 * C;
    static constexpr int WQ = WAVES * END, WPT = WQ + 3 * C * LD_W, BQ = WPT + C * LD_W, SLOT = BQ + 3 * C, BWD_FLOATS = SLOT + 4 * WAVES;
    static constexpr int F_WQ = 0, F_WP = 3 * C * LD_W, F_BQ = 4 * C * LD_W, FWD_FLOATS = F_BQ + 3 * C;  // forward: Wqkv[j][ci] | Wp[co][c] | bqkv
};

template <int C, bool NORM>
__global__ __launch_bounds__(256, 2) void attn_big_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wqkv,
                                                              const float* __restrict__ bqkv, const float* __restrict__ wp,
                                                              const float* __restrict__ bp, float* __restrict__ y, int N, int H, int W,
                                                              const float* __restrict__ in_stats) {
    constexpr int NF = C / 16;
    typedef BigLds<C> S;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    for (int e = threadIdx.x; e < 3 * C * C; e += 256) {
        const int j = e / C, ci = e - j * C;
        sm[S::F_WQ + j * S::LD_W + ci] = wqkv[e];
    }
    for (int e = threadIdx.x; e < C * C; e += 256) {
        const int co = e / C, c = e - co * C;
        sm[S::F_WP + co * S::LD_W + c] = wp[e];
    }
    for (int e = threadIdx.x; e < 3 * C; e += 256) sm[S::F_BQ + e] = bqkv[e];
    __syncthreads();
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwx * nwy;
    const int wv = uniform(blockIdx.x * 4 + (threadIdx.x >> 6)), nwv = gridDim.x * 4;
    const int per = (nwin + nwv - 1) / nwv, w0 = wv * per, w1 = w0 + per < nwin ? w0 + per : nwin;
    if (w0 >= nwin) return;
    f32x4 bpv[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bpv[f] = *reinterpret_cast<const f32x4*>(bp + 16 * f + 4 * g);
    f32x4 xr[NF];
    WinWalk cur, nxt;
    cur.start(w0, nwx, nwy);
    nxt = cur;
    fetch_cp<C>(xr, x, H, W, cur.n, cur.wy, cur.wx, i, g);
    for (int win = w0; win < w1; ++win) {
        const int n = cur.n, wy = cur.wy, wx = cur.wx;
        f32x4 xn[NF];
        if (NORM) {
            NormQ<C> nq;
            nq.load(in_stats, n, g);
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = nq.apply(xr[h], h);
        } else {
#pragma unroll
            for (int h = 0; h < NF; ++h) xn[h] = xr[h];
        }
        if (win + 1 < w1) nxt.next();
        fetch_cp<C>(xr, x, H, W, nxt.n, nxt.wy, nxt.wx, i, g);
        cur = nxt;
        Chain<C> ch;
        forward_chain_lds<C>(ch, xn, sm + S::F_WQ, S::LD_W, sm + S::F_BQ, i, g);
        f32x4 ot[NF];
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) ot[n1] = splat(0.f);
#pragma unroll
        for (int m = 0; m < NF; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) ot[n1] = mfma16(ch.pt[m][n1][r], ch.vt[m][r], ot[n1]);
        f32x4 yt[NF];
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) yt[cf] = bpv[cf];
#pragma unroll
        for (int n1 = 0; n1 < NF; ++n1) {
            f32x4 wv[NF];  // Wp[16cf + i][16n1 + 4g + r]   L(c1|co)
#pragma unroll
            for (int cf = 0; cf < NF; ++cf) wv[cf] = *reinterpret_cast<const f32x4*>(&sm[S::F_WP + (16 * cf + i) * S::LD_W + 16 * n1 + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) yt[cf] = mfma16(wv[cf][r], ot[n1][r], yt[cf]);
        }
        float* dst = y + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(((i >> 2) * W + (i & 3)) * C + 4 * g);
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) *reinterpret_cast<f32x4*>(dst + 16 * cf) = yt[cf];
    }
}

template <int C, bool NORM>
__global__ __launch_bounds__(256, 1) void attn_big_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wqkv,
                                                              const float* __restrict__ bqkv, const float* __restrict__ wp,
                                                              const float* __restrict__ dy, float* __restrict__ dx,
                                                              float* __restrict__ partial, int N, int H, int W,
                                                              const float* __restrict__ in_stats, float* __restrict__ nsum, int kblk) {
    constexpr int NF = C / 16, WAVES = 4, NT = 256;
    constexpr int JOWN = (3 * NF + WAVES - 1) / WAVES, FOWN = (NF + WAVES - 1) / WAVES;  // dWqkv row / dWp row fragments a wave owns
    typedef BigLds<C> S;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    float* my = sm + wave * S::END;
    int* slot = reinterpret_cast<int*>(sm + S::SLOT);  // [wave][4]: image, window row, window column of the wave's current window (-1: none)
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwx * nwy, nrun = nwin / kblk;
    const int wv = uniform(blockIdx.x * WAVES + wave), nwv = gridDim.x * WAVES;
    const int perr = (nrun + nwv - 1) / nwv, r0 = wv * perr < nrun ? wv * perr : nrun, r1 = r0 + perr < nrun ? r0 + perr : nrun;
    const int w0 = r0 * kblk, w1 = r1 * kblk, iters = perr * kblk;  // every wave of the grid makes `iters` trips (the barriers below)

    for (int e = threadIdx.x; e < 3 * C * C; e += NT) {
        const int j = e / C, ci = e - j * C;
        sm[S::WQ + j * S::LD_W + ci] = wqkv[e];
    }
    for (int e = threadIdx.x; e < C * C; e += NT) {
        const int co = e / C, c = e - co * C;
        sm[S::WPT + c * S::LD_W + co] = wp[e];
    }
    for (int e = threadIdx.x; e < 3 * C; e += NT) sm[S::BQ + e] = bqkv[e];
    __syncthreads();

    // Parameter gradients.  A wave does NOT accumulate its own windows' products (dWqkv alone would be 192 registers): after every
    // window the four waves meet, and wave w adds ALL FOUR windows' contributions to the row fragments it owns -- dWqkv rows
    // 16 (w + 4k) .., dWp rows 16 (w + 4k) .. -- reading the other waves' dQKV and O tiles from their LDS images and x / dy from
    // L1.  Same MFMA count, 64 accumulator registers instead of 256, and the workgroup's slab needs no cross-wave sum.
    f32x4 gW[JOWN][NF], gWp[FOWN][NF];  // dWqkv L(j|ci) rows of fragment w + 4k; dWp L(co|c) rows of fragment w + 4k
    float gb[JOWN], gbp[FOWN];          // per lane (i = j / co), still to be summed over g
#pragma unroll
    for (int k = 0; k < JOWN; ++k) {
        gb[k] = 0.f;
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) gW[k][cf] = splat(0.f);
    }
#pragma unroll
    for (int k = 0; k < FOWN; ++k) {
        gbp[k] = 0.f;
#pragma unroll
        for (int n = 0; n < NF; ++n) gWp[k][n] = splat(0.f);
    }
    float ns1[NORM ? NF : 1], ns2[NORM ? NF : 1];  // lane (i, g): channel 16f + i, pixels of window row g (the dX tile is L(p|ci) here)
#pragma unroll
    for (int f = 0; f < (NORM ? NF : 1); ++f) ns1[f] = ns2[f] = 0.f;

    WinWalk cur;
    cur.start(w0 < nwin ? w0 : 0, nwx, nwy);
    for (int it = 0, run = r0, j = 0; it < iters; ++it) {
        const bool active = w0 + it < w1;  // wave-uniform
        if (active) {
            const int n = cur.n, wy = cur.wy, wx = cur.wx;
            if (lane == 0) {
                slot[4 * wave] = n;
                slot[4 * wave + 1] = wy;
                slot[4 * wave + 2] = wx;
            }
            const bool run_ends = j + 1 == kblk;
            f32x4 dyt[NF];
            float warm = 0.f;  // destination of the cache-warming load below: stays allocated until the window's last statement
            // ---- recompute the forward -----------------------------------------------------------------------------------
            Chain<C> ch;
            {
                f32x4 xn[NF];
                fetch_cp<C>(xn, x, H, W, n, wy, wx, i, g);
                fetch_cp<C>(dyt, dy, H, W, n, wy, wx, i, g);
                cur.next();
                if (w0 + it + 1 < w1) {  // the next window's 2 x 32 cache lines on their way: lanes 0-31 x, lanes 32-63 dy, one line each
                    const int l5 = lane & 31;
                    const float* src = (lane < 32 ? x : dy) + (((size_t)cur.n * H + 4 * cur.wy + (l5 >> 3)) * W + 4 * cur.wx + ((l5 >> 1) & 3)) * C + 32 * (l5 & 1);
                    asm volatile("global_load_dword %0, %1, off" : "+v"(warm) : "v"(src) : "memory");
                }
                if (NORM) {
                    NormQ<C> nq;
                    nq.load(in_stats, n, g);
#pragma unroll
                    for (int h = 0; h < NF; ++h) xn[h] = nq.apply(xn[h], h);
                }
                forward_chain_lds<C>(ch, xn, sm + S::WQ, S::LD_W, sm + S::BQ, i, g);
            }
            put_pc<NF>(my + S::Q, S::LD_RC, ch.q, i, g);
            put_pc<NF>(my + S::K, S::LD_RC, ch.k, i, g);
            put_cp<NF>(my + S::V, S::LD_CR, ch.vt, i, g);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 o_pc[NF];  // O[p][c1] = sum_c2 v^T[c2][p] P^T[c2][c1]     L(p|c1)
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1) o_pc[n1] = splat(0.f);
#pragma unroll
            for (int m = 0; m < NF; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n1 = 0; n1 < NF; ++n1) o_pc[n1] = mfma16(ch.vt[m][r], ch.pt[m][n1][r], o_pc[n1]);
            // ---- proj backward: dO = dY Wp (L(p|c1)) -------------------------------------------------------------------------
            f32x4 dO[NF];
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1) dO[n1] = splat(0.f);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                f32x4 wv[NF];  // Wp[16f + 4g + e][16n1 + i]   L(co|c)
#pragma unroll
                for (int n1 = 0; n1 < NF; ++n1) wv[n1] = *reinterpret_cast<const f32x4*>(&sm[S::WPT + (16 * n1 + i) * S::LD_W + 16 * f + 4 * g]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n1 = 0; n1 < NF; ++n1) dO[n1] = mfma16(dyt[f][e], wv[n1][e], dO[n1]);
            }
            put_pc<NF>(my + S::DO, S::LD_RC, dO, i, g);
            wave_lds_fence();
            f32x4 v_pc[NF];  // v in L(p|c2): read before the transpose image (the same bytes) is first written
            get_pc<NF>(v_pc, my + S::V, S::LD_CR, i, g);
            // ---- dV^T[c2][p] = sum_c1 P[c1][c2] dO^T[c1][p], one c1 fragment row of P at a time ------------------------------------
            f32x4 dqkv[3 * NF];  // dq | dk | dv in L(j|p)
            {
                f32x4 dOt[NF];
                get_cp<NF>(dOt, my + S::DO, S::LD_RC, i, g);  // L(c1|p)
#pragma unroll
                for (int m = 0; m < NF; ++m) dqkv[2 * NF + m] = splat(0.f);
#pragma unroll
                for (int a = 0; a < NF; ++a) {
                    wave_lds_fence();
#pragma unroll
                    for (int m = 0; m < NF; ++m) *reinterpret_cast<f32x4*>(&my[S::SQ + i * S::LD_CR + 16 * m + 4 * g]) = ch.pt[m][a];
                    wave_lds_fence();
                    f32x4 prow[NF];  // P[16a + 4g + r][16m + i]
#pragma unroll
                    for (int m = 0; m < NF; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) prow[m][r] = my[S::SQ + (4 * g + r) * S::LD_CR + 16 * m + i];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int m = 0; m < NF; ++m) dqkv[2 * NF + m] = mfma16(prow[m][r], dOt[a][r], dqkv[2 * NF + m]);
                }
            }
            // O -> the dO image (its L(c1|p) read is done): the weight-gradient stage reads it as L(p|c1), maybe from another wave
            wave_lds_fence();
            put_pc<NF>(my + S::DO, S::LD_RC, o_pc, i, g);
            // ---- dP^T[c2][c1] = sum_p v[p][c2] dO[p][c1];  dS^T = P^T (dP^T - sum_c2 dP^T P^T) ------------------------------------
            f32x4 dSt[NF][NF];  // one c1 fragment column at a time: P^T's column dies as dS^T's is born
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1) {
#pragma unroll
                for (int m = 0; m < NF; ++m) dSt[m][n1] = splat(0.f);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < NF; ++m) dSt[m][n1] = mfma16(v_pc[m][r], dO[n1][r], dSt[m][n1]);
                float d = 0.f;
#pragma unroll
                for (int m = 0; m < NF; ++m) d += hsum(dSt[m][n1] * ch.pt[m][n1]);
                d = xg_sum(d);
#pragma unroll
                for (int m = 0; m < NF; ++m) dSt[m][n1] = ch.pt[m][n1] * (dSt[m][n1] - splat(d));
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- q, k in L(c|p); norms again in that orientation ------------------------------------------------------------------
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
            f32x4 dqh[NF], dkh[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) dqh[f] = dkh[f] = splat(0.f);
            // dq^[c1][p] = sum_c2 dS^T[c2][c1] k^[c2][p]
#pragma unroll
            for (int m = 0; m < NF; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n1 = 0; n1 < NF; ++n1) dqh[n1] = mfma16(dSt[m][n1][r], kh_cp[m][r], dqh[n1]);
            // dk^[c2][p] = sum_c1 dS[c1][c2] q^[c1][p], one c1 fragment row of dS at a time
#pragma unroll
            for (int a = 0; a < NF; ++a) {
                wave_lds_fence();
#pragma unroll
                for (int m = 0; m < NF; ++m) *reinterpret_cast<f32x4*>(&my[S::SQ + i * S::LD_CR + 16 * m + 4 * g]) = dSt[m][a];
                wave_lds_fence();
                f32x4 srow[NF];  // dS[16a + 4g + r][16m + i]
#pragma unroll
                for (int m = 0; m < NF; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) srow[m][r] = my[S::SQ + (4 * g + r) * S::LD_CR + 16 * m + i];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < NF; ++m) dkh[m] = mfma16(srow[m][r], qh_cp[a][r], dkh[m]);
            }
            {   // backward of F.normalize, per pixel (= per lane i)
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
            wave_lds_fence();
            put_cp<3 * NF>(my + S::DQKV, S::LD_J, dqkv, i, g);  // dQKV -> L(p|j) later; over q / k / v, all consumed
            // ---- qkv conv backward: dX[p][ci] = sum_j dQKV^T[j][p] Wqkv[j][ci], as L(p|ci): the layout the norm's backward sums want
            //      (one register per channel fragment, no cross-lane work until the run ends) ------------------------------------
            f32x4 x_pc[NF];
            if constexpr (NORM) fetch_pc<C>(x_pc, x, H, W, n, wy, wx, i, g);
            f32x4 dXp[NF];
#pragma unroll
            for (int cf = 0; cf < NF; ++cf) dXp[cf] = splat(0.f);
#pragma unroll
            for (int jf = 0; jf < 3 * NF; ++jf) {
                f32x4 w2[NF];  // w2[cf][r] = Wqkv[16jf + 4g + r][16cf + i]   L(j|ci): 4-byte loads of the [j][ci] image
#pragma unroll
                for (int cf = 0; cf < NF; ++cf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w2[cf][r] = sm[S::WQ + (16 * jf + 4 * g + r) * S::LD_W + 16 * cf + i];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cf = 0; cf < NF; ++cf) dXp[cf] = mfma16(dqkv[jf][r], w2[cf][r], dXp[cf]);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                float* dst = dx + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(g * W * C + i);
#pragma unroll
                for (int cf = 0; cf < NF; ++cf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * C + 16 * cf] = dXp[cf][r];
            }
            if (NORM) {  // z = relu(x^) in L(p|ci); where z > 0 it IS x^, elsewhere the element contributes nothing
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const f32x2 st2 = *reinterpret_cast<const f32x2*>(in_stats + ((size_t)n * C + 16 * f + i) * 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = fmaxf((x_pc[f][r] - st2[0]) * st2[1], 0.f);
                        const float gg = z > 0.f ? dXp[f][r] : 0.f;
                        ns1[f] += gg;
                        ns2[f] += gg * z;
                    }
                }
                if (run_ends) {
                    float* row = nsum + (size_t)run * 2 * C;
#pragma unroll
                    for (int cf = 0; cf < NF; ++cf) {
                        const float a = xg_sum(ns1[cf]), b = xg_sum(ns2[cf]);
                        if (g == 0) {
                            row[16 * cf + i] = a;
                            row[C + 16 * cf + i] = b;
                        }
                        ns1[cf] = ns2[cf] = 0.f;
                    }
                }
            }
            if (run_ends) {
                ++run;
                j = 0;
            } else {
                ++j;
            }
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(warm) : "memory");  // the warming load has landed; its register is free again
        } else if (lane == 0) {
            slot[4 * wave] = -1;
        }
        __syncthreads();  // every wave's dQKV and O images and window coordinates are in LDS
        // ---- dWqkv[j][ci] += sum_p dQKV[p][j] X[p][ci]; dbqkv += colsum dQKV; dWp[co][c] += sum_p dY[p][co] O[p][c]; dbp += colsum dY ----
        // all four windows' x and dY fragments (and, NORM, their statistics) are requested before the first product: one memory
        // latency per window round, not four
        int vn[WAVES];
        f32x4 x4[WAVES][NF], dy4[WAVES][FOWN];
        f32x2 st4[WAVES][NORM ? NF : 1];
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            vn[ww] = uniform(slot[4 * ww]);
            const int n = vn[ww] < 0 ? 0 : vn[ww], wy = vn[ww] < 0 ? 0 : uniform(slot[4 * ww + 1]), wx = vn[ww] < 0 ? 0 : uniform(slot[4 * ww + 2]);
            fetch_pc<C>(x4[ww], x, H, W, n, wy, wx, i, g);
#pragma unroll
            for (int k = 0; k < FOWN; ++k) {  // dY[4g + r][16 (wave + 4k) + i]
                const int f = wave + WAVES * k;
                const float* p = dy + (((size_t)n * H + 4 * wy) * W + 4 * wx) * C + (unsigned)(g * W * C + 16 * (f < NF ? f : 0) + i);
#pragma unroll
                for (int r = 0; r < 4; ++r) dy4[ww][k][r] = p[r * C];
            }
            if (NORM) {
#pragma unroll
                for (int f = 0; f < NF; ++f) st4[ww][f] = *reinterpret_cast<const f32x2*>(in_stats + ((size_t)n * C + 16 * f + i) * 2);
            }
        }
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            if (vn[ww] < 0) continue;  // that wave had no window this round: its images are stale
            const float* img = sm + ww * S::END;
            if (NORM) {
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) x4[ww][f][r] = fmaxf((x4[ww][f][r] - st4[ww][f][0]) * st4[ww][f][1], 0.f);
            }
#pragma unroll
            for (int k = 0; k < JOWN; ++k) {
                const int jf = wave + WAVES * k;
                if (jf < 3 * NF) {
                    f32x4 dq_pj;  // dQKV[4g + r][16jf + i]
#pragma unroll
                    for (int r = 0; r < 4; ++r) dq_pj[r] = img[S::DQKV + (4 * g + r) * S::LD_J + 16 * jf + i];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cf = 0; cf < NF; ++cf) gW[k][cf] = mfma16(dq_pj[r], x4[ww][cf][r], gW[k][cf]);
                    gb[k] += hsum(dq_pj);
                }
            }
#pragma unroll
            for (int k = 0; k < FOWN; ++k) {
                if (wave + WAVES * k < NF) {
#pragma unroll
                    for (int n1 = 0; n1 < NF; ++n1) {
                        f32x4 o;  // O[4g + r][16n1 + i]
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = img[S::DO + (4 * g + r) * S::LD_RC + 16 * n1 + i];
#pragma unroll
                        for (int r = 0; r < 4; ++r) gWp[k][n1] = mfma16(dy4[ww][k][r], o[r], gWp[k][n1]);
                    }
                    gbp[k] += hsum(dy4[ww][k]);
                }
            }
        }
        __syncthreads();  // the images are free for the next window
    }
    // ---- this workgroup's slab: dWqkv (3C x C) | dWp (C x C) | dbqkv | dbp; every row fragment has one owner, nothing to add up ----
    float* out = partial + (size_t)blockIdx.x * S::SLAB;
#pragma unroll
    for (int k = 0; k < JOWN; ++k) {
        const int jf = wave + WAVES * k;
        if (jf < 3 * NF) {
#pragma unroll
            for (int cf = 0; cf < NF; ++cf)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * jf + 4 * g + r) * C + 16 * cf + i] = gW[k][cf][r];
            const float v = xg_sum(gb[k]);
            if (g == 0) out[4 * C * C + 16 * jf + i] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < FOWN; ++k) {
        const int f = wave + WAVES * k;
        if (f < NF) {
#pragma unroll
            for (int n1 = 0; n1 < NF; ++n1)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[3 * C * C + (16 * f + 4 * g + r) * C + 16 * n1 + i] = gWp[k][n1][r];
            const float v = xg_sum(gbp[k]);
            if (g == 0) out[4 * C * C + 3 * C + 16 * f + i] = v;
        }
    }
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

// C = 32 through the C = 64 code path (test switch: the two implementations of one chain check each other)
static bool big32() {
    const char* e = env_get(ENV_ATTN_BIG32);
    return e && e[0] == '1';
}
static bool use_big(int C) { return C == 64 || (C == 32 && big32()); }

// waves per workgroup / workgroups per CU of the backward
constexpr int bwd_waves(int C) { return C == 16 ? 4 : 8; }
constexpr int bwd_wgs_per_cu(int C) { return C == 16 ? 4 : 1; }
// most workgroups the backward uses (= slabs in its workspace)
int attn_reg_bwd_blocks(int C) { return use_big(C) ? reg_cus() : reg_cus() * bwd_wgs_per_cu(C); }
int attn_reg_bwd_waves(int C) { return use_big(C) ? 4 : bwd_waves(C); }

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

template <typename K>
static int set_lds(K kern, size_t lds, const char* what) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e == hipSuccess ? MSTG_OK : fail_launch(e, what);
}

template <int C>
static int big_fwd(const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* bp, float* y,
                   int N, int H, int W, hipStream_t st) {
    typedef BigLds<C> S;
    const size_t lds = (size_t)S::FWD_FLOATS * sizeof(float);
    const int nwin = N * (H / 4) * (W / 4);
    int nb = reg_cus() * 2;
    if (nb * 4 > nwin) nb = cdiv(nwin, 4);
    static bool ready = false;
    if (!ready) {
        if (int rc = set_lds(attn_big_fwd_kernel<C, true>, lds, "hipFuncSetAttribute(attn_big_fwd<norm>)")) return rc;
        if (int rc = set_lds(attn_big_fwd_kernel<C, false>, lds, "hipFuncSetAttribute(attn_big_fwd)")) return rc;
        ready = true;
    }
    if (in_stats)
        MSTG_LAUNCH((attn_big_fwd_kernel<C, true>), dim3(nb), dim3(256), lds, st, x, wqkv, bqkv, wp, bp, y, N, H, W, in_stats);
    else
        MSTG_LAUNCH((attn_big_fwd_kernel<C, false>), dim3(nb), dim3(256), lds, st, x, wqkv, bqkv, wp, bp, y, N, H, W, in_stats);
    MSTG_CHECK_LAUNCH("attn_big_fwd_kernel");
    return MSTG_OK;
}

template <int C>
static int big_bwd(const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* dy, float* dx,
                   float* partial, float* nsum, int kblk, int nb, int N, int H, int W, hipStream_t st) {
    typedef BigLds<C> S;
    const size_t lds = (size_t)S::BWD_FLOATS * sizeof(float);
    static bool ready = false;
    if (!ready) {
        if (int rc = set_lds(attn_big_bwd_kernel<C, true>, lds, "hipFuncSetAttribute(attn_big_bwd<norm>)")) return rc;
        if (int rc = set_lds(attn_big_bwd_kernel<C, false>, lds, "hipFuncSetAttribute(attn_big_bwd)")) return rc;
        ready = true;
    }
    if (in_stats)
        MSTG_LAUNCH((attn_big_bwd_kernel<C, true>), dim3(nb), dim3(256), lds, st, x, wqkv, bqkv, wp, dy, dx, partial, N, H, W, in_stats, nsum, kblk);
    else
        MSTG_LAUNCH((attn_big_bwd_kernel<C, false>), dim3(nb), dim3(256), lds, st, x, wqkv, bqkv, wp, dy, dx, partial, N, H, W, in_stats, nsum, 1);
    MSTG_CHECK_LAUNCH("attn_big_bwd_kernel");
    return MSTG_OK;
}

int attn_reg_fwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* bp,
                 float* y, int N, int H, int W, hipStream_t st) {
    if (C == 64) return big_fwd<64>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st);
    if (C == 32 && big32()) return big_fwd<32>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st);
    return C == 16 ? reg_fwd<16>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st) : reg_fwd<32>(x, in_stats, wqkv, bqkv, wp, bp, y, N, H, W, st);
}
int attn_reg_bwd(int C, const float* x, const float* in_stats, const float* wqkv, const float* bqkv, const float* wp, const float* dy,
                 float* dx, float* partial, float* nsum, int kblk, int nblocks, int N, int H, int W, hipStream_t st) {
    if (C == 64) return big_bwd<64>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st);
    if (C == 32 && big32()) return big_bwd<32>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st);
    return C == 16 ? reg_bwd<16>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st)
                   : reg_bwd<32>(x, in_stats, wqkv, bqkv, wp, dy, dx, partial, nsum, kblk, nblocks, N, H, W, st);
}

}  // namespace mstg
