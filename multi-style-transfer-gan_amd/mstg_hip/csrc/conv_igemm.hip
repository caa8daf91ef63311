// Implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_16x16x4_f32) of gfx950.
//
// One kernel serves Conv2d forward, ConvTranspose2d(k4,s2,p1) forward and every input-gradient on the path:
//   gather mode : y[n,oy,ox,co] = sum_{ky,kx,cr} src[n, oy*s - p + ky*d, ox*s - p + kx*d, cr] * W(co,cr,ky,kx)
//   phase mode  : the stride-2 transposed convolution split into its 4 output-parity classes, each a 2x2
//                 stride-1 gather over the source (no zero insertion, no scatter, no atomics).
// GEMM view per workgroup: M = 16*NFW output channels (the A operand = filter rows), N = 8x16 output pixels
// (the B operand = LDS-staged source patch with halo), K = taps x 4V source channels.  With the filter on the
// MFMA's M side the accumulator of a lane is 4 consecutive output channels of one pixel, so the NHWC store is
// one 16-byte write per lane and fragment.
//
// Reference sites replaced: every nn.Conv2d / nn.ConvTranspose2d forward and the dgrad half of their autograd
// (enhanced_generator.py:10-11,53-73,92,99,106,121,128,137,237-265; pretrain.py:65-91).
#include "common.h"
#include "igemm_args.h"
#include <stdlib.h>

namespace mstg {


constexpr int TILE_H = 8, TILE_W = 16;

// Phase time stamps of sampled workgroups (tools/diag_stamps.py); compiled in only with -DMSTG_STAMPS.
#ifdef MSTG_STAMPS
__device__ unsigned long long g_dbg_stamps[64 * 8];
#define MSTG_STAMP(k)                                                                                      \
    if (threadIdx.x == 0 && (blockIdx.x % 11) == 0 && blockIdx.x / 11 < 64 && blockIdx.y == 0 && blockIdx.z == 0) \
        g_dbg_stamps[(blockIdx.x / 11) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define MSTG_STAMP(k)
#endif
constexpr int W_BUDGET_FLOATS = 6144;  // 24 KiB of LDS for one tap group's filter slice

template <int V> struct Frag;
template <> struct Frag<1> { typedef float T; };
template <> struct Frag<2> { typedef f32x2 T; };
template <> struct Frag<4> { typedef f32x4 T; };
template <int V> __device__ __forceinline__ float frag_get(const typename Frag<V>::T& f, int j);
template <> __device__ __forceinline__ float frag_get<1>(const float& f, int) { return f; }
template <> __device__ __forceinline__ float frag_get<2>(const f32x2& f, int j) { return f[j]; }
template <> __device__ __forceinline__ float frag_get<4>(const f32x4& f, int j) { return f[j]; }

// LDS stride (floats) of one pixel's / one filter row's 4V-channel chunk.  16-channel chunks: 20 suits the light kernels, whose
// b128 patch reads walk 16 neighbouring pixels; the pipelined (heavy) kernel also reads its filter slice at this stride, and 24
// makes those reads conflict-free (measured on MI355X, discriminator layers: 3.21 -> 2.52 ms per train step).
template <int V> __host__ __device__ constexpr int ckp_of() { return V == 4 ? 20 : (V == 2 ? 12 : 4); }
template <int V> __host__ __device__ constexpr int ckp_heavy_of() { return V == 4 ? 24 : ckp_of<V>(); }

// ---- filter pre-pack ------------------------------------------------------------------------------------------------
// wp[cls][chunk][tap][co (padded to CoP)][ci (CK)] : exactly the order the main kernels stage into LDS, so staging is a
// run of 16-byte copies with no index arithmetic on the filter's own (OIHW / IOHW, flipped, parity-class) layout.
__global__ void pack_filter_kernel(const IGemmArgs a, float* __restrict__ wp, int CK, int CoP, int nchunks, int ncls) {
    const int total = ncls * nchunks * a.ntaps * CoP * CK;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int ci = idx % CK;
        int rest = idx / CK;
        const int co = rest % CoP; rest /= CoP;
        const int t = rest % a.ntaps; rest /= a.ntaps;
        const int chunk = rest % nchunks, cls = rest / nchunks;
        int widx, oc = co;
        bool ok = true;
        if (a.phase) {
            const int pa = cls >> 1, pb = cls & 1, u = t >> 1, v = t & 1;
            widx = ((1 - pa) + 2 * u) * 4 + ((1 - pb) + 2 * v);
        } else if (a.dpack) {  // row = 4*delta + channel ; packed tap t = (ky, j) holds the real tap kx = 4j + 3 - delta
            const int delta = co >> 2, kx = 4 * (t % a.tapsx) + 3 - delta, real = (t / a.tapsx) * a.KW + kx;
            oc = co & 3;
            ok = co < 16 && kx >= 0 && kx < a.KW;
            widx = a.flip ? (a.KH * a.KW - 1 - real) : real;
        } else {
            widx = a.flip ? (a.ntaps - 1 - t) : t;
        }
        const int cr = chunk * CK + ci;
        wp[idx] = (ok && oc < a.Co && cr < a.Cr) ? a.w[(size_t)oc * a.w_so + (size_t)cr * a.w_sr + widx] : 0.f;
    }
}

// tap t -> LDS offset of its first pixel inside the patch
__device__ __forceinline__ int tap_patch_offset(const IGemmArgs& a, int t, int pa, int pb, int ckp) {
    int pro, pco;
    if (a.phase) {
        pro = 1 + pa - (t >> 1);
        pco = 1 + pb - (t & 1);
    } else if (a.dpack) {
        pro = t / a.tapsx;
        pco = 4 * (t % a.tapsx) + 3;
    } else {
        pro = (t / a.KW) * a.dil;
        pco = (t % a.KW) * a.dil;
    }
    return (pro * a.PW + pco) * ckp;
}

// bias, optional accumulate, activation, store of one workgroup tile
template <int NFW, int PF = 2>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& a, const f32x4 (&acc)[NFW][PF], int n, int ty0, int tx0, int co0, int pa,
                                               int pb, int wave, int i, int g) {
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int gy = ty0 * (4 * PF) + PF * wave + pf, gx = tx0 * TILE_W + i;
        if (gy >= a.Gh || gx >= a.Gw) continue;
        const int oy = a.phase ? 2 * gy + pa : gy, ox = a.phase ? 2 * gx + pb : gx;
#pragma unroll
        for (int wf = 0; wf < NFW; ++wf) {
            const int co = co0 + 16 * wf + 4 * g;
            if (co >= a.Co) continue;
            f32x4 v = acc[wf][pf];
            if (!a.y_nchw && co + 3 < a.Co && ((a.y_ctot | a.y_coff) & 3) == 0) {
                float* p = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + co;
                if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + co);
                if (a.accumulate) v += *reinterpret_cast<const f32x4*>(p);
                if (a.act != MSTG_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                }
                *reinterpret_cast<f32x4*>(p) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e >= a.Co) continue;
                    float* p = a.y_nchw ? a.y + (((size_t)n * a.y_ctot + a.y_coff + co + e) * a.Ho + oy) * a.Wo + ox
                                        : a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + co + e;
                    float val = v[e] + (a.bias ? a.bias[co + e] : 0.f);
                    if (a.accumulate) val += *p;
                    *p = apply_act(val, a.act);
                }
            }
        }
    }
}

// =====================================================================================================================
// "light" kernel: one tile per workgroup, many workgroups per CU (small LDS, few registers) -- for layers whose MFMA
// work per staged byte is small (1x1 convolutions, tiny filters): they are bound by memory latency/bandwidth and want
// occupancy, not a deep per-workgroup pipeline.
// V   : source channels per MFMA k-slot (a lane reads V consecutive channels; K chunk = 4V channels)
// NFW : 16-channel output fragments per workgroup (BN = 16*NFW)
// PF  : tile rows per wave (tile = 4*PF x 16 pixels).  Four rows halve, per pixel, everything a workgroup does once (index
//       set-up, filter staging, halo) and reuse each filter fragment over four pixel fragments; two rows keep the patch small.
// =====================================================================================================================
template <int V, int NFW, int PF>
__global__ __launch_bounds__(256) void igemm_light_kernel(const IGemmArgs a, const float* __restrict__ wp, const int CoP) {
    constexpr int CK = 4 * V, CKP = ckp_of<V>(), BN = 16 * NFW, TH = 4 * PF;
    typedef typename Frag<V>::T frag_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    float* wl = smem + ((a.PH * a.PW * CKP + 3) & ~3);
    int* tapo = reinterpret_cast<int*>(wl + (a.wglob ? 0 : a.TG * BN * CKP));  // LDS offset of every tap inside the patch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int co0 = blockIdx.y * BN;
    const int cls = blockIdx.z, pa = cls >> 1, pb = cls & 1;
    const int s = a.phase ? 1 : a.stride;
    const int y0 = a.phase ? ty0 * TH - 1 : ty0 * TH * s - a.pad;
    // dpack: tiles advance by 13 output columns; the 16 accumulator columns start 3 pixels to their left
    const int x0 = a.phase ? tx0 * TILE_W - 1 : (a.dpack ? tx0 * 13 - 3 - a.pad : tx0 * TILE_W * s - a.pad);
    const int nchunks = (a.Cr + CK - 1) / CK;
    const unsigned m_pw = magic_u32(a.PW);
    MSTG_STAMP(0)
    if (tid < a.ntaps) tapo[tid] = tap_patch_offset(a, tid, pa, pb, CKP);  // visible after the first barrier below

    f32x4 acc[NFW][PF];
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk) __syncthreads();
        // ---- stage the source patch (halo included, zero outside the image) -------------------------------------
        if (a.x_nchw) {  // 3-channel image tensor, V == 1: channel 3 of the k-slot group is zero
            stage_window_c4(a.x + ((size_t)n * a.x_ctot + a.x_coff) * a.H * a.W, patch, a.PH, a.PW, m_pw, y0, x0, a.H, a.W, (unsigned)a.W, 1u,
                            (unsigned)(a.H * a.W), a.Cr, tid);
        } else if (((a.x_ctot | a.x_coff | a.Cr) & 3) == 0) {
            stage_window(a.x + (size_t)n * a.H * a.W * a.x_ctot + a.x_coff + chunk * CK, patch, a.PH, a.PW, V, m_pw, 0xFFFFFFFFu / V + 1u, y0,
                         x0, a.H, a.W, a.x_ctot, min(V, (a.Cr - chunk * CK) >> 2), CKP, tid);
        } else {
            for (int pr = wave; pr < a.PH; pr += 4) {
                const int iy = y0 + pr;
                for (int e = lane; e < a.PW * V; e += 64) {
                    const int pc = e / V, q = e % V;
                    const int ix = x0 + pc;
                    const bool inb = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    const int c0 = chunk * CK + 4 * q;
                    if (inb && c0 < a.Cr) {
                        const float* src = a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.x_ctot + a.x_coff + c0;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c0 + c < a.Cr) v[c] = src[c];
                    }
                    *reinterpret_cast<f32x4*>(&patch[(pr * a.PW + pc) * CKP + 4 * q]) = v;
                }
            }
        }
        MSTG_STAMP(1)
        if (a.wglob) {
            // ---- filter fragments from L2: no filter slice in LDS, no second barrier; tap t+2's fragment is in flight --------------
            __syncthreads();
            const float* wb = wp + ((size_t)((cls * nchunks + chunk) * a.ntaps) * CoP + co0 + i) * CK + V * g;
            const int wts = CoP * CK;  // floats between consecutive taps
            const int bbase0 = ((PF * wave) * s * a.PW + i * s) * CKP + V * g, bstep = s * a.PW * CKP;
            frag_t a0[NFW], a1[NFW], a2[NFW], bf[PF], bfn[PF];
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) {
                a0[wf] = *reinterpret_cast<const frag_t*>(wb + 16 * wf * CK);
                a1[wf] = *reinterpret_cast<const frag_t*>(wb + min(1, a.ntaps - 1) * wts + 16 * wf * CK);
            }
            {
                const int po = tapo[0];
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
            }
            for (int tl = 0; tl < a.ntaps; ++tl) {
                const int t2 = min(tl + 2, a.ntaps - 1), t1 = min(tl + 1, a.ntaps - 1);
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) a2[wf] = *reinterpret_cast<const frag_t*>(wb + t2 * wts + 16 * wf * CK);
                const int po = tapo[t1];
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bfn[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf)
                            acc[wf][pf] = mfma16(frag_get<V>(a0[wf], j), frag_get<V>(bf[pf], j), acc[wf][pf]);
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) { a0[wf] = a1[wf]; a1[wf] = a2[wf]; }
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = bfn[pf];
            }
            continue;
        }
        for (int t0 = 0; t0 < a.ntaps; t0 += a.TG) {
            __syncthreads();  // patch staged / previous tap group consumed
            MSTG_STAMP(2)
            const int tn = min(a.TG, a.ntaps - t0);
            // ---- stage this tap group's filter slice: 16-byte copies out of the packed filter ---------------------
            const float* base = wp + ((size_t)((cls * nchunks + chunk) * a.ntaps + t0) * CoP + co0) * CK;
            for (int e0 = 0; e0 < tn * BN * V; e0 += 1024) {
                f32x4 wv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = min(e0 + 256 * k + tid, tn * BN * V - 1);
                    const int row = e / V, q = e % V, tl = row / BN, col = row % BN;
                    wv[k] = *reinterpret_cast<const f32x4*>(base + (unsigned)((tl * CoP + col) * CK + 4 * q));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = e0 + 256 * k + tid;
                    if (e < tn * BN * V) *reinterpret_cast<f32x4*>(&wl[(e / V) * CKP + 4 * (e % V)]) = wv[k];
                }
            }
            MSTG_STAMP(3)
            __syncthreads();
            MSTG_STAMP(4)
            // ---- MFMA over the group's taps -------------------------------------------------------------------------
            // fragments of tap tl+1 are read while the MFMAs of tap tl run
            const int bbase0 = ((PF * wave) * s * a.PW + i * s) * CKP + V * g, bstep = s * a.PW * CKP;
            frag_t af[NFW], bf[PF], afn[NFW], bfn[PF];
            {
                const int po = tapo[t0];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) af[wf] = *reinterpret_cast<const frag_t*>(&wl[(16 * wf + i) * CKP + V * g]);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
            }
            for (int tl = 0; tl < ((a.dbg & 1) ? 1 : tn); ++tl) {
                const int tnx = min(tl + 1, tn - 1);
                const int po = tapo[t0 + tnx];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) afn[wf] = *reinterpret_cast<const frag_t*>(&wl[(tnx * BN + 16 * wf + i) * CKP + V * g]);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bfn[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf)
                            acc[wf][pf] = mfma16(frag_get<V>(af[wf], j), frag_get<V>(bf[pf], j), acc[wf][pf]);
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) af[wf] = afn[wf];
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = bfn[pf];
            }
        }
    }
    MSTG_STAMP(5)
    if (!a.dpack) {
        igemm_epilogue<NFW, PF>(a, acc, n, ty0, tx0, co0, pa, pb, wave, i, g);
        MSTG_STAMP(6)
        return;
    }
    // ---- dpack epilogue.  Accumulator row 4*delta + c of pixel column p is a partial sum of output column p + delta:
    //      y[q][c] = sum_delta D[delta][q - delta].  Exchange through LDS, then lanes (q < 13, pf) finish 13 columns. ---------
    __syncthreads();  // everybody is done with the filter / patch tiles: reuse the front of LDS
    float* comb = smem + wave * (256 * PF);  // [pf][delta][p][4]
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) *reinterpret_cast<f32x4*>(&comb[((pf * 4 + g) * 16 + i) * 4]) = acc[0][pf];
    __syncthreads();
    if (g < PF && i < 13) {
        const int pf = g;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d) v += *reinterpret_cast<const f32x4*>(&comb[((pf * 4 + d) * 16 + i + 3 - d) * 4]);
        const int oy = ty0 * TH + PF * wave + pf, ox = tx0 * 13 + i;
        if (oy < a.Gh && ox < a.Gw) {
            if (!a.y_nchw && a.Co == 4 && ((a.y_ctot | a.y_coff) & 3) == 0) {  // a whole 4-channel slice: one 16-byte store
                float* p = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff;
                if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias);
                if (a.accumulate) v += *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                *reinterpret_cast<f32x4*>(p) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e >= a.Co) continue;
                    float* p = a.y_nchw ? a.y + (((size_t)n * a.y_ctot + a.y_coff + e) * a.Ho + oy) * a.Wo + ox
                                        : a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + e;
                    float val = v[e] + (a.bias ? a.bias[e] : 0.f);
                    if (a.accumulate) val += *p;
                    *p = apply_act(val, a.act);
                }
            }
        }
    }
    MSTG_STAMP(6)
}

// =====================================================================================================================
// "stream" kernel: the layers with few channels are bound by memory latency, not by MFMA or bandwidth (measured with
// in-kernel time stamps: of a light workgroup's ~16k cycles, 7k wait for the patch, 1.5k for the filter, 2-3k for the
// stores, and only 2-5k do MFMAs).  Here a workgroup is persistent and walks tiles with a stride of gridDim.x:
//   * the WHOLE filter (all channel chunks and taps of this workgroup's output columns) is staged into LDS once;
//   * a stage = (tile, channel chunk).  The loads of stage k+1's patch are issued into registers before stage k's MFMAs
//     and written to the (single) LDS patch after them: the memory round trip runs under the MFMAs and the epilogue;
//   * one LDS patch (not two) keeps three or more workgroups per CU, i.e. three patches permanently in flight per CU.
// SRC : 0 = NHWC, aligned channel quads;  2 = NCHW tensor with <= 4 channels (V == 1).  NB patch slots per thread.
// =====================================================================================================================
constexpr int SNB = 8;

template <int V, int NFW, int PF, int SRC>
__global__ __launch_bounds__(256) void igemm_stream_kernel(const IGemmArgs a, const float* __restrict__ wp, const int CoP) {
    constexpr int CK = 4 * V, CKP = ckp_of<V>(), BN = 16 * NFW, TH = 4 * PF;
    typedef typename Frag<V>::T frag_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nchunks = (a.Cr + CK - 1) / CK;
    const int wsz = nchunks * a.ntaps * BN * CKP;
    float* patch = smem;
    float* wl = smem + a.psz;
    int* tapo = reinterpret_cast<int*>(wl + wsz);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int co0 = blockIdx.y * BN;
    const int cls = blockIdx.z, pa = cls >> 1, pb = cls & 1;
    const int s = a.phase ? 1 : a.stride;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_tiles * nchunks;
    if (total == 0) return;

    // ---- once: tap table, resident filter -----------------------------------------------------------------------------
    if (tid < a.ntaps) tapo[tid] = tap_patch_offset(a, tid, pa, pb, CKP);
    {
        // packed filter wp[cls][chunk][tap][CoP][CK]  ->  wl[((chunk * ntaps + tap) * BN + col) * CKP + ci]
        const int nrow = nchunks * a.ntaps * BN;  // rows of CK floats
        const float* base = wp + (size_t)cls * nchunks * a.ntaps * CoP * CK + (size_t)co0 * CK;
        for (int e0 = 0; e0 < nrow * V; e0 += 1024) {
            f32x4 wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = min(e0 + 256 * k + tid, nrow * V - 1);
                const int row = e / V, q = e % V, ct = row / BN, col = row % BN;
                wv[k] = *reinterpret_cast<const f32x4*>(base + (unsigned)((ct * CoP + col) * CK + 4 * q));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 256 * k + tid;
                if (e < nrow * V) *reinterpret_cast<f32x4*>(&wl[(e / V) * CKP + 4 * (e % V)]) = wv[k];
            }
        }
    }
    // ---- per-thread patch slots (tile independent): element e = tid + 256 j  ->  (row, col, quad) -------------------
    const int npatch = a.PH * a.PW * V;
    const unsigned m_pw = magic_u32(a.PW);
    int p_lds[SNB], p_rc[SNB];
    unsigned p_off[SNB], used = 0;  // p_off: offset of the slot's element from the patch's first pixel (0 for unused slots)
#pragma unroll
    for (int j = 0; j < SNB; ++j) {
        const int e = tid + 256 * j;
        const int pp = e / V, q = e % V;
        const int r = (int)__umulhi((unsigned)pp, m_pw), c = pp - r * a.PW;
        const bool u = e < npatch;
        p_lds[j] = u ? pp * CKP + 4 * q : -1;
        p_rc[j] = r | (c << 8) | (q << 16);
        p_off[j] = !u ? 0u : (SRC == 2 ? (unsigned)(r * a.W + c) : (unsigned)(r * a.W + c) * (unsigned)a.x_ctot + 4u * q);
        used |= u ? (1u << j) : 0u;
    }
    const unsigned plane = (unsigned)(a.H * a.W);
    // bias of this lane's output channels, once (the epilogue must not wait for a dependent global load per tile)
    f32x4 breg[NFW];
    unsigned cok = 0;  // bit wf: channels co0 + 16 wf + 4 g .. + 3 all exist
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf) {
        const int co = a.dpack ? 0 : co0 + 16 * wf + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) breg[wf][e] = (a.bias && co + e < a.Co) ? a.bias[co + e] : 0.f;
        cok |= (co + 3 < a.Co) ? (1u << wf) : 0u;
    }
    const bool yvec = !a.y_nchw && ((a.y_ctot | a.y_coff) & 3) == 0;
    const int yrs = a.phase ? 2 * a.Wo : a.Wo, ycs = a.phase ? 2 : 1;  // output row / column step of the walked grid
    const unsigned y_lane = (unsigned)((PF * wave) * yrs + i * ycs) * (unsigned)a.y_ctot + 4u * g;

    // stage cursor -> (tile, chunk)
    auto tile_of = [&](int seq, int& tx0, int& ty0, int& n) {
        const int tile = xcd_swizzle((int)blockIdx.x + seq * (int)gridDim.x, ntiles);
        tx0 = tile % a.tiles_x;
        ty0 = (tile / a.tiles_x) % a.tiles_y;
        n = tile / (a.tiles_x * a.tiles_y);
    };
    f32x4 preg[SNB];
    unsigned pmask = 0;  // bit j: slot j holds a real (in-image, existing channel) element
    auto load_stage = [&](int seq, int chunk) {
        int tx0, ty0, n;
        tile_of(seq, tx0, ty0, n);
        const int y0 = a.phase ? ty0 * TH - 1 : ty0 * TH * s - a.pad;
        const int x0 = a.phase ? tx0 * TILE_W - 1 : (a.dpack ? tx0 * 13 - 3 - a.pad : tx0 * TILE_W * s - a.pad);
        pmask = 0;
        const bool interior = y0 >= 0 && x0 >= 0 && y0 + a.PH <= a.H && x0 + a.PW <= a.W;  // wave-uniform
        if (interior && SRC == 2) {
            const float* base = a.x + ((size_t)n * a.x_ctot + a.x_coff) * plane + (unsigned)(y0 * a.W + x0);
            pmask = used;
#pragma unroll
            for (int j = 0; j < SNB; ++j) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) preg[j][ch] = base[p_off[j] + (ch < a.Cr ? ch * plane : 0u)];
            }
        } else if (interior && SRC == 0 && (a.Cr - chunk * CK) >= CK) {
            const float* base = a.x + ((size_t)n * plane + (unsigned)(y0 * a.W + x0)) * a.x_ctot + a.x_coff + chunk * CK;
            pmask = used;
#pragma unroll
            for (int j = 0; j < SNB; ++j) preg[j] = *reinterpret_cast<const f32x4*>(base + p_off[j]);
        } else if (SRC == 2) {
            const float* img = a.x + ((size_t)n * a.x_ctot + a.x_coff) * plane;
#pragma unroll
            for (int j = 0; j < SNB; ++j) {
                const int iy = y0 + (p_rc[j] & 255), ix = x0 + ((p_rc[j] >> 8) & 255);
                const bool ok = p_lds[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = ok ? (unsigned)(iy * a.W + ix) : 0u;
                pmask |= ok ? (1u << j) : 0u;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) preg[j][ch] = img[off + (ch < a.Cr ? ch * plane : 0u)];
            }
        } else {
            const float* img = a.x + (size_t)n * plane * a.x_ctot + a.x_coff + chunk * CK;
            const int nqv = min(V, (a.Cr - chunk * CK) >> 2);
#pragma unroll
            for (int j = 0; j < SNB; ++j) {
                const int iy = y0 + (p_rc[j] & 255), ix = x0 + ((p_rc[j] >> 8) & 255), q = p_rc[j] >> 16;
                const bool ok = p_lds[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && q < nqv;
                const unsigned off = ok ? (unsigned)(iy * a.W + ix) * (unsigned)a.x_ctot + 4u * q : 0u;
                pmask |= ok ? (1u << j) : 0u;
                preg[j] = *reinterpret_cast<const f32x4*>(img + off);
            }
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int j = 0; j < SNB; ++j) {
            if (p_lds[j] >= 0) {
                f32x4 v = preg[j];
                if (!((pmask >> j) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (SRC == 2) {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (ch >= a.Cr) v[ch] = 0.f;
                }
                *reinterpret_cast<f32x4*>(&patch[p_lds[j]]) = v;
            }
        }
    };

    load_stage(0, 0);
    store_stage();
    __syncthreads();

    f32x4 acc[NFW][PF];
    int seq = 0, chunk = 0;
    for (int k = 0; k < total; ++k) {
        const bool more = k + 1 < total;
        int nseq = seq, nchunk = chunk + 1;
        if (nchunk == nchunks) { nchunk = 0; ++nseq; }
        if (k == 2) { MSTG_STAMP(0) }
        if (more) load_stage(nseq, nchunk);  // in flight during the MFMAs and the epilogue below
        if (k == 2) { MSTG_STAMP(1) }
        if (chunk == 0) {
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- MFMA over every tap of this chunk; fragments of tap t+1 are read while tap t's MFMAs run -----------------
        {
            const float* wc = wl + chunk * a.ntaps * BN * CKP;
            const int bbase0 = ((PF * wave) * s * a.PW + i * s) * CKP + V * g, bstep = s * a.PW * CKP;
            const int arow = i * CKP + V * g;
            frag_t af[NFW], bf[PF], afn[NFW], bfn[PF];
            {
                const int po = tapo[0];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) af[wf] = *reinterpret_cast<const frag_t*>(&wc[16 * wf * CKP + arow]);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
            }
            for (int t = 0; t < a.ntaps; ++t) {
                const int tnx = min(t + 1, a.ntaps - 1);
                const int po = tapo[tnx];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) afn[wf] = *reinterpret_cast<const frag_t*>(&wc[(tnx * BN + 16 * wf) * CKP + arow]);
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bfn[pf] = *reinterpret_cast<const frag_t*>(&patch[bbase0 + pf * bstep + po]);
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                        for (int pf = 0; pf < PF; ++pf)
                            acc[wf][pf] = mfma16(frag_get<V>(af[wf], j), frag_get<V>(bf[pf], j), acc[wf][pf]);
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf) af[wf] = afn[wf];
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) bf[pf] = bfn[pf];
            }
        }
        if (k == 2) { MSTG_STAMP(2) }
        if (chunk == nchunks - 1) {
            int tx0, ty0, n;
            tile_of(seq, tx0, ty0, n);
            if (!a.dpack && yvec && ty0 * TH + TH <= a.Gh && tx0 * TILE_W + TILE_W <= a.Gw) {
                // whole tile inside the grid, 16-byte channel quads: no per-lane bounds work, bias from registers
                const int oy0 = a.phase ? 2 * ty0 * TH + pa : ty0 * TH, ox0 = a.phase ? 2 * tx0 * TILE_W + pb : tx0 * TILE_W;
                float* yb = a.y + (((size_t)n * a.Ho + oy0) * a.Wo + ox0) * a.y_ctot + a.y_coff + co0;
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) {
#pragma unroll
                    for (int wf = 0; wf < NFW; ++wf) {
                        if ((cok >> wf) & 1u) {
                            float* p = yb + y_lane + (unsigned)(pf * yrs * a.y_ctot + 16 * wf);
                            f32x4 v = acc[wf][pf] + breg[wf];
                            if (a.accumulate) v += *reinterpret_cast<const f32x4*>(p);
                            if (a.act != MSTG_ACT_NONE) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                            }
                            *reinterpret_cast<f32x4*>(p) = v;
                        } else {
                            const int co = co0 + 16 * wf + 4 * g;
                            const int oy = oy0 + (PF * wave + pf) * (a.phase ? 2 : 1), ox = ox0 + i * ycs;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (co + e >= a.Co) continue;
                                float* p = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + co + e;
                                float val = acc[wf][pf][e] + breg[wf][e];
                                if (a.accumulate) val += *p;
                                *p = apply_act(val, a.act);
                            }
                        }
                    }
                }
            } else if (!a.dpack) {
                igemm_epilogue<NFW, PF>(a, acc, n, ty0, tx0, co0, pa, pb, wave, i, g);
            } else {
                // dpack: y[q][c] = sum_delta D[delta][q - delta] through an LDS exchange in the (now idle) patch area
                __syncthreads();
                float* comb = smem + wave * (256 * PF);  // [pf][delta][p][4]
#pragma unroll
                for (int pf = 0; pf < PF; ++pf) *reinterpret_cast<f32x4*>(&comb[((pf * 4 + g) * 16 + i) * 4]) = acc[0][pf];
                __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own exchange tile is complete (wave-private)
                if (g < PF && i < 13) {
                    const int pf = g;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int d = 0; d < 4; ++d) v += *reinterpret_cast<const f32x4*>(&comb[((pf * 4 + d) * 16 + i + 3 - d) * 4]);
                    const int oy = ty0 * TH + PF * wave + pf, ox = tx0 * 13 + i;
                    if (oy < a.Gh && ox < a.Gw) {
                        v += breg[0];
                        if (!a.y_nchw && a.Co == 4 && ((a.y_ctot | a.y_coff) & 3) == 0) {
                            float* p = a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff;
                            if (a.accumulate) v += *reinterpret_cast<const f32x4*>(p);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                            *reinterpret_cast<f32x4*>(p) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (e >= a.Co) continue;
                                float* p = a.y_nchw ? a.y + (((size_t)n * a.y_ctot + a.y_coff + e) * a.Ho + oy) * a.Wo + ox
                                                    : a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + e;
                                float val = v[e];
                                if (a.accumulate) val += *p;
                                *p = apply_act(val, a.act);
                            }
                        }
                    }
                }
            }
        }
        if (k == 2) { MSTG_STAMP(3) }
        __syncthreads();  // every wave is done with the patch (and the exchange tiles)
        if (k == 2) { MSTG_STAMP(4) }
        if (more) store_stage();
        if (k == 2) { MSTG_STAMP(5) }
        __syncthreads();
        if (k == 2) { MSTG_STAMP(6) }
        seq = nseq;
        chunk = nchunk;
    }
}

// =====================================================================================================================
// "heavy" kernel: persistent workgroups with a software pipeline (issue-early / write-late).  A stage = (tile, channel
// chunk, tap group).  While the MFMAs of stage k run, the global loads of stage k+1 (filter slice, and the source patch
// when a new (tile, chunk) starts) are already in flight into registers; they are written to the OTHER half of the
// double-buffered LDS after the MFMA loop, followed by the stage's single barrier.  Every prefetch load is unconditional
// (clamped address, value zeroed at the LDS write): nothing sits between a load and its first use, so the compiler
// issues them back to back and waits only in front of the LDS writes.
// SRC : 0 = NHWC source, 16-byte aligned channel quads; 1 = NHWC arbitrary channel slice (scalar loads);
//       2 = NCHW 3-channel image tensor (V == 1)
// =====================================================================================================================
constexpr int NPQ = 10;  // patch float4 slots per thread  (2560 >= PH*PW*V for every geometry the host admits)
constexpr int NWQ = 5;   // filter float4 slots per thread

template <int V, int NFW, int SRC>
__global__ __launch_bounds__(256) void igemm_heavy_kernel(const IGemmArgs a, const float* __restrict__ wp, const int CoP) {
    constexpr int CK = 4 * V, CKP = ckp_heavy_of<V>(), BN = 16 * NFW;
    typedef typename Frag<V>::T frag_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PSZ = (a.PH * a.PW * CKP + 3) & ~3, WSZ = a.TG * BN * CKP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int co0 = blockIdx.y * BN;
    const int cls = blockIdx.z, pa = cls >> 1, pb = cls & 1;
    const int s = a.phase ? 1 : a.stride;
    const int nchunks = (a.Cr + CK - 1) / CK;
    const int ngroups = (a.ntaps + a.TG - 1) / a.TG;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nchunks * ngroups;
    const int npatch = a.PH * a.PW * V;

    // per-thread slot descriptors (tile independent).  patch: LDS float offset, row, col, quad; unused slots read pixel 0.
    int p_lds[NPQ], p_rc[NPQ];
#pragma unroll
    for (int j = 0; j < NPQ; ++j) {
        const int e = tid + 256 * j;
        const bool used = e < npatch;
        const int ee = used ? e : 0;
        const int pr = ee / (a.PW * V), rem = ee % (a.PW * V), pc = rem / V, q = rem % V;
        p_lds[j] = used ? (pr * a.PW + pc) * CKP + 4 * q : -1;
        p_rc[j] = pr | (pc << 8) | (q << 16);
    }
    int w_src[NWQ], w_lds[NWQ];  // filter: offset inside the stage's packed slice (tap-local), LDS float offset, or -1
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int e = tid + 256 * j, row = e / V, q = e % V, tl = row / BN, col = row % BN;
        const bool used = tl < a.TG;
        w_src[j] = used ? (tl * CoP + col) * CK + 4 * q : 0;
        w_lds[j] = used ? row * CKP + 4 * q : -1;
    }

    // two cursors over (tile sequence number, chunk, tap group): cur = stage being computed, nxt = stage being loaded
    int c_seq = 0, c_chunk = 0, c_tg = 0, c_tx0, c_ty0, c_n;
    int n_seq = 0, n_chunk = 0, n_tg = 0, n_tx0, n_ty0, n_n;
    {
        const int tile = xcd_swizzle((int)blockIdx.x, ntiles);
        c_tx0 = n_tx0 = tile % a.tiles_x;
        c_ty0 = n_ty0 = (tile / a.tiles_x) % a.tiles_y;
        c_n = n_n = tile / (a.tiles_x * a.tiles_y);
    }

    f32x4 preg[NPQ], wreg[NWQ];

#define IG_LOAD_STAGE()                                                                                                     \
    {                                                                                                                       \
        if (n_tg == 0) {                                                                                                    \
            const int y0 = a.phase ? n_ty0 * TILE_H - 1 : n_ty0 * TILE_H * s - a.pad;                                       \
            const int x0 = a.phase ? n_tx0 * TILE_W - 1 : n_tx0 * TILE_W * s - a.pad;                                       \
            _Pragma("unroll") for (int j = 0; j < NPQ; ++j) {                                                               \
                const int iy = min(max(y0 + (p_rc[j] & 255), 0), a.H - 1), ix = min(max(x0 + ((p_rc[j] >> 8) & 255), 0), a.W - 1); \
                if (SRC == 2) {                                                                                             \
                    const size_t cs = (size_t)a.H * a.W;                                                                    \
                    const float* src = a.x + (((size_t)n_n * a.x_ctot + a.x_coff) * a.H + iy) * a.W + ix;                   \
                    preg[j][0] = src[0];                                                                                    \
                    preg[j][1] = src[(a.Cr > 1 ? 1 : 0) * cs];                                                              \
                    preg[j][2] = src[(a.Cr > 2 ? 2 : 0) * cs];                                                              \
                    preg[j][3] = src[(a.Cr > 3 ? 3 : 0) * cs];                                                              \
                } else if (SRC == 0) {                                                                                      \
                    const int c0 = min(n_chunk * CK + 4 * (p_rc[j] >> 16), a.Cr - 4);                                       \
                    preg[j] = *reinterpret_cast<const f32x4*>(a.x + (((size_t)n_n * a.H + iy) * a.W + ix) * a.x_ctot + a.x_coff + c0); \
                } else {                                                                                                    \
                    const int c0 = min(n_chunk * CK + 4 * (p_rc[j] >> 16), a.Cr - 1), rem = a.Cr - 1 - c0;                  \
                    const float* src = a.x + (((size_t)n_n * a.H + iy) * a.W + ix) * a.x_ctot + a.x_coff + c0;              \
                    preg[j][0] = src[0];                                                                                    \
                    preg[j][1] = src[min(1, rem)];                                                                          \
                    preg[j][2] = src[min(2, rem)];                                                                          \
                    preg[j][3] = src[min(3, rem)];                                                                          \
                }                                                                                                           \
            }                                                                                                               \
        }                                                                                                                   \
        {                                                                                                                   \
            const int t0 = n_tg * a.TG, tlast = a.ntaps - 1 - t0;                                                           \
            const float* base = wp + ((size_t)((cls * nchunks + n_chunk) * a.ntaps + t0) * CoP + co0) * CK;                 \
            const int lim = tlast * CoP * CK;  /* rows of taps beyond the filter are clamped to its last tap */             \
            _Pragma("unroll") for (int j = 0; j < NWQ; ++j) {                                                               \
                const int off = w_src[j] > lim + (CoP - 1) * CK + CK - 4 ? 0 : w_src[j];                                    \
                wreg[j] = *reinterpret_cast<const f32x4*>(base + off);                                                      \
            }                                                                                                               \
        }                                                                                                                   \
    }

#define IG_STORE_STAGE(KBUF)                                                                                                \
    {                                                                                                                       \
        if (n_tg == 0) {                                                                                                    \
            float* patch = smem + ((n_seq * nchunks + n_chunk) & 1) * PSZ;                                                  \
            const int y0 = a.phase ? n_ty0 * TILE_H - 1 : n_ty0 * TILE_H * s - a.pad;                                       \
            const int x0 = a.phase ? n_tx0 * TILE_W - 1 : n_tx0 * TILE_W * s - a.pad;                                       \
            _Pragma("unroll") for (int j = 0; j < NPQ; ++j) {                                                               \
                if (p_lds[j] >= 0) {                                                                                        \
                    const int iy = y0 + (p_rc[j] & 255), ix = x0 + ((p_rc[j] >> 8) & 255);                                  \
                    const bool inb = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;                          \
                    const int c0 = SRC == 2 ? 0 : n_chunk * CK + 4 * (p_rc[j] >> 16);                                       \
                    const int nreal = inb ? a.Cr - c0 : 0;                                                                  \
                    f32x4 v = preg[j];                                                                                      \
                    v[0] = nreal > 0 ? v[0] : 0.f;                                                                          \
                    v[1] = nreal > 1 ? v[1] : 0.f;                                                                          \
                    v[2] = nreal > 2 ? v[2] : 0.f;                                                                          \
                    v[3] = nreal > 3 ? v[3] : 0.f;                                                                          \
                    *reinterpret_cast<f32x4*>(&patch[p_lds[j]]) = v;                                                        \
                }                                                                                                           \
            }                                                                                                               \
        }                                                                                                                   \
        {                                                                                                                   \
            float* wl = smem + 2 * PSZ + (KBUF)*WSZ;                                                                        \
            const int tn = min(a.TG, a.ntaps - n_tg * a.TG);                                                                \
            _Pragma("unroll") for (int j = 0; j < NWQ; ++j) {                                                               \
                if (w_lds[j] >= 0 && w_lds[j] < tn * BN * CKP) *reinterpret_cast<f32x4*>(&wl[w_lds[j]]) = wreg[j];          \
            }                                                                                                               \
        }                                                                                                                   \
    }

#define IG_ADVANCE(P)                                                                                                       \
    {                                                                                                                       \
        if (++P##_tg == ngroups) {                                                                                          \
            P##_tg = 0;                                                                                                     \
            if (++P##_chunk == nchunks) {                                                                                   \
                P##_chunk = 0;                                                                                              \
                ++P##_seq;                                                                                                  \
                if (P##_seq < my_tiles) {                                                                                   \
                    const int tile = xcd_swizzle((int)blockIdx.x + P##_seq * (int)gridDim.x, ntiles);                       \
                    P##_tx0 = tile % a.tiles_x;                                                                             \
                    P##_ty0 = (tile / a.tiles_x) % a.tiles_y;                                                               \
                    P##_n = tile / (a.tiles_x * a.tiles_y);                                                                 \
                }                                                                                                           \
            }                                                                                                               \
        }                                                                                                                   \
    }

    if (total > 0) {
        IG_LOAD_STAGE();
        IG_STORE_STAGE(0);
        IG_ADVANCE(n);
    }
    __syncthreads();

    f32x4 acc[NFW][2];
    for (int k = 0; k < total; ++k) {
        const bool more = k + 1 < total;
        if (more) IG_LOAD_STAGE();  // in flight during the MFMAs below
        const float* wl = smem + 2 * PSZ + (k & 1) * WSZ;
        const float* patch = smem + ((c_seq * nchunks + c_chunk) & 1) * PSZ;
        if (c_chunk == 0 && c_tg == 0) {
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- MFMA over the group's taps; the fragments of tap t+1 are read from LDS while tap t's MFMAs issue ------
        const int t0 = c_tg * a.TG, tn = min(a.TG, a.ntaps - t0);
        const int brow0 = ((2 * wave) * s * a.PW + i * s) * CKP + V * g, brow1 = brow0 + s * a.PW * CKP;
        const int arow = i * CKP + V * g;
        frag_t af[NFW], bf[2], afn[NFW], bfn[2];
        {
            const int po = tap_patch_offset(a, t0, pa, pb, CKP);
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) af[wf] = *reinterpret_cast<const frag_t*>(&wl[16 * wf * CKP + arow]);
            bf[0] = *reinterpret_cast<const frag_t*>(&patch[brow0 + po]);
            bf[1] = *reinterpret_cast<const frag_t*>(&patch[brow1 + po]);
        }
        for (int tl = 0; tl < tn; ++tl) {
            const int tnext = min(tl + 1, tn - 1);
            const int po = tap_patch_offset(a, t0 + tnext, pa, pb, CKP);
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) afn[wf] = *reinterpret_cast<const frag_t*>(&wl[(tnext * BN + 16 * wf) * CKP + arow]);
            bfn[0] = *reinterpret_cast<const frag_t*>(&patch[brow0 + po]);
            bfn[1] = *reinterpret_cast<const frag_t*>(&patch[brow1 + po]);
#pragma unroll
            for (int j = 0; j < V; ++j)
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                    for (int pf = 0; pf < 2; ++pf)
                        acc[wf][pf] = mfma16(frag_get<V>(af[wf], j), frag_get<V>(bf[pf], j), acc[wf][pf]);
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) af[wf] = afn[wf];
            bf[0] = bfn[0];
            bf[1] = bfn[1];
        }
        if (c_chunk == nchunks - 1 && c_tg == ngroups - 1) igemm_epilogue<NFW>(a, acc, c_n, c_ty0, c_tx0, co0, pa, pb, wave, i, g);
        if (more) {
            IG_STORE_STAGE((k + 1) & 1);
            IG_ADVANCE(n);
        }
        IG_ADVANCE(c);
        __syncthreads();
    }
#undef IG_LOAD_STAGE
#undef IG_STORE_STAGE
#undef IG_ADVANCE
}

struct IGemmPlan {
    int pf;      // light / stream kernel: tile rows per wave (2 or 4)
    int stream;  // persistent streaming kernel (filter resident in LDS, next patch prefetched into registers)
    int gx;      // stream kernel: persistent workgroups along x
    int V, nfw, src, CK, CKP, BN, CoP, nchunks, ncls, TG, heavy;
    size_t lds, ws_bytes;
};

// tile / patch geometry for a tile height of th grid rows
static void igemm_geometry(IGemmArgs& a, int th) {
    a.TH = th;
    a.tiles_x = cdiv(a.Gw, TILE_W);
    a.tiles_y = cdiv(a.Gh, th);
    a.tapsx = cdiv(a.KW, 4);
    if (a.phase) {
        a.PH = th + 2;
        a.PW = TILE_W + 2;
        a.ntaps = 4;
    } else if (a.dpack) {
        a.tiles_x = cdiv(a.Gw, 13);
        a.PH = (th - 1) + (a.KH - 1) + 1;
        a.PW = (TILE_W - 1) + 4 * a.tapsx;
        a.ntaps = a.KH * a.tapsx;
    } else {
        a.PH = (th - 1) * a.stride + (a.KH - 1) * a.dil + 1;
        a.PW = (TILE_W - 1) * a.stride + (a.KW - 1) * a.dil + 1;
        a.ntaps = a.KH * a.KW;
    }
}

static int plan_igemm(IGemmArgs& a, IGemmPlan& p) {
    // <= 4 output channels (the RGB head, the stem's input gradient, the discriminator heads, 4-channel branches): pack four
    // horizontally adjacent taps into the 16 filter rows of the MFMA tile instead of padding 4 channels to 16
    a.dpack = !a.phase && a.Co <= 4 && a.stride == 1 && a.dil == 1 && a.KW > 1;
    { const char* e = env_get(ENV_NO_DPACK); if (e && e[0] == '1') a.dpack = 0; }
    igemm_geometry(a, TILE_H);
    p.pf = 2;
    p.stream = 0;
    a.wglob = 0;
    p.gx = 0;
    if (a.N <= 0 || a.Gh <= 0 || a.Gw <= 0 || a.Co <= 0 || a.Cr <= 0) return fail_arg(MSTG_E_BADARG, "conv: empty tensor");
    if (a.x_nchw) {
        if (a.Cr > 4) return fail_arg(MSTG_E_UNSUPPORTED, "conv: NCHW source supports at most 4 channels");
        p.V = 1;
    } else if (a.Cr % 16 == 0) p.V = 4;
    else if (a.Cr % 8 == 0) p.V = 2;
    else p.V = 1;  // any channel count: 4-channel k-slots, the tail quad zero-filled
    p.src = a.x_nchw ? 2 : ((((a.x_ctot | a.x_coff | a.Cr) & 3) == 0) ? 0 : 1);
    p.nfw = a.Co <= 16 ? 1 : (a.Co <= 32 ? 2 : 4);
    // Deep-channel layers on small maps (the discriminator's 32x32 / 16x16 tail: 64 -> 128 k4 s2, 128 -> 128 3x3) make ~128 workgroups
    // of 64 output channels -- half the chip's SIMDs get nothing.  Narrower workgroups (32 / 16 output channels) re-stage the (L2-resident)
    // patch more often but fill the machine: halve the fragment count until there are >= 512 workgroups.  MSTG_NFW_ADAPT=0 keeps 64.
    {
        const char* e = env_get(ENV_NFW_ADAPT);
        if (!(e && e[0] == '0') && !a.dpack && a.Cr >= 64) {
            const long per = (long)a.N * a.tiles_x * a.tiles_y * (a.phase ? 4 : 1);
            while (p.nfw > 1 && per * cdiv(a.Co, 16 * p.nfw) < 512) p.nfw >>= 1;
        }
    }
    p.CK = 4 * p.V;
    p.CKP = p.V == 4 ? ckp_of<4>() : (p.V == 2 ? 12 : 4);
    const int ckp_heavy = p.V == 4 ? ckp_heavy_of<4>() : p.CKP;
    p.BN = 16 * p.nfw;
    p.CoP = cdiv(a.Co, p.BN) * p.BN;
    p.nchunks = cdiv(a.Cr, p.CK);
    p.ncls = a.phase ? 4 : 1;
    p.ws_bytes = ((size_t)p.ncls * p.nchunks * a.ntaps * p.CoP * p.CK + 64) * sizeof(float);
    // filter slice per stage: bounded by the LDS budget (and, for the pipelined kernel, by its prefetch registers)
    int tg = W_BUDGET_FLOATS / (p.BN * p.CKP);
    if (tg < 1) tg = 1;
    if (tg > a.ntaps) tg = a.ntaps;
    int tgh = (NWQ * 256) / (p.BN * p.V);
    if (tgh > tg) tgh = tg;
    const size_t patch_floats = (size_t)((a.PH * a.PW * ckp_heavy + 3) & ~3);
    const size_t lds_heavy = 2 * (patch_floats + (size_t)tgh * p.BN * ckp_heavy) * sizeof(float);
    // heavy = enough MFMAs per stage and wave to cover the latency of the next stage's loads
    const int mfma_per_stage = tgh * p.nfw * 2 * p.V;
    p.heavy = !a.dpack && tgh >= 1 && mfma_per_stage >= 64 && a.PH <= 255 && a.PW <= 255 && a.PH * a.PW * p.V <= NPQ * 256 &&
              lds_heavy <= 160 * 1024 && !(p.src == 2 && p.V != 1) && !(p.src == 0 && a.Cr < 4);
    // measured on MI355X: with the packed filter the high-occupancy kernel wins everywhere except on deep-channel layers
    // with few tiles (the discriminator's 32x32 / 16x16 maps), where a workgroup has too few neighbours to hide behind
    {
        const char* e = env_get(ENV_IGEMM);
        if (!(e && e[0] == 'h')) p.heavy = p.heavy && a.Cr >= 64 && a.N * a.tiles_x * a.tiles_y * (p.CoP / p.BN) <= 1024;
        if (e && e[0] == 'l') p.heavy = 0;
    }
    if (p.heavy) {
        p.TG = a.TG = tgh;
        p.CKP = ckp_heavy;
        p.lds = lds_heavy;
    } else {
        // ---- streaming kernel: aligned NHWC or <= 4-channel NCHW source, the whole filter of one workgroup resident in
        //      LDS (<= 28 KiB), the patch within SNB slots per thread, >= 3 workgroups per CU ---------------------------------
        p.stream = 0;
        {
            const char* e = env_get(ENV_STREAM);
            // Off by default: measured on MI355X (tools/diag_stamps.py) it wins only on the tap-heavy 16-channel layers
            // (3x3 dilation 4: 0.165 -> 0.140 ms, 7x7 head: 0.278 -> 0.245 ms) and loses on the rest, because those layers are
            // bound by the L2 -> CU traffic of the halo re-reads, not by the serialised phases.  MSTG_STREAM=1 enables it.
            const bool allow = (e && e[0] == '1') && (p.src == 0 || (p.src == 2 && p.V == 1));
            const size_t wsz = (size_t)p.nchunks * a.ntaps * p.BN * p.CKP;
            const int ny = p.CoP / p.BN;
            const char* epf = env_get(ENV_PF);
            const int force = epf ? atoi(epf) : 0;
            for (int th = 16; allow && th >= 8 && !p.stream; th -= 8) {
                if ((th == 16 && force == 2) || (th == 8 && force == 4)) continue;
                if (a.Gh < th) continue;
                igemm_geometry(a, th);
                const int pf = th / 4;
                size_t psz = (size_t)((a.PH * a.PW * p.CKP + 3) & ~3);
                if (a.dpack && psz < (size_t)1024 * pf) psz = (size_t)1024 * pf;
                const size_t lds = (psz + wsz + 64) * sizeof(float);
                const long ntiles = (long)a.N * a.tiles_x * a.tiles_y;
                int per_cu = (int)((160 * 1024) / lds);
                if (per_cu > 6) per_cu = 6;
                if (a.PH * a.PW * p.V > SNB * 256 || a.PH > 255 || a.PW > 255 || wsz * sizeof(float) > 28 * 1024 || per_cu < 3) continue;
                int gx = (256 * per_cu) / (ny * p.ncls);
                gx = gx < 8 ? 8 : (gx & ~7);
                if (ntiles < 3L * gx && !(e && e[1] == 'f')) continue;  // a persistent workgroup needs a few tiles to pipeline ("1f" forces: tests)
                p.stream = 1; p.pf = pf; p.gx = gx; a.psz = (int)psz; p.lds = lds;
            }
            if (!p.stream) igemm_geometry(a, TILE_H);
        }
        if (p.stream) return MSTG_OK;
        // four rows per wave where the taller patch still leaves >= 3 workgroups per CU and the grid stays >= 4 per CU
        {
            const char* e = env_get(ENV_PF);
            const int force = e ? atoi(e) : 0;
            const int ph16 = a.phase ? 18 : (a.dpack ? 15 + a.KH : 15 * a.stride + (a.KH - 1) * a.dil + 1);
            const char* ew = env_get(ENV_WGLOB);
            const int wg16 = ew ? atoi(ew) : (p.V == 4 ? 1 : 0);
            const size_t lds16 = ((size_t)((ph16 * a.PW * p.CKP + 3) & ~3) + (wg16 ? 0 : (size_t)tg * p.BN * p.CKP) + 64) * sizeof(float);
            const long blocks16 = (long)a.N * a.tiles_x * cdiv(a.Gh, 16) * (p.CoP / p.BN) * p.ncls;
            if (force == 4 || (force != 2 && lds16 <= 52 * 1024 && blocks16 >= 1024 && a.Gh >= 16)) {
                p.pf = 4;
                igemm_geometry(a, 16);
            }
        }
        const size_t patch_floats_l = (size_t)((a.PH * a.PW * p.CKP + 3) & ~3);
        p.TG = a.TG = tg;
        {
            // filter fragments from L2 where a tap's MFMAs (V * NFW * PF of them) are long enough to cover the fetch of the tap
            // after next: frees the LDS of the filter slice (more workgroups per CU) and its staging barrier
            const char* e = env_get(ENV_WGLOB);
            a.wglob = e ? atoi(e) : (p.V == 4 ? 1 : 0);
        }
        p.lds = (patch_floats_l + (a.wglob ? 0 : (size_t)tg * p.BN * p.CKP) + 64) * sizeof(float);  // + tap-offset table (<= 64 taps)
        if (a.dpack && p.lds < (size_t)4 * 256 * p.pf * sizeof(float)) p.lds = (size_t)4 * 256 * p.pf * sizeof(float);
        if (p.lds > 160 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "conv: LDS patch too large for this geometry");
    }
    return MSTG_OK;
}

static int launch_pack(IGemmArgs& a, const IGemmPlan& p, float* wp, hipStream_t st) {
    const int npack = p.ncls * p.nchunks * a.ntaps * p.CoP * p.CK;
    const int nb = cdiv(npack, 256) > 256 ? 256 : cdiv(npack, 256);
    MSTG_PACK_LAUNCH(pack_filter_kernel, dim3(nb), dim3(256), 0, st, a, wp, p.CK, p.CoP, p.nchunks, p.ncls);
    MSTG_CHECK_LAUNCH("pack_filter_kernel");
    return MSTG_OK;
}

template <int V, int NFW, int PF>
static int launch_light_t(IGemmArgs& a, const IGemmPlan& p, float* wp, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_light_kernel<V, NFW, PF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(igemm_light)");
        attr_set = true;
    }
    dim3 grid(a.N * a.tiles_x * a.tiles_y, p.CoP / p.BN, p.ncls);
    MSTG_LAUNCH((igemm_light_kernel<V, NFW, PF>), grid, dim3(256), p.lds, st, a, (const float*)wp, p.CoP);
    MSTG_CHECK_LAUNCH("igemm_light_kernel");
    return MSTG_OK;
}

template <int V, int NFW, int PF, int SRC>
static int launch_stream_t(IGemmArgs& a, const IGemmPlan& p, float* wp, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_stream_kernel<V, NFW, PF, SRC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(igemm_stream)");
        attr_set = true;
    }
    dim3 grid(p.gx, p.CoP / p.BN, p.ncls);
    MSTG_LAUNCH((igemm_stream_kernel<V, NFW, PF, SRC>), grid, dim3(256), p.lds, st, a, (const float*)wp, p.CoP);
    MSTG_CHECK_LAUNCH("igemm_stream_kernel");
    return MSTG_OK;
}

template <int V, int NFW, int SRC>
static int launch_heavy_t(IGemmArgs& a, const IGemmPlan& p, float* wp, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_heavy_kernel<V, NFW, SRC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(igemm_heavy)");
        attr_set = true;
    }
    // persistent workgroups: as many as fit on the chip at once (LDS / register bound), a multiple of 8 so that one
    // workgroup's tiles all fall into one XCD's contiguous run of the tile order (xcd_swizzle)
    const int ntiles = a.N * a.tiles_x * a.tiles_y, ny = p.CoP / p.BN, nz = p.ncls;
    int per_cu = p.lds <= 80 * 1024 ? 2 : 1;
    { const char* e = env_get(ENV_HEAVY_PER_CU); if (e) per_cu = atoi(e); }
    int gx = (256 * per_cu) / (ny * nz);
    gx = gx < 8 ? 8 : (gx & ~7);
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, ny, nz);
    MSTG_LAUNCH((igemm_heavy_kernel<V, NFW, SRC>), grid, dim3(256), p.lds, st, a, (const float*)wp, p.CoP);
    MSTG_CHECK_LAUNCH("igemm_heavy_kernel");
    return MSTG_OK;
}

size_t igemm_workspace_bytes(IGemmArgs a) {
    IGemmPlan p;
    if (plan_igemm(a, p)) return 0;
    const size_t alt = p32_workspace_bytes(a);  // the persistent kernel packs its filter differently (conv_p32.hip)
    return p.ws_bytes > alt ? p.ws_bytes : alt;
}

int launch_igemm(IGemmArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (img_dgrad_eligible(a)) return launch_img_dgrad(a, st);  // 3-channel image gradient of a 4x4 stride-2 layer: no MFMA tile to fill
    if (p32_eligible(a)) return launch_p32(a, workspace, workspace_bytes, st);  // 4x4 stride-2 family at 16 / 32 / 64 channels
    a.dbg = 0;
    IGemmPlan p;
    if (int rc = plan_igemm(a, p)) return rc;
    { const char* e = env_get(ENV_DBG); if (e) a.dbg = atoi(e); }                       // experiments only: bit 0 = one tap only
    { const char* e = env_get(ENV_DBG_LDS_KB); if (e && !p.heavy) p.lds += (size_t)atoi(e) * 1024; }  // lower the occupancy
    if (!workspace || workspace_bytes < p.ws_bytes) return fail_arg(MSTG_E_WORKSPACE, "conv: workspace too small for the packed filter");
    float* wp = (float*)workspace;
    if (int rc = launch_pack(a, p, wp, st)) return rc;
#define MSTG_DISPATCH(VV, NN)                                                              \
    if (p.V == VV && p.nfw == NN) {                                                        \
        if (p.stream && p.src == 0 && p.pf == 4) return launch_stream_t<VV, NN, 4, 0>(a, p, wp, st); \
        if (p.stream && p.src == 0) return launch_stream_t<VV, NN, 2, 0>(a, p, wp, st);   \
        if (VV == 1 && p.stream && p.src == 2 && p.pf == 4) return launch_stream_t<1, NN, 4, 2>(a, p, wp, st); \
        if (VV == 1 && p.stream && p.src == 2) return launch_stream_t<1, NN, 2, 2>(a, p, wp, st); \
        if (!p.heavy && p.pf == 4) return launch_light_t<VV, NN, 4>(a, p, wp, st);          \
        if (!p.heavy) return launch_light_t<VV, NN, 2>(a, p, wp, st);                      \
        if (p.src == 0) return launch_heavy_t<VV, NN, 0>(a, p, wp, st);                    \
        if (p.src == 1) return launch_heavy_t<VV, NN, 1>(a, p, wp, st);                    \
        if (VV == 1 && p.src == 2) return launch_heavy_t<1, NN, 2>(a, p, wp, st);          \
    }
    MSTG_DISPATCH(1, 1) MSTG_DISPATCH(1, 2) MSTG_DISPATCH(1, 4)
    MSTG_DISPATCH(2, 1) MSTG_DISPATCH(2, 2) MSTG_DISPATCH(2, 4)
    MSTG_DISPATCH(4, 1) MSTG_DISPATCH(4, 2) MSTG_DISPATCH(4, 4)
#undef MSTG_DISPATCH
    return fail_arg(MSTG_E_UNSUPPORTED, "conv: no kernel variant");
}

int check_desc(const mstg_conv_desc* d) {
    if (!d) return fail_arg(MSTG_E_BADARG, "conv: null descriptor");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0 || d->stride <= 0 ||
        d->dil <= 0 || d->pad < 0)
        return fail_arg(MSTG_E_BADARG, "conv: non-positive dimension");
    if (d->transposed) {
        if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad != 1 || d->dil != 1)
            return fail_arg(MSTG_E_UNSUPPORTED, "conv_transpose: only k4 s2 p1 is implemented");
        if (d->Ho != 2 * d->H || d->Wo != 2 * d->W) return fail_arg(MSTG_E_BADARG, "conv_transpose: Ho,Wo must be 2H,2W");
    } else {
        const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
        const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
        if (ho != d->Ho || wo != d->Wo) return fail_arg(MSTG_E_BADARG, "conv: Ho,Wo inconsistent with H,W,k,s,p,d");
        if (d->stride != 1 && !(d->stride == 2 && d->KH == 4 && d->KW == 4 && d->pad == 1 && d->dil == 1 && d->H % 2 == 0 &&
                                d->W % 2 == 0))
            return fail_arg(MSTG_E_UNSUPPORTED, "conv: strided convolution only as k4 s2 p1 on even H,W");
    }
    if (d->x_ctot < d->x_coff + d->Cin || d->y_ctot < d->y_coff + d->Cout) return fail_arg(MSTG_E_BADARG, "conv: channel slice out of range");
    // the kernels index inside one image with 32-bit offsets (the image base is a 64-bit pointer)
    if ((uint64_t)d->H * d->W * d->x_ctot >= (1ull << 30) || (uint64_t)d->Ho * d->Wo * d->y_ctot >= (1ull << 30))
        return fail_arg(MSTG_E_UNSUPPORTED, "conv: one image must stay below 2^30 elements");
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

void fill_fwd_args(const mstg_conv_desc* d, IGemmArgs& a) {
    const int T = d->KH * d->KW;
    a.N = d->N;
    a.H = d->H; a.W = d->W; a.x_ctot = d->x_ctot; a.x_coff = d->x_coff; a.x_nchw = d->x_nchw; a.Cr = d->Cin;
    a.Ho = d->Ho; a.Wo = d->Wo; a.y_ctot = d->y_ctot; a.y_coff = d->y_coff; a.y_nchw = d->y_nchw; a.Co = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.flip = 0;
    a.act = d->act; a.accumulate = d->accumulate;
    if (d->transposed) {
        a.phase = 1; a.Gh = d->H; a.Gw = d->W;
        a.w_so = T; a.w_sr = d->Cout * T;  // IOHW
    } else {
        a.phase = 0; a.Gh = d->Ho; a.Gw = d->Wo;
        a.w_so = d->Cin * T; a.w_sr = T;   // OIHW
    }
}

int fill_dgrad_args(const mstg_conv_desc* d, IGemmArgs& a) {
    const int T = d->KH * d->KW;
    a.N = d->N;
    // source = module output gradient, destination = module input gradient
    a.H = d->Ho; a.W = d->Wo; a.x_ctot = d->y_ctot; a.x_coff = d->y_coff; a.x_nchw = d->y_nchw; a.Cr = d->Cout;
    a.Ho = d->H; a.Wo = d->W; a.y_ctot = d->x_ctot; a.y_coff = d->x_coff; a.y_nchw = d->x_nchw; a.Co = d->Cin;
    a.KH = d->KH; a.KW = d->KW; a.dil = d->dil;
    a.act = MSTG_ACT_NONE; a.accumulate = d->accumulate;
    if (d->transposed) {
        // dX[iy] = sum_t dY[2 iy - 1 + t] W(ci,co,t): a stride-2 gather over dY
        a.phase = 0; a.Gh = d->H; a.Gw = d->W; a.stride = 2; a.pad = 1; a.flip = 0;
        a.w_so = d->Cout * T; a.w_sr = T;
    } else if (d->stride == 1) {
        a.phase = 0; a.Gh = d->H; a.Gw = d->W; a.stride = 1; a.pad = d->dil * (d->KH - 1) - d->pad; a.flip = 1;
        if (d->KH != d->KW) return fail_arg(MSTG_E_UNSUPPORTED, "conv_dgrad: square kernels only");
        if (a.pad < 0) return fail_arg(MSTG_E_UNSUPPORTED, "conv_dgrad: pad larger than dil*(k-1)");
        a.w_so = T; a.w_sr = d->Cin * T;
    } else {
        // stride-2 k4 p1: same parity-class structure as the transposed forward, over dY
        a.phase = 1; a.Gh = d->Ho; a.Gw = d->Wo; a.stride = 2; a.pad = 1; a.flip = 0;
        a.w_so = T; a.w_sr = d->Cin * T;
    }
    return MSTG_OK;
}

extern "C" size_t mstg_conv2d_workspace_bytes(const mstg_conv_desc* d) {
    if (check_desc(d)) return 0;
    IGemmArgs f{}, b{};
    fill_fwd_args(d, f);
    if (fill_dgrad_args(d, b)) return 0;
    const size_t wf = igemm_workspace_bytes(f), wb = igemm_workspace_bytes(b);
    return wf > wb ? wf : wb;
}

extern "C" int mstg_conv2d_fwd(const mstg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!x || !w || !y) return fail_arg(MSTG_E_BADARG, "conv_fwd: null pointer");
    IGemmArgs a{};
    fill_fwd_args(d, a);
    a.x = x; a.y = y; a.w = w; a.bias = bias;
    return launch_igemm(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- forward with InstanceNorm folded in on either side (only where the persistent kernel runs: conv_p32.hip) ----------------------
extern "C" int mstg_conv2d_fwd_norm_supported(const mstg_conv_desc* d) {
    if (check_desc(d)) return 0;
    IGemmArgs a{};
    fill_fwd_args(d, a);
    return p32_eligible(a) && !a.x_nchw && !a.y_nchw && a.Co != 1 ? 1 : 0;
}

// 1 where taking the output's statistics in the epilogue is cheaper than a statistics pass over the output (MSTG_BSUMS_ALL=1: wherever
// mstg_conv2d_fwd_norm_supported)
extern "C" int mstg_conv2d_fwd_stats_pays(const mstg_conv_desc* d) {
    if (!mstg_conv2d_fwd_norm_supported(d)) return 0;
    IGemmArgs a{};
    fill_fwd_args(d, a);
    const char* e = env_get(ENV_BSUMS_ALL);
    return (e && e[0] == '1') || p32_stats_pays(a) ? 1 : 0;
}

extern "C" size_t mstg_conv2d_fwd_norm_workspace_bytes(const mstg_conv_desc* d) {
    if (check_desc(d)) return 0;
    IGemmArgs a{};
    fill_fwd_args(d, a);
    return p32_eligible(a) && !a.x_nchw && !a.y_nchw && a.Co != 1 ? p32_norm_workspace_bytes(a) : 0;
}

extern "C" int mstg_conv2d_fwd_norm(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* w, const float* bias,
                                    float* y, float* out_stats, void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!x || !w || !y) return fail_arg(MSTG_E_BADARG, "conv_fwd_norm: null pointer");
    IGemmArgs a{};
    fill_fwd_args(d, a);
    a.x = x; a.y = y; a.w = w; a.bias = bias;
    if (!p32_eligible(a) || a.x_nchw || a.y_nchw || a.Co == 1) return fail_arg(MSTG_E_UNSUPPORTED, "conv_fwd_norm: only the layers mstg_conv2d_fwd_norm_supported() reports");
    return launch_p32_norm(a, in_stats, out_stats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mstg_conv2d_dgrad(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!dy || !w || !dx) return fail_arg(MSTG_E_BADARG, "conv_dgrad: null pointer");
    IGemmArgs a{};
    if (int rc = fill_dgrad_args(d, a)) return rc;
    a.x = dy; a.y = dx; a.w = w; a.bias = nullptr;
    return launch_igemm(a, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- input gradient + the reductions of the norm backward it feeds (conv_p32.hip, STATS == 2) ----------------------------------------
extern "C" int mstg_conv2d_dgrad_bsums_supported(const mstg_conv_desc* d) {
    if (check_desc(d)) return 0;
    IGemmArgs a{};
    if (fill_dgrad_args(d, a)) return 0;
    if (!(p32_generic(a) && a.Co != 1 && !d->accumulate && d->x_ctot == d->Cin && d->x_coff == 0)) return 0;
    const char* e = env_get(ENV_BSUMS_ALL);  // MSTG_BSUMS_ALL=1: wherever the kernel can, not only where it pays (tests, A/B)
    return (e && e[0] == '1') || p32_bsums_pays(a) ? 1 : 0;
}

extern "C" size_t mstg_conv2d_dgrad_bsums_workspace_bytes(const mstg_conv_desc* d) {
    if (!mstg_conv2d_dgrad_bsums_supported(d)) return 0;
    IGemmArgs a{};
    fill_dgrad_args(d, a);
    return p32_norm_workspace_bytes(a);
}

extern "C" int mstg_conv2d_dgrad_bsums(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, const float* x_raw,
                                       const float* x_stats, float* sums, void* workspace, size_t workspace_bytes, int workspace_packed,
                                       void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!dy || !w || !dx || !x_raw || !x_stats || !sums) return fail_arg(MSTG_E_BADARG, "conv_dgrad_bsums: null pointer");
    if (!mstg_conv2d_dgrad_bsums_supported(d)) return fail_arg(MSTG_E_UNSUPPORTED, "conv_dgrad_bsums: only the layers mstg_conv2d_dgrad_bsums_supported() reports");
    IGemmArgs a{};
    if (int rc = fill_dgrad_args(d, a)) return rc;
    a.x = dy; a.y = dx; a.w = w; a.bias = nullptr;
    mstg::t_ws_packed = workspace_packed != 0;
    const int rc = launch_p32_bsums(a, x_raw, x_stats, sums, workspace, workspace_bytes, (hipStream_t)stream);
    mstg::t_ws_packed = false;
    return rc;
}

// ---- the same entry points for a caller that caches filter packs (common.h: t_ws_packed) -------------------------------------------
namespace {
struct PackedScope {
    explicit PackedScope(int packed) { mstg::t_ws_packed = packed != 0; }
    ~PackedScope() { mstg::t_ws_packed = false; }
};
}  // namespace
extern "C" int mstg_conv2d_fwd_cached(const mstg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace,
                                      size_t workspace_bytes, int workspace_packed, void* stream) {
    PackedScope scope(workspace_packed);
    return mstg_conv2d_fwd(d, x, w, bias, y, workspace, workspace_bytes, stream);
}
extern "C" int mstg_conv2d_fwd_norm_cached(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* w, const float* bias,
                                           float* y, float* out_stats, void* workspace, size_t workspace_bytes, int workspace_packed,
                                           void* stream) {
    PackedScope scope(workspace_packed);
    return mstg_conv2d_fwd_norm(d, x, in_stats, w, bias, y, out_stats, workspace, workspace_bytes, stream);
}
extern "C" int mstg_conv2d_dgrad_cached(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                                        size_t workspace_bytes, int workspace_packed, void* stream) {
    PackedScope scope(workspace_packed);
    return mstg_conv2d_dgrad(d, dy, w, dx, workspace, workspace_bytes, stream);
}

// kernel symbol (as rocprofv3 prints it, without namespace/arguments) a forward (0) / dgrad (1) call would launch
const char* igemm_kernel_name(const mstg_conv_desc* d, int pass) {
    static thread_local char name[64];
    IGemmArgs a{};
    IGemmPlan p;
    if (check_desc(d)) return "";
    if (pass == 0) fill_fwd_args(d, a);
    else if (fill_dgrad_args(d, a)) return "";
    if (img_dgrad_eligible(a)) { snprintf(name, sizeof(name), "conv_img_dgrad_kernel<%d>", a.Co); return name; }
    if (plan_igemm(a, p)) return "";
    if (p32_eligible(a)) return p32_kernel_name(a);
    if (p.stream) snprintf(name, sizeof(name), "igemm_stream_kernel<%d, %d, %d, %d>", p.V, p.nfw, p.pf, p.src);
    else if (p.heavy) snprintf(name, sizeof(name), "igemm_heavy_kernel<%d, %d, %d>", p.V, p.nfw, p.src);
    else snprintf(name, sizeof(name), "igemm_light_kernel<%d, %d, %d>", p.V, p.nfw, p.pf);
    return name;
}

#ifdef MSTG_STAMPS
extern "C" int mstg_debug_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mstg::g_dbg_stamps), sizeof(unsigned long long) * 64 * 8);
}
#endif
