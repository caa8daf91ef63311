// MultiScaleBlock branch convolutions (enhanced_generator.py:52-71,79-83), fused across the four branches.
//
// The block runs a 1x1 and three dilated 3x3 convolutions (dilation 1, 2, 4), each ch -> ch/4, on the SAME input and
// concatenates the results.  As four separate weight-gradient launches the input is staged four times (with four different
// halos) and each launch pads its ch/4 output-gradient channels to a 16-wide MFMA fragment.  Seen together the four branches
// are ONE sparse 9x9-footprint convolution with 25 distinct tap offsets: the centre (shared by all four branches) and three
// rings of eight.  Here one workgroup stages the input patch (halo 4) and the full output-gradient tile once and accumulates
// all 25 taps:
//   GEMM view: M = 16 input channels (one chunk per blockIdx.y), N = the 16-channel fragment of dy the tap's branch lives in,
//   K = pixels.  A "unit" = (tap, fragment): the centre tap needs every fragment (ch/16 units), a ring tap only its branch's
//   fragment (24 units).  Units are dealt round-robin to the four waves; every wave walks all pixels of the tile for its own
//   units (no cross-wave reduction).  Per-workgroup partial slabs are reduced in a fixed order by a second kernel, which also
//   scatters the accumulators into the four PyTorch-layout weight gradients and the four bias gradients.
#include "common.h"
#include <stdlib.h>

namespace mstg {

constexpr int MS_TH = 8, MS_PH = MS_TH + 8, MS_PW = 16 + 8, MS_CKP = 20;

template <int CH>
__global__ __launch_bounds__(256) void wgrad_ms_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ partial, int N, int H, int W, int tiles_x, int tiles_y,
                                                       int ntiles) {
    constexpr int NFH = CH / 16, C4 = CH / 4, U = NFH + 24, UW = (U + 3) / 4, BNP = CH + 4, NG = CH / 16;
    constexpr int PSTRIDE = NG * U * 256 + CH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                              // [PH][PW][CKP]   16 input channels of this chunk
    float* ht = smem + MS_PH * MS_PW * MS_CKP;        // [TH*16][BNP]    all CH output-gradient channels

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int gchunk = blockIdx.y, g0 = 16 * gchunk;

    f32x4 acc[UW];
    int toff[UW], hfo[UW];
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        const int u = min(wave + 4 * k, U - 1);
        int dyp = 0, dxp = 0, frag = u;  // centre tap: fragment u
        if (u >= NFH) {
            const int r = (u - NFH) >> 3, i8 = (u - NFH) & 7, t9 = i8 < 4 ? i8 : i8 + 1, d = 1 << r;
            dyp = (t9 / 3 - 1) * d;
            dxp = (t9 % 3 - 1) * d;
            frag = ((r + 1) * C4) / 16;
        }
        toff[k] = ((4 + dyp) * MS_PW + 4 + dxp) * MS_CKP;
        hfo[k] = 16 * frag;
        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nu = (U - wave + 3) / 4;  // units this wave really owns

    const bool do_bias = gchunk == 0;
    float bsum = 0.f;  // thread c < CH: running column sum of dy channel c
    const unsigned m_pw = magic_u32(MS_PW), m_nq = magic_u32(CH / 4);
    const size_t plane = (size_t)H * W;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        __syncthreads();
        stage_window(x + (size_t)n * plane * CH + g0, patch, MS_PH, MS_PW, 4, m_pw, 0x40000000u, ty0 * MS_TH - 4, tx0 * 16 - 4, H, W, CH, 4,
                     MS_CKP, tid);
        stage_window(dy + (size_t)n * plane * CH, ht, MS_TH, 16, CH / 4, 0x10000000u, m_nq, ty0 * MS_TH, tx0 * 16, H, W, CH, CH / 4, BNP, tid);
        __syncthreads();
        if (do_bias && tid < CH) {
#pragma unroll 8
            for (int p = 0; p < MS_TH * 16; ++p) bsum += ht[p * BNP + tid];
        }
#pragma unroll 1
        for (int r = 0; r < MS_TH; ++r) {
#pragma unroll 2
            for (int xs = 0; xs < 4; ++xs) {
                const int c = 4 * xs + g;  // this lane's k-slot pixel column
                const int abase = (r * MS_PW + c) * MS_CKP + i;
                const int hbase = (r * 16 + c) * BNP + i;
                float af[UW], bf[UW];  // units beyond nu repeat the last real unit: their accumulators are never written out
#pragma unroll
                for (int k = 0; k < UW; ++k) { af[k] = patch[abase + toff[k]]; bf[k] = ht[hbase + hfo[k]]; }
#pragma unroll
                for (int k = 0; k < UW; ++k) acc[k] = mfma16(af[k], bf[k], acc[k]);
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * PSTRIDE;
    if (do_bias && tid < CH) out[NG * U * 256 + tid] = bsum;
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        if (k >= nu) continue;
        const int u = wave + 4 * k;
        // accumulator element e of lane (i, g): row m = 4g + e (input channel g0 + m), column n = i
        float* o = out + ((size_t)gchunk * U + u) * 256;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[(4 * g + e) * 16 + i] = acc[k][e];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward of the four branches in one launch.  GEMM view per 8x16 tile: M = output channels (filter rows, 16 per fragment),
// N = 16 pixels, K = (tap, input channel).  The input patch (halo 4, 16 channels per chunk) is staged ONCE for all 25 taps; a
// "unit" = (tap, output fragment) exactly as in the weight gradient: the centre tap feeds every fragment, a ring tap only the
// fragment its branch lives in.  The unit loop is fully unrolled (all patch offsets are immediates) and the filter fragments
// come straight from the packed filter in L2 (1 KiB per unit and chunk, lane-linear), so LDS holds only the patch and five
// workgroups fit on a CU.  The output is written as whole pixels (all CH channels), not as four strided channel slices.
// ---------------------------------------------------------------------------------------------------------------------
template <int CH>
struct MsUnits {
    static constexpr int NFW = CH / 16, C4 = CH / 4, U = NFW + 24;
    // unit u -> patch offset of its tap and the output fragment it feeds
    __host__ __device__ static constexpr int ring(int u) { return (u - NFW) >> 3; }
    __host__ __device__ static constexpr int t9(int u) { return ((u - NFW) & 7) < 4 ? ((u - NFW) & 7) : ((u - NFW) & 7) + 1; }
    __host__ __device__ static constexpr int oy(int u) { return u < NFW ? 0 : (t9(u) / 3 - 1) * (1 << ring(u)); }
    __host__ __device__ static constexpr int ox(int u) { return u < NFW ? 0 : (t9(u) % 3 - 1) * (1 << ring(u)); }
    __host__ __device__ static constexpr int frag(int u) { return u < NFW ? u : ((ring(u) + 1) * C4) / 16; }
};

struct MsParamPtrs {
    const float* w[4];
    const float* b[4];
};

// packed forward filter: wp[((chunk * U + u) * 64 + lane) * 4 + j] = A[row i = lane & 15][channel 16*chunk + 4*(lane >> 4) + j]
// of unit u (zero where the row's output channel does not belong to the unit's branch); then CH concatenated biases.
template <int CH>
__global__ void ms_pack_fwd_kernel(MsParamPtrs prm, float* __restrict__ wp) {
    typedef MsUnits<CH> G;
    constexpr int NCH = CH / 16, TOTAL = NCH * G::U * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < TOTAL + CH; idx += gridDim.x * blockDim.x) {
        if (idx >= TOTAL) {
            const int co = idx - TOTAL;
            wp[idx] = prm.b[co / G::C4][co % G::C4];
            continue;
        }
        const int j = idx & 3, lane = (idx >> 2) & 63, u = (idx >> 8) % G::U, chunk = idx / (G::U * 256);
        const int i = lane & 15, g = lane >> 4, ci = 16 * chunk + 4 * g + j;
        float v = 0.f;
        if (u < G::NFW) {  // centre tap of every branch
            const int co = 16 * u + i, br = co / G::C4, cj = co % G::C4;
            v = br == 0 ? prm.w[0][cj * CH + ci] : prm.w[br][(cj * CH + ci) * 9 + 4];
        } else {
            const int r = (u - G::NFW) >> 3, i8 = (u - G::NFW) & 7, t9 = i8 < 4 ? i8 : i8 + 1, br = r + 1;
            const int co = 16 * ((br * G::C4) / 16) + i;
            if (co >= br * G::C4 && co < (br + 1) * G::C4) v = prm.w[br][((co - br * G::C4) * CH + ci) * 9 + t9];
        }
        wp[idx] = v;
    }
}

template <int CH, int UB, int UE>
struct MsFwdUnitLoop {
    template <typename ACC>
    static __device__ __forceinline__ void run(ACC& acc, const float* __restrict__ wpc, const float* __restrict__ patch, int pbase, int lane) {
        typedef MsUnits<CH> G;
        constexpr int u = UB;
        const f32x4 a = *reinterpret_cast<const f32x4*>(wpc + u * 256 + lane * 4);
        constexpr int off = (G::oy(u) * MS_PW + G::ox(u)) * MS_CKP;
        f32x4 b[2];
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) b[pf] = *reinterpret_cast<const f32x4*>(&patch[pbase + pf * MS_PW * MS_CKP + off]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) acc[G::frag(u)][pf] = mfma16(a[j], b[pf][j], acc[G::frag(u)][pf]);
        MsFwdUnitLoop<CH, UB + 1, UE>::run(acc, wpc, patch, pbase, lane);
    }
};
template <int CH, int UE>
struct MsFwdUnitLoop<CH, UE, UE> {
    template <typename ACC>
    static __device__ __forceinline__ void run(ACC&, const float*, const float*, int, int) {}
};

template <int CH>
__global__ __launch_bounds__(256) void ms_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                     int N, int H, int W, int tiles_x, int tiles_y) {
    typedef MsUnits<CH> G;
    constexpr int NFW = G::NFW, NCH = CH / 16, PF = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;  // [PH][PW][CKP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const size_t plane = (size_t)H * W;
    const unsigned m_pw = magic_u32(MS_PW);
    // bias of this lane's output channels (in flight while the patch is staged)
    f32x4 bq[NFW];
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf) bq[wf] = *reinterpret_cast<const f32x4*>(wp + NCH * G::U * 256 + 16 * wf + 4 * g);

    f32x4 acc[NFW][PF];
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragment of (pixel row PF*wave + pf, column i) at tap (0,0): patch row +4, column +4
    const int pbase = ((PF * wave + 4) * MS_PW + i + 4) * MS_CKP + 4 * g;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk) __syncthreads();
        stage_window(x + (size_t)n * plane * CH + 16 * chunk, patch, MS_PH, MS_PW, 4, m_pw, 0x40000000u, ty0 * MS_TH - 4, tx0 * 16 - 4, H, W,
                     CH, 4, MS_CKP, tid);
        __syncthreads();
        MsFwdUnitLoop<CH, 0, G::U>::run(acc, wp + (size_t)chunk * G::U * 256, patch, pbase, lane);
    }
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int gy = ty0 * MS_TH + PF * wave + pf, gx = tx0 * 16 + i;
        if (gy < H && gx < W) {
            float* p = y + (((size_t)n * H + gy) * W + gx) * CH + 4 * g;
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) *reinterpret_cast<f32x4*>(p + 16 * wf) = acc[wf][pf] + bq[wf];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Input gradient of the four branches in one launch: dx = sum over branches of conv_transpose(dy_branch, W_branch).
// GEMM view per 8x16 tile: M = input channels (16 per fragment), N = 16 pixels, K = (tap, output channel of that tap's branch).
// The reduction runs over the channels of dy, and a tap only touches the channel slice of its own branch, so K is walked in
// groups of 4 consecutive dy channels (one MFMA each): the centre tap uses every group, a ring tap the groups of its branch.
// A "slot" = (tap, channel group); slots are compile-time tables per 16-channel chunk of dy, the filter comes from L2 packed
// four slots per 16-byte load, and dx is written once as whole pixels (the per-branch path reads and rewrites dx three times).
// ---------------------------------------------------------------------------------------------------------------------
template <int CH, int C>
struct MsDgChunk {
    static constexpr int C4 = CH / 4;
    static constexpr int FG = (C4 / 4 - 4 * C) < 0 ? 0 : ((C4 / 4 - 4 * C) > 4 ? 4 : (C4 / 4 - 4 * C));  // groups of branch 0 in this chunk
    static constexpr int NRG = 4 - FG, NS = 4 + 8 * NRG, NQ = (NS + 3) / 4;
    __host__ __device__ static constexpr int grp(int s) { return s < 4 ? s : FG + (s - 4) % (NRG > 0 ? NRG : 1); }
    __host__ __device__ static constexpr int br(int s) { return (16 * C + 4 * grp(s)) / C4; }
    __host__ __device__ static constexpr int t9(int s) { return ((s - 4) / (NRG > 0 ? NRG : 1)) < 4 ? (s - 4) / (NRG > 0 ? NRG : 1) : (s - 4) / (NRG > 0 ? NRG : 1) + 1; }
    __host__ __device__ static constexpr int dil(int s) { return 1 << (br(s) - 1); }
    // dx[p] += W[ky][kx]^T dy[p - (ky-1) d, p - (kx-1) d]
    __host__ __device__ static constexpr int oy(int s) { return s < 4 ? 0 : -(t9(s) / 3 - 1) * dil(s); }
    __host__ __device__ static constexpr int ox(int s) { return s < 4 ? 0 : -(t9(s) % 3 - 1) * dil(s); }
};
template <int CH>
__host__ __device__ constexpr int ms_dg_quads_before(int c) {
    return (c > 0 ? MsDgChunk<CH, 0>::NQ : 0) + (c > 1 ? MsDgChunk<CH, 1>::NQ : 0) + (c > 2 ? MsDgChunk<CH, 2>::NQ : 0) +
           (c > 3 ? MsDgChunk<CH, 3>::NQ : 0);
}

// wp[((quads_before(c) * NFW + wf * NQ(c) + quad) * 64 + lane) * 4 + e] = W_br[cj][ci = 16 wf + (lane & 15)][tap] of slot 4 quad + e,
// cj = (16 c + 4 grp + (lane >> 4)) - br * C4
template <int CH, int C>
__device__ void ms_pack_dgrad_chunk(const MsParamPtrs& prm, float* __restrict__ wp, int tid, int nthreads) {
    typedef MsDgChunk<CH, C> K;
    constexpr int NFW = CH / 16, C4 = CH / 4;
    float* base = wp + (size_t)ms_dg_quads_before<CH>(C) * NFW * 256;
    for (int idx = tid; idx < NFW * K::NQ * 256; idx += nthreads) {
        const int e = idx & 3, lane = (idx >> 2) & 63, quad = (idx >> 8) % K::NQ, wf = idx / (K::NQ * 256);
        const int s = 4 * quad + e, i = lane & 15, g = lane >> 4;
        float v = 0.f;
        if (s < K::NS) {
            const int br = K::br(s), cj = 16 * C + 4 * K::grp(s) + g - br * C4, ci = 16 * wf + i;
            if (s < 4) v = br == 0 ? prm.w[0][cj * CH + ci] : prm.w[br][(cj * CH + ci) * 9 + 4];
            else v = prm.w[br][(cj * CH + ci) * 9 + K::t9(s)];
        }
        base[idx] = v;
    }
}
template <int CH>
__global__ void ms_pack_dgrad_kernel(MsParamPtrs prm, float* __restrict__ wp) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    ms_pack_dgrad_chunk<CH, 0>(prm, wp, tid, nt);
    if (CH >= 32) ms_pack_dgrad_chunk<CH, (CH >= 32 ? 1 : 0)>(prm, wp, tid, nt);
    if (CH >= 64) {
        ms_pack_dgrad_chunk<CH, (CH >= 64 ? 2 : 0)>(prm, wp, tid, nt);
        ms_pack_dgrad_chunk<CH, (CH >= 64 ? 3 : 0)>(prm, wp, tid, nt);
    }
}

template <int CH, int C, int QB, int QE>
struct MsDgQuadLoop {
    template <typename ACC>
    static __device__ __forceinline__ void run(ACC& acc, const float* __restrict__ wpc, const float* __restrict__ patch, int pbase, int lane) {
        typedef MsDgChunk<CH, C> K;
        constexpr int NFW = CH / 16, Q = QB;
        f32x4 a[NFW];
#pragma unroll
        for (int wf = 0; wf < NFW; ++wf) a[wf] = *reinterpret_cast<const f32x4*>(wpc + ((wf * K::NQ + Q) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            constexpr int s0 = 4 * Q;
            const int s = s0 + e;
            if (s < K::NS) {
                const int off = (K::oy(s) * MS_PW + K::ox(s)) * MS_CKP + 4 * K::grp(s);
                float b[2];
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) b[pf] = patch[pbase + pf * MS_PW * MS_CKP + off];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                    for (int pf = 0; pf < 2; ++pf) acc[wf][pf] = mfma16(a[wf][e], b[pf], acc[wf][pf]);
            }
        }
        MsDgQuadLoop<CH, C, QB + 1, QE>::run(acc, wpc, patch, pbase, lane);
    }
};
template <int CH, int C, int QE>
struct MsDgQuadLoop<CH, C, QE, QE> {
    template <typename ACC>
    static __device__ __forceinline__ void run(ACC&, const float*, const float*, int, int) {}
};

template <int CH, int C, typename ACC>
__device__ __forceinline__ void ms_dgrad_chunk(ACC& acc, const float* __restrict__ dy_img, const float* __restrict__ wp, float* patch,
                                               int ty0, int tx0, int H, int W, unsigned m_pw, int pbase, int tid, int lane) {
    typedef MsDgChunk<CH, C> K;
    constexpr int NFW = CH / 16;
    if (C) __syncthreads();
    stage_window(dy_img + 16 * C, patch, MS_PH, MS_PW, 4, m_pw, 0x40000000u, ty0 * MS_TH - 4, tx0 * 16 - 4, H, W, CH, 4, MS_CKP, tid);
    __syncthreads();
    MsDgQuadLoop<CH, C, 0, K::NQ>::run(acc, wp + (size_t)ms_dg_quads_before<CH>(C) * NFW * 256, patch, pbase, lane);
}

template <int CH>
__global__ __launch_bounds__(256) void ms_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wp,
                                                       const float* __restrict__ dres, float* __restrict__ dx, int N, int H, int W,
                                                       int tiles_x, int tiles_y) {
    constexpr int NFW = CH / 16, PF = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const size_t plane = (size_t)H * W;
    const unsigned m_pw = magic_u32(MS_PW);
    f32x4 acc[NFW][PF];
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
        for (int pf = 0; pf < PF; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B value of (pixel row PF*wave + pf, column i), k-slot g, at tap (0,0) and channel group 0
    const int pbase = ((PF * wave + 4) * MS_PW + i + 4) * MS_CKP + g;
    const float* dy_img = dy + (size_t)n * plane * CH;
    ms_dgrad_chunk<CH, 0>(acc, dy_img, wp, patch, ty0, tx0, H, W, m_pw, pbase, tid, lane);
    if (CH >= 32) ms_dgrad_chunk<CH, (CH >= 32 ? 1 : 0)>(acc, dy_img, wp, patch, ty0, tx0, H, W, m_pw, pbase, tid, lane);
    if (CH >= 64) {
        ms_dgrad_chunk<CH, (CH >= 64 ? 2 : 0)>(acc, dy_img, wp, patch, ty0, tx0, H, W, m_pw, pbase, tid, lane);
        ms_dgrad_chunk<CH, (CH >= 64 ? 3 : 0)>(acc, dy_img, wp, patch, ty0, tx0, H, W, m_pw, pbase, tid, lane);
    }
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        const int gy = ty0 * MS_TH + PF * wave + pf, gx = tx0 * 16 + i;
        if (gy < H && gx < W) {
            const size_t o = (((size_t)n * H + gy) * W + gx) * CH + 4 * g;
#pragma unroll
            for (int wf = 0; wf < NFW; ++wf) {
                f32x4 v = acc[wf][pf];
                if (dres) v += *reinterpret_cast<const f32x4*>(dres + o + 16 * wf);  // gradient of the block's residual path
                *reinterpret_cast<f32x4*>(dx + o + 16 * wf) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tap-packed weight gradient (CH = 16 and 32).  In the unit kernel above a ring tap uses only the CH/4 columns of its branch:
// at CH = 16 three quarters of every ring MFMA are wasted.  Writing
//     dW[tap][ci][co] = sum_p x[p + tap][ci] dy[p][co] = sum_p' x[p'][ci] dy[p' - tap][co]
// moves the tap shift from x to dy, and the MFMA's B operand is a per-lane gather: column n = (tap slot s, co) reads
// dy[p' - tap_s][co of the branch].  One MFMA then covers 16/(CH/4) taps of a branch: 7 MFMAs per pixel group instead of 25
// at CH = 16, 14 instead of 26 at CH = 32.  x needs no halo here; the dy patch carries it (halo 4).
// Units: u < NFH: centre tap, dy fragment u (all branches at tap 0);  then branch j = 1..3, TPU = 16 / C4 ring taps each.
// ---------------------------------------------------------------------------------------------------------------------
template <int CH>
struct MsPk {
    static constexpr int NFH = CH / 16, C4 = CH / 4, TPU = 16 / C4, UPB = 8 / TPU, U = NFH + 3 * UPB, UW = (U + 3) / 4;
    static constexpr int NG = CH / 16, PSTRIDE = NG * U * 256 + CH, LDY = CH + 4;
};

template <int CH>
__global__ __launch_bounds__(256) void wgrad_msp_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ partial, int N, int H, int W, int tiles_x, int tiles_y,
                                                        int ntiles) {
    typedef MsPk<CH> G;
    constexpr int UW = G::UW, U = G::U, LDY = G::LDY, MG = G::NG, LDX = CH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                          // [TH*16][LDX]   ALL input channels (one workgroup owns every 16-channel chunk: the
                                               //                dy patch, the expensive operand, is staged once, not once per chunk)
    float* dyp = smem + MS_TH * 16 * LDX;      // [PH][PW][LDY]  all CH output-gradient channels, halo 4

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;

    f32x4 acc[UW][MG];
    int boff[UW];  // per-lane offset of this unit's B element inside the dy patch, relative to the pixel (r + 4, c + 4)
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        const int u = min(wave + 4 * k, U - 1);
        int off;
        if (u < G::NFH) {
            off = 16 * u + i;  // centre tap: dy channel 16 u + i
        } else {
            const int j = 1 + (u - G::NFH) / G::UPB, m = (u - G::NFH) % G::UPB;  // branch, unit inside the ring
            const int sidx = i / G::C4, co = i % G::C4, i8 = m * G::TPU + sidx, t9 = i8 < 4 ? i8 : i8 + 1, d = 1 << (j - 1);
            const int oy = (t9 / 3 - 1) * d, ox = (t9 % 3 - 1) * d;
            off = (-oy * MS_PW - ox) * LDY + j * G::C4 + co;  // dy[p' - tap][branch channel]
        }
        boff[k] = off;
#pragma unroll
        for (int mf = 0; mf < MG; ++mf) acc[k][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nu = (U - wave + 3) / 4;

    float bsum = 0.f;
    const unsigned m_pw = magic_u32(MS_PW), m_nq = magic_u32(CH / 4);
    const size_t plane = (size_t)H * W;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        __syncthreads();
        stage_window(x + (size_t)n * plane * CH, xs, MS_TH, 16, CH / 4, 0x10000000u, m_nq, ty0 * MS_TH, tx0 * 16, H, W, CH, CH / 4, LDX, tid);
        stage_window(dy + (size_t)n * plane * CH, dyp, MS_PH, MS_PW, CH / 4, m_pw, m_nq, ty0 * MS_TH - 4, tx0 * 16 - 4, H, W, CH, CH / 4, LDY,
                     tid);
        __syncthreads();
        if (tid < CH) {
            for (int r = 0; r < MS_TH; ++r)
#pragma unroll 8
                for (int c = 0; c < 16; ++c) bsum += dyp[((r + 4) * MS_PW + c + 4) * LDY + tid];
        }
#pragma unroll 1
        for (int r = 0; r < MS_TH; ++r) {
#pragma unroll 2
            for (int xs4 = 0; xs4 < 4; ++xs4) {
                const int c = 4 * xs4 + g;  // this lane's k-slot pixel column
                float af[MG], bf[UW];
#pragma unroll
                for (int mf = 0; mf < MG; ++mf) af[mf] = xs[(r * 16 + c) * LDX + 16 * mf + i];
                const int bbase = ((r + 4) * MS_PW + c + 4) * LDY;
#pragma unroll
                for (int k = 0; k < UW; ++k) bf[k] = dyp[bbase + boff[k]];
#pragma unroll
                for (int k = 0; k < UW; ++k)
#pragma unroll
                    for (int mf = 0; mf < MG; ++mf) acc[k][mf] = mfma16(af[mf], bf[k], acc[k][mf]);
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * G::PSTRIDE;
    if (tid < CH) out[G::NG * U * 256 + tid] = bsum;
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        if (k >= nu) continue;
        const int u = wave + 4 * k;
#pragma unroll
        for (int mf = 0; mf < MG; ++mf) {
            float* o = out + ((size_t)mf * U + u) * 256;  // [chunk mf][unit][m = input channel 4g + e][n = i]
#pragma unroll
            for (int e = 0; e < 4; ++e) o[(4 * g + e) * 16 + i] = acc[k][mf][e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward at CH = 16 on the 4x4x1 MFMA (v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products per instruction).
// A 16-row MFMA tile wastes three quarters of every ring tap here (a branch has only 4 output channels); with 4x4 blocks
// the rows ARE the branch's 4 output channels and the columns 4 pixels: lane (block b, j) supplies the weight of output
// channel j and the input value of pixel (b, j), and ends up holding its own pixel's 4 output channels (one per accumulator
// register) -- exactly an NHWC store.  One wave = 64 pixels; a tap and input channel cost one 8-cycle instruction for all
// four rows x 64 pixels, i.e. 3.6x fewer MFMA cycles than ms_fwd_kernel<16>.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int M4_TH = 16, M4_PH = M4_TH + 8;

// one (branch, tap, channel quad of the current 16-channel chunk): RS row sets (4 output channels each) share the B value
template <int CH, int JB, int T9, int Q>
struct Ms4Step {
    static constexpr int C4 = CH / 4, RS = C4 / 4;
    static __device__ __forceinline__ void run(f32x4 (&acc)[4][RS], const float* __restrict__ wl, const float* __restrict__ patch, int pbase,
                                               int wbase) {
        constexpr int d = JB == 0 ? 0 : (1 << (JB - 1));
        constexpr int oy = JB == 0 ? 0 : (T9 / 3 - 1) * d, ox = JB == 0 ? 0 : (T9 % 3 - 1) * d;
        constexpr int t = JB == 0 ? 0 : 1 + 9 * (JB - 1) + T9;  // tap slot in the LDS filter
        const f32x4 b = *reinterpret_cast<const f32x4*>(&patch[pbase + (oy * MS_PW + ox) * MS_CKP + 4 * Q]);
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&wl[wbase + (t * C4 + 4 * rs) * CH + 4 * Q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[JB][rs] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], b[e], acc[JB][rs], 0, 0, 0);
        }
    }
};

template <int CH, int JB>
struct Ms4Branch {
    static constexpr int RS = CH / 16;
    static __device__ __forceinline__ void run(f32x4 (&acc)[4][RS], const float* wl, const float* patch, int pbase, int wbase) {
#define M4_TAP(T9)                                                  \
    Ms4Step<CH, JB, T9, 0>::run(acc, wl, patch, pbase, wbase);      \
    Ms4Step<CH, JB, T9, 1>::run(acc, wl, patch, pbase, wbase);      \
    Ms4Step<CH, JB, T9, 2>::run(acc, wl, patch, pbase, wbase);      \
    Ms4Step<CH, JB, T9, 3>::run(acc, wl, patch, pbase, wbase);
        if (JB == 0) {
            M4_TAP(0)
        } else {
            M4_TAP(0) M4_TAP(1) M4_TAP(2) M4_TAP(3) M4_TAP(4) M4_TAP(5) M4_TAP(6) M4_TAP(7) M4_TAP(8)
        }
#undef M4_TAP
    }
};

template <int CH>
__global__ __launch_bounds__(256) void ms_fwd4_kernel(const float* __restrict__ x, MsParamPtrs prm, float* __restrict__ y, int N, int H,
                                                      int W, int tiles_x, int tiles_y) {
    constexpr int C4 = CH / 4, RS = C4 / 4, NCHK = CH / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                  // [M4_PH][PW][CKP]   one 16-channel chunk of x at a time
    float* wl = smem + M4_PH * MS_PW * MS_CKP;            // [28 taps][C4 co][CH ci]   the whole filter, staged once
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const size_t plane = (size_t)H * W;
    // ---- filter: wl[(t * C4 + co) * CH + ci] straight from the four PyTorch-layout tensors ----------------------------------
    for (int e = tid; e < 28 * C4 * CH; e += 256) {
        const int ci = e % CH, co = (e / CH) % C4, t = e / (CH * C4);
        float v;
        if (t == 0) v = prm.w[0][co * CH + ci];
        else v = prm.w[1 + (t - 1) / 9][(co * CH + ci) * 9 + (t - 1) % 9];
        wl[e] = v;
    }
    // lane (block b, j): pixel row 4*wave + b/4, column 4*(b%4) + j ; supplies the weights of output channel 4*rs + j
    const int b = lane >> 2, j = lane & 3;
    const int prow = 4 * wave + (b >> 2), pcol = 4 * (b & 3) + j;
    const int pbase = ((prow + 4) * MS_PW + pcol + 4) * MS_CKP;
    f32x4 acc[4][RS];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) acc[jb][rs] = *reinterpret_cast<const f32x4*>(prm.b[jb] + 4 * rs);  // bias: register r = channel 4 rs + r
    const unsigned m_pw = magic_u32(MS_PW);
#pragma unroll
    for (int chunk = 0; chunk < NCHK; ++chunk) {
        if (chunk) __syncthreads();
        stage_window(x + (size_t)n * plane * CH + 16 * chunk, patch, M4_PH, MS_PW, 4, m_pw, 0x40000000u, ty0 * M4_TH - 4, tx0 * 16 - 4, H, W,
                     CH, 4, MS_CKP, tid);
        __syncthreads();
        const int wbase = j * CH + 16 * chunk;
        Ms4Branch<CH, 0>::run(acc, wl, patch, pbase, wbase);
        Ms4Branch<CH, 1>::run(acc, wl, patch, pbase, wbase);
        Ms4Branch<CH, 2>::run(acc, wl, patch, pbase, wbase);
        Ms4Branch<CH, 3>::run(acc, wl, patch, pbase, wbase);
    }
    const int gy = ty0 * M4_TH + prow, gx = tx0 * 16 + pcol;
    if (gy < H && gx < W) {
        float* p = y + (((size_t)n * H + gy) * W + gx) * CH;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int rs = 0; rs < RS; ++rs) *reinterpret_cast<f32x4*>(p + C4 * jb + 4 * rs) = acc[jb][rs];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 4x4x1 forward with ONE BRANCH PER WAVE.  ms_fwd4_kernel gives every wave 64 pixels and all four branches: per
// (tap, channel quad) it reads one 16-byte patch operand and one 16-byte filter operand per row set from LDS for four 8-cycle
// MFMAs -- twice (CH = 16) the LDS bandwidth a CU has per MFMA cycle, which is what bounds it (profiles/r03: LDS 50 % busy, matrix
// pipe 32 %).  Here a wave owns one branch for the whole 16 x 16 tile (four 64-pixel sets): the filter operand of a (tap, quad) is
// read ONCE and used against four patch operands, 5 (CH = 16) or 6 (CH = 32) LDS reads per 16 / 32 MFMAs instead of 8 / 12, and
// at CH = 32 the 4-row blocks carry no dead rows (the 16-row kernel multiplies 8 zero rows on every ring tap).  The 1x1 branch
// is a ninth of a 3x3 branch's work, so the 16 (branch, pixel set) units of a tile are dealt 28 tap-sets to each wave (see the kernel).
// ---------------------------------------------------------------------------------------------------------------------
// LDS layout.  ds_read_b128 serves a wave in four 16-lane groups -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
// + 32 -- over 64 four-byte banks.  With 20 floats between pixel columns, 16 consecutive columns of ONE patch row fall on 16 different
// 4-bank slots; two different rows collide unless the row stride is 0 mod 64 floats (the dense 480 is 32 mod 64: with lanes mapped
// (row, column) = (lane >> 4, lane & 15) every read of every multi-scale kernel is a two-way conflict, SQ_LDS_BANK_CONFLICT = half of
// SQ_LDS_IDX_ACTIVE).  The 4x4x1 MFMA does not care which pixel a lane holds (lane (b, j) only has to supply output channel j's
// weights as its A operand), so the lanes of one hardware group take the 16 columns of one row: conflict-free on the dense layout,
// which keeps the patch at 46 KB (three workgroups per CU at CH = 16; padded rows would be 49 KB: two).  The filter operand has four
// distinct addresses per group (output channel j = lane & 3), LDW floats apart so that they fall on different banks.
constexpr int M4B_ROW = MS_PW * MS_CKP;
__device__ __forceinline__ void m4b_lane_pixel(int lane, int& row, int& col) {  // row 0..3 of the 64-pixel set, column 0..15
    const int l5 = lane & 31;
    int grp, pos;
    if (l5 < 4) { grp = 0; pos = l5; }
    else if (l5 < 12) { grp = 1; pos = l5 - 4; }
    else if (l5 < 16) { grp = 0; pos = l5 - 8; }
    else if (l5 < 20) { grp = 1; pos = l5 - 8; }
    else if (l5 < 28) { grp = 0; pos = l5 - 12; }
    else { grp = 1; pos = l5 - 16; }
    row = 2 * (lane >> 5) + grp;
    col = pos;
}

template <int CH, int JB, int NS>
struct Ms4bBranch {  // branch JB on NS consecutive 64-pixel sets (acc[0 .. NS-1]); pbase addresses the first set
    static constexpr int C4 = CH / 4, RS = C4 / 4, ROW = M4B_ROW, SET = 4 * ROW, LDW = (CH == 16 ? CH : CH + 4);
    static constexpr int NSTEP = (JB == 0 ? 1 : 9) * 4;  // (tap, channel quad) steps of this branch per 16-channel chunk
    struct Ops {
        f32x4 a[RS], b[NS];
    };
    template <int S>
    static __device__ __forceinline__ void load(Ops& o, const float* __restrict__ wl, const float* __restrict__ patch, int pbase, int wbase) {
        constexpr int T9 = S / 4, Q = S % 4;
        constexpr int d = JB == 0 ? 0 : (1 << (JB - 1));
        constexpr int oy = JB == 0 ? 0 : (T9 / 3 - 1) * d, ox = JB == 0 ? 0 : (T9 % 3 - 1) * d;
        constexpr int t = JB == 0 ? 0 : 1 + 9 * (JB - 1) + T9;  // tap slot in the LDS filter
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) o.a[rs] = *reinterpret_cast<const f32x4*>(&wl[wbase + (t * C4 + 4 * rs) * LDW + 4 * Q]);
#pragma unroll
        for (int s = 0; s < NS; ++s) o.b[s] = *reinterpret_cast<const f32x4*>(&patch[pbase + s * SET + oy * ROW + ox * MS_CKP + 4 * Q]);
    }
    // one accumulator per (set, row set), products added in the order tap, channel: the summation order of ms_fwd4_kernel, bit for bit
    // (splitting the chain over two accumulators to shorten the dependency bought nothing: the 8-cycle products of the other sets and
    // the operand reads fill the gaps)
    static __device__ __forceinline__ void mma(f32x4 (*acc)[RS], const Ops& o) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int rs = 0; rs < RS; ++rs) acc[s][rs] = __builtin_amdgcn_mfma_f32_4x4x1f32(o.a[rs][e], o.b[s][e], acc[s][rs], 0, 0, 0);
    }
    // step S multiplies the operands in `cur` while step S + 1's are on their way into `nxt`; the scheduling barriers keep the reads
    // in front of this step's products and the compiler from hoisting every later read as well (it otherwise fills 240 registers)
    template <int S>
    static __device__ __forceinline__ void steps(f32x4 (*acc)[RS], Ops& cur, Ops& nxt, const float* wl, const float* patch, int pbase, int wbase) {
        if constexpr (S + 1 < NSTEP) load<S + 1>(nxt, wl, patch, pbase, wbase);
        __builtin_amdgcn_sched_barrier(0);
        mma(acc, cur);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S + 1 < NSTEP) steps<S + 1>(acc, nxt, cur, wl, patch, pbase, wbase);
    }
    static __device__ __forceinline__ void run(f32x4 (*acc)[RS], const float* wl, const float* patch, int pbase, int wbase) {
        Ops o0, o1;
        load<0>(o0, wl, patch, pbase, wbase);
        steps<0>(acc, o0, o1, wl, patch, pbase, wbase);
    }
};

// the filter in the LDS layout of ms_fwd4b_kernel, [28 taps][C4 co][CH + 4] (pad zero), then the CH concatenated biases: packed once
// per weight update (the caller's pack cache), so that a workgroup stages it with a few 16-byte loads instead of gathering
// 28 * C4 * CH scalars from the four PyTorch-layout tensors per tile
template <int CH>
__global__ void ms_pack_fwd4b_kernel(MsParamPtrs prm, float* __restrict__ wp) {
    constexpr int C4 = CH / 4, LDW = (CH == 16 ? CH : CH + 4), TOTAL = 28 * C4 * LDW;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < TOTAL + CH; idx += gridDim.x * blockDim.x) {
        if (idx >= TOTAL) {
            const int co = idx - TOTAL;
            wp[idx] = prm.b[co / C4][co % C4];
            continue;
        }
        const int ci = idx % LDW, co = (idx / LDW) % C4, t = idx / (LDW * C4);
        float v = 0.f;
        if (ci < CH) v = t == 0 ? prm.w[0][co * CH + ci] : prm.w[1 + (t - 1) / 9][(co * CH + ci) * 9 + (t - 1) % 9];
        wp[idx] = v;
    }
}

// Persistent: a workgroup stages the filter once and walks tiles blockIdx.x, + gridDim.x, ...; the NEXT (tile, chunk)'s patch -- nine
// 16-byte loads per thread -- is in flight in registers while the current one is multiplied, so the only exposed memory latency is
// the first patch's.
template <int CH>
__global__ __launch_bounds__(256, CH == 16 ? 3 : 2) void ms_fwd4b_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y,
                                                          int N, int H, int W, int tiles_x, int tiles_y, int ntiles) {
    constexpr int C4 = CH / 4, RS = C4 / 4, NCHK = CH / 16, NS = 4, LDW = (CH == 16 ? CH : CH + 4), NPF = M4_PH * MS_PW * 4 / 256;
    static_assert(M4_PH * MS_PW * 4 == NPF * 256, "patch quads divide evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                  // [M4_PH][PW][CKP]   one 16-channel chunk of x at a time
    float* wl = smem + M4_PH * M4B_ROW;                   // [28 taps][C4 co][LDW]   the whole filter, staged once
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t plane = (size_t)H * W;
    for (int e = tid; e < 28 * C4 * LDW / 4; e += 256) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(wp)[e];
    // this thread's NPF patch quads: (row, column, channel quad) -> LDS offset, offset inside the image, row / column for the bounds
    int pdst[NPF], prel[NPF], prc[NPF];
    const unsigned m_pw = magic_u32(MS_PW);
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int e = 256 * k + tid, pix = e >> 2, q = e & 3;
        const int r = (int)__umulhi((unsigned)pix, m_pw), c = pix - r * MS_PW;
        pdst[k] = r * M4B_ROW + c * MS_CKP + 4 * q;
        prel[k] = (r * W + c) * CH + 4 * q;
        prc[k] = (r << 16) | c;
    }
    f32x4 pv[NPF];
    unsigned pok = 0;
    auto fetch = [&](int tile, int chunk) {  // issue the loads of (tile, chunk)'s patch
        const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int y0 = ty0 * M4_TH - 4, x0 = tx0 * 16 - 4;
        const float* img = x + (size_t)n * plane * CH + 16 * chunk + ((ptrdiff_t)y0 * W + x0) * CH;
        pok = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int iy = y0 + (prc[k] >> 16), ix = x0 + (prc[k] & 0xffff);
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            pv[k] = *reinterpret_cast<const f32x4*>(ok ? img + prel[k] : x);
            pok |= (ok ? 1u : 0u) << k;
        }
    };
    // lane (block b, j = lane & 3) supplies the weights of output channel 4*rs + j of its branch; its pixel: m4b_lane_pixel
    const int j = lane & 3;
    int prow, pcol;
    m4b_lane_pixel(lane, prow, pcol);
    const int pbase = (prow + 4) * M4B_ROW + (pcol + 4) * MS_CKP;
    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile, 0);
    f32x4 acc[NS][RS];
    for (int it = 0; tile < ntiles; ++it) {
        // Work of a tile = (branch, 64-pixel set) units of 9, 9, 9 and 1 tap: roles 0..2 take a 3x3 branch on sets 0..2 plus the 1x1
        // branch on one set; role 3 takes set 3 of all four branches.  28 tap-sets each (one branch per wave would be 36 / 36 / 36 / 4);
        // the role rotates with the tile so that the wave re-reading the filter per set (role 3) is not always on the same SIMD.
        const int role = __builtin_amdgcn_readfirstlane((wave + blockIdx.x + it) & 3);
        // acc[k]'s (branch, set): roles 0..2: (role + 1, k) for k < 3, (0, role) for k = 3; role 3: (k + 1, 3) for k < 3, (0, 3)
#pragma unroll 1
        for (int chunk = 0; chunk < NCHK; ++chunk) {
            __syncthreads();  // the previous chunk's products are done with the patch
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                *reinterpret_cast<f32x4*>(&patch[pdst[k]]) = ((pok >> k) & 1) ? pv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            if (chunk + 1 < NCHK) fetch(tile, chunk + 1);
            else if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x, 0);
            if (chunk == 0) {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int br = k == 3 ? 0 : (role == 3 ? k + 1 : role + 1);
                    const float* bias = wp + 28 * C4 * LDW + C4 * br;
#pragma unroll
                    for (int rs = 0; rs < RS; ++rs) acc[k][rs] = *reinterpret_cast<const f32x4*>(bias + 4 * rs);  // register r = channel 4 rs + r
                }
            }
            const int wbase = j * LDW + 16 * chunk;
            constexpr int SET = 4 * M4B_ROW;
            if (role == 0) {
                Ms4bBranch<CH, 1, 3>::run(acc, wl, patch, pbase, wbase);
                Ms4bBranch<CH, 0, 1>::run(acc + 3, wl, patch, pbase, wbase);
            } else if (role == 1) {
                Ms4bBranch<CH, 2, 3>::run(acc, wl, patch, pbase, wbase);
                Ms4bBranch<CH, 0, 1>::run(acc + 3, wl, patch, pbase + SET, wbase);
            } else if (role == 2) {
                Ms4bBranch<CH, 3, 3>::run(acc, wl, patch, pbase, wbase);
                Ms4bBranch<CH, 0, 1>::run(acc + 3, wl, patch, pbase + 2 * SET, wbase);
            } else {
                Ms4bBranch<CH, 1, 1>::run(acc, wl, patch, pbase + 3 * SET, wbase);
                Ms4bBranch<CH, 2, 1>::run(acc + 1, wl, patch, pbase + 3 * SET, wbase);
                Ms4bBranch<CH, 3, 1>::run(acc + 2, wl, patch, pbase + 3 * SET, wbase);
                Ms4bBranch<CH, 0, 1>::run(acc + 3, wl, patch, pbase + 3 * SET, wbase);
            }
        }
        const int tx0 = tile % tiles_x, ty0 = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int gx = tx0 * 16 + pcol;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int br = k == 3 ? 0 : (role == 3 ? k + 1 : role + 1), set = role == 3 ? 3 : (k == 3 ? role : k);
            const int gy = ty0 * M4_TH + 4 * set + prow;
            if (gy < H && gx < W) {
                float* p = y + (((size_t)n * H + gy) * W + gx) * CH + C4 * br;
#pragma unroll
                for (int rs = 0; rs < RS; ++rs) *reinterpret_cast<f32x4*>(p + 4 * rs) = acc[k][rs];
            }
        }
        tile += gridDim.x;
    }
}

struct MsGradPtrs {
    float* dw[4];
    float* db[4];
};

// fixed-order sum over the S partial slabs, then scatter: slab index -> (branch, out channel, in channel, tap) of the four
// PyTorch-layout gradients  dw1 (C4, CH, 1, 1), dw2..4 (C4, CH, 3, 3)  and  db1..4 (C4)
template <int CH>
__global__ __launch_bounds__(256) void wgrad_ms_reduce_kernel(const float* __restrict__ partial, MsGradPtrs out, int S, int accumulate) {
    constexpr int NFH = CH / 16, C4 = CH / 4, U = NFH + 24, NG = CH / 16, NACC = NG * U * 256, PSTRIDE = NACC + CH;
    __shared__ float sh[16][17];
    const int e = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e;
    float sum = 0.f;
    if (idx < PSTRIDE) {  // eight loads in flight per trip; the additions keep the order sp = row, row + 16, ...
        int sp = row;
        for (; sp + 7 * 16 < S; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(sp + 16 * k) * PSTRIDE + idx];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; sp < S; sp += 16) sum += partial[(size_t)sp * PSTRIDE + idx];
    }
    sh[row][e] = sum;
    __syncthreads();
    if (row != 0 || idx >= PSTRIDE) return;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += sh[k][e];
    if (idx >= NACC) {  // bias gradient of output channel co
        const int co = idx - NACC;
        float* o = &out.db[co / C4][co % C4];
        *o = accumulate ? *o + r : r;
        return;
    }
    const int n = idx & 15, m = (idx >> 4) & 15, u = (idx >> 8) % U, gch = idx / (U * 256);
    const int ci = 16 * gch + m;
    if (u < NFH) {  // centre tap, fragment u
        const int co = 16 * u + n, j = co / C4, cj = co % C4;
        float* o = j == 0 ? &out.dw[0][cj * CH + ci] : &out.dw[j][(cj * CH + ci) * 9 + 4];
        *o = accumulate ? *o + r : r;
    } else {
        const int rr = (u - NFH) >> 3, i8 = (u - NFH) & 7, t9 = i8 < 4 ? i8 : i8 + 1, j = rr + 1;
        const int co = 16 * ((j * C4) / 16) + n;
        if (co >= j * C4 && co < (j + 1) * C4) {
            float* o = &out.dw[j][((co - j * C4) * CH + ci) * 9 + t9];
            *o = accumulate ? *o + r : r;
        }
    }
}

template <int CH>
__global__ __launch_bounds__(256) void wgrad_msp_reduce_kernel(const float* __restrict__ partial, MsGradPtrs out, int S, int accumulate) {
    typedef MsPk<CH> G;
    constexpr int NACC = G::NG * G::U * 256, PSTRIDE = G::PSTRIDE, C4 = G::C4;
    __shared__ float sh[16][17];
    const int e = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + e;
    float sum = 0.f;
    if (idx < PSTRIDE) {  // eight loads in flight per trip; the additions keep the order sp = row, row + 16, ...
        int sp = row;
        for (; sp + 7 * 16 < S; sp += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partial[(size_t)(sp + 16 * k) * PSTRIDE + idx];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; sp < S; sp += 16) sum += partial[(size_t)sp * PSTRIDE + idx];
    }
    sh[row][e] = sum;
    __syncthreads();
    if (row != 0 || idx >= PSTRIDE) return;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) r += sh[k][e];
    float* o;
    if (idx >= NACC) {
        const int co = idx - NACC;
        o = &out.db[co / C4][co % C4];
    } else {
        const int n = idx & 15, m = (idx >> 4) & 15, u = (idx >> 8) % G::U, gch = idx / (G::U * 256);
        const int ci = 16 * gch + m;
        if (u < G::NFH) {
            const int co = 16 * u + n, j = co / C4, cj = co % C4;
            o = j == 0 ? &out.dw[0][cj * CH + ci] : &out.dw[j][(cj * CH + ci) * 9 + 4];
        } else {
            const int j = 1 + (u - G::NFH) / G::UPB, mm = (u - G::NFH) % G::UPB;
            const int sidx = n / C4, co = n % C4, i8 = mm * G::TPU + sidx, t9 = i8 < 4 ? i8 : i8 + 1;
            o = &out.dw[j][(co * CH + ci) * 9 + t9];
        }
    }
    *o = accumulate ? *o + r : r;
}

template <int CH>
static bool ms_wgrad_packed() {
    const char* e = env_get(ENV_MS_WGRAD_PACKED);
    return CH <= 32 && !(e && e[0] == '0');
}

template <int CH>
static size_t ms_wgrad_plan(int N, int H, int W, int& S, int& tiles_x, int& tiles_y, int& ntiles) {
    constexpr int NFH = CH / 16, U = NFH + 24, NG = CH / 16, PSTRIDE = NG * U * 256 + CH;  // the unit layout is the larger one
    tiles_x = cdiv(W, 16);
    tiles_y = cdiv(H, MS_TH);
    ntiles = N * tiles_x * tiles_y;
    S = 768 / NG;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    size_t need = (size_t)S * PSTRIDE;
    if (CH <= 32) {  // the tap-packed kernel: S * NG workgroups, smaller slabs
        const size_t sp = (size_t)S * NG > (size_t)ntiles ? (size_t)ntiles : (size_t)S * NG;
        const size_t np = sp * MsPk<(CH <= 32 ? CH : 16)>::PSTRIDE;
        if (np > need) need = np;
    }
    return need * sizeof(float);
}

template <int CH>
static int launch_ms_wgrad(const float* x, const float* dy, const MsGradPtrs& out, int accumulate, int N, int H, int W, void* ws,
                           size_t ws_bytes, hipStream_t st) {
    constexpr int NFH = CH / 16, U = NFH + 24, NG = CH / 16, PSTRIDE = NG * U * 256 + CH;
    int S, tiles_x, tiles_y, ntiles;
    const size_t need = ms_wgrad_plan<CH>(N, H, W, S, tiles_x, tiles_y, ntiles);
    if (!ws || ws_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "msblock_wgrad: workspace too small");
    const size_t lds = (size_t)(MS_PH * MS_PW * MS_CKP + MS_TH * 16 * (CH + 4)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ms_kernel<CH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(wgrad_ms)");
        attr_set = true;
    }
    if (ms_wgrad_packed<CH>()) {
        typedef MsPk<CH> P;
        const size_t ldsp = (size_t)(MS_TH * 16 * (CH + 4) + MS_PH * MS_PW * P::LDY) * sizeof(float);
        static bool attr_set_p = false;
        if (!attr_set_p) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_msp_kernel<CH>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(wgrad_msp)");
            attr_set_p = true;
        }
        const int Sp = S * NG > ntiles ? ntiles : S * NG;  // one workgroup covers every input-channel chunk: keep the workgroup count
        MSTG_LAUNCH((wgrad_msp_kernel<CH>), dim3(Sp, 1, 1), dim3(256), ldsp, st, x, dy, (float*)ws, N, H, W, tiles_x, tiles_y, ntiles);
        MSTG_CHECK_LAUNCH("wgrad_msp_kernel");
        MSTG_LAUNCH((wgrad_msp_reduce_kernel<CH>), dim3(cdiv(P::PSTRIDE, 16)), dim3(256), 0, st, (const float*)ws, out, Sp, accumulate);
        MSTG_CHECK_LAUNCH("wgrad_msp_reduce_kernel");
        return MSTG_OK;
    }
    MSTG_LAUNCH((wgrad_ms_kernel<CH>), dim3(S, NG, 1), dim3(256), lds, st, x, dy, (float*)ws, N, H, W, tiles_x, tiles_y, ntiles);
    MSTG_CHECK_LAUNCH("wgrad_ms_kernel");
    MSTG_LAUNCH((wgrad_ms_reduce_kernel<CH>), dim3(cdiv(PSTRIDE, 16)), dim3(256), 0, st, (const float*)ws, out, S, accumulate);
    MSTG_CHECK_LAUNCH("wgrad_ms_reduce_kernel");
    return MSTG_OK;
}

static int ms_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int CH>
static int launch_ms_fwd(const float* x, const MsParamPtrs& prm, float* y, int N, int H, int W, void* ws, size_t ws_bytes, hipStream_t st) {
    typedef MsUnits<CH> G;
    {
        const char* e = env_get(ENV_MS_FWD4);
        bool aligned = true;
        for (int k = 0; k < 4; ++k) aligned = aligned && (reinterpret_cast<uintptr_t>(prm.b[k]) & 15) == 0;
        // MSTG_MS_FWD4: unset / 3 = the branch-per-wave 4x4x1 kernel at CH = 16 and 32; 1 = the pixel-per-wave 4x4x1 kernel at CH = 16
        // (round 2); 2 = that kernel at CH = 32 too; 0 = the 16-row-tile kernel everywhere
        const bool use4b = CH <= 32 && (!e || e[0] == '3');
        const bool use4 = !use4b && ((CH == 16 && !(e && e[0] == '0')) || (CH == 32 && e && e[0] == '2'));
        if ((use4 || use4b) && aligned && H >= 16) {
            constexpr int C4K = (CH <= 32 ? CH : 16);
            const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, M4_TH);
            constexpr int LDW4B = C4K == 16 ? C4K : C4K + 4;
            const size_t lds = use4b ? (size_t)(M4_PH * M4B_ROW + 28 * (C4K / 4) * LDW4B) * sizeof(float)
                                     : (size_t)(M4_PH * MS_PW * MS_CKP + 28 * (C4K / 4) * C4K) * sizeof(float);
            static bool attr4 = false;
            if (!attr4) {
                hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(&ms_fwd4_kernel<C4K>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (er == hipSuccess)
                    er = hipFuncSetAttribute(reinterpret_cast<const void*>(&ms_fwd4b_kernel<C4K>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024);
                if (er != hipSuccess) return fail_launch(er, "hipFuncSetAttribute(ms_fwd4)");
                attr4 = true;
            }
            if (use4b) {
                constexpr int PACKED = 28 * (C4K / 4) * LDW4B + C4K;
                if (!ws || ws_bytes < (size_t)PACKED * sizeof(float)) return fail_arg(MSTG_E_WORKSPACE, "msblock_fwd: workspace too small");
                MSTG_PACK_LAUNCH((ms_pack_fwd4b_kernel<C4K>), dim3(cdiv(PACKED, 256)), dim3(256), 0, st, prm, (float*)ws);
                MSTG_CHECK_LAUNCH("ms_pack_fwd4b_kernel");
                const int ntiles = N * tiles_x * tiles_y, slots = (C4K == 16 ? 3 : 2) * ms_cus();  // workgroups a CU's LDS holds
                MSTG_LAUNCH((ms_fwd4b_kernel<C4K>), dim3(ntiles < slots ? ntiles : slots), dim3(256), lds, st, x, (const float*)ws, y, N, H, W,
                            tiles_x, tiles_y, ntiles);
            } else MSTG_LAUNCH((ms_fwd4_kernel<C4K>), dim3(N * tiles_x * tiles_y), dim3(256), lds, st, x, prm, y, N, H, W, tiles_x, tiles_y);
            MSTG_CHECK_LAUNCH("ms_fwd4_kernel");
            return MSTG_OK;
        }
    }
    constexpr int NCH = CH / 16;
    const size_t need = (size_t)(NCH * G::U * 256 + CH) * sizeof(float);
    if (!ws || ws_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "msblock_fwd: workspace too small");
    float* wp = (float*)ws;
    MSTG_PACK_LAUNCH((ms_pack_fwd_kernel<CH>), dim3(cdiv(NCH * G::U * 256 + CH, 256)), dim3(256), 0, st, prm, wp);
    MSTG_CHECK_LAUNCH("ms_pack_fwd_kernel");
    const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, MS_TH);
    const size_t lds = (size_t)(MS_PH * MS_PW * MS_CKP) * sizeof(float);
    MSTG_LAUNCH((ms_fwd_kernel<CH>), dim3(N * tiles_x * tiles_y), dim3(256), lds, st, x, (const float*)wp, y, N, H, W, tiles_x,
                       tiles_y);
    MSTG_CHECK_LAUNCH("ms_fwd_kernel");
    return MSTG_OK;
}

template <int CH>
static size_t ms_dgrad_ws_floats() {
    return (size_t)ms_dg_quads_before<CH>(CH / 16) * (CH / 16) * 256;
}

template <int CH>
static int launch_ms_dgrad(const float* dy, const MsParamPtrs& prm, const float* dres, float* dx, int N, int H, int W, void* ws,
                           size_t ws_bytes, hipStream_t st) {
    const size_t need = ms_dgrad_ws_floats<CH>() * sizeof(float);
    if (!ws || ws_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "msblock_dgrad: workspace too small");
    float* wp = (float*)ws;
    MSTG_PACK_LAUNCH((ms_pack_dgrad_kernel<CH>), dim3(cdiv((int)ms_dgrad_ws_floats<CH>(), 1024)), dim3(256), 0, st, prm, wp);
    MSTG_CHECK_LAUNCH("ms_pack_dgrad_kernel");
    const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, MS_TH);
    const size_t lds = (size_t)(MS_PH * MS_PW * MS_CKP) * sizeof(float);
    MSTG_LAUNCH((ms_dgrad_kernel<CH>), dim3(N * tiles_x * tiles_y), dim3(256), lds, st, dy, (const float*)wp, dres, dx, N, H, W,
                       tiles_x, tiles_y);
    MSTG_CHECK_LAUNCH("ms_dgrad_kernel");
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

extern "C" size_t mstg_msblock_dgrad_workspace_bytes(int CH) {
    if (CH == 16) return ms_dgrad_ws_floats<16>() * sizeof(float);
    if (CH == 32) return ms_dgrad_ws_floats<32>() * sizeof(float);
    if (CH == 64) return ms_dgrad_ws_floats<64>() * sizeof(float);
    return 0;
}

extern "C" int mstg_msblock_dgrad(const float* dy, const float* w1, const float* w2, const float* w3, const float* w4, const float* dres,
                                  float* dx, int N, int H, int W, int CH, void* workspace, size_t workspace_bytes, void* stream) {
    if (!dy || !dx || !w1 || !w2 || !w3 || !w4) return fail_arg(MSTG_E_BADARG, "msblock_dgrad: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "msblock_dgrad: bad shape");
    if ((uint64_t)H * W * CH >= (1ull << 30)) return fail_arg(MSTG_E_UNSUPPORTED, "msblock_dgrad: one image must stay below 2^30 elements");
    MsParamPtrs prm{{w1, w2, w3, w4}, {nullptr, nullptr, nullptr, nullptr}};
    hipStream_t st = (hipStream_t)stream;
    if (CH == 16) return launch_ms_dgrad<16>(dy, prm, dres, dx, N, H, W, workspace, workspace_bytes, st);
    if (CH == 32) return launch_ms_dgrad<32>(dy, prm, dres, dx, N, H, W, workspace, workspace_bytes, st);
    if (CH == 64) return launch_ms_dgrad<64>(dy, prm, dres, dx, N, H, W, workspace, workspace_bytes, st);
    return fail_arg(MSTG_E_UNSUPPORTED, "msblock_dgrad: fused path exists for 16, 32 and 64 channels");
}

extern "C" int mstg_msblock_fused_supported(int CH) { return CH == 16 || CH == 32 || CH == 64; }

extern "C" size_t mstg_msblock_fwd_workspace_bytes(int CH) {
    if (!mstg_msblock_fused_supported(CH)) return 0;
    return (size_t)((CH / 16) * (CH / 16 + 24) * 256 + CH) * sizeof(float);
}

extern "C" int mstg_msblock_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                const float* b3, const float* w4, const float* b4, float* y, int N, int H, int W, int CH, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!x || !y || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !w4 || !b4) return fail_arg(MSTG_E_BADARG, "msblock_fwd: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "msblock_fwd: bad shape");
    if ((uint64_t)H * W * CH >= (1ull << 30)) return fail_arg(MSTG_E_UNSUPPORTED, "msblock_fwd: one image must stay below 2^30 elements");
    MsParamPtrs prm{{w1, w2, w3, w4}, {b1, b2, b3, b4}};
    hipStream_t st = (hipStream_t)stream;
    if (CH == 16) return launch_ms_fwd<16>(x, prm, y, N, H, W, workspace, workspace_bytes, st);
    if (CH == 32) return launch_ms_fwd<32>(x, prm, y, N, H, W, workspace, workspace_bytes, st);
    if (CH == 64) return launch_ms_fwd<64>(x, prm, y, N, H, W, workspace, workspace_bytes, st);
    return fail_arg(MSTG_E_UNSUPPORTED, "msblock_fwd: fused path exists for 16, 32 and 64 channels");
}

// the same two entry points for a caller that caches the packed filters (common.h: t_ws_packed)
extern "C" int mstg_msblock_fwd_cached(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                       const float* b3, const float* w4, const float* b4, float* y, int N, int H, int W, int CH,
                                       void* workspace, size_t workspace_bytes, int workspace_packed, void* stream) {
    mstg::t_ws_packed = workspace_packed != 0;
    const int rc = mstg_msblock_fwd(x, w1, b1, w2, b2, w3, b3, w4, b4, y, N, H, W, CH, workspace, workspace_bytes, stream);
    mstg::t_ws_packed = false;
    return rc;
}
extern "C" int mstg_msblock_dgrad_cached(const float* dy, const float* w1, const float* w2, const float* w3, const float* w4,
                                         const float* dres, float* dx, int N, int H, int W, int CH, void* workspace, size_t workspace_bytes,
                                         int workspace_packed, void* stream) {
    mstg::t_ws_packed = workspace_packed != 0;
    const int rc = mstg_msblock_dgrad(dy, w1, w2, w3, w4, dres, dx, N, H, W, CH, workspace, workspace_bytes, stream);
    mstg::t_ws_packed = false;
    return rc;
}

extern "C" size_t mstg_msblock_wgrad_workspace_bytes(int N, int H, int W, int CH) {
    int S, tx, ty, nt;
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    if (CH == 16) return ms_wgrad_plan<16>(N, H, W, S, tx, ty, nt);
    if (CH == 32) return ms_wgrad_plan<32>(N, H, W, S, tx, ty, nt);
    if (CH == 64) return ms_wgrad_plan<64>(N, H, W, S, tx, ty, nt);
    return 0;
}

extern "C" int mstg_msblock_wgrad(const float* x, const float* dy, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                                  float* dw4, float* db4, int accumulate, int N, int H, int W, int CH, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw1 || !db1 || !dw2 || !db2 || !dw3 || !db3 || !dw4 || !db4) return fail_arg(MSTG_E_BADARG, "msblock_wgrad: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "msblock_wgrad: bad shape");
    if ((uint64_t)H * W * CH >= (1ull << 30)) return fail_arg(MSTG_E_UNSUPPORTED, "msblock_wgrad: one image must stay below 2^30 elements");
    MsGradPtrs out{{dw1, dw2, dw3, dw4}, {db1, db2, db3, db4}};
    hipStream_t st = (hipStream_t)stream;
    if (CH == 16) return launch_ms_wgrad<16>(x, dy, out, accumulate, N, H, W, workspace, workspace_bytes, st);
    if (CH == 32) return launch_ms_wgrad<32>(x, dy, out, accumulate, N, H, W, workspace, workspace_bytes, st);
    if (CH == 64) return launch_ms_wgrad<64>(x, dy, out, accumulate, N, H, W, workspace, workspace_bytes, st);
    return fail_arg(MSTG_E_UNSUPPORTED, "msblock_wgrad: fused path exists for 16, 32 and 64 channels");
}
