// Weight gradient of Conv2d / ConvTranspose2d(k4,s2,p1) on the fp32 MFMA of gfx950.
//
//   dW(gch, hch, tap) = sum_{n, gy, gx} G[n, gy*s - p + ky*d, gx*s - p + kx*d, gch] * Hh[n, gy, gx, hch]
//
// "G" is the tensor the filter taps are gathered from (Conv2d: the module input; ConvTranspose2d: the output
// gradient), "Hh" the tensor living on the walked grid (Conv2d: the output gradient; ConvTranspose2d: the module
// input).  GEMM view: M = 16 gathered channels, N = 16*NFH grid channels, K = pixels (4 per MFMA).  A workgroup
// walks a strided set of 8x16 pixel tiles, keeping one accumulator fragment per tap in registers; the four waves
// split the tile rows, are summed through LDS once at the end, and the per-workgroup partials are reduced by a
// second tiny kernel in a fixed order (bit-reproducible: no atomics).
//
// Three operand packings keep the 16x16 MFMA tile full when a tensor has only 3-4 channels (the RGB image at the stem,
// the RGB gradient at the head), where a plain channel mapping would leave 13 of 16 rows or columns idle:
//   MODE_PLAIN : M = 16 gathered channels, N = 16 grid channels.
//   MODE_PACKX : gathered tensor has <= 4 channels.  The LDS patch is stored [row][col][4] with no padding, so the 16
//                floats starting at a pixel are (4 consecutive pixels) x (4 channels): M row m = 4*kx_low + c.  One MFMA
//                covers four horizontally adjacent taps; a 7x7 filter needs 7 x 2 "taps" instead of 49.
//   MODE_DPACK : grid tensor has <= 4 channels (stride 1, dilation 1).  The grid tile is stored [row][col][4] with three
//                extra columns, and N column n = 4*delta + co is the gradient of the pixel delta columns to the right:
//                D[ci][(delta,co)] accumulates x[p + tap] * dy[p + delta] = the gradient of tap (kx - delta).  The walked
//                grid starts three columns left of the image so every (pixel, delta) pair is visited exactly once.
//
// Reference sites replaced: the wgrad half of convolution_backward for every conv on the path
// (enhanced_generator.py:10-11,53-73,92,99,106,121,128,137,237-265; pretrain.py:65-91).
#include "common.h"
#include <stdlib.h>

namespace mstg {

enum { MODE_PLAIN = 0, MODE_PACKX = 1, MODE_DPACK = 2 };

struct WGradArgs {
    const float* g;   // gathered tensor
    const float* h;   // grid tensor
    float* partial;   // [S][T][Cg][Ch] (+ [Ch] bias tail)
    int N;
    int gH, gW, g_ctot, g_coff, g_nchw, Cg;
    int hH, hW, h_ctot, h_coff, h_nchw, Ch;  // hH x hW is the walked grid
    int KH, KW, stride, pad, dil;
    int tiles_x, tiles_y, ntiles;
    int PH, PW;     // gathered patch extent
    int T;          // real taps KH*KW
    int Teff;       // accumulator "taps": T, or KH * ceil(KW/4) in the packed modes
    int tapsx;      // ceil(KW/4) in the packed modes
    int TGn;        // accumulator taps per z-slice
    int n_gchunks;
    int mode;
    int ckp;        // LDS floats per patch pixel (20 plain/dpack: 16 channels + pad; 4 packx)
    int htw;        // LDS grid-tile row length in pixels (16, or 20 for dpack)
    int xshift;     // walked-grid column offset (3 for dpack, else 0)
    int with_bias;  // also emit per-split column sums of the grid tensor (Conv2d bias gradient) after the T*Cg*Ch block
    const float* g_stats;  // wgrad_p32_kernel: the gathered tensor is RAW, in front of InstanceNorm + ReLU; (mean, rstd) [N][Cg][2]
};

constexpr int WT_H = 8, WT_W = 16;

// Phase time stamps of sampled tap-split workgroups, third tile of each (tools/diag_stamps_wgrad.py); only with -DMSTG_STAMPS.
#ifdef MSTG_STAMPS
__device__ unsigned long long g_wg_stamps[2 * 64 * 8];
#define WG_STAMP(k)                                                                                                          \
    if (tcount == 2 && (threadIdx.x == 0 || threadIdx.x == 192) && (blockIdx.x % 11) == 0 && blockIdx.x / 11 < 64 && blockIdx.y == 0) \
        g_wg_stamps[((threadIdx.x != 0) * 64 + blockIdx.x / 11) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define WG_STAMP(k)
#endif
#ifndef WG_WAVES
#define WG_WAVES 3  // waves per SIMD the register allocation must allow (3 -> <= 168 VGPRs, no scratch spills) = workgroups per CU the
                    // launches create.  Uncapped the compiler took up to 250 registers (one or two waves per SIMD: 10-15 % slower on
                    // the small-channel layers); at 4 (128 VGPRs) the 7x7 and 3x3 variants spilled 8-47 registers around the staging
#endif

// Stage one tile's gathered patch and grid-tensor tile into LDS (shared by both kernels).  The aligned NHWC cases and the
// <= 4-channel cases go through the batched helpers of common.h; the rest (odd channel counts, wide NCHW) keeps simple loops.
__device__ __forceinline__ void wgrad_stage(const WGradArgs& a, float* patch, float* ht, int n, int ty0, int gx0, int y0, int x0, int g0,
                                            int h0, int TH, int nqh, int BNP, unsigned m_pw, unsigned m_htw, unsigned m_nqh, int tid) {
    const int lane = tid & 63, wave = tid >> 6, ckp = a.ckp, mode = a.mode;
    const size_t gplane = (size_t)a.gH * a.gW, hplane = (size_t)a.hH * a.hW;
    // ---- gathered patch, zero beyond Cg and outside the image ---------------------------------------------------------
    if (mode == MODE_PACKX) {  // <= 4 channels, [row][col][4]
        if (a.g_nchw)
            stage_window_c4(a.g + ((size_t)n * a.g_ctot + a.g_coff) * gplane, patch, a.PH, a.PW, m_pw, y0, x0, a.gH, a.gW, a.gW, 1u,
                            (unsigned)gplane, a.Cg, tid);
        else
            stage_window_c4(a.g + (size_t)n * gplane * a.g_ctot + a.g_coff, patch, a.PH, a.PW, m_pw, y0, x0, a.gH, a.gW,
                            (unsigned)(a.gW * a.g_ctot), (unsigned)a.g_ctot, 1u, a.Cg, tid);
    } else if (a.g_nchw) {
        for (int pr = wave; pr < a.PH; pr += 4) {
            const int iy = y0 + pr;
            for (int pc = lane; pc < a.PW; pc += 64) {
                const int ix = x0 + pc;
                const bool inb = (unsigned)iy < (unsigned)a.gH && (unsigned)ix < (unsigned)a.gW;
                float* dst = &patch[(pr * a.PW + pc) * ckp];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    float v = 0.f;
                    if (inb && c < a.Cg) v = a.g[(((size_t)n * a.g_ctot + a.g_coff + c) * a.gH + iy) * a.gW + ix];
                    dst[c] = v;
                }
            }
        }
    } else if (((a.g_ctot | a.g_coff | a.Cg) & 3) == 0) {
        stage_window(a.g + (size_t)n * gplane * a.g_ctot + a.g_coff + g0, patch, a.PH, a.PW, 4, m_pw, 0x40000000u, y0, x0, a.gH, a.gW,
                     a.g_ctot, min(4, (a.Cg - g0) >> 2), ckp, tid);
    } else {
        for (int pr = wave; pr < a.PH; pr += 4) {
            const int iy = y0 + pr;
            for (int e = lane; e < a.PW * 4; e += 64) {
                const int pc = e >> 2, q = e & 3;
                const int ix = x0 + pc;
                const bool inb = (unsigned)iy < (unsigned)a.gH && (unsigned)ix < (unsigned)a.gW;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                const int c = g0 + 4 * q;
                if (inb && c < a.Cg) {
                    const float* src = a.g + (((size_t)n * a.gH + iy) * a.gW + ix) * a.g_ctot + a.g_coff + c;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k < a.Cg) v[k] = src[k];
                }
                *reinterpret_cast<f32x4*>(&patch[(pr * a.PW + pc) * ckp + 4 * q]) = v;
            }
        }
    }
    // ---- grid tensor tile, zero outside ------------------------------------------------------------------------------------
    if (mode == MODE_DPACK) {  // <= 4 channels, [TH][htw][4], columns gx0 .. gx0 + htw
        if (a.h_nchw)
            stage_window_c4(a.h + ((size_t)n * a.h_ctot + a.h_coff) * hplane, ht, TH, a.htw, m_htw, ty0 * TH, gx0, a.hH, a.hW, a.hW, 1u,
                            (unsigned)hplane, a.Ch, tid);
        else if (a.Ch == 4 && ((a.h_ctot | a.h_coff) & 3) == 0)
            stage_window(a.h + (size_t)n * hplane * a.h_ctot + a.h_coff, ht, TH, a.htw, 1, m_htw, 0u, ty0 * TH, gx0, a.hH, a.hW, a.h_ctot, 1,
                         4, tid);
        else
            stage_window_c4(a.h + (size_t)n * hplane * a.h_ctot + a.h_coff, ht, TH, a.htw, m_htw, ty0 * TH, gx0, a.hH, a.hW,
                            (unsigned)(a.hW * a.h_ctot), (unsigned)a.h_ctot, 1u, a.Ch, tid);
    } else if (a.h_nchw) {
        for (int c = wave; c < 4 * nqh; c += 4) {
            for (int p = lane; p < TH * 16; p += 64) {
                const int gy = ty0 * TH + (p >> 4), gx = gx0 + (p & 15);
                float v = 0.f;
                if (gy < a.hH && gx < a.hW && h0 + c < a.Ch)
                    v = a.h[(((size_t)n * a.h_ctot + a.h_coff + h0 + c) * a.hH + gy) * a.hW + gx];
                ht[p * BNP + c] = v;
            }
        }
    } else if (((a.h_ctot | a.h_coff | a.Ch) & 3) == 0) {
        stage_window(a.h + (size_t)n * hplane * a.h_ctot + a.h_coff + h0, ht, TH, 16, nqh, 0x10000000u, m_nqh, ty0 * TH, gx0, a.hH, a.hW,
                     a.h_ctot, min(nqh, (a.Ch - h0) >> 2), BNP, tid);
    } else {
        for (int e = tid; e < TH * 16 * nqh; e += 256) {
            const int q = e % nqh, p = e / nqh;
            const int gy = ty0 * TH + (p >> 4), gx = gx0 + (p & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int c = h0 + 4 * q;
            if (gy < a.hH && gx < a.hW && c < a.Ch) {
                const float* src = a.h + (((size_t)n * a.hH + gy) * a.hW + gx) * a.h_ctot + a.h_coff + c;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c + k < a.Ch) v[k] = src[k];
            }
            *reinterpret_cast<f32x4*>(&ht[p * BNP + 4 * q]) = v;
        }
    }
}


// TG  : accumulator fragments (taps) per workgroup z-slice;  NFH : 16-wide grid-channel fragments per workgroup
template <int TG, int NFH>
__global__ __launch_bounds__(256, (TG >= 16 ? 2 : WG_WAVES)) void wgrad_kernel(const WGradArgs a) {
    constexpr int BN = 16 * NFH, BNP = BN + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                        // [PH][PW][ckp]
    float* ht = smem + ((a.PH * a.PW * a.ckp + 3) & ~3);        // plain/packx: [128][BNP]; dpack: [8][htw][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int gchunk = blockIdx.y % a.n_gchunks, htile = blockIdx.y / a.n_gchunks;
    const int g0 = gchunk * 16, h0 = htile * BN;
    const int t0 = blockIdx.z * a.TGn;
    const int tn = min(a.TGn, a.Teff - t0);
    const int s = a.stride, ckp = a.ckp;
    const int mode = a.mode;

    f32x4 acc[TG][NFH];
    int toff[TG];  // LDS offset of each (packed) tap inside the patch (wave-uniform, hoisted out of the pixel loops)
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tt = min(t0 + t, a.Teff - 1);
        int ky, kxo;
        if (mode == MODE_PLAIN) { ky = tt / a.KW; kxo = (tt % a.KW) * a.dil; }
        else if (mode == MODE_PACKX) { ky = tt / a.tapsx; kxo = 4 * (tt % a.tapsx); }
        else { ky = tt / a.tapsx; kxo = 4 * (tt % a.tapsx) + 3; }
        toff[t] = (ky * a.dil * a.PW + kxo) * ckp;
#pragma unroll
        for (int hf = 0; hf < NFH; ++hf) acc[t][hf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Conv2d bias gradient = column sums of the grid tensor.  All 256 threads share the tile's 128 pixels (thread = column
    // bcol, pixel phase bpart); a single wave summing them serially held the other three at the next barrier for ~5k cycles
    // per tile (tools/diag_stamps_wgrad.py).  The phases are combined in fixed order after the tile loop.
    const bool do_bias = a.with_bias && gchunk == 0 && blockIdx.z == 0;
    const int bcols = mode == MODE_DPACK ? 4 : BN, bparts = 256 / bcols, bcol = tid % bcols, bpart = tid / bcols;
    float bsum = 0.f;
    const unsigned m_pw = magic_u32(a.PW), m_htw = magic_u32(a.htw);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
        const int gx0 = tx0 * WT_W - a.xshift;  // first walked-grid column of this tile
        const int y0 = ty0 * WT_H * s - a.pad, x0 = gx0 * s - a.pad;
        __syncthreads();
        wgrad_stage(a, patch, ht, n, ty0, gx0, y0, x0, g0, h0, WT_H, BN / 4, BNP, m_pw, m_htw, magic_u32(BN / 4), tid);
        __syncthreads();
        if (do_bias && mode != MODE_DPACK) {
#pragma unroll 4
            for (int p = bpart; p < 128; p += bparts) bsum += ht[p * BNP + bcol];
        }
        if (do_bias && mode == MODE_DPACK) {  // the tile's own 16 columns sit 3 pixels into the [8][htw][4] tile
#pragma unroll
            for (int p = bpart; p < 128; p += 64) bsum += ht[((p >> 4) * a.htw + (p & 15) + 3) * 4 + bcol];
        }
        // ---- MFMA: this wave's two tile rows, 4 pixels per k-step --------------------------------------------------
        const int hrow = mode == MODE_DPACK ? a.htw * 4 : 16 * BNP, hcol = mode == MODE_DPACK ? 4 : BNP;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
#pragma unroll 1
            for (int xs = 0; xs < 4; ++xs) {
                const int c = 4 * xs + g;  // this lane's k-slot pixel column
                float bf[NFH];
#pragma unroll
                for (int hf = 0; hf < NFH; ++hf) bf[hf] = ht[r * hrow + c * hcol + 16 * hf + i];
                const int abase = (r * s * a.PW + c * s) * ckp + i;
                float af[TG];  // taps beyond tn repeat the last real tap: their accumulators are never written out
#pragma unroll
                for (int t = 0; t < TG; ++t) af[t] = patch[abase + toff[t]];
#pragma unroll
                for (int t = 0; t < TG; ++t)
#pragma unroll
                    for (int hf = 0; hf < NFH; ++hf) acc[t][hf] = mfma16(af[t], bf[hf], acc[t][hf]);
            }
        }
    }

    // ---- sum the four waves through LDS (fixed order), then write this workgroup's partial --------------------
    __syncthreads();
    if (do_bias) {  // wave-uniform
        smem[tid] = bsum;
        __syncthreads();
        if (tid < bcols) {
            bsum = 0.f;
            for (int j = 0; j < bparts; ++j) bsum += smem[tid + bcols * j];
        }
        __syncthreads();
    }
    float* red = smem;  // [TG*NFH][256]
    for (int wv = 1; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int hf = 0; hf < NFH; ++hf) *reinterpret_cast<f32x4*>(&red[((t * NFH + hf) * 64 + lane) * 4]) = acc[t][hf];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int hf = 0; hf < NFH; ++hf) acc[t][hf] += *reinterpret_cast<const f32x4*>(&red[((t * NFH + hf) * 64 + lane) * 4]);
        }
        __syncthreads();
    }
    const size_t pstride = (size_t)a.T * a.Cg * a.Ch + (a.with_bias ? a.Ch : 0);
    if (do_bias && tid < BN && h0 + tid < a.Ch) a.partial[(size_t)blockIdx.x * pstride + (size_t)a.T * a.Cg * a.Ch + h0 + tid] = bsum;
    if (wave == 0) {
        float* out = a.partial + (size_t)blockIdx.x * pstride;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            if (t >= tn) continue;
            const int tt = t0 + t;
#pragma unroll
            for (int hf = 0; hf < NFH; ++hf) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // accumulator element: row m = 4g + e, column n = 16hf + i
                    int tap, gch, hch;
                    bool ok;
                    if (mode == MODE_PLAIN) {
                        tap = tt; gch = g0 + 4 * g + e; hch = h0 + 16 * hf + i;
                        ok = gch < a.Cg && hch < a.Ch;
                    } else if (mode == MODE_PACKX) {  // m = 4*kx_low + c
                        const int kx = 4 * (tt % a.tapsx) + g;
                        tap = (tt / a.tapsx) * a.KW + kx; gch = e; hch = h0 + 16 * hf + i;
                        ok = kx < a.KW && gch < a.Cg && hch < a.Ch;
                    } else {                          // n = 4*delta + co ; tap kx = 4j + 3 - delta
                        const int kx = 4 * (tt % a.tapsx) + 3 - (i >> 2);
                        tap = (tt / a.tapsx) * a.KW + kx; gch = g0 + 4 * g + e; hch = i & 3;
                        ok = kx >= 0 && kx < a.KW && gch < a.Cg && hch < a.Ch;
                    }
                    if (ok) out[((size_t)tap * a.Cg + gch) * a.Ch + hch] = acc[t][hf][e];
                }
            }
        }
    }
}

// =====================================================================================================================
// Tap-split variant (every filter with more than one tap).  The accumulator "units" (tap x 16-wide grid-channel fragment)
// of a (gathered-chunk, grid-channel-group) pair are dealt round-robin to the four waves and every wave walks ALL pixels
// of the tile.  Compared with splitting the pixels: one workgroup covers up to 64 grid channels with one staging of the
// patch (4x less re-staging), a wave keeps only UW accumulators (occupancy), and no cross-wave reduction is needed.
// TH (4 or 8) is the tile height: stride-2 patches are large, a 4-row tile keeps three workgroups per CU.
// =====================================================================================================================
// Register budget of three waves per SIMD: the launch puts three workgroups on a CU (768 in all), and at four (128 VGPRs) the
// compiler spilled 45 registers around the staging (scratch traffic on every tile).
template <int UW>
__global__ __launch_bounds__(256, 3) void wgrad_ts_kernel(const WGradArgs a, const int TH, const int NFHT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ckp = a.ckp, s = a.stride, mode = a.mode;
    const int BNP = 16 * NFHT + 4;
    float* patch = smem;                                        // [PH][PW][ckp]
    float* ht = smem + ((a.PH * a.PW * ckp + 3) & ~3);          // plain/packx: [TH*16][BNP]; dpack: [TH][htw][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int gchunk = blockIdx.y % a.n_gchunks, hgroup = blockIdx.y / a.n_gchunks;
    const int g0 = gchunk * 16, h0 = hgroup * 16 * NFHT;
    const int U = a.Teff * NFHT;  // units of this workgroup; unit u -> tap u / NFHT, fragment u % NFHT

    f32x4 acc[UW];
    int toff[UW];
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        const int u = min(wave + 4 * k, U - 1);
        const int tt = u / NFHT, hf = u % NFHT;
        int ky, kxo;
        if (mode == MODE_PLAIN) { ky = tt / a.KW; kxo = (tt % a.KW) * a.dil; }
        else if (mode == MODE_PACKX) { ky = tt / a.tapsx; kxo = 4 * (tt % a.tapsx); }
        else { ky = tt / a.tapsx; kxo = 4 * (tt % a.tapsx) + 3; }
        toff[k] = (ky * a.dil * a.PW + kxo) * ckp;
        acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int nu = (U - wave + 3) / 4;  // units this wave really owns
    const int hfo0 = 16 * (min(wave, U - 1) % NFHT);

    const bool do_bias = a.with_bias && gchunk == 0;
    const int bcols = 16 * NFHT, bparts = 256 / bcols, bcol = tid % bcols, bpart = tid / bcols;  // as in wgrad_kernel
    float bsum = 0.f;
    const unsigned m_pw = magic_u32(a.PW), m_htw = magic_u32(a.htw), m_nqh = magic_u32(4 * NFHT);
    const int hrow = mode == MODE_DPACK ? a.htw * 4 : 16 * BNP, hcol = mode == MODE_DPACK ? 4 : BNP;
#ifdef MSTG_STAMPS
    int tcount = 0;
#endif
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
        const int gx0 = tx0 * WT_W - a.xshift;
        const int y0 = ty0 * TH * s - a.pad, x0 = gx0 * s - a.pad;
        WG_STAMP(0)
        __syncthreads();
        WG_STAMP(1)
        wgrad_stage(a, patch, ht, n, ty0, gx0, y0, x0, g0, h0, TH, 4 * NFHT, BNP, m_pw, m_htw, m_nqh, tid);
        WG_STAMP(2)
        __syncthreads();
        WG_STAMP(3)
        if (do_bias && bpart < bparts) {
#pragma unroll 4
            for (int p = bpart; p < TH * 16; p += bparts) bsum += ht[p * BNP + bcol];
        }
        WG_STAMP(4)
        // ---- MFMA: every wave walks all TH x 16 pixels (4 per k-step) for its own units -----------------------------------
#pragma unroll 1
        for (int r = 0; r < TH; ++r) {
#pragma unroll 2
            for (int xs = 0; xs < 4; ++xs) {  // two k-steps per iteration: the second step's LDS reads are in flight behind the first's MFMAs
                const int c = 4 * xs + g;
                const int abase = (r * s * a.PW + c * s) * ckp + i;
                const int hbase = r * hrow + c * hcol + i;
                float af[UW];  // units beyond nu repeat the last real unit: their accumulators are never written out
                // NFHT divides 4 (launch_ts_t checks): unit u = wave + 4k uses fragment u % NFHT = wave % NFHT for every k,
                // so ONE B read serves all of this wave's units
                const float b = ht[hbase + hfo0];
#pragma unroll
                for (int k = 0; k < UW; ++k) af[k] = patch[abase + toff[k]];
#pragma unroll
                for (int k = 0; k < UW; ++k) acc[k] = mfma16(af[k], b, acc[k]);
            }
        }
        WG_STAMP(5)
#ifdef MSTG_STAMPS
        ++tcount;
#endif
    }
    // ---- each wave writes the partial of its own units; no cross-wave reduction ------------------------------------------
    if (do_bias) {  // wave-uniform: combine the pixel phases in fixed order
        __syncthreads();
        smem[tid] = bpart < bparts ? bsum : 0.f;
        __syncthreads();
        if (tid < bcols) {
            bsum = 0.f;
            for (int j = 0; j < bparts; ++j) bsum += smem[tid + bcols * j];
        }
    }
    const size_t pstride = (size_t)a.T * a.Cg * a.Ch + (a.with_bias ? a.Ch : 0);
    if (do_bias && tid < 16 * NFHT && h0 + tid < a.Ch) a.partial[(size_t)blockIdx.x * pstride + (size_t)a.T * a.Cg * a.Ch + h0 + tid] = bsum;
    float* out = a.partial + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        if (k >= nu) continue;
        const int u = wave + 4 * k, tt = u / NFHT, hf = u % NFHT;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int tap, gch, hch;
            bool ok;
            if (mode == MODE_PLAIN) {
                tap = tt; gch = g0 + 4 * g + e; hch = h0 + 16 * hf + i;
                ok = gch < a.Cg && hch < a.Ch;
            } else if (mode == MODE_PACKX) {
                const int kx = 4 * (tt % a.tapsx) + g;
                tap = (tt / a.tapsx) * a.KW + kx; gch = e; hch = h0 + 16 * hf + i;
                ok = kx < a.KW && gch < a.Cg && hch < a.Ch;
            } else {
                const int kx = 4 * (tt % a.tapsx) + 3 - (i >> 2);
                tap = (tt / a.tapsx) * a.KW + kx; gch = g0 + 4 * g + e; hch = i & 3;
                ok = kx >= 0 && kx < a.KW && gch < a.Cg && hch < a.Ch;
            }
            if (ok) out[((size_t)tap * a.Cg + gch) * a.Ch + hch] = acc[k][e];
        }
    }
}

// =====================================================================================================================
// Persistent form of the tap-split kernel for the 4x4 stride-2 family (aligned NHWC on both sides, gathered channels in chunks
// of 16, grid channels in groups of 32): the workgroup fetches the NEXT tile's patch and grid tile into registers before it
// starts on the current tile's MFMAs, so the global round trip (a third of a tile's time in wgrad_ts_kernel, time stamps in
// DESIGN.md) runs behind the arithmetic instead of in front of it.  Same LDS layouts, same unit dealing, same partial-slab
// layout and reduce kernel as wgrad_ts_kernel.
// =====================================================================================================================
#ifndef WP_CKP
#define WP_CKP 24
#endif
#ifndef WP_BNP
#define WP_BNP 48
#endif
// LDS strides (floats) of a patch pixel / a grid-tile pixel: the b32 operand reads of a half-wave touch 16 consecutive floats at two
// pixels, conflict-free when the two pixels sit 16 banks apart: 2 * 24 = 48 = 16 (mod 32) for the stride-2 patch, 48 for the tile
constexpr int WP_TH = 4, WP_NFHT = 2, WP_UW = 8, WP_NG = 6, WP_NH = 2;  // 10 x 34 x 4 patch quads = 5.3 per thread; 64 x 8 tile quads = 2

struct WpRegs {
    f32x4 gv[WP_NG], hv[WP_NH];
    f32x4 st_lo, st_hi;  // g_stats: (mean, rstd) x 4 channels of this thread's patch quad (the same quad for all its elements), image of the tile
    unsigned gok, hok;
};

// element k of thread tid: patch quad e = 256 k + tid -> pixel e >> 2 = (pr, pc), channel quad e & 3; tile quad -> pixel e >> 3, quad e & 7.
// Offsets are recomputed per tile (a few integer instructions against ~4000 MFMA cycles) rather than held in registers.
__device__ __forceinline__ void wp_fetch(const WGradArgs& a, int tile, int tid, int g0, int h0, unsigned m_pw, WpRegs& R) {
    const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int y0 = ty0 * WP_TH * 2 - 1, x0 = tx0 * WT_W * 2 - 1;
    const char* gimg = reinterpret_cast<const char*>(a.g + (size_t)n * a.gH * a.gW * a.Cg + g0);
    const char* himg = reinterpret_cast<const char*>(a.h + (size_t)n * a.hH * a.hW * a.Ch + h0);
    const bool gin = y0 >= 0 && x0 >= 0 && y0 + a.PH <= a.gH && x0 + a.PW <= a.gW;  // uniform
    const int total = a.PH * a.PW * 4;
    R.gok = 0;
#pragma unroll
    for (int k = 0; k < WP_NG; ++k) {
        const int e = 256 * k + tid, pix = e >> 2;
        const int pr = (int)__umulhi((unsigned)pix, m_pw), pc = pix - pr * a.PW;
        const bool ok = e < total && (gin || ((unsigned)(y0 + pr) < (unsigned)a.gH && (unsigned)(x0 + pc) < (unsigned)a.gW));
        R.gok |= (unsigned)ok << k;
        const unsigned off = ok ? (unsigned)((((y0 + pr) * a.gW + x0 + pc) * a.Cg + 4 * (e & 3)) * 4) : 0u;
        R.gv[k] = *reinterpret_cast<const f32x4*>(gimg + off);
    }
    if (a.g_stats) {
        const float* st = a.g_stats + ((size_t)n * a.Cg + g0 + 4 * (tid & 3)) * 2;
        R.st_lo = *reinterpret_cast<const f32x4*>(st);
        R.st_hi = *reinterpret_cast<const f32x4*>(st + 4);
    }
    const int gy0 = ty0 * WP_TH, gx0 = tx0 * WT_W;
    R.hok = 0;
#pragma unroll
    for (int k = 0; k < WP_NH; ++k) {
        const int e = 256 * k + tid, pix = e >> 3;
        const bool ok = gy0 + (pix >> 4) < a.hH && gx0 + (pix & 15) < a.hW;
        R.hok |= (unsigned)ok << k;
        const unsigned off = ok ? (unsigned)((((gy0 + (pix >> 4)) * a.hW + gx0 + (pix & 15)) * a.Ch + 4 * (e & 7)) * 4) : 0u;
        R.hv[k] = *reinterpret_cast<const f32x4*>(himg + off);
    }
}

__global__ __launch_bounds__(256, 3) void wgrad_p32_kernel(const WGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TH = WP_TH, NFHT = WP_NFHT, UW = WP_UW, BNP = WP_BNP;
    constexpr int ckp = WP_CKP;
    float* patch = smem;                                // [PH][PW][ckp]
    float* ht = smem + ((a.PH * a.PW * ckp + 3) & ~3);  // [TH*16][BNP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int gchunk = blockIdx.y % a.n_gchunks, hgroup = blockIdx.y / a.n_gchunks;
    const int g0 = gchunk * 16, h0 = hgroup * 16 * NFHT;
    f32x4 acc[UW];
#pragma unroll
    for (int k = 0; k < UW; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // unit u = wave + 4k: tap tt = u / 2 = 2k + wave / 2 -> ky = tt >> 2 = k >> 1, kx = tt & 3 = 2 (k & 1) + wave / 2
    const int toff0 = (wave >> 1) * ckp;
    const int hfo0 = 16 * (wave % NFHT);
    const bool do_bias = a.with_bias && gchunk == 0;
    const int bcols = 16 * NFHT, bparts = 256 / bcols, bcol = tid % bcols, bpart = tid / bcols;
    float bsum = 0.f;
    const unsigned m_pw = magic_u32(a.PW);
    const int total = a.PH * a.PW * 4;
    WpRegs R;
    int tile = blockIdx.x;
    if (tile < a.ntiles) wp_fetch(a, tile, tid, g0, h0, m_pw, R);
    for (; tile < a.ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < WP_NG; ++k) {
            const int e = 256 * k + tid;
            if (e < total) {
                f32x4 w = ((R.gok >> k) & 1) ? R.gv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.g_stats && ((R.gok >> k) & 1)) {  // norm_apply_kernel's arithmetic; the zero padding is of the NORMALISED tensor
                    w[0] = fmaxf((w[0] - R.st_lo[0]) * R.st_lo[1], 0.f);
                    w[1] = fmaxf((w[1] - R.st_lo[2]) * R.st_lo[3], 0.f);
                    w[2] = fmaxf((w[2] - R.st_hi[0]) * R.st_hi[1], 0.f);
                    w[3] = fmaxf((w[3] - R.st_hi[2]) * R.st_hi[3], 0.f);
                }
                *reinterpret_cast<f32x4*>(&patch[(e >> 2) * ckp + 4 * (e & 3)]) = w;
            }
        }
#pragma unroll
        for (int k = 0; k < WP_NH; ++k) {
            const int e = 256 * k + tid;
            *reinterpret_cast<f32x4*>(&ht[(e >> 3) * BNP + 4 * (e & 7)]) = ((R.hok >> k) & 1) ? R.hv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) wp_fetch(a, tile + gridDim.x, tid, g0, h0, m_pw, R);
        if (do_bias && bpart < bparts) {
#pragma unroll 4
            for (int p = bpart; p < TH * 16; p += bparts) bsum += ht[p * BNP + bcol];
        }
#pragma unroll 1
        for (int r = 0; r < TH; ++r) {
#pragma unroll 2
            for (int xs = 0; xs < 4; ++xs) {  // two k-steps per iteration: the second step's LDS reads are in flight behind the first's MFMAs
                const int c = 4 * xs + g;
                const float* ap = patch + (r * 2 * a.PW + c * 2) * ckp + i + toff0;
                const float b = ht[(r * 16 + c) * BNP + i + hfo0];
                float af[UW];
#pragma unroll
                for (int k = 0; k < UW; ++k) af[k] = ap[((k >> 1) * a.PW + 2 * (k & 1)) * ckp];
#pragma unroll
                for (int k = 0; k < UW; ++k) acc[k] = mfma16(af[k], b, acc[k]);
            }
        }
        __syncthreads();  // every wave is done with both tiles
    }
    if (do_bias) {
        smem[tid] = bpart < bparts ? bsum : 0.f;
        __syncthreads();
        if (tid < bcols) {
            bsum = 0.f;
            for (int j = 0; j < bparts; ++j) bsum += smem[tid + bcols * j];
        }
    }
    const size_t pstride = (size_t)a.T * a.Cg * a.Ch + (a.with_bias ? a.Ch : 0);
    if (do_bias && tid < 16 * NFHT) a.partial[(size_t)blockIdx.x * pstride + (size_t)a.T * a.Cg * a.Ch + h0 + tid] = bsum;
    float* out = a.partial + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int k = 0; k < UW; ++k) {
        const int u = wave + 4 * k, tt = u / NFHT, hf = u % NFHT;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[((size_t)tt * a.Cg + g0 + 4 * g + e) * a.Ch + h0 + 16 * hf + i] = acc[k][e];
    }
}

// =====================================================================================================================
// Persistent form of the pixel-split kernel for the two 7x7 layers with a 3-channel side (stem: the RGB image is the gathered
// tensor, MODE_PACKX; head: the RGB gradient is the grid tensor, MODE_DPACK), 16 channels on the other side: the next tile's
// patch and grid tile travel into registers behind the current tile's 112 MFMAs per wave, and the k-loop keeps the next step's
// LDS reads in flight behind the current step's MFMAs.  LDS layouts, tap packing, slab layout and reduce kernel are those of
// wgrad_kernel<14, 1>.
// =====================================================================================================================
template <int MODE>
struct W7 {
    static constexpr int NBIG = MODE == MODE_PACKX ? 2 : 6;  // 16-byte NHWC quads per thread: 128 x 4 tile quads / 322 x 4 patch quads
    static constexpr int NSM = MODE == MODE_PACKX ? 2 : 1;   // NCHW pixels (3 planes each) per thread: 322 patch / 160 tile pixels
    static constexpr int PH = 14, PW = 23, HTW = 20, TG = 14;
};

template <int MODE>
struct W7Regs {
    f32x4 big[W7<MODE>::NBIG];
    float sm[W7<MODE>::NSM][3];
    unsigned okb, oks;
};

template <int MODE>
__device__ __forceinline__ void w7_fetch(const WGradArgs& a, int tile, int tid, W7Regs<MODE>& R) {
    typedef W7<MODE> K;
    const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int gx0 = tx0 * WT_W - a.xshift;
    const int y0 = ty0 * WT_H - 3, x0 = gx0 - 3;
    R.okb = R.oks = 0;
    if (MODE == MODE_PACKX) {
        const float* himg = a.h + (size_t)n * a.hH * a.hW * 16;
#pragma unroll
        for (int k = 0; k < K::NBIG; ++k) {
            const int e = 256 * k + tid, pix = e >> 2;
            const int gy = ty0 * WT_H + (pix >> 4), gx = gx0 + (pix & 15);
            const bool ok = gy < a.hH && gx < a.hW;
            R.okb |= (unsigned)ok << k;
            R.big[k] = *reinterpret_cast<const f32x4*>(himg + (ok ? (unsigned)((gy * a.hW + gx) * 16 + 4 * (e & 3)) : 0u));
        }
        const size_t plane = (size_t)a.gH * a.gW;
        const float* gimg = a.g + ((size_t)n * a.g_ctot + a.g_coff) * plane;
#pragma unroll
        for (int k = 0; k < K::NSM; ++k) {
            const int e = 256 * k + tid;
            const int r = e / K::PW, c = e - r * K::PW;
            const bool ok = e < K::PH * K::PW && (unsigned)(y0 + r) < (unsigned)a.gH && (unsigned)(x0 + c) < (unsigned)a.gW;
            R.oks |= (unsigned)ok << k;
            const unsigned off = ok ? (unsigned)((y0 + r) * a.gW + x0 + c) : 0u;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) R.sm[k][ch] = gimg[off + (ch < a.Cg ? ch * plane : 0)];
        }
    } else {
        const float* gimg = a.g + (size_t)n * a.gH * a.gW * 16;
#pragma unroll
        for (int k = 0; k < K::NBIG; ++k) {
            const int e = 256 * k + tid, pix = e >> 2;
            const int r = pix / K::PW, c = pix - r * K::PW;
            const bool ok = e < K::PH * K::PW * 4 && (unsigned)(y0 + r) < (unsigned)a.gH && (unsigned)(x0 + c) < (unsigned)a.gW;
            R.okb |= (unsigned)ok << k;
            R.big[k] = *reinterpret_cast<const f32x4*>(gimg + (ok ? (unsigned)(((y0 + r) * a.gW + x0 + c) * 16 + 4 * (e & 3)) : 0u));
        }
        const size_t plane = (size_t)a.hH * a.hW;
        const float* himg = a.h + ((size_t)n * a.h_ctot + a.h_coff) * plane;
        {
            const int e = tid;
            const int r = e / K::HTW, c = e - r * K::HTW;
            const int gy = ty0 * WT_H + r, gx = gx0 + c;
            const bool ok = e < WT_H * K::HTW && gy < a.hH && (unsigned)gx < (unsigned)a.hW;
            R.oks = (unsigned)ok;
            const unsigned off = ok ? (unsigned)(gy * a.hW + gx) : 0u;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) R.sm[0][ch] = himg[off + (ch < a.Ch ? ch * plane : 0)];
        }
    }
}

template <int MODE>
__device__ __forceinline__ void w7_commit(const WGradArgs& a, int tid, const W7Regs<MODE>& R, float* patch, float* ht) {
    typedef W7<MODE> K;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_PACKX) {
#pragma unroll
        for (int k = 0; k < K::NBIG; ++k) {
            const int e = 256 * k + tid;
            *reinterpret_cast<f32x4*>(&ht[(e >> 2) * 20 + 4 * (e & 3)]) = ((R.okb >> k) & 1) ? R.big[k] : zero;
        }
#pragma unroll
        for (int k = 0; k < K::NSM; ++k) {
            const int e = 256 * k + tid;
            if (e < K::PH * K::PW) {
                f32x4 w = zero;
                if ((R.oks >> k) & 1) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) w[ch] = ch < a.Cg ? R.sm[k][ch] : 0.f;
                }
                *reinterpret_cast<f32x4*>(&patch[e * 4]) = w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K::NBIG; ++k) {
            const int e = 256 * k + tid;
            if (e < K::PH * K::PW * 4) *reinterpret_cast<f32x4*>(&patch[(e >> 2) * 20 + 4 * (e & 3)]) = ((R.okb >> k) & 1) ? R.big[k] : zero;
        }
        if (tid < WT_H * K::HTW) {
            f32x4 w = zero;
            if (R.oks & 1) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) w[ch] = ch < a.Ch ? R.sm[0][ch] : 0.f;
            }
            *reinterpret_cast<f32x4*>(&ht[tid * 4]) = w;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void wgrad7_kernel(const WGradArgs a) {
    typedef W7<MODE> K;
    constexpr int TG = K::TG, BNP = 20;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ckp = MODE == MODE_PACKX ? 4 : 20;
    float* patch = smem;                                          // [14][23][ckp]
    float* ht = smem + ((K::PH * K::PW * ckp + 3) & ~3);          // packx: [128][20]; dpack: [8][20][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    f32x4 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // packed tap t = (ky = t / 2, four columns from 4 (t % 2)); dpack reads them three pixels to the right (the tile's shift)
    constexpr int kx_shift = MODE == MODE_DPACK ? 3 : 0;
    const bool do_bias = a.with_bias;
    constexpr int bcols = MODE == MODE_DPACK ? 4 : 16, bparts = 256 / bcols;
    const int bcol = tid % bcols, bpart = tid / bcols;
    float bsum = 0.f;
    constexpr int hrow = MODE == MODE_DPACK ? K::HTW * 4 : 16 * BNP, hcol = MODE == MODE_DPACK ? 4 : BNP;
    W7Regs<MODE> R;
    int tile = blockIdx.x;
    if (tile < a.ntiles) w7_fetch<MODE>(a, tile, tid, R);
    for (; tile < a.ntiles; tile += gridDim.x) {
        w7_commit<MODE>(a, tid, R, patch, ht);
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) w7_fetch<MODE>(a, tile + gridDim.x, tid, R);
        if (do_bias) {
            if (MODE != MODE_DPACK) {
#pragma unroll 4
                for (int p = bpart; p < 128; p += bparts) bsum += ht[p * BNP + bcol];
            } else {
#pragma unroll
                for (int p = bpart; p < 128; p += 64) bsum += ht[((p >> 4) * K::HTW + (p & 15) + 3) * 4 + bcol];
            }
        }
        // this wave's two tile rows, 4 pixels per k-step; two steps per trip: the second step's LDS reads fly behind the first's MFMAs
#pragma unroll 2
        for (int st = 0; st < 8; ++st) {
            const int r = 2 * wave + (st >> 2), c = 4 * (st & 3) + g;
            const float bf = ht[r * hrow + c * hcol + i];
            const float* ap = patch + (r * K::PW + c + kx_shift) * ckp + i;
            float af[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t) af[t] = ap[((t >> 1) * K::PW + 4 * (t & 1)) * ckp];
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t] = mfma16(af[t], bf, acc[t]);
        }
        __syncthreads();
    }
    // ---- sum the four waves through LDS (fixed order), then write this workgroup's partial ------------------------------------
    if (do_bias) {  // uniform
        smem[tid] = bsum;
        __syncthreads();
        if (tid < bcols) {
            bsum = 0.f;
            for (int j = 0; j < bparts; ++j) bsum += smem[tid + bcols * j];
        }
        __syncthreads();
    }
    float* red = smem;  // [TG][256]
    for (int wv = 1; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < TG; ++t) *reinterpret_cast<f32x4*>(&red[(t * 64 + lane) * 4]) = acc[t];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t] += *reinterpret_cast<const f32x4*>(&red[(t * 64 + lane) * 4]);
        }
        __syncthreads();
    }
    const size_t pstride = (size_t)a.T * a.Cg * a.Ch + (a.with_bias ? a.Ch : 0);
    if (do_bias && tid < bcols && tid < a.Ch) a.partial[(size_t)blockIdx.x * pstride + (size_t)a.T * a.Cg * a.Ch + tid] = bsum;
    if (wave == 0) {
        float* out = a.partial + (size_t)blockIdx.x * pstride;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int tap, gch, hch;
                bool ok;
                if (MODE == MODE_PACKX) {  // row m = 4 * kx_low + c
                    const int kx = 4 * (t & 1) + g;
                    tap = (t >> 1) * 7 + kx; gch = e; hch = i;
                    ok = kx < 7 && gch < a.Cg;
                } else {                   // column n = 4 * delta + co ; tap kx = 4 j + 3 - delta
                    const int kx = 4 * (t & 1) + 3 - (i >> 2);
                    tap = (t >> 1) * 7 + kx; gch = 4 * g + e; hch = i & 3;
                    ok = kx >= 0 && kx < 7 && hch < a.Ch;
                }
                if (ok) out[((size_t)tap * a.Cg + gch) * a.Ch + hch] = acc[t][e];
            }
        }
    }
}

// =====================================================================================================================
// 1x1 convolutions: the weight gradient is a plain GEMM  dW[ci][co] = sum_p x[p][ci] dy[p][co]  over all P = N*H*W pixels,
// with no spatial structure at all.  The generic kernel above still tiles it in 2-D and gives every (16 input channels,
// 32 output channels) pair its own workgroup column, so x is re-read Ch/32 times and dy Cg/16 times.  Here a workgroup owns
// a run of pixels and ALL channels: it stages [32 pixels][Cg] and [32 pixels][Ch] once (rows are contiguous in NHWC), the
// four waves split the output-channel fragments, and each wave keeps MF x NW accumulators (MF = Cg/16 fragments of input
// channels, NW = its share of the Ch/16 output fragments).  x and dy are read exactly once.
// =====================================================================================================================
// pixels per staged tile: about eight 16-byte loads per thread (32 pixels at 64 + 192 channels, 256 at 16 + 16)
__host__ __device__ constexpr int w1x1_ld(int frags) { return (frags & 1) ? 16 * frags : 16 * frags + 16; }

static int wgrad_1x1_tile(int Cg, int Ch) {
    int p1 = (2048 / ((Cg + Ch) >> 2)) & ~3;
    return p1 < 32 ? 32 : (p1 > 256 ? 256 : p1);
}

// (16 -> 16 channels: four workgroups per CU -- <= 128 registers instead of the 136 the compiler takes when left alone; the same bound
// on the 32-channel variant made its batch-32 launches 65 % slower)
template <int MF, int NW>
__global__ __launch_bounds__(256, (MF * NW == 1 ? 4 : 1)) void wgrad_1x1_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                        long P, int Cg, int g_ctot, int g_coff, int Ch, int h_ctot, int h_coff,
                                                        int with_bias, int P1, const float* __restrict__ in_stats, int HW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // row strides = 16 mod 32 floats: the operand reads (4 bytes, lanes (i, g) -> row k0 + g, column 16 f + i; banks mod 32 over the two
    // 32-lane halves) then put rows g and g + 1 on the two halves of the banks (+ 4 made every read a two-way conflict)
    const int NF = (Ch + 15) / 16, lda = w1x1_ld(MF), ldb = w1x1_ld(NF);
    float* As = smem;             // [P1][lda]
    float* Bs = smem + P1 * lda;  // [P1][ldb]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    f32x4 acc[MF][NW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int k = 0; k < NW; ++k) acc[mf][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fewer than four output fragments: the spare waves take every KS-th pixel group of the same fragments instead
    const int KS = NF >= 3 ? 1 : (NF == 2 ? 2 : 4), wslot = wave % (4 / KS), kpart = wave / (4 / KS);
    const int qa = Cg >> 2, qb = Ch >> 2;          // channel quads per pixel
    const int ea = P1 * qa, eb = P1 * qb;          // float4 elements of the two tiles
    const unsigned m_qa = magic_u32(qa), m_qb = magic_u32(qb);
    float bsum = 0.f;
    const bool bsplit = Ch <= 64 && (256 % Ch) == 0;  // bias gradient: 256 / Ch threads per channel, combined in a fixed order at the end
    const int nbg = bsplit ? 256 / Ch : 1, bch = bsplit ? tid % Ch : tid, bgrp = bsplit ? tid / Ch : 0;
    const long ntiles = (P + P1 - 1) / P1;
    // Both tiles of the NEXT pixel run travel into registers behind the current run's MFMAs (ea + eb <= 2048 float4 elements:
    // wgrad_1x1_tile): element e of a thread is row pr, channel quad q of the x tile (e < ea) or of the dy tile.
    f32x4 v[8];
    unsigned okm = 0;
    // in_stats: (mean, rstd) of this thread's channel quad of the tile's image travel with the tile (256 is a multiple of the quads per
    // pixel, so every x element of a thread has the same quad): two 16-byte loads per tile instead of eight scalar loads per element
    f32x4 st_lo = {0.f, 1.f, 0.f, 1.f}, st_hi = {0.f, 1.f, 0.f, 1.f};
    const bool st_fast = in_stats && (256 % qa) == 0;
    // where this thread's eight elements live: tile-invariant, so the divisions happen once (they were a quarter of the kernel's
    // instruction stream when redone per tile in both the fetch and the commit)
    int dst8[8], pr8[8];
    unsigned rel8[8], isa_m = 0, use_m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = 256 * k + tid;
        const bool isa = e < ea, use = e < ea + eb;
        const int eb_ = isa ? 0 : e - ea;
        const int pr = isa ? (qa == 1 ? e : (int)__umulhi((unsigned)e, m_qa)) : (qb == 1 ? eb_ : (int)__umulhi((unsigned)eb_, m_qb));
        const int q = isa ? e - pr * qa : eb_ - pr * qb;
        pr8[k] = use ? pr : 0;
        dst8[k] = isa ? pr * lda + 4 * q : P1 * lda + pr * ldb + 4 * q;
        rel8[k] = use ? (isa ? (unsigned)pr * g_ctot + g_coff + 4 * q : (unsigned)pr * h_ctot + h_coff + 4 * q) : 0u;
        isa_m |= (unsigned)isa << k;
        use_m |= (unsigned)use << k;
    }
    auto fetch = [&](long tile) {
        const long p0 = tile * P1;
        const float* xa = x + (size_t)p0 * g_ctot;
        const float* yb = dy + (size_t)p0 * h_ctot;
        okm = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = ((use_m >> k) & 1) && p0 + pr8[k] < P;
            okm |= (unsigned)ok << k;
            const f32x4* src = reinterpret_cast<const f32x4*>(ok ? (((isa_m >> k) & 1) ? xa : yb) + rel8[k] : x);
            // 16 -> 16 channels at 256 x 256: two 268 MB streams read once -- non-temporal (3.8 -> 5.3 TB/s); the smaller tensors of the
            // other widths are partly served by the 256 MB cache and lose with the hint
            v[k] = (MF * NW == 1) ? __builtin_nontemporal_load(src) : *src;
        }
        if (st_fast) {
            const float* st = in_stats + ((size_t)(p0 / HW) * g_ctot + g_coff + 4 * (tid % qa)) * 2;
            st_lo = *reinterpret_cast<const f32x4*>(st);
            st_hi = *reinterpret_cast<const f32x4*>(st + 4);
        }
    };
    if ((long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((use_m >> k) & 1) {
                const bool isa = (isa_m >> k) & 1;
                f32x4 w = ((okm >> k) & 1) ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
                if (in_stats && isa && ((okm >> k) & 1)) {
                    // x is the RAW tensor in front of InstanceNorm + ReLU: normalise here, as norm_apply_kernel would have (the host
                    // guarantees that a pixel run stays inside one image: HW % P1 == 0)
                    if (st_fast) {
                        w[0] = fmaxf((w[0] - st_lo[0]) * st_lo[1], 0.f);
                        w[1] = fmaxf((w[1] - st_lo[2]) * st_lo[3], 0.f);
                        w[2] = fmaxf((w[2] - st_hi[0]) * st_hi[1], 0.f);
                        w[3] = fmaxf((w[3] - st_hi[2]) * st_hi[3], 0.f);
                    } else {
                        const int q = (256 * k + tid) % qa;
                        const float* st = in_stats + ((size_t)(tile * P1 / HW) * g_ctot + g_coff + 4 * q) * 2;
#pragma unroll
                        for (int c = 0; c < 4; ++c) w[c] = fmaxf((w[c] - st[2 * c]) * st[2 * c + 1], 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(&smem[dst8[k]]) = w;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        if (with_bias) {  // column sums of the dy tile: every thread takes the pixels p = bgrp, bgrp + nbg, ... of channel bch
            if (bsplit) {
#pragma unroll 4
                for (int p = bgrp; p < P1; p += nbg) bsum += Bs[p * ldb + bch];
            } else if (tid < Ch) {
#pragma unroll 8
                for (int p = 0; p < P1; ++p) bsum += Bs[p * ldb + tid];
            }
        }
        // ---- MFMA: K = the tile's 32 pixels, 4 per step ---------------------------------------------------------------------------
#pragma unroll 2
        for (int k0 = 4 * kpart; k0 < P1; k0 += 4 * KS) {
            float af[MF], bf[NW];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) af[mf] = As[(k0 + g) * lda + 16 * mf + i];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int nf = min(wslot + 4 * k, NF - 1);
                bf[k] = Bs[(k0 + g) * ldb + 16 * nf + i];
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int k = 0; k < NW; ++k) acc[mf][k] = mfma16(af[mf], bf[k], acc[mf][k]);
        }
        __syncthreads();  // every wave is done with both tiles
    }
    if (KS > 1) {  // sum the k-parts of a fragment through LDS in a fixed order (NW == 1 here)
        __syncthreads();
        float* red = smem;  // [wave][MF][64 lanes][4]
        if (kpart > 0) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) *reinterpret_cast<f32x4*>(&red[((wave * MF + mf) * 64 + lane) * 4]) = acc[mf][0];
        }
        __syncthreads();
        if (kpart == 0) {
            for (int kp = 1; kp < KS; ++kp) {
                const int w2 = kp * (4 / KS) + wslot;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) acc[mf][0] += *reinterpret_cast<const f32x4*>(&red[((w2 * MF + mf) * 64 + lane) * 4]);
            }
        }
    }
    // partial slab [Cg][Ch] (+ [Ch] bias tail), the layout wgrad_reduce_kernel expects for T = 1
    const size_t pstride = (size_t)Cg * Ch + (with_bias ? Ch : 0);
    float* out = partial + (size_t)blockIdx.x * pstride;
    if (with_bias && bsplit) {  // groups 0, 1, ... of a channel, in that order
        __syncthreads();
        float* red = smem;
        red[tid] = bsum;
        __syncthreads();
        if (tid < Ch) {
            float t = 0.f;
            for (int k = 0; k < nbg; ++k) t += red[k * Ch + tid];
            bsum = t;
        }
        __syncthreads();
    }
    if (with_bias && tid < Ch) out[(size_t)Cg * Ch + tid] = bsum;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int nf = wslot + 4 * k;
        if (nf >= NF || kpart != 0) continue;
        const int hch = 16 * nf + i;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gch = 16 * mf + 4 * g + e;
                if (gch < Cg && hch < Ch) out[(size_t)gch * Ch + hch] = acc[mf][k][e];
            }
    }
}

// *S in: the slabs the workspace holds (<= 1024); out: the workgroups launched = what the CUs hold at once (a persistent kernel with
// 1024 workgroups at three per CU ran one full wave of 768 and a tail of 256: 16 -> 16 channels 3.5 -> 5.3 TB/s with the tail gone)
static int w11_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus;
}

template <int MF, int NW>
static int launch_wgrad_1x1(const WGradArgs& a, long P, int* S, hipStream_t st, const float* in_stats = nullptr) {
    const int NF = cdiv(a.Ch, 16), P1 = wgrad_1x1_tile(a.Cg, a.Ch);
    const size_t lds = (size_t)P1 * (w1x1_ld(MF) + w1x1_ld(NF)) * sizeof(float);
    {
        static size_t c_lds = 0;
        static int c_occ = 0;
        if (c_lds != lds || !c_occ) {
            int nb = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&wgrad_1x1_kernel<MF, NW>), 256, lds) != hipSuccess || nb < 1) nb = 1;
            c_occ = nb > 4 ? 4 : nb;
            c_lds = lds;
        }
        const int fit = w11_cus() * c_occ;
        if (*S > fit) *S = fit;
    }
    MSTG_LAUNCH((wgrad_1x1_kernel<MF, NW>), dim3(*S), dim3(256), lds, st, a.g, a.h, a.partial, P, a.Cg, a.g_ctot, a.g_coff, a.Ch,
                       a.h_ctot, a.h_coff, a.with_bias, P1, in_stats, a.hH * a.hW);
    MSTG_CHECK_LAUNCH("wgrad_1x1_kernel");
    return MSTG_OK;
}

// eligible: 1x1, stride 1, both tensors NHWC with 16-byte aligned channel slices, <= 64 input and <= 192 output channels
static bool wgrad_1x1_ok(const WGradArgs& a) {
    const char* e = env_get(ENV_WGRAD_1X1);
    if (e && e[0] == '0') return false;
    return a.T == 1 && a.stride == 1 && !a.g_nchw && !a.h_nchw && ((a.g_ctot | a.g_coff | a.Cg | a.h_ctot | a.h_coff | a.Ch) & 3) == 0 &&
           a.Cg * a.Ch >= 256;
}
static int wgrad_1x1_splits(const WGradArgs& a) {
    const int P1 = wgrad_1x1_tile(a.Cg, a.Ch);
    const long P = (long)a.N * a.hH * a.hW, ntiles = (P + P1 - 1) / P1;
    long S = 1024;
    if (S > ntiles) S = ntiles;
    return (int)S;
}
// Wider layers (the class-default generator has 128 -> 384 and 256 -> 768) are cut into (<= 64 input, <= 192 output) channel
// blocks, one launch each on the same pixels: x is then read Ch/192 times and dy Cg/64 times, against Ch/32 and Cg/16 times in
// the generic kernel.  Block widths are multiples of 16 and as even as the channel counts allow.
struct W11Chunks {
    int ng, nh, wg, wh;  // number of blocks and block width (last one may be narrower) along Cg / Ch
};
static W11Chunks wgrad_1x1_chunks(const WGradArgs& a) {
    W11Chunks c;
    c.ng = cdiv(a.Cg, 64);
    c.nh = cdiv(a.Ch, 192);
    c.wg = c.ng == 1 ? a.Cg : cdiv(cdiv(a.Cg, c.ng), 16) * 16;
    c.wh = c.nh == 1 ? a.Ch : cdiv(cdiv(a.Ch, c.nh), 16) * 16;
    c.ng = cdiv(a.Cg, c.wg);
    c.nh = cdiv(a.Ch, c.wh);
    return c;
}
static WGradArgs wgrad_1x1_block(const WGradArgs& a, const W11Chunks& c, int ig, int ih) {
    WGradArgs b = a;
    b.g_coff = a.g_coff + ig * c.wg;
    b.Cg = a.Cg - ig * c.wg < c.wg ? a.Cg - ig * c.wg : c.wg;
    b.h_coff = a.h_coff + ih * c.wh;
    b.Ch = a.Ch - ih * c.wh < c.wh ? a.Ch - ih * c.wh : c.wh;
    b.with_bias = a.with_bias && ig == 0;
    return b;
}
static size_t wgrad_1x1_workspace(const WGradArgs& a) {
    const W11Chunks c = wgrad_1x1_chunks(a);
    const WGradArgs b = wgrad_1x1_block(a, c, 0, 0);  // the widest block
    return (size_t)wgrad_1x1_splits(b) * ((size_t)b.Cg * b.Ch + b.Ch) * sizeof(float);
}

// dw[gch*s_g + hch*s_h + t] = sum_split partial[split][t][gch][hch]  (+ dbias[hch] from the tail of each split's block).
// A workgroup owns 64 consecutive outputs: thread (row = tid >> 4, q = tid & 15) sums outputs 4q .. 4q+3 of every 16th split with
// one 16-byte load per split (a row's 16 threads read 256 contiguous bytes; the round-2 kernel read 64), the 16 rows are combined
// through LDS in a fixed order: per output the summation order is split row, row + 16, ... then rows 0..15 -- the order the
// narrower kernel used, so results are bit-identical to it and independent of scheduling.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           float* __restrict__ dbias, int S, int T, int Cg, int Ch, int s_g,
                                                           int s_h, int pstride, int accumulate) {
    __shared__ f32x4 sh[16][17];
    const int q = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int total = T * Cg * Ch, nout = pstride;
    const int idx = (blockIdx.x * 16 + q) * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if ((pstride & 3) == 0) {
        if (idx < nout) {  // eight loads in flight per trip; the additions keep the order sp = row, row + 16, ...
            int sp = row;
            for (; sp + 7 * 16 < S; sp += 8 * 16) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(partial + (size_t)(sp + 16 * k) * pstride + idx);
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += v[k];
            }
            for (; sp < S; sp += 16) sum += *reinterpret_cast<const f32x4*>(partial + (size_t)sp * pstride + idx);
        }
    } else if (idx < nout) {  // slab stride not a multiple of four floats (three-channel layers): 4-byte loads, same batching and order
        const int nc = nout - idx < 4 ? nout - idx : 4;
        int sp = row;
        for (; sp + 7 * 16 < S; sp += 8 * 16) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) v[k][c] = c < nc ? partial[(size_t)(sp + 16 * k) * pstride + idx + c] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; sp < S; sp += 16)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nc) sum[c] += partial[(size_t)sp * pstride + idx + c];
    }
    sh[row][q] = sum;
    __syncthreads();
    if (row == 0 && idx < nout) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k) r += sh[k][q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int id = idx + c;
            if (id < total) {
                const int hch = id % Ch, gch = (id / Ch) % Cg, t = id / (Ch * Cg);
                float* o = dw + (size_t)gch * s_g + (size_t)hch * s_h + t;
                *o = accumulate ? *o + r[c] : r[c];
            } else if (id < nout && dbias) {
                dbias[id - total] = accumulate ? dbias[id - total] + r[c] : r[c];
            }
        }
    }
}

struct WGradPlan {
    int S, TGn, nz, nfh, tg;
    size_t ws_bytes, lds;
};

static WGradPlan plan_wgrad(const WGradArgs& a) {
    WGradPlan p;
    if (a.Teff == 1) p.tg = 1;
    else if (a.Teff <= 3) p.tg = 3;
    else if (a.Teff <= 9) p.tg = 9;
    else if (a.Teff == 14) p.tg = 14;
    else p.tg = 16;
    // 16 taps x 2 column fragments = 128 accumulator + 152 other registers = one wave per SIMD; one fragment keeps three
    p.nfh = (a.mode == MODE_DPACK || a.Ch <= 16 || p.tg >= 14) ? 1 : 2;
    p.nz = cdiv(a.Teff, p.tg);
    p.TGn = cdiv(a.Teff, p.nz);
    const int ny = a.n_gchunks * (a.mode == MODE_DPACK ? 1 : cdiv(a.Ch, 16 * p.nfh));
    int S = (p.tg >= 16 ? 1024 : 256 * WG_WAVES) / (ny * p.nz);  // as many workgroups as the register budget keeps resident
    // keep the partial slabs small: they are written once and re-read once by the reduce kernel
    const size_t slab = ((size_t)a.T * a.Cg * a.Ch + a.Ch) * sizeof(float);
    while (S > 256 && (size_t)S * slab > ((size_t)48 << 20)) S >>= 1;
    if (S < 1) S = 1;
    if (S > a.ntiles) S = a.ntiles;
    p.S = S;
    p.ws_bytes = (size_t)S * slab;
    const int bnp = 16 * p.nfh + 4;
    const size_t htile = a.mode == MODE_DPACK ? (size_t)WT_H * a.htw * 4 : (size_t)128 * bnp;
    p.lds = ((size_t)((a.PH * a.PW * a.ckp + 3) & ~3) + htile) * sizeof(float);
    const size_t red = (size_t)p.tg * p.nfh * 256 * sizeof(float);
    if (red > p.lds) p.lds = red;
    return p;
}

template <int TG, int NFH>
static int launch_wgrad_t(WGradArgs& a, const WGradPlan& p, hipStream_t st) {
    if (p.lds > 160 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "wgrad: LDS patch too large");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TG, NFH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(wgrad)");
        attr_set = true;
    }
    const int ny = a.n_gchunks * (a.mode == MODE_DPACK ? 1 : cdiv(a.Ch, 16 * NFH));
    dim3 grid(p.S, ny, p.nz);
    MSTG_LAUNCH((wgrad_kernel<TG, NFH>), grid, dim3(256), p.lds, st, a);
    MSTG_CHECK_LAUNCH("wgrad_kernel");
    return MSTG_OK;
}

static int wgrad_ts_max_ch() {
    const char* e = env_get(ENV_WGRAD_TS_MAXCH);
    return e ? atoi(e) : 128;  // measured: 8-18 % faster than the pixel-split kernel up to 128 grid channels (two fragments per workgroup)
}

struct TsPlan {
    int TH, NFHT, UW, S, ngroups;
    size_t lds, ws_bytes;
};

static TsPlan plan_ts(WGradArgs& a) {
    TsPlan p;
    p.TH = a.stride >= 2 ? 4 : 8;
    // re-derive the tile geometry for this tile height
    a.tiles_y = cdiv(a.hH, p.TH);
    a.ntiles = a.N * a.tiles_x * a.tiles_y;
    a.PH = (p.TH - 1) * a.stride + (a.KH - 1) * a.dil + 1;
    const int nfh_all = a.mode == MODE_DPACK ? 1 : cdiv(a.Ch, 16);
    p.NFHT = nfh_all > 4 ? 4 : nfh_all;
    while (p.NFHT > 1 && a.Teff * p.NFHT > 32) --p.NFHT;  // at most 8 units per wave (16 would need > 128 VGPRs)
    p.ngroups = cdiv(nfh_all, p.NFHT);
    const int U = a.Teff * p.NFHT;
    p.UW = U <= 16 ? 4 : (U <= 32 ? 8 : 16);
    const int ny = a.n_gchunks * p.ngroups;
    int S = 768 / ny;
    const size_t slab = ((size_t)a.T * a.Cg * a.Ch + a.Ch) * sizeof(float);
    while (S > 128 && (size_t)S * slab > ((size_t)48 << 20)) S >>= 1;
    if (S < 1) S = 1;
    if (S > a.ntiles) S = a.ntiles;
    p.S = S;
    p.ws_bytes = (size_t)S * slab;
    const size_t htile = a.mode == MODE_DPACK ? (size_t)p.TH * a.htw * 4 : (size_t)p.TH * 16 * (16 * p.NFHT + 4);
    p.lds = ((size_t)((a.PH * a.PW * a.ckp + 3) & ~3) + htile) * sizeof(float);
    return p;
}

template <int UW>
static int launch_ts_t(WGradArgs& a, const TsPlan& p, hipStream_t st) {
    if (p.lds > 160 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "wgrad: LDS patch too large");
    if (4 % p.NFHT) return fail_arg(MSTG_E_UNSUPPORTED, "wgrad: tap-split kernel needs 1, 2 or 4 grid-channel fragments per workgroup");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ts_kernel<UW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(wgrad_ts)");
        attr_set = true;
    }
    dim3 grid(p.S, a.n_gchunks * p.ngroups, 1);
    MSTG_LAUNCH((wgrad_ts_kernel<UW>), grid, dim3(256), p.lds, st, a, p.TH, p.NFHT);
    MSTG_CHECK_LAUNCH("wgrad_ts_kernel");
    return MSTG_OK;
}

// the persistent kernel takes the 4x4 stride-2 family on aligned, unsliced NHWC tensors (call after plan_ts set PH for TH = 4)
static bool wp_ok(const WGradArgs& a) {
    const char* e = env_get(ENV_P32);
    if (e && e[0] == '0') return false;
    return a.Teff == 16 && a.mode == MODE_PLAIN && a.stride == 2 && a.pad == 1 && a.dil == 1 && !a.g_nchw && !a.h_nchw && !a.g_coff &&
           !a.h_coff && a.g_ctot == a.Cg && a.h_ctot == a.Ch && a.Cg % 16 == 0 && a.Ch % (16 * WP_NFHT) == 0 && a.gH == 2 * a.hH &&
           a.gW == 2 * a.hW && (size_t)a.gH * a.gW * a.Cg * 4 < ((size_t)1 << 32) && (size_t)a.hH * a.hW * a.Ch * 4 < ((size_t)1 << 32);
}
struct WpPlan { int S, ny; size_t lds, ws_bytes; };
static WpPlan wp_plan(const WGradArgs& a) {  // a: after plan_ts (tiles for TH = 4)
    WpPlan p;
    p.ny = a.n_gchunks * (a.Ch / (16 * WP_NFHT));
    p.lds = ((size_t)((a.PH * a.PW * WP_CKP + 3) & ~3) + (size_t)WP_TH * 16 * WP_BNP) * sizeof(float);
    const size_t slab = ((size_t)a.T * a.Cg * a.Ch + a.Ch) * sizeof(float);
    int S = 1024 / p.ny;  // at most four workgroups per CU; the launch clamps to what the kernel's registers / LDS really allow
    while (S > 128 && (size_t)S * slab > ((size_t)48 << 20)) S >>= 1;
    if (S < 1) S = 1;
    if (S > a.ntiles) S = a.ntiles;
    p.S = S;
    p.ws_bytes = (size_t)S * slab;
    return p;
}
static int launch_wp(WGradArgs& a, WpPlan& p, hipStream_t st) {
    static int occ = 0;
    static size_t occ_lds = 0;
    if (!occ || occ_lds != p.lds) {
        const void* kptr = reinterpret_cast<const void*>(&wgrad_p32_kernel);
        if (p.lds > 64 * 1024) (void)hipFuncSetAttribute(kptr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int nb = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kptr, 256, p.lds) != hipSuccess || nb < 1) nb = 1;
        occ = nb > 4 ? 4 : nb;
        occ_lds = p.lds;
    }
    int occ_ = occ;
    { const char* e = env_get(ENV_P32_OCC); if (e && atoi(e) >= 1 && atoi(e) < occ_) occ_ = atoi(e); }
    int S = (256 * occ_) / p.ny;
    if (S > p.S) S = p.S;
    if (S < 1) S = 1;
    p.S = S;
    MSTG_LAUNCH(wgrad_p32_kernel, dim3(S, p.ny, 1), dim3(256), p.lds, st, a);
    MSTG_CHECK_LAUNCH("wgrad_p32_kernel");
    return MSTG_OK;
}

// the two 7x7 layers with a 3-channel side (see wgrad7_kernel); everything else about them is fixed by the module
static bool w7_ok(const WGradArgs& a) {
    const char* e = env_get(ENV_P32);
    if (e && e[0] == '0') return false;
    if (a.KH != 7 || a.KW != 7 || a.stride != 1 || a.pad != 3 || a.dil != 1 || a.gH != a.hH || a.gW != a.hW) return false;
    if ((size_t)a.hH * a.hW * 16 >= ((size_t)1 << 30)) return false;
    if (a.mode == MODE_PACKX) return a.g_nchw && a.Cg <= 3 && !a.h_nchw && a.Ch == 16 && a.h_ctot == 16 && a.h_coff == 0;
    if (a.mode == MODE_DPACK) return a.h_nchw && a.Ch <= 3 && !a.g_nchw && a.Cg == 16 && a.g_ctot == 16 && a.g_coff == 0;
    return false;
}
static size_t w7_lds(const WGradArgs& a) {
    const size_t stage = ((size_t)((14 * 23 * (a.mode == MODE_PACKX ? 4 : 20) + 3) & ~3) + (a.mode == MODE_PACKX ? 128 * 20 : 8 * 20 * 4)) * sizeof(float);
    const size_t red = (size_t)14 * 256 * sizeof(float);
    return stage > red ? stage : red;
}
static int w7_splits(const WGradArgs& a) {
    int cap = 768;
    { const char* e = env_get(ENV_P32_OCC); if (e && atoi(e) >= 1 && atoi(e) < 3) cap = 256 * atoi(e); }
    return a.ntiles < cap ? a.ntiles : cap;
}
static int launch_w7(WGradArgs& a, int S, hipStream_t st) {
    const size_t lds = w7_lds(a);
    if (a.mode == MODE_PACKX) MSTG_LAUNCH((wgrad7_kernel<MODE_PACKX>), dim3(S), dim3(256), lds, st, a);
    else MSTG_LAUNCH((wgrad7_kernel<MODE_DPACK>), dim3(S), dim3(256), lds, st, a);
    MSTG_CHECK_LAUNCH("wgrad7_kernel");
    return MSTG_OK;
}

static int fill_wgrad_args(const mstg_conv_desc* d, const float* x, const float* dy, WGradArgs& a) {
    a.N = d->N;
    a.KH = d->KH; a.KW = d->KW; a.dil = d->dil; a.T = d->KH * d->KW;
    if (d->transposed) {  // gathered = dY (stride 2, pad 1 over the module-input grid), grid tensor = x
        a.g = dy; a.gH = d->Ho; a.gW = d->Wo; a.g_ctot = d->y_ctot; a.g_coff = d->y_coff; a.g_nchw = d->y_nchw; a.Cg = d->Cout;
        a.h = x; a.hH = d->H; a.hW = d->W; a.h_ctot = d->x_ctot; a.h_coff = d->x_coff; a.h_nchw = d->x_nchw; a.Ch = d->Cin;
        a.stride = 2; a.pad = 1;
    } else {
        a.g = x; a.gH = d->H; a.gW = d->W; a.g_ctot = d->x_ctot; a.g_coff = d->x_coff; a.g_nchw = d->x_nchw; a.Cg = d->Cin;
        a.h = dy; a.hH = d->Ho; a.hW = d->Wo; a.h_ctot = d->y_ctot; a.h_coff = d->y_coff; a.h_nchw = d->y_nchw; a.Ch = d->Cout;
        a.stride = d->stride; a.pad = d->pad;
    }
    // operand packing: see the file header
    a.mode = MODE_PLAIN;
    if (a.Cg <= 4 && a.dil == 1 && a.KW > 1) a.mode = MODE_PACKX;
    else if (a.Ch <= 4 && a.dil == 1 && a.stride == 1 && a.KW > 1) a.mode = MODE_DPACK;
    { const char* e = env_get(ENV_WGRAD_PLAIN); if (e && e[0] == '1') a.mode = MODE_PLAIN; }
    a.tapsx = cdiv(a.KW, 4);
    a.xshift = a.mode == MODE_DPACK ? 3 : 0;
    a.htw = a.mode == MODE_DPACK ? 20 : 16;
    a.ckp = a.mode == MODE_PACKX ? 4 : 20;
    a.Teff = a.mode == MODE_PLAIN ? a.T : a.KH * a.tapsx;
    a.tiles_x = cdiv(a.hW + a.xshift, WT_W);
    a.tiles_y = cdiv(a.hH, WT_H);
    a.ntiles = a.N * a.tiles_x * a.tiles_y;
    a.PH = (WT_H - 1) * a.stride + (a.KH - 1) * a.dil + 1;
    a.PW = a.mode == MODE_PLAIN ? (WT_W - 1) * a.stride + (a.KW - 1) * a.dil + 1 : (WT_W - 1) * a.stride + 4 * a.tapsx;
    a.n_gchunks = a.mode == MODE_PACKX ? 1 : cdiv(a.Cg, 16);
    if (a.g_nchw && a.Cg > 16) return fail_arg(MSTG_E_UNSUPPORTED, "wgrad: NCHW gathered tensor supports <= 16 channels");
    return MSTG_OK;
}

int check_desc(const mstg_conv_desc* d);  // conv_igemm.hip

}  // namespace mstg

using namespace mstg;

const char* igemm_kernel_name(const mstg_conv_desc* d, int pass);  // conv_igemm.hip

extern "C" const char* mstg_conv2d_kernel_name(const mstg_conv_desc* d, int pass) {
    if (pass != 2) return igemm_kernel_name(d, pass);
    static thread_local char name[64];
    WGradArgs a{};
    if (check_desc(d) || fill_wgrad_args(d, nullptr, nullptr, a)) return "";
    const bool use_ts = a.Teff == 16 && a.mode == MODE_PLAIN && a.Ch > 16 && a.Ch <= wgrad_ts_max_ch();
    if (wgrad_1x1_ok(a)) {
        const WGradArgs b = wgrad_1x1_block(a, wgrad_1x1_chunks(a), 0, 0);
        snprintf(name, sizeof(name), "wgrad_1x1_kernel<%d, %d>", cdiv(b.Cg, 16), cdiv(cdiv(b.Ch, 16), 4));
    } else if (use_ts) {
        const int uw = plan_ts(a).UW;
        if (wp_ok(a)) snprintf(name, sizeof(name), "wgrad_p32_kernel");
        else snprintf(name, sizeof(name), "wgrad_ts_kernel<%d>", uw);
    } else if (w7_ok(a)) {
        snprintf(name, sizeof(name), "wgrad7_kernel<%d>", a.mode);
    } else {
        const WGradPlan p = plan_wgrad(a);
        snprintf(name, sizeof(name), "wgrad_kernel<%d, %d>", p.tg, p.nfh);
    }
    return name;
}

extern "C" size_t mstg_conv2d_wgrad_workspace_bytes(const mstg_conv_desc* d) {
    if (check_desc(d)) return 0;
    WGradArgs a{};
    if (fill_wgrad_args(d, nullptr, nullptr, a)) return 0;
    const size_t w_old = plan_wgrad(a).ws_bytes;
    size_t w_ts = (a.Teff == 16 && a.mode == MODE_PLAIN) ? plan_ts(a).ws_bytes : 0;  // the only shapes the tap-split kernel takes
    if (w_ts && wp_ok(a)) { const size_t w_p = wp_plan(a).ws_bytes; if (w_p > w_ts) w_ts = w_p; }
    const size_t w_11 = wgrad_1x1_ok(a) ? wgrad_1x1_workspace(a) : 0;
    size_t w = w_old > w_ts ? w_old : w_ts;
    if (w7_ok(a)) { const size_t w_7 = (size_t)w7_splits(a) * ((size_t)a.T * a.Cg * a.Ch + a.Ch) * sizeof(float); if (w_7 > w) w = w_7; }
    return w > w_11 ? w : w_11;
}

static bool wgrad_ts_path(const WGradArgs& a) {
    return a.Teff == 16 && a.mode == MODE_PLAIN && a.Ch > 16 && a.Ch <= wgrad_ts_max_ch() &&
           !(env_get(ENV_WGRAD_OLD) && env_get(ENV_WGRAD_OLD)[0] == '1');
}
static bool wgrad_norm_ok(const WGradArgs& a) {  // normalise-on-load exists in the persistent 4x4 stride-2 kernel and in the 1x1 kernel
    if (!wgrad_1x1_ok(a)) return wgrad_ts_path(a) && wp_ok(a);
    if (a.g_coff != 0) return false;  // 1x1: for pixel runs that stay inside one image
    const W11Chunks c = wgrad_1x1_chunks(a);
    for (int ig = 0; ig < c.ng; ++ig)
        for (int ih = 0; ih < c.nh; ++ih) {
            const WGradArgs b = wgrad_1x1_block(a, c, ig, ih);
            if ((a.hH * a.hW) % wgrad_1x1_tile(b.Cg, b.Ch)) return false;
        }
    return true;
}

extern "C" int mstg_conv2d_wgrad_norm_supported(const mstg_conv_desc* d) {
    if (check_desc(d) || d->transposed) return 0;
    WGradArgs a{};
    if (fill_wgrad_args(d, nullptr, nullptr, a)) return 0;
    return wgrad_norm_ok(a) ? 1 : 0;
}

static int conv2d_wgrad_impl(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* dy, float* dw, float* dbias,
                             void* workspace, size_t workspace_bytes, void* stream);

extern "C" int mstg_conv2d_wgrad(const mstg_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl(d, x, nullptr, dy, dw, dbias, workspace, workspace_bytes, stream);
}

extern "C" int mstg_conv2d_wgrad_norm(const mstg_conv_desc* d, const float* x_raw, const float* in_stats, const float* dy, float* dw,
                                      float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!in_stats) return fail_arg(MSTG_E_BADARG, "conv_wgrad_norm: null statistics");
    return conv2d_wgrad_impl(d, x_raw, in_stats, dy, dw, dbias, workspace, workspace_bytes, stream);
}

static int conv2d_wgrad_impl(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* dy, float* dw, float* dbias,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!x || !dy || !dw || !workspace) return fail_arg(MSTG_E_BADARG, "conv_wgrad: null pointer");
    if (dbias && d->transposed)
        return fail_arg(MSTG_E_UNSUPPORTED, "conv_wgrad: fused bias gradient only for Conv2d (use mstg_channel_sum for ConvTranspose2d)");
    WGradArgs a{};
    if (int rc = fill_wgrad_args(d, x, dy, a)) return rc;
    hipStream_t st = (hipStream_t)stream;
    a.partial = (float*)workspace;
    a.with_bias = dbias != nullptr;
    int S;
    // measured on MI355X: the tap-split kernel wins where a workgroup gets 32 units (16 taps x 2 column fragments: the
    // stride-2 / transposed 4x4 layers with 17..32 grid channels); the pixel-split kernel elsewhere
    const bool use_ts = a.Teff == 16 && a.mode == MODE_PLAIN && a.Ch > 16 && a.Ch <= wgrad_ts_max_ch();
    if (in_stats && (d->transposed || !wgrad_norm_ok(a)))
        return fail_arg(MSTG_E_UNSUPPORTED, "conv_wgrad_norm: only the layers mstg_conv2d_wgrad_norm_supported() reports");
    if (wgrad_1x1_ok(a)) {
        if (workspace_bytes < wgrad_1x1_workspace(a)) return fail_arg(MSTG_E_WORKSPACE, "conv_wgrad: workspace too small");
        const long P = (long)a.N * a.hH * a.hW;
        const W11Chunks c = wgrad_1x1_chunks(a);
        const int s_g = 1, s_h = d->transposed ? d->Cout : d->Cin;  // T == 1
        for (int ig = 0; ig < c.ng; ++ig)
            for (int ih = 0; ih < c.nh; ++ih) {
                const WGradArgs b = wgrad_1x1_block(a, c, ig, ih);
                int Sb = wgrad_1x1_splits(b);
                const int MF = cdiv(b.Cg, 16), NW = cdiv(cdiv(b.Ch, 16), 4);
                int rc = MSTG_E_UNSUPPORTED;
#define MSTG_W11(M_, N_) if (MF == M_ && NW == N_) rc = launch_wgrad_1x1<M_, N_>(b, P, &Sb, st, in_stats);
                MSTG_W11(1, 1) MSTG_W11(1, 2) MSTG_W11(1, 3) MSTG_W11(2, 1) MSTG_W11(2, 2) MSTG_W11(2, 3) MSTG_W11(3, 1) MSTG_W11(3, 2)
                MSTG_W11(3, 3) MSTG_W11(4, 1) MSTG_W11(4, 2) MSTG_W11(4, 3)
#undef MSTG_W11
                if (rc) return rc;
                // the partial slabs are reused by the next block: stream order keeps this reduce ahead of the next launch
                const int pstride = b.Cg * b.Ch + (b.with_bias ? b.Ch : 0);
                MSTG_LAUNCH(wgrad_reduce_kernel, dim3(cdiv(pstride, 64)), dim3(256), 0, st, a.partial,
                                   dw + (size_t)(ig * c.wg) * s_g + (size_t)(ih * c.wh) * s_h, b.with_bias ? dbias + ih * c.wh : nullptr, Sb, 1,
                                   b.Cg, b.Ch, s_g, s_h, pstride, d->accumulate);
                MSTG_CHECK_LAUNCH("wgrad_reduce_kernel");
            }
        return MSTG_OK;
    } else if (use_ts && !(env_get(ENV_WGRAD_OLD) && env_get(ENV_WGRAD_OLD)[0] == '1')) {
        const TsPlan p = plan_ts(a);
        if (wp_ok(a)) {  // persistent, prefetching form of the same kernel
            a.g_stats = in_stats;
            WpPlan q = wp_plan(a);
            if (workspace_bytes < q.ws_bytes) return fail_arg(MSTG_E_WORKSPACE, "conv_wgrad: workspace too small");
            if (int rc = launch_wp(a, q, st)) return rc;
            S = q.S;
        } else {
            if (workspace_bytes < p.ws_bytes) return fail_arg(MSTG_E_WORKSPACE, "conv_wgrad: workspace too small");
            int rc = p.UW == 4 ? launch_ts_t<4>(a, p, st) : (p.UW == 8 ? launch_ts_t<8>(a, p, st) : launch_ts_t<16>(a, p, st));
            if (rc) return rc;
            S = p.S;
        }
    } else if (w7_ok(a)) {
        S = w7_splits(a);
        if (workspace_bytes < (size_t)S * ((size_t)a.T * a.Cg * a.Ch + a.Ch) * sizeof(float)) return fail_arg(MSTG_E_WORKSPACE, "conv_wgrad: workspace too small");
        if (int rc = launch_w7(a, S, st)) return rc;
    } else {
        const WGradPlan p = plan_wgrad(a);
        if (workspace_bytes < p.ws_bytes) return fail_arg(MSTG_E_WORKSPACE, "conv_wgrad: workspace too small");
        a.TGn = p.TGn;
        int rc = MSTG_E_UNSUPPORTED;
        if (p.tg == 1 && p.nfh == 1) rc = launch_wgrad_t<1, 1>(a, p, st);
        else if (p.tg == 1 && p.nfh == 2) rc = launch_wgrad_t<1, 2>(a, p, st);
        else if (p.tg == 3 && p.nfh == 1) rc = launch_wgrad_t<3, 1>(a, p, st);
        else if (p.tg == 3 && p.nfh == 2) rc = launch_wgrad_t<3, 2>(a, p, st);
        else if (p.tg == 14 && p.nfh == 1) rc = launch_wgrad_t<14, 1>(a, p, st);
        else if (p.tg == 9 && p.nfh == 1) rc = launch_wgrad_t<9, 1>(a, p, st);
        else if (p.tg == 9 && p.nfh == 2) rc = launch_wgrad_t<9, 2>(a, p, st);
        else if (p.tg == 16 && p.nfh == 1) rc = launch_wgrad_t<16, 1>(a, p, st);
        else if (p.tg == 16 && p.nfh == 2) rc = launch_wgrad_t<16, 2>(a, p, st);
        if (rc) return rc;
        S = p.S;
    }
    const int T = a.T, total = T * a.Cg * a.Ch;
    // Conv2d: dw[co][ci][t] (gch = ci, hch = co) ; ConvTranspose2d: dw[ci][co][t] (gch = co, hch = ci)
    const int s_g = T;
    const int s_h = d->transposed ? d->Cout * T : d->Cin * T;
    const int pstride = total + (a.with_bias ? a.Ch : 0);
    MSTG_LAUNCH(wgrad_reduce_kernel, dim3(cdiv(pstride, 64)), dim3(256), 0, st, a.partial, dw, dbias, S, T, a.Cg, a.Ch, s_g, s_h,
                       pstride, d->accumulate);
    MSTG_CHECK_LAUNCH("wgrad_reduce_kernel");
    return MSTG_OK;
}

#ifdef MSTG_STAMPS
extern "C" int mstg_debug_stamps_wgrad(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mstg::g_wg_stamps), sizeof(unsigned long long) * 2 * 64 * 8);
}
#endif
