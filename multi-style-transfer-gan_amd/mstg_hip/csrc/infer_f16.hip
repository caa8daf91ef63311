// Inference-only forward path in fp16 storage / fp16 MFMA / fp32 accumulation (BASELINE config #5: batch 64 at 1024x1024,
// the HBM-bound point of the workload).  Callers: the reference's inference scripts (direct_transform.py:44-79,
// batch_process_images.py:176-217) run EnhancedGenerator.forward under torch.no_grad(); this file is that forward with
//   * activations NHWC fp16 between kernels, every accumulation / statistic / softmax in fp32;
//   * InstanceNorm taken out of HBM: a convolution's epilogue emits per-tile sum / sum-of-squares of what it writes
//     (f16_norm_finalize_kernel folds them into mean / rstd per (image, channel)), and the CONSUMER applies
//     (x - mean) * rstd -> ReLU while it stages its input tile into LDS.  The only normalisation that still runs as a pass of
//     its own is the one in front of the residual add (f16_norm_residual_kernel: two reads, one write);
//   * ONE implicit-GEMM kernel for every convolution shape of the generator -- 7x7 stem (NCHW fp32 image in), 4x4 stride 2,
//     ConvTranspose 4x4 stride 2 as four parity classes, 1x1, the four MultiScaleBlock branches as one 25-tap sparse filter,
//     7x7 head with tanh (NCHW out) -- driven by a small table of K-steps built on the host.
//
// GEMM view: v_mfma_f32_16x16x32_f16, A = packed filter (16 output channels x 32 k), B = LDS patch (32 k x 16 pixels of one
// output row), D = 4 consecutive output channels of one pixel per lane -> one 8-byte NHWC store per lane and fragment.
// A K-step is four lane groups of 8 k-elements; a group is 8 consecutive input channels of one tap (or, for the 3-channel stem
// padded to 4, two horizontally adjacent taps x 4 channels): one 16-byte LDS read per lane.  The LDS pixel stride is
// 2*Cin + 16 bytes, which makes those reads bank-conflict free for Cin = 16 / 32 / 64 (MI355X_MICROARCH.md, LDS).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "common.h"

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace mstg {

constexpr int F16_MAX_STEPS = 64;
constexpr int F16_MAX_SEG = 8;
constexpr int F16_TW = 16;  // tile width in output pixels of the compute grid (= one MFMA N fragment)
constexpr int F16_MAX_GRID = 1024;  // persistent workgroups: 256 CUs x at most 4
constexpr int F16_TABLE_BYTES = (64 + F16_MAX_SEG) * 16;  // K-step table + segment table at the start of the kernel's LDS

// how the pack kernel fills one half (4 k-elements) of a lane group
enum : int8_t { PK_ZERO = 0, PK_CONV = 1, PK_CONVT = 2, PK_MS_CENTER = 3, PK_MS_RING1 = 4 /* +0,1,2 = branches 2,3,4 */, PK_DPACK = 7 };
struct PackHalf { int8_t mode, ky, kx; int8_t pad; int16_t cb; };
struct PackTable {
    PackHalf h[F16_MAX_STEPS][4][2];
    uint8_t pair_step[F16_MAX_STEPS * 4], pair_frag[F16_MAX_STEPS * 4];  // stored fragment -> (K-step, output fragment)
};

struct F16Plan {
    int nsteps, ncls;
    int cls_begin[5];
    // segments: runs of K-steps that feed the same set of output fragments; a class (ConvTranspose parity, or the whole layer)
    // is a run of segments that share accumulators: `first` starts them from the bias, `last` runs the epilogue
    int nseg;
    struct Seg { int16_t s0, s1; uint8_t mask, first, last, cls; int wbase; } seg[F16_MAX_SEG];
    uint16_t koff[F16_MAX_STEPS][4];  // byte offset of the group's 16 bytes relative to the lane's pixel base in the patch
    uint8_t fmask[F16_MAX_STEPS];     // output-channel fragments the step feeds
    uint16_t wofs[F16_MAX_STEPS];     // index of the step's first stored filter fragment (only the fragments in fmask are stored)
    int nwfrag, wlds;                 // stored fragments in all; 1: the kernel keeps the filter in LDS
    unsigned m_pw;                    // magic multiplier for division by PW
    unsigned m_ntile, m_tx;           // ... by tiles per image and by tiles per row (exact while t * d < 2^32: checked by the host)
    int npf;                          // 16-byte patch elements per thread
    int8_t cls_oy[4], cls_ox[4];      // ConvTranspose: output parity of the class
    int PH, PW, pixstride;            // patch rows / cols / bytes per pixel
    int oy0, ox0;                     // patch origin = tile origin * stride + (oy0, ox0)
    int stride, up;                   // compute-grid stride in the source; up = 1: outputs at 2*y+oy, 2*x+ox
    int NF, TH;
    int tstep, dpack;                 // tile advance in compute-grid columns (16; 13 with dpack), <= 4 output channels as 4 shifts x 4 channels
};

struct F16ConvArgs {
    const void* x;          // NHWC fp16, or NCHW fp32 (src = 1)
    const h16* res;         // src = 2: the layer's input is relu((x - mean) * rstd) + res, formed while staging (both NHWC fp16)
    h16* y;                 // NHWC fp16, or NCHW fp16 (dst = 1)
    const h16* wpk;         // [step][frag][lane][8]
    const float* bias;      // [16 * NF]
    const float* in_stats;  // nullable [N][Cin][2]: normalise + ReLU while staging
    float* partial;         // nullable [N][tiles][2][16 * NF]
    int N, H, W, Cin, Ho, Wo, Cout;
    int Gh, Gw;             // compute grid per image (= Ho x Wo, or H x W for up)
    int tiles_x, tiles_y;
    int act;
    int dbg;  // experiments (MSTG_F16_DBG): 1 skip K-steps, 2 skip stores, 4 skip patch fetch, 8 skip commit
};

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum_f(float v) {
    v += dppf<0xB1>(v);
    v += dppf<0x4E>(v);
    v += dppf<0x141>(v);
    v += dppf<0x140>(v);
    return v;
}

// -------------------------------------------------------------------------------------------------------------------------
// filter packing: fp32 PyTorch layouts -> [step][frag][lane][8] fp16, + fp32 bias vector padded to 16 * NF
// -------------------------------------------------------------------------------------------------------------------------
struct PackSrc {
    const float* w[4];
    const float* b[4];
    int Cin, Cout, KH, KW, c4;  // c4: channels per MultiScaleBlock branch
};

__global__ void f16_pack_kernel(PackTable t, PackSrc s, int nwfrag, int NF, h16* __restrict__ wpk, float* __restrict__ bias) {
    const int total = nwfrag * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, f = t.pair_frag[e >> 9], step = t.pair_step[e >> 9];
        const int m = lane & 15, g = lane >> 4, hf = j >> 2, jj = j & 3;
        const PackHalf ph = t.h[step][g][hf];
        const int co = 16 * f + m, ci = ph.cb + jj;
        float v = 0.f;
        if (ph.mode != PK_ZERO && (co < s.Cout || ph.mode == PK_DPACK) && ci < s.Cin && ph.cb >= 0) {
            if (ph.mode == PK_CONV) {
                v = s.w[0][((size_t)(co * s.Cin + ci) * s.KH + ph.ky) * s.KW + ph.kx];
            } else if (ph.mode == PK_CONVT) {
                v = s.w[0][((size_t)(ci * s.Cout + co) * s.KH + ph.ky) * s.KW + ph.kx];
            } else if (ph.mode == PK_DPACK) {  // row m = 4 * delta + c carries the real tap kx = 4 j + 3 - delta of output channel c
                const int delta = m >> 2, c = m & 3, kx = 4 * ph.kx + 3 - delta;
                v = (f == 0 && c < s.Cout && kx >= 0 && kx < s.KW) ? s.w[0][((size_t)(c * s.Cin + ci) * s.KH + ph.ky) * s.KW + kx] : 0.f;
            } else {
                const int br = co / s.c4, cb = co - br * s.c4;  // branch of this output channel
                if (ph.mode == PK_MS_CENTER) {
                    v = br == 0 ? s.w[0][cb * s.Cin + ci] : s.w[br][((size_t)(cb * s.Cin + ci) * 3 + 1) * 3 + 1];
                } else if (br == ph.mode - PK_MS_RING1 + 1) {
                    v = s.w[br][((size_t)(cb * s.Cin + ci) * 3 + ph.ky) * 3 + ph.kx];
                }
            }
        }
        wpk[e] = (h16)v;
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 16 * NF; c += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (c < s.Cout) {
            if (s.c4 > 0) { const int br = c / s.c4; v = s.b[br] ? s.b[br][c - br * s.c4] : 0.f; }
            else v = s.b[0] ? s.b[0][c] : 0.f;
        }
        bias[c] = v;
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// the convolution kernel.  Persistent workgroups of 256 threads; wave w computes rows RPW*w .. RPW*w + RPW-1 of a (4*RPW) x 16
// tile of the compute grid for all 16*NF output channels.  Per workgroup: the packed filter is copied into LDS once (when it
// fits next to the patch: p.wlds), then for every tile
//     registers (prefetched patch) -> [InstanceNorm + ReLU] -> LDS | barrier | issue the NEXT tile's patch loads |
//     K-steps (operands of step s+1 are read while the MFMAs of step s run) | epilogue: bias, statistics, store | barrier
// so the global round trip of a patch hides behind the previous tile's arithmetic.
// -------------------------------------------------------------------------------------------------------------------------
template <unsigned M> struct MaskT { static constexpr unsigned value = M; };
struct TrueT { static constexpr bool value = true; };
struct FalseT { static constexpr bool value = false; };

template <int NPF, int SRC>
struct PatchRegs {
    h16x8 v[SRC != 1 ? NPF : 1];
    h16x8 r2[SRC == 2 ? NPF : 1];  // the residual operand of the same patch elements
    f32x4 f[SRC == 1 ? NPF : 1];
    unsigned okmask;
};

// What a thread stages is the same for every tile: element k of thread tid is patch pixel (r, c) [and channel octet o], i.e. a
// fixed byte offset from the patch origin in the source and a fixed LDS address.  Computed once per kernel; an interior tile
// (the whole patch inside the image: almost all of them) then costs one load and one LDS write per element and no arithmetic.
template <int NPF>
struct PatchGeom {
    unsigned rel[NPF];   // byte offset in the source image relative to the patch origin (SRC 1: within one plane)
    unsigned vmask;      // element exists
};

template <int NPF, int SRC>
__device__ __forceinline__ void patch_geom(const F16ConvArgs& a, const F16Plan& p, int tid, PatchGeom<NPF>& Gm) {
    Gm.vmask = 0;
    const int oct = SRC != 1 ? (a.Cin >> 3) : 1, o = tid & (oct - 1), sh = oct == 1 ? 0 : (oct == 2 ? 1 : (oct == 4 ? 2 : 3));
    const int total = p.PH * p.PW * oct;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int e = 256 * k + tid;
        const int pix = e >> sh;
        const int r = (int)__umulhi((unsigned)pix, p.m_pw), c = pix - r * p.PW;
        if (e < total) Gm.vmask |= 1u << k;
        if (SRC != 1) {
            Gm.rel[k] = (unsigned)(((r * a.W + c) * a.Cin + 8 * o) * 2);
        } else {
            Gm.rel[k] = (unsigned)((r * a.W + c) * 4);
        }
    }
}

// tile index -> (image, tile row, tile column); consecutive tiles stay on one XCD (blocks b and b + 8 share an XCD)
__device__ __forceinline__ int persistent_tile(int it, int b, int G) { return it * G + (b & 7) * (G >> 3) + (b >> 3); }

template <int NPF, int SRC>
__device__ __forceinline__ void patch_fetch(const F16ConvArgs& a, const F16Plan& p, int t, int TH, int tid, const PatchGeom<NPF>& Gm,
                                            PatchRegs<NPF, SRC>& R) {
    const int ntile = a.tiles_x * a.tiles_y;
    const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
    const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
    const int sy0 = ty * TH * p.stride + p.oy0, sx0 = tx * p.tstep * p.stride + p.ox0;
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + p.PH <= a.H && sx0 + p.PW <= a.W;  // uniform
    if (SRC != 1) {
        const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * a.H * a.W * a.Cin * 2;
        const char* rimg = reinterpret_cast<const char*>(a.res) + (size_t)n * a.H * a.W * a.Cin * 2;
        if (interior) {
            const size_t oo = ((size_t)sy0 * a.W + sx0) * a.Cin * 2;
            const char* org = img + oo;
            R.okmask = Gm.vmask;
#pragma unroll
            for (int k = 0; k < NPF; ++k) R.v[k] = *reinterpret_cast<const h16x8*>(org + (((Gm.vmask >> k) & 1) ? Gm.rel[k] : 0u));
            if (SRC == 2) {
#pragma unroll
                for (int k = 0; k < NPF; ++k) R.r2[k] = *reinterpret_cast<const h16x8*>(rimg + oo + (((Gm.vmask >> k) & 1) ? Gm.rel[k] : 0u));
            }
        } else {
            R.okmask = 0;
            const long org = ((long)sy0 * a.W + sx0) * a.Cin * 2;
            const int oct = a.Cin >> 3, sh = oct == 2 ? 1 : (oct == 4 ? 2 : 3);
#pragma unroll
            for (int k = 0; k < NPF; ++k) {  // border tile (rare): recompute (r, c) instead of keeping them in registers
                const int pix = (256 * k + tid) >> sh;
                const int r = (int)__umulhi((unsigned)pix, p.m_pw), c = pix - r * p.PW;
                const int iy = sy0 + r, ix = sx0 + c;
                const bool ok = ((Gm.vmask >> k) & 1) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                R.okmask |= (unsigned)ok << k;
                R.v[k] = *reinterpret_cast<const h16x8*>(img + (ok ? org + (long)Gm.rel[k] : 0L));
                if (SRC == 2) R.r2[k] = *reinterpret_cast<const h16x8*>(rimg + (ok ? org + (long)Gm.rel[k] : 0L));
            }
        }
    } else {
        const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * a.Cin * a.H * a.W * 4;
        const size_t plane = (size_t)a.H * a.W * 4;
        R.okmask = 0;
        const long org = ((long)sy0 * a.W + sx0) * 4;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int e = 256 * k + tid;
            const int r = (int)__umulhi((unsigned)e, p.m_pw), c = e - r * p.PW;
            const int iy = sy0 + r, ix = sx0 + c;
            const bool ok = ((Gm.vmask >> k) & 1) && (interior || ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W));
            R.okmask |= (unsigned)ok << k;
            const char* q = img + (ok ? org + (long)Gm.rel[k] : 0L);
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            w[0] = *reinterpret_cast<const float*>(q);
            if (a.Cin > 1) w[1] = *reinterpret_cast<const float*>(q + plane);
            if (a.Cin > 2) w[2] = *reinterpret_cast<const float*>(q + 2 * plane);
            if (a.Cin > 3) w[3] = *reinterpret_cast<const float*>(q + 3 * plane);
            R.f[k] = w;
        }
    }
}

template <int NPF, int SRC>
__device__ __forceinline__ void patch_commit(const F16ConvArgs& a, const F16Plan& p, int n, int tid, const PatchGeom<NPF>& Gm,
                                             const PatchRegs<NPF, SRC>& R, unsigned char* patch) {
    if (SRC != 1) {
        const int oct = a.Cin >> 3, o = tid & (oct - 1), sh = oct == 2 ? 1 : (oct == 4 ? 2 : 3);
        const bool norm = SRC == 2 || a.in_stats != nullptr;  // src = 2 always carries statistics (checked by the host)
        float sc[8], nb[8];  // (x - mean) * rstd = x * sc + nb
        if (norm) {
            const float* st = a.in_stats + ((size_t)n * a.Cin + 8 * o) * 2;
#pragma unroll
            for (int c = 0; c < 8; ++c) { sc[c] = st[2 * c + 1]; nb[c] = -st[2 * c] * st[2 * c + 1]; }
        }
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if ((Gm.vmask >> k) & 1) {
                h16x8 w = R.v[k];
                if (SRC == 2) {  // the arithmetic of f16_norm_residual_kernel, so that folding the pass changes no bit
                    const h16x8 rr = R.r2[k];
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = (h16)(fmaxf(fmaf((float)w[c], sc[c], nb[c]), 0.f) + (float)rr[c]);
                } else if (norm) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = (h16)fmaxf(fmaf((float)w[c], sc[c], nb[c]), 0.f);
                }
                if (!((R.okmask >> k) & 1)) w = h16x8{0, 0, 0, 0, 0, 0, 0, 0};  // zero padding applies to the NORMALISED activation
                *reinterpret_cast<h16x8*>(patch + (unsigned)(((256 * k + tid) >> sh) * p.pixstride + 16 * o)) = w;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            if ((Gm.vmask >> k) & 1) {
                h16x4 w;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) w[ch] = ((R.okmask >> k) & 1) ? (h16)R.f[k][ch] : (h16)0;
                *reinterpret_cast<h16x4*>(patch + (unsigned)((256 * k + tid) * 8)) = w;
            }
        }
    }
}

// register budget: the light single-fragment kernels want four workgroups per CU (<= 128 VGPRs); the others are LDS-bound to 1-2
// (three workgroups per CU for the light kernels, <= 168 registers: left alone the compiler takes 160-212, told so it needs 97-161
// without spilling; the variants that would spill keep the default)
template <int RPW, int NF, int SRC, int DST, int NPF, bool WLDS>
__global__ __launch_bounds__(256, (NF == 1 && RPW <= 4 && NPF <= (SRC == 2 ? 4 : 6)) ? 3 : ((NF == 2 && NPF <= 6 && !(RPW == 4 && SRC == 2 && NPF == 6)) || (NF == 4 && RPW == 2 && NPF <= 4 && SRC != 2)) ? 2 : 1) void conv_f16_kernel(const F16ConvArgs a, const F16Plan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 4 * RPW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int ntile = a.tiles_x * a.tiles_y, total_tiles = a.N * ntile, G = gridDim.x;
    const bool want_stats = a.partial != nullptr;
    // LDS: [K-step table: 64 x 16 B | class table: 4 x 16 B] [filter (p.wlds)] [patch] ; the statistics scratch aliases the patch.
    // The tables live in LDS, not in the kernel-argument segment: a dynamically indexed kernarg read is a scalar load whose wait
    // (lgkmcnt(0): scalar loads return out of order) also drains every LDS read in flight -- measured 700-3000 cycles per K-step.
    unsigned char* wl = smem + F16_TABLE_BYTES;
    unsigned char* patch = wl + (WLDS ? (size_t)p.nwfrag * 1024 : 0);
    if (tid < F16_MAX_STEPS) {
        uint16_t* e = reinterpret_cast<uint16_t*>(smem + 16 * tid);
        e[0] = p.koff[tid][0]; e[1] = p.koff[tid][1]; e[2] = p.koff[tid][2]; e[3] = p.koff[tid][3];
    } else if (tid < F16_MAX_STEPS + F16_MAX_SEG) {
        const int c = tid - F16_MAX_STEPS;
        int* e = reinterpret_cast<int*>(smem + 16 * tid);
        const int cl = p.seg[c].cls;
        e[0] = p.seg[c].s0; e[1] = p.seg[c].s1;
        e[2] = p.seg[c].mask | (p.seg[c].first << 8) | (p.seg[c].last << 9) | (p.cls_oy[cl] << 10) | (p.cls_ox[cl] << 11);
        e[3] = p.seg[c].wbase;
    }
    if (WLDS) {
        const int nchunk = p.nwfrag * 64;
        for (int e = tid; e < nchunk; e += 256) reinterpret_cast<h16x8*>(wl)[e] = reinterpret_cast<const h16x8*>(a.wpk)[e];
    }
    // the address space must be static: a pointer that is "LDS or global" compiles to flat loads, whose waits drain both counters
    const h16x8* wglob = reinterpret_cast<const h16x8*>(a.wpk) + lane;
    const unsigned char* wlds_lane = wl + 16 * lane;
    const int nseg = p.nseg, up = p.up, pixstride = p.pixstride;
    unsigned base[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) base[r] = (unsigned)(((RPW * wv + r) * p.stride * p.PW + nl * p.stride) * p.pixstride);
    (void)pixstride;
    f32x4 b4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) b4[f] = (DST == 1 && p.dpack) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.bias + 16 * f + 4 * g);

    // statistics of what this workgroup writes, per channel: kept in registers across the tiles of one image and flushed to
    // partial[image][workgroup] when the image changes (the buffer is zeroed by the host: not every workgroup sees every image)
    float ssum[NF][4], ssq[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) ssum[f][q] = ssq[f][q] = 0.f;
    auto flush_stats = [&](int n_img) {  // call only between tiles (the scratch aliases the patch); ends with a barrier
        float* red = reinterpret_cast<float*>(patch);  // [4 waves][2][16 * NF]
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = row16_sum_f(ssum[f][q]), s2 = row16_sum_f(ssq[f][q]);
                if (nl == 0) {
                    red[(wv * 2 + 0) * 16 * NF + 16 * f + 4 * g + q] = s1;
                    red[(wv * 2 + 1) * 16 * NF + 16 * f + 4 * g + q] = s2;
                }
                ssum[f][q] = ssq[f][q] = 0.f;
            }
        __syncthreads();
        if (tid < 2 * 16 * NF) {
            const float sm = red[tid] + red[2 * 16 * NF + tid] + red[4 * 16 * NF + tid] + red[6 * 16 * NF + tid];
            a.partial[((size_t)n_img * G + blockIdx.x) * 2 * 16 * NF + tid] = sm;
        }
        __syncthreads();
    };

    PatchGeom<NPF> Gm;
    patch_geom<NPF, SRC>(a, p, tid, Gm);
    // this lane's output elements relative to the tile's first output pixel (NHWC destination): row r of the wave, column nl
    const size_t out_row = (size_t)(up ? 2 : 1) * a.Wo * a.Cout * 2;  // bytes between this wave's consecutive rows
    PatchRegs<NPF, SRC> R;
    int it = 0, cur_n = -1;
    int t = persistent_tile(it, blockIdx.x, G);
    if (t < total_tiles) patch_fetch<NPF, SRC>(a, p, t, TH, tid, Gm, R);
    while (t < total_tiles) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
        const int gy0 = ty * TH, gx0 = tx * p.tstep;
        if (want_stats && n != cur_n) {
            if (cur_n >= 0) flush_stats(cur_n);
            cur_n = n;
        }
        if (!(a.dbg & 8)) patch_commit<NPF, SRC>(a, p, n, tid, Gm, R, patch);
        __syncthreads();
        const int tnext = persistent_tile(it + 1, blockIdx.x, G);
        if (tnext < total_tiles && !(a.dbg & 4)) patch_fetch<NPF, SRC>(a, p, tnext, TH, tid, Gm, R);

        f32x4 acc[RPW][NF];
        for (int sg = 0; sg < nseg; ++sg) {
            const int4 ci = *reinterpret_cast<const int4*>(smem + 16 * (F16_MAX_STEPS + sg));
            const int s0 = __builtin_amdgcn_readfirstlane(ci.x), s1 = __builtin_amdgcn_readfirstlane(ci.y);
            const int sflags = __builtin_amdgcn_readfirstlane(ci.z), wbase = __builtin_amdgcn_readfirstlane(ci.w);
            const int smask = sflags & 0xff;
            // a single output fragment never splits into segments: first / last are compile-time there (keeps the light kernels lean)
            const bool first = NF == 1 ? true : (bool)((sflags >> 8) & 1), last = NF == 1 ? true : (bool)((sflags >> 9) & 1);
            // One segment = K-steps that all feed the fragment set MASK (compile-time inside `run`): no per-step branches, no per-step
            // metadata beyond the lane's patch offset.  Three-stage pipeline over LDS (reads return in order: counted waits only):
            //   patch offset of step s+2 | operand fragments of step s+1 | MFMAs of step s
            auto run = [&](auto MT) {
                constexpr unsigned MASK = decltype(MT)::value;
                constexpr int NLIVE = __builtin_popcount(MASK);
                auto load_ko = [&](int s) -> unsigned { return *reinterpret_cast<const uint16_t*>(smem + 16 * s + 2 * g); };
                auto load_ops = [&](unsigned ko, int s, h16x8 (&bf)[RPW], h16x8 (&af)[NF]) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r) {
                        if (SRC == 1) {  // 8-byte pixels: the 16 bytes of a tap pair are only 8-byte aligned -> two 8-byte reads
                            const h16x4 lo = *reinterpret_cast<const h16x4*>(patch + base[r] + ko);
                            const h16x4 hi = *reinterpret_cast<const h16x4*>(patch + base[r] + ko + 8);
                            bf[r] = h16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        } else {
                            bf[r] = *reinterpret_cast<const h16x8*>(patch + base[r] + ko);
                        }
                    }
                    int wi = wbase + (s - s0) * NLIVE;
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        if ((MASK >> f) & 1) {
                            if (WLDS) af[f] = *reinterpret_cast<const h16x8*>(wlds_lane + (size_t)wi * 1024);
                            else af[f] = wglob[(size_t)wi * 64];
                            ++wi;
                        }
                };
                auto mma_step = [&](const h16x8 (&bf)[RPW], const h16x8 (&af)[NF], auto from_bias) {
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        if ((MASK >> f) & 1) {
#pragma unroll
                            for (int r = 0; r < RPW; ++r)
                                acc[r][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[f], bf[r], decltype(from_bias)::value ? b4[f] : acc[r][f], 0, 0, 0);
                        }
                };
                auto clip = [&](int s) { return s < s1 ? s : s1 - 1; };  // past the end: re-read the last step (loaded, never used)
                h16x8 bA[RPW], aA[NF], bB[RPW], aB[NF];
                unsigned k0 = load_ko(s0), k1 = load_ko(clip(s0 + 1));
                load_ops(k0, s0, bA, aA);
                k0 = load_ko(clip(s0 + 2));
                load_ops(k1, clip(s0 + 1), bB, aB);
                if (first) mma_step(bA, aA, TrueT{});   // accumulators start from the bias (MFMA C operand)
                else mma_step(bA, aA, FalseT{});
                int s = s0 + 1;  // invariant: B holds the operands of step s (if s < s1), k0 the offset of step s + 1
                for (; s + 1 < s1; s += 2) {
                    k1 = load_ko(clip(s + 2));
                    load_ops(k0, s + 1, bA, aA);
                    mma_step(bB, aB, FalseT{});
                    k0 = load_ko(clip(s + 3));
                    load_ops(k1, clip(s + 2), bB, aB);
                    mma_step(bA, aA, FalseT{});
                }
                if (s < s1) mma_step(bB, aB, FalseT{});
            };
            if (first) {  // fragments the first segment does not feed start from the bias as well
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    if (!((smask >> f) & 1) || (a.dbg & 1)) {
#pragma unroll
                        for (int r = 0; r < RPW; ++r) acc[r][f] = b4[f];
                    }
            }
            if (!(a.dbg & 1)) {
                if (NF == 1) run(MaskT<1>{});
                else if (NF == 2) {
                    if (smask == 3) run(MaskT<3>{});
                    else if (smask == 1) run(MaskT<1>{});
                    else run(MaskT<2>{});
                } else {
                    if (smask == 15) run(MaskT<15>{});
                    else if (smask == 1) run(MaskT<1>{});
                    else if (smask == 2) run(MaskT<2>{});
                    else if (smask == 4) run(MaskT<4>{});
                    else if (smask == 8) run(MaskT<8>{});
                    else if (smask == 3) run(MaskT<3>{});
                    else run(MaskT<12>{});
                }
            }
            if (!last) continue;
            // ---- epilogue of this class: lane holds output channels 16f + 4g + {0..3} of compute-grid pixel (gy, gx0 + nl) -------
            const int oyc = up ? (sflags >> 10) & 1 : 0, oxc = up ? (sflags >> 11) & 1 : 0, mul = up ? 2 : 1;
            if (DST == 0 && gy0 + TH <= a.Gh && gx0 + F16_TW <= a.Gw && (a.Cout & 15) == 0 && !(a.dbg & 2)) {
                // whole tile inside the image (almost all of them): no per-lane bounds, addresses = tile base + a per-lane constant
                char* ybase = reinterpret_cast<char*>(a.y) + ((((size_t)n * a.Ho + gy0 * mul + oyc) * a.Wo + gx0 * mul + oxc) * a.Cout) * 2;
                const unsigned out_off0 = (unsigned)(((RPW * wv * mul) * a.Wo + nl * mul) * a.Cout + 4 * g) * 2u;  // recomputed per tile: a register less
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const f32x4 v = acc[r][f];
                        if (want_stats) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { ssum[f][q] += v[q]; ssq[f][q] += v[q] * v[q]; }
                        }
                        h16x4 hv;
#pragma unroll
                        for (int q = 0; q < 4; ++q) hv[q] = (h16)v[q];
                        *reinterpret_cast<h16x4*>(ybase + r * out_row + out_off0 + 32 * f) = hv;
                    }
                continue;
            }
            if (DST == 1 && p.dpack) {
                // 4-shift packing: accumulator row 4 delta + c of column p is a partial sum of output column p + delta.
                // y[q][c] = sum_delta D[delta][q - delta] through LDS (the patch is dead once every wave is past its K-steps)
                __syncthreads();
                float* comb = reinterpret_cast<float*>(patch) + wv * (256 * RPW);  // [row][delta][p][4]
#pragma unroll
                for (int r = 0; r < RPW; ++r) *reinterpret_cast<f32x4*>(&comb[((r * 4 + g) * 16 + nl) * 4]) = acc[r][0];
                __syncthreads();
                if (g < RPW && nl < 13 && !(a.dbg & 2)) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(a.bias);
#pragma unroll
                    for (int d = 0; d < 4; ++d) v += *reinterpret_cast<const f32x4*>(&comb[((g * 4 + d) * 16 + nl + 3 - d) * 4]);
                    const int gy = gy0 + RPW * wv + g, gx = gx0 + nl;
                    if (gy < a.Gh && gx < a.Gw) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < a.Cout) {
                                const float o = a.act == MSTG_ACT_TANH ? 1.f - 2.f * __frcp_rn(__expf(2.f * v[q]) + 1.f) : v[q];
                                a.y[(((size_t)n * a.Cout + q) * a.Ho + gy) * a.Wo + gx] = (h16)o;
                            }
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int gy = gy0 + RPW * wv + r, gx = gx0 + nl;
                const bool inside = gy < a.Gh && gx < a.Gw;
                const int oy = gy * mul + oyc, ox = gx * mul + oxc;
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const f32x4 v = acc[r][f];
                    if (want_stats && inside) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { ssum[f][q] += v[q]; ssq[f][q] += v[q] * v[q]; }
                    }
                    if (inside && !(a.dbg & 2)) {
                        if (DST == 0) {
                            if (16 * f + 4 * g < a.Cout) {
                                h16x4 hv;
#pragma unroll
                                for (int q = 0; q < 4; ++q) hv[q] = (h16)v[q];
                                *reinterpret_cast<h16x4*>(a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + 16 * f + 4 * g) = hv;
                            }
                        } else if (f == 0 && g == 0) {  // NCHW output with <= 4 channels: planes
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (q < a.Cout) {
                                    // tanh(x) = 1 - 2 / (exp(2x) + 1): exact limits at +-inf, ~1e-6 absolute error (the output is fp16)
                                    const float o = a.act == MSTG_ACT_TANH ? 1.f - 2.f * __frcp_rn(__expf(2.f * v[q]) + 1.f) : v[q];
                                    a.y[(((size_t)n * a.Cout + q) * a.Ho + oy) * a.Wo + ox] = (h16)o;
                                }
                        }
                    }
                }
            }
        }
        __syncthreads();  // every wave is done with the patch
        ++it;
        t = tnext;
    }
    if (want_stats && cur_n >= 0) flush_stats(cur_n);
}

// -------------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions without the LDS patch.  For a 1x1 filter the MFMA's B fragment IS the memory layout: lane (pixel n, group g)
// needs channels 8g .. 8g+7 of its pixel = 16 contiguous bytes of the NHWC tensor.  conv_f16_kernel spends ~45 % of such a
// layer in per-tile fixed costs (LDS commit, table reads, pipeline prologue, barriers) for ONE K-step per tile; here a wave loads
// its fragments straight from global memory (next tile's in flight behind the current tile's MFMAs), normalises them in registers,
// keeps the whole filter in registers and only touches LDS to reduce the statistics.  Same tiles, same tile order, same partial rows
// as conv_f16_kernel, so the statistics stay batch-independent bit for bit.
// -------------------------------------------------------------------------------------------------------------------------
template <int NF, int KS>  // output fragments (Cout / 16), K-steps (ceil(Cin / 32))
__global__ __launch_bounds__(256) void conv1x1_f16_kernel(const F16ConvArgs a, const F16Plan p) {
    __shared__ float red[8 * 16 * NF];
    constexpr int RPW = 4, TH = 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nl = lane & 15, g = lane >> 4;
    const int ntile = a.tiles_x * a.tiles_y, total_tiles = a.N * ntile, G = gridDim.x;
    const bool want_stats = a.partial != nullptr, norm = a.in_stats != nullptr;
    h16x8 af[KS][NF];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int f = 0; f < NF; ++f) af[s][f] = reinterpret_cast<const h16x8*>(a.wpk)[(size_t)(s * NF + f) * 64 + lane];
    f32x4 b4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) b4[f] = *reinterpret_cast<const f32x4*>(a.bias + 16 * f + 4 * g);
    bool chan_ok[KS];  // this lane's 8-channel group of step s exists (Cin = 16 fills only groups 0 and 1 of the single step)
#pragma unroll
    for (int s = 0; s < KS; ++s) chan_ok[s] = 8 * (4 * s + g) < a.Cin;
    float ssum[NF][4], ssq[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) ssum[f][q] = ssq[f][q] = 0.f;
    auto flush_stats = [&](int n_img) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = row16_sum_f(ssum[f][q]), s2 = row16_sum_f(ssq[f][q]);
                if (nl == 0) {
                    red[(wv * 2 + 0) * 16 * NF + 16 * f + 4 * g + q] = s1;
                    red[(wv * 2 + 1) * 16 * NF + 16 * f + 4 * g + q] = s2;
                }
                ssum[f][q] = ssq[f][q] = 0.f;
            }
        __syncthreads();
        if (tid < 2 * 16 * NF)
            a.partial[((size_t)n_img * G + blockIdx.x) * 2 * 16 * NF + tid] = red[tid] + red[2 * 16 * NF + tid] + red[4 * 16 * NF + tid] + red[6 * 16 * NF + tid];
        __syncthreads();
    };
    struct Frag { h16x8 v[RPW][KS]; unsigned ok; };
    auto fetch = [&](int t, Frag& F) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
        const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * a.H * a.W * a.Cin * 2;
        F.ok = 0;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int gy = ty * TH + RPW * wv + r, gx = tx * F16_TW + nl;
            const bool in = gy < a.H && gx < a.W;
            F.ok |= (unsigned)in << r;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bool ld = in && chan_ok[s];
                F.v[r][s] = *reinterpret_cast<const h16x8*>(img + (ld ? ((size_t)(gy * a.W + gx) * a.Cin + 8 * (4 * s + g)) * 2 : 0));
                if (!ld) F.v[r][s] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };
    Frag F;
    int it = 0, cur_n = -1, cur_in = -1;
    float sc[KS][8], nb[KS][8];  // (x - mean) * rstd = x * sc + nb for this lane's channels of the current image
    int t = persistent_tile(it, blockIdx.x, G);
    if (t < total_tiles) fetch(t, F);
    while (t < total_tiles) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
        if (want_stats && n != cur_n) {
            if (cur_n >= 0) flush_stats(cur_n);
            cur_n = n;
        }
        if (norm && n != cur_in) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float* st = a.in_stats + ((size_t)n * a.Cin + (chan_ok[s] ? 8 * (4 * s + g) : 0)) * 2;
#pragma unroll
                for (int c = 0; c < 8; ++c) { sc[s][c] = st[2 * c + 1]; nb[s][c] = -st[2 * c] * st[2 * c + 1]; }
            }
            cur_in = n;
        }
        // this tile's operands out of the prefetch registers (normalised), then the next tile's loads go out
        h16x8 bf[RPW][KS];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                h16x8 w = F.v[r][s];
                if (norm) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c] = (h16)fmaxf(fmaf((float)w[c], sc[s][c], nb[s][c]), 0.f);
                    if (!((F.ok >> r) & 1) || !chan_ok[s]) w = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
                bf[r][s] = w;
            }
        const unsigned okr = F.ok;
        const int tnext = persistent_tile(it + 1, blockIdx.x, G);
        if (tnext < total_tiles) fetch(tnext, F);
        f32x4 acc[RPW][NF];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[r][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s][f], bf[r][s], s == 0 ? b4[f] : acc[r][f], 0, 0, 0);
        char* yimg = reinterpret_cast<char*>(a.y) + (size_t)n * a.Ho * a.Wo * a.Cout * 2;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (!((okr >> r) & 1)) continue;
            const int gy = ty * TH + RPW * wv + r, gx = tx * F16_TW + nl;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 v = acc[r][f];
                if (want_stats) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ssum[f][q] += v[q]; ssq[f][q] += v[q] * v[q]; }
                }
                h16x4 hv;
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[q] = (h16)v[q];
                *reinterpret_cast<h16x4*>(yimg + ((size_t)(gy * a.Wo + gx) * a.Cout + 16 * f + 4 * g) * 2) = hv;
            }
        }
        ++it;
        t = tnext;
    }
    if (want_stats && cur_n >= 0) flush_stats(cur_n);
}

// partial [N][tiles][2][CP] -> stats [N][C][2] = (mean, rstd); one workgroup per image, double accumulation, fixed order
__global__ __launch_bounds__(256) void f16_norm_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                                int tiles, int CP, int C, float count) {
    __shared__ double red[256 * 2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int lanes_per_c = 256 / CP;            // CP in {16, 32, 64} -> 16 / 8 / 4 threads share a channel
    const int c = tid % CP, sub = tid / CP;
    double s1 = 0.0, s2 = 0.0;
    {   // an image has thousands of tile rows at 1024 x 1024: sixteen loads in flight per trip, the additions in their old order
        int tl = sub;
        for (; tl + 7 * lanes_per_c < tiles; tl += 8 * lanes_per_c) {
            float u[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* pp = partial + ((size_t)n * tiles + tl + k * lanes_per_c) * 2 * CP;
                u[k] = pp[c];
                w[k] = pp[CP + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1 += (double)u[k];
                s2 += (double)w[k];
            }
        }
        for (; tl < tiles; tl += lanes_per_c) {
            const float* pp = partial + ((size_t)n * tiles + tl) * 2 * CP;
            s1 += (double)pp[c];
            s2 += (double)pp[CP + c];
        }
    }
    red[tid * 2] = s1;
    red[tid * 2 + 1] = s2;
    __syncthreads();
    if (sub == 0 && c < C) {
        for (int k = 1; k < lanes_per_c; ++k) { s1 += red[(k * CP + c) * 2]; s2 += red[(k * CP + c) * 2 + 1]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((size_t)n * C + c) * 2] = (float)mean;
        stats[((size_t)n * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// y = relu((x - mean) * rstd) + residual   (the block's `+ x`, enhanced_generator.py:84), NHWC fp16, C in {16, 32, 64}
__global__ __launch_bounds__(256) void f16_norm_residual_kernel(const h16* __restrict__ x, const h16* __restrict__ res,
                                                                const float* __restrict__ stats, h16* __restrict__ y, size_t HW, int C,
                                                                int blocks_per_image) {
    const int n = blockIdx.x / blocks_per_image, b = blockIdx.x - n * blocks_per_image;
    const int oct = C >> 3, o = threadIdx.x & (oct - 1);
    float sc[8], nb[8];  // (x - mean) * rstd = x * sc + nb: the form (and the bits) of the consumers' normalise-on-load
    const float* st = stats + ((size_t)n * C + 8 * o) * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] = st[2 * c + 1]; nb[c] = -st[2 * c] * st[2 * c + 1]; }
    const size_t total = HW * oct;  // 16-byte chunks of this image
    const size_t base = (size_t)n * total;
    const h16x8* xv = reinterpret_cast<const h16x8*>(x) + base;
    const h16x8* rv = res ? reinterpret_cast<const h16x8*>(res) + base : nullptr;
    h16x8* yv = reinterpret_cast<h16x8*>(y) + base;
    for (size_t e = (size_t)b * 256 + threadIdx.x; e < total; e += (size_t)blocks_per_image * 256) {
        const h16x8 a = xv[e];
        h16x8 r = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (rv) r = rv[e];
        h16x8 w;
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = (h16)(fmaxf(fmaf((float)a[c], sc[c], nb[c]), 0.f) + (float)r[c]);
        yv[e] = w;
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// host: plan tables
// -------------------------------------------------------------------------------------------------------------------------
struct GroupList {
    struct G { int dy, dx, cb; PackHalf h[2]; uint8_t fm; };
    G g[F16_MAX_STEPS * 4];
    int n = 0;
};

static int build_plan(const mstg_f16_conv_desc* d, F16Plan& p, PackTable& pt) {
    memset(&p, 0, sizeof(p));
    memset(&pt, 0, sizeof(pt));
    const int Cin = d->Cin, K = d->K;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || Cin <= 0 || d->Cout <= 0) return fail_arg(MSTG_E_BADARG, "f16 conv: empty tensor");
    if (d->Cout > 64) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: more than 64 output channels");
    p.NF = (d->Cout + 15) / 16;
    const bool image_src = d->src_nchw_f32 != 0;
    if (image_src) {
        if (Cin > 4) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: NCHW fp32 source needs <= 4 channels");
        if (d->kind != 0 || d->stride != 1 || d->dil != 1) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: image source only for stride-1 convs");
    } else if (Cin != 16 && Cin != 32 && Cin != 64) {
        return fail_arg(MSTG_E_ALIGN, "f16 conv: NHWC source needs 16, 32 or 64 channels");
    }
    if (d->dst_nchw && d->Cout > 4) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: NCHW destination needs <= 4 channels");
    if (!d->dst_nchw && (d->Cout & 3)) return fail_arg(MSTG_E_ALIGN, "f16 conv: NHWC destination needs a multiple of 4 channels");
    p.pixstride = 0;  // set once the compute stride is known
    p.stride = 1;
    p.up = 0;
    p.ncls = 1;
    const int ngrp = image_src ? 1 : Cin / 8;  // lane groups per tap (image source: one group per PAIR of taps)
    // ---- list the lane groups class by class --------------------------------------------------------------------------------
    GroupList* gl = new GroupList[4];
    int halo_lo = 0, halo_hi = 0, extra_w = 0;
    if (d->kind == 0) {  // Conv2d
        if (d->stride != 1 && d->stride != 2) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: stride must be 1 or 2"); }
        if (d->Ho != (d->H + 2 * d->pad - d->dil * (K - 1) - 1) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - d->dil * (K - 1) - 1) / d->stride + 1) {
            delete[] gl; return fail_arg(MSTG_E_BADARG, "f16 conv: Ho / Wo do not match the geometry");
        }
        p.stride = d->stride;
        halo_lo = -d->pad;
        halo_hi = d->dil * (K - 1) - d->pad;
        const char* e_dp = getenv("MSTG_F16_DPACK");
        p.dpack = d->dst_nchw && !image_src && d->stride == 1 && d->dil == 1 && d->Cout <= 4 && K >= 3 && K <= 7 && !(e_dp && e_dp[0] == '0');
        if (p.dpack) {  // rows of the MFMA tile = 4 column shifts x 4 channels: a packed tap (ky, j) reads patch column p + 4 j + 3
            const int tapsx = (K + 3) / 4;
            for (int ky = 0; ky < K; ++ky)
                for (int j = 0; j < tapsx; ++j)
                    for (int gi = 0; gi < ngrp; ++gi) {
                        GroupList::G& g = gl[0].g[gl[0].n++];
                        g.dy = ky; g.dx = 4 * j + 3; g.cb = 8 * gi; g.fm = 0x1;
                        g.h[0] = PackHalf{PK_DPACK, (int8_t)ky, (int8_t)j, 0, (int16_t)(8 * gi)};
                        g.h[1] = PackHalf{PK_DPACK, (int8_t)ky, (int8_t)j, 0, (int16_t)(8 * gi + 4)};
                    }
        } else
        for (int ky = 0; ky < K; ++ky) {
            if (image_src) {
                extra_w = 1;
                for (int kx = 0; kx < K; kx += 2) {
                    GroupList::G& g = gl[0].g[gl[0].n++];
                    g.dy = ky; g.dx = kx; g.cb = 0; g.fm = 0xF;
                    g.h[0] = PackHalf{PK_CONV, (int8_t)ky, (int8_t)kx, 0, 0};
                    g.h[1] = kx + 1 < K ? PackHalf{PK_CONV, (int8_t)ky, (int8_t)(kx + 1), 0, 0} : PackHalf{PK_ZERO, 0, 0, 0, -1};
                }
            } else {
                for (int kx = 0; kx < K; ++kx)
                    for (int gi = 0; gi < ngrp; ++gi) {
                        GroupList::G& g = gl[0].g[gl[0].n++];
                        g.dy = ky * d->dil; g.dx = kx * d->dil; g.cb = 8 * gi; g.fm = 0xF;
                        g.h[0] = PackHalf{PK_CONV, (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi)};
                        g.h[1] = PackHalf{PK_CONV, (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi + 4)};
                    }
            }
            if (gl[0].n > F16_MAX_STEPS * 4 - 16) break;
        }
        if (!p.dpack && (image_src ? K * ((K + 1) / 2) : K * K * ngrp) > F16_MAX_STEPS * 4) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: too many taps x channels"); }
    } else if (d->kind == 1) {  // ConvTranspose2d k4 s2 p1: class (py, px) reads source rows y + {-1,0} (py = 0) or y + {0,1} (py = 1)
        if (K != 4 || d->stride != 2 || d->pad != 1 || d->Ho != 2 * d->H || d->Wo != 2 * d->W) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 convT: only k4 s2 p1"); }
        p.up = 1;
        p.ncls = 4;
        halo_lo = -1;
        halo_hi = 1;
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            p.cls_oy[cls] = (int8_t)py;
            p.cls_ox[cls] = (int8_t)px;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    // py = 0: source row y-1 <-> ky = 3, row y <-> ky = 1 ; py = 1: row y <-> ky = 2, row y+1 <-> ky = 0
                    const int dyy = py == 0 ? a - 1 : a, ky = py == 0 ? (a == 0 ? 3 : 1) : (a == 0 ? 2 : 0);
                    const int dxx = px == 0 ? b - 1 : b, kx = px == 0 ? (b == 0 ? 3 : 1) : (b == 0 ? 2 : 0);
                    for (int gi = 0; gi < ngrp; ++gi) {
                        GroupList::G& g = gl[cls].g[gl[cls].n++];
                        g.dy = dyy + 1; g.dx = dxx + 1; g.cb = 8 * gi; g.fm = 0xF;
                        g.h[0] = PackHalf{PK_CONVT, (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi)};
                        g.h[1] = PackHalf{PK_CONVT, (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi + 4)};
                    }
                }
        }
    } else if (d->kind == 2) {  // MultiScaleBlock branches: 1x1 | 3x3 d1 | 3x3 d2 | 3x3 d4, each Cin -> Cin / 4
        if (d->Cout != Cin || d->Ho != d->H || d->Wo != d->W) { delete[] gl; return fail_arg(MSTG_E_BADARG, "f16 msblock: output must match the input"); }
        halo_lo = -4;
        halo_hi = 4;
        const int c4 = Cin / 4;
        auto frag_of_branch = [&](int br) { uint8_t m = 0; for (int c = br * c4; c < (br + 1) * c4; ++c) m |= (uint8_t)(1u << (c / 16)); return m; };
        for (int gi = 0; gi < ngrp; ++gi) {  // shared centre tap
            GroupList::G& g = gl[0].g[gl[0].n++];
            g.dy = 4; g.dx = 4; g.cb = 8 * gi; g.fm = 0xF;
            g.h[0] = PackHalf{PK_MS_CENTER, 1, 1, 0, (int16_t)(8 * gi)};
            g.h[1] = PackHalf{PK_MS_CENTER, 1, 1, 0, (int16_t)(8 * gi + 4)};
        }
        for (int br = 1; br < 4; ++br) {
            const int dil = 1 << (br - 1);
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    if (ky == 1 && kx == 1) continue;
                    for (int gi = 0; gi < ngrp; ++gi) {
                        GroupList::G& g = gl[0].g[gl[0].n++];
                        g.dy = 4 + (ky - 1) * dil; g.dx = 4 + (kx - 1) * dil; g.cb = 8 * gi; g.fm = frag_of_branch(br);
                        g.h[0] = PackHalf{(int8_t)(PK_MS_RING1 + br - 1), (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi)};
                        g.h[1] = PackHalf{(int8_t)(PK_MS_RING1 + br - 1), (int8_t)ky, (int8_t)kx, 0, (int16_t)(8 * gi + 4)};
                    }
                }
        }
    } else {
        delete[] gl;
        return fail_arg(MSTG_E_BADARG, "f16 conv: unknown kind");
    }
    // LDS pixel stride: conflict-free 16-byte operand reads (ds_read_b128 serves lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} ...
    // together): for a compute stride of 1 the stride must be 32 mod 64 bytes, for a compute stride of 2 it must be 16 mod 32
    if (image_src) p.pixstride = 8;
    else if (p.stride == 1) p.pixstride = ((2 * Cin + 31) / 64) * 64 + 32;
    else p.pixstride = 2 * Cin + 16;
    // ---- tile height: the tallest of 16 / 8 rows whose patch (plus room for the statistics scratch) fits 64 KiB ----------------
    const int ext0 = halo_hi - halo_lo;  // rows / cols beyond (T - 1) * stride + 1
    p.PW = (F16_TW - 1) * p.stride + 1 + ext0 + extra_w;  // PH / TH are chosen below, once the filter size is known
    p.oy0 = p.ox0 = halo_lo;
    p.tstep = F16_TW;
    if (p.dpack) {  // 16 accumulator columns start 3 pixels left of the tile's 13 output columns
        p.PW = 15 + 4 * ((K + 3) / 4);
        p.ox0 = halo_lo - 3;
        p.tstep = 13;
    }
    // ---- four groups per K-step ---------------------------------------------------------------------------------------------
    int s = 0;
    for (int cls = 0; cls < p.ncls; ++cls) {
        p.cls_begin[cls] = s;
        for (int i = 0; i < gl[cls].n; i += 4, ++s) {
            if (s >= F16_MAX_STEPS) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: too many K-steps"); }
            uint8_t fm = 0;
            for (int k = 0; k < 4; ++k) {
                if (i + k < gl[cls].n) {
                    const GroupList::G& g = gl[cls].g[i + k];
                    const size_t off = ((size_t)g.dy * p.PW + g.dx) * p.pixstride + 2 * g.cb;
                    if (off > 65535) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: tap offset beyond 64 KiB"); }
                    p.koff[s][k] = (uint16_t)off;
                    pt.h[s][k][0] = g.h[0];
                    pt.h[s][k][1] = g.h[1];
                    fm |= g.fm;
                } else {
                    p.koff[s][k] = 0;
                    pt.h[s][k][0] = pt.h[s][k][1] = PackHalf{PK_ZERO, 0, 0, 0, -1};
                }
            }
            {
                // the kernel instantiates its K loop for these fragment sets; anything else is widened to all fragments (the packed
                // filter then holds zero rows where a tap has no business)
                const unsigned full = (1u << p.NF) - 1, m = fm & full;
                const bool ok = m == full || m == 1 || m == 2 || m == 4 || m == 8 || (p.NF == 4 && (m == 3 || m == 12));
                p.fmask[s] = (uint8_t)(ok ? m : full);
            }
        }
    }
    p.cls_begin[p.ncls] = s;
    p.nsteps = s;
    delete[] gl;
    // ---- stored filter fragments: only the (step, fragment) pairs some tap feeds ------------------------------------------------
    int nw = 0;
    for (int st = 0; st < p.nsteps; ++st) {
        p.wofs[st] = (uint16_t)nw;
        for (int f = 0; f < p.NF; ++f)
            if ((p.fmask[st] >> f) & 1) {
                pt.pair_step[nw] = (uint8_t)st;
                pt.pair_frag[nw] = (uint8_t)f;
                ++nw;
            }
    }
    p.nwfrag = nw;
    // segments: maximal runs of steps of one class with one fragment mask (masks outside the kernel's instantiated set are widened)
    p.nseg = 0;
    for (int cls = 0; cls < p.ncls; ++cls) {
        for (int st = p.cls_begin[cls]; st < p.cls_begin[cls + 1];) {
            int e = st + 1;
            while (e < p.cls_begin[cls + 1] && p.fmask[e] == p.fmask[st]) ++e;
            if (p.nseg >= F16_MAX_SEG) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: too many fragment-mask segments");
            F16Plan::Seg& sg = p.seg[p.nseg++];
            sg.s0 = (int16_t)st; sg.s1 = (int16_t)e; sg.mask = p.fmask[st]; sg.cls = (uint8_t)cls;
            sg.first = st == p.cls_begin[cls]; sg.last = e == p.cls_begin[cls + 1];
            sg.wbase = p.wofs[st];
            st = e;
        }
    }
    // tile height (16 or 8 rows) and where the filter lives: prefer two workgroups per CU with the filter in LDS (<= 78 KiB each),
    // then one workgroup with the filter in LDS, then the filter from global memory
    const int ext = halo_hi - halo_lo;
    // a candidate height is admissible only if its patch can be prefetched in at most 12 16-byte registers per thread
    auto patch_bytes_of = [&](int th) {
        const int ph = (th - 1) * p.stride + 1 + ext;
        const int elems = image_src ? ph * p.PW : ph * p.PW * (Cin / 8);
        return cdiv(elems, 256) > 12 ? (size_t)1 << 30 : (size_t)ph * p.PW * p.pixstride;
    };
    const size_t wb = (size_t)nw * 1024 + F16_TABLE_BYTES;
    int TH;
    int th_max = 16;  // measured (tools/f16_ab.sh): 32-row tiles lose 10-30 % on every NHWC layer and are a wash on the stem
    { const char* e = getenv("MSTG_F16_TH"); if (e) th_max = atoi(e); }  // experiments: cap the tile height
    // one output fragment: 32-row tiles (eight rows per wave) halve everything a tile does once per step / per tile and cut the
    // halo share; taken when two workgroups per CU still fit and the image is tall enough to have whole tiles
    const int gh_plan = d->kind == 1 ? d->H : d->Ho;
    if (p.NF == 1 && !p.dpack && th_max >= 32 && gh_plan >= 32 && wb + patch_bytes_of(32) <= 78 * 1024) { TH = 32; p.wlds = 1; }
    else if (wb + patch_bytes_of(16) <= 78 * 1024) { TH = 16; p.wlds = 1; }
    else if (wb + patch_bytes_of(8) <= 78 * 1024) { TH = 8; p.wlds = 1; }
    else if (wb + patch_bytes_of(16) <= 156 * 1024) { TH = 16; p.wlds = 1; }
    else if (wb + patch_bytes_of(8) <= 156 * 1024) { TH = 8; p.wlds = 1; }
    else if (patch_bytes_of(16) + F16_TABLE_BYTES <= 78 * 1024) { TH = 16; p.wlds = 0; }
    else { TH = 8; p.wlds = 0; }
    p.TH = TH;
    p.PH = (TH - 1) * p.stride + 1 + ext;
    const size_t patch_bytes = patch_bytes_of(TH);
    if (patch_bytes + F16_TABLE_BYTES > 156 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: patch does not fit LDS");
    p.m_pw = magic_u32((unsigned)p.PW);
    const int elems = image_src ? p.PH * p.PW : p.PH * p.PW * (Cin / 8);
    p.npf = cdiv(elems, 256);
    if (p.npf > 12) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: patch needs more than 12 prefetch registers per thread");
    if (p.PH * p.PW >= 65536) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: patch too large");
    return MSTG_OK;
}

static size_t plan_blob_bytes(const F16Plan& p) { return 256 + (size_t)p.nwfrag * 64 * 16; }

template <int RPW, int NF, int NPF>
static int launch_conv(const F16ConvArgs& a, const F16Plan& p, int src, int dst, size_t lds, long grid /* tiles */, hipStream_t st, int* grid_out) {
#define MSTG_F16_LAUNCH(SRC, DST)                                                                                          \
    do {                                                                                                                  \
        auto kern = p.wlds ? conv_f16_kernel<RPW, NF, SRC, DST, NPF, true> : conv_f16_kernel<RPW, NF, SRC, DST, NPF, false>;      \
        const void* kptr = reinterpret_cast<const void*>(kern);                                                           \
        /* persistent workgroups: what the CU really holds (registers, LDS), at most 4, a multiple of 8 in all */       \
        static const void* c_kern = nullptr;                                                                              \
        static size_t c_lds = 0;                                                                                          \
        static int c_occ = 1;                                                                                             \
        if (c_kern != kptr || c_lds != lds) {                                                                             \
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(kptr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            int nb = 1;                                                                                                   \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kptr, 256, lds) != hipSuccess || nb < 1) nb = 1;        \
            c_occ = nb > 4 ? 4 : nb;                                                                                      \
            c_kern = kptr;                                                                                                \
            c_lds = lds;                                                                                                  \
        }                                                                                                                 \
        long g_ = 256L * c_occ;                                                                                           \
        if (g_ > grid) g_ = (grid + 7) & ~7L;                                                                             \
        if (a.partial && hipMemsetAsync(a.partial, 0, (size_t)a.N * g_ * 2 * 16 * NF * sizeof(float), st) != hipSuccess)   \
            return fail_arg(MSTG_E_LAUNCH, "f16 conv: clearing the statistics partials failed");                           \
        *grid_out = (int)g_;                                                                                              \
        MSTG_LAUNCH(kern, dim3((unsigned)g_), dim3(256), lds, st, a, p);                                           \
    } while (0)
    if (src == 0 && dst == 0) MSTG_F16_LAUNCH(0, 0);
    else if (src == 1 && dst == 0) MSTG_F16_LAUNCH(1, 0);
    else if (src == 0 && dst == 1) MSTG_F16_LAUNCH(0, 1);
    else if (src == 2 && dst == 0) MSTG_F16_LAUNCH(2, 0);
    else if (src == 2 && dst == 1) MSTG_F16_LAUNCH(2, 1);
    else return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: NCHW source and destination in one layer");
#undef MSTG_F16_LAUNCH
    MSTG_CHECK_LAUNCH("conv_f16_kernel");
    return MSTG_OK;
}

template <int RPW, int NF>
static int launch_conv_npf(const F16ConvArgs& a, const F16Plan& p, int src, int dst, size_t lds, long grid, hipStream_t st, int* grid_out) {
    if (p.npf <= 2) return launch_conv<RPW, NF, 2>(a, p, src, dst, lds, grid, st, grid_out);
    if (p.npf <= 4) return launch_conv<RPW, NF, 4>(a, p, src, dst, lds, grid, st, grid_out);
    if (p.npf <= 6) return launch_conv<RPW, NF, 6>(a, p, src, dst, lds, grid, st, grid_out);
    if (p.npf <= 8) return launch_conv<RPW, NF, 8>(a, p, src, dst, lds, grid, st, grid_out);
    return launch_conv<RPW, NF, 12>(a, p, src, dst, lds, grid, st, grid_out);
}

}  // namespace mstg

using namespace mstg;

extern "C" size_t mstg_f16_conv_plan_bytes(const mstg_f16_conv_desc* d) {
    F16Plan p;
    PackTable* pt = new PackTable;
    const int rc = d ? build_plan(d, p, *pt) : MSTG_E_BADARG;
    delete pt;
    return rc ? 0 : plan_blob_bytes(p);
}

extern "C" int mstg_f16_conv_pack(const mstg_f16_conv_desc* d, const float* w0, const float* b0, const float* w1, const float* b1,
                                  const float* w2, const float* b2, const float* w3, const float* b3, void* blob, size_t blob_bytes,
                                  void* stream) {
    if (!d || !w0 || !blob) return fail_arg(MSTG_E_BADARG, "f16 conv pack: null pointer");
    F16Plan p;
    PackTable* pt = new PackTable;
    if (int rc = build_plan(d, p, *pt)) { delete pt; return rc; }
    if (blob_bytes < plan_blob_bytes(p)) { delete pt; return fail_arg(MSTG_E_WORKSPACE, "f16 conv pack: blob too small"); }
    if (d->kind == 2 && (!w1 || !w2 || !w3)) { delete pt; return fail_arg(MSTG_E_BADARG, "f16 msblock pack: four weight tensors needed"); }
    PackSrc s;
    s.w[0] = w0; s.w[1] = w1; s.w[2] = w2; s.w[3] = w3;
    s.b[0] = b0; s.b[1] = b1; s.b[2] = b2; s.b[3] = b3;
    s.Cin = d->Cin; s.Cout = d->Cout; s.KH = s.KW = d->K; s.c4 = d->kind == 2 ? d->Cin / 4 : 0;
    float* bias = (float*)blob;
    h16* wpk = (h16*)((char*)blob + 256);
    MSTG_LAUNCH(f16_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, *pt, s, p.nwfrag, p.NF, wpk, bias);
    delete pt;
    MSTG_CHECK_LAUNCH("f16_pack_kernel");
    return MSTG_OK;
}

extern "C" size_t mstg_f16_conv_partial_bytes(const mstg_f16_conv_desc* d) {
    F16Plan p;
    PackTable* pt = new PackTable;
    const int rc = d ? build_plan(d, p, *pt) : MSTG_E_BADARG;
    delete pt;
    if (rc) return 0;
    return (size_t)d->N * F16_MAX_GRID * 2 * 16 * p.NF * sizeof(float);
}

extern "C" int mstg_f16_conv_fwd(const mstg_f16_conv_desc* d, const void* blob, const void* x, const float* in_stats, void* y,
                                 float* out_stats, void* workspace, size_t workspace_bytes, void* stream) {
    return mstg_f16_conv_fwd_res(d, blob, x, in_stats, nullptr, y, out_stats, workspace, workspace_bytes, stream);
}

extern "C" int mstg_f16_conv_fwd_res(const mstg_f16_conv_desc* d, const void* blob, const void* x, const float* in_stats,
                                     const void* residual, void* y, float* out_stats, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!d || !blob || !x || !y) return fail_arg(MSTG_E_BADARG, "f16 conv: null pointer");
    if (residual && (!in_stats || d->src_nchw_f32)) return fail_arg(MSTG_E_BADARG, "f16 conv: a residual operand needs NHWC input and its statistics");
    F16Plan p;
    PackTable* pt = new PackTable;
    const int rc = build_plan(d, p, *pt);
    delete pt;
    if (rc) return rc;
    if (in_stats && d->src_nchw_f32) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: no normalise-on-load for the image source");
    F16ConvArgs a;
    a.x = x; a.y = (h16*)y;
    a.res = (const h16*)residual;
    a.bias = (const float*)blob;
    a.wpk = (const h16*)((const char*)blob + 256);
    a.in_stats = in_stats;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.Gh = p.up ? d->H : d->Ho;
    a.Gw = p.up ? d->W : d->Wo;
    a.tiles_y = cdiv(a.Gh, p.TH);
    a.tiles_x = cdiv(a.Gw, p.tstep);
    a.act = d->act;
    { const char* e = getenv("MSTG_F16_DBG"); a.dbg = e ? atoi(e) : 0; }
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffL) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: too many tiles");
    // the kernel divides tile indices by multiply-high: exact while index * divisor < 2^32
    if ((unsigned long long)(tiles + 4096) * (unsigned long long)(a.tiles_x * a.tiles_y) >= (1ull << 32))
        return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: batch x tiles too large for the tile-index arithmetic");
    p.m_ntile = magic_u32((unsigned)(a.tiles_x * a.tiles_y));
    p.m_tx = magic_u32((unsigned)a.tiles_x);
    a.partial = nullptr;
    if (out_stats) {
        const size_t need = (size_t)a.N * F16_MAX_GRID * 2 * 16 * p.NF * sizeof(float);
        if (!workspace || workspace_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "f16 conv: workspace too small for the statistics partials");
        a.partial = (float*)workspace;
    }
    size_t lds = (size_t)p.PH * p.PW * p.pixstride;
    if (lds < (size_t)8 * 16 * p.NF * sizeof(float)) lds = (size_t)8 * 16 * p.NF * sizeof(float);
    lds = (lds + 15) & ~(size_t)15;
    if (p.wlds) lds += (size_t)p.nwfrag * 1024;
    lds += F16_TABLE_BYTES;
    const long grid = tiles;
    int launched = 0;
    hipStream_t st = (hipStream_t)stream;
    {   // 1x1: fragments straight from global memory, no LDS patch (conv1x1_f16_kernel)
        const char* e = getenv("MSTG_F16_DIRECT");
        const bool direct = !residual && d->kind == 0 && d->K == 1 && d->stride == 1 && !d->src_nchw_f32 && !d->dst_nchw && d->act == MSTG_ACT_NONE &&
                            (d->Cout & 15) == 0 && p.TH == 16 && !(e && e[0] == '0');
        if (direct) {
            const int KS = (d->Cin + 31) / 32;
            auto go = [&](auto kern) -> int {
                static int occ = 0;
                if (!occ) {
                    int nb = 1;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, 0) != hipSuccess || nb < 1) nb = 1;
                    occ = nb > 4 ? 4 : nb;
                }
                long g_ = 256L * occ;
                if (g_ > grid) g_ = (grid + 7) & ~7L;
                if (a.partial && hipMemsetAsync(a.partial, 0, (size_t)a.N * g_ * 2 * 16 * p.NF * sizeof(float), st) != hipSuccess)
                    return fail_arg(MSTG_E_LAUNCH, "f16 conv: clearing the statistics partials failed");
                launched = (int)g_;
                MSTG_LAUNCH(kern, dim3((unsigned)g_), dim3(256), 0, st, a, p);
                MSTG_CHECK_LAUNCH("conv1x1_f16_kernel");
                return MSTG_OK;
            };
            int rc = MSTG_E_UNSUPPORTED;
            if (p.NF == 1 && KS == 1) rc = go(conv1x1_f16_kernel<1, 1>);
            else if (p.NF == 2 && KS == 1) rc = go(conv1x1_f16_kernel<2, 1>);
            else if (p.NF == 4 && KS == 2) rc = go(conv1x1_f16_kernel<4, 2>);
            else if (p.NF == 4 && KS == 1) rc = go(conv1x1_f16_kernel<4, 1>);
            else if (p.NF == 2 && KS == 2) rc = go(conv1x1_f16_kernel<2, 2>);
            else if (p.NF == 1 && KS == 2) rc = go(conv1x1_f16_kernel<1, 2>);
            if (rc != MSTG_E_UNSUPPORTED) {
                if (rc) return rc;
                if (out_stats) {
                    MSTG_LAUNCH(f16_norm_finalize_kernel, dim3(a.N), dim3(256), 0, st, (const float*)a.partial, out_stats, launched,
                                       16 * p.NF, d->Cout, (float)((size_t)d->Ho * d->Wo));
                    MSTG_CHECK_LAUNCH("f16_norm_finalize_kernel");
                }
                return MSTG_OK;
            }
        }
    }
    const int src = d->src_nchw_f32 ? 1 : (residual ? 2 : 0), dst = d->dst_nchw ? 1 : 0;
    int lrc;
    if (p.TH == 32) {
        lrc = launch_conv_npf<8, 1>(a, p, src, dst, lds, grid, st, &launched);
    } else if (p.TH == 16) {
        switch (p.NF) {
            case 1: lrc = launch_conv_npf<4, 1>(a, p, src, dst, lds, grid, st, &launched); break;
            case 2: lrc = launch_conv_npf<4, 2>(a, p, src, dst, lds, grid, st, &launched); break;
            case 4: lrc = launch_conv_npf<4, 4>(a, p, src, dst, lds, grid, st, &launched); break;
            default: return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: output channels must pad to 16, 32 or 64");
        }
    } else {
        switch (p.NF) {
            case 1: lrc = launch_conv_npf<2, 1>(a, p, src, dst, lds, grid, st, &launched); break;
            case 2: lrc = launch_conv_npf<2, 2>(a, p, src, dst, lds, grid, st, &launched); break;
            case 4: lrc = launch_conv_npf<2, 4>(a, p, src, dst, lds, grid, st, &launched); break;
            default: return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: output channels must pad to 16, 32 or 64");
        }
    }
    if (lrc) return lrc;
    if (out_stats) {
        const float count = (float)((size_t)d->Ho * d->Wo);
        // ConvTranspose: each compute-grid tile wrote 4 classes; the per-tile sums already cover all of them
        MSTG_LAUNCH(f16_norm_finalize_kernel, dim3(a.N), dim3(256), 0, st, (const float*)a.partial, out_stats,
                           launched, 16 * p.NF, d->Cout, count);
        MSTG_CHECK_LAUNCH("f16_norm_finalize_kernel");
    }
    return MSTG_OK;
}

extern "C" int mstg_f16_norm_residual(const void* x, const void* residual, const float* stats, void* y, int N, int HW, int C,
                                      void* stream) {
    if (!x || !stats || !y) return fail_arg(MSTG_E_BADARG, "f16 norm_residual: null pointer");
    if (C != 16 && C != 32 && C != 64) return fail_arg(MSTG_E_ALIGN, "f16 norm_residual: C must be 16, 32 or 64");
    if (N <= 0 || HW <= 0) return fail_arg(MSTG_E_BADARG, "f16 norm_residual: empty tensor");
    const size_t chunks = (size_t)HW * (C / 8);
    int bpi = (int)((chunks + 256 * 8 - 1) / (256 * 8));  // ~8 chunks per thread
    if (bpi > 4096) bpi = 4096;
    if (bpi < 1) bpi = 1;
    MSTG_LAUNCH(f16_norm_residual_kernel, dim3((unsigned)N * bpi), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
                       (const h16*)residual, stats, (h16*)y, (size_t)HW, C, bpi);
    MSTG_CHECK_LAUNCH("f16_norm_residual_kernel");
    return MSTG_OK;
}

// -------------------------------------------------------------------------------------------------------------------------
// LocalAttention (enhanced_generator.py:13-47) in one kernel: qkv 1x1 conv, window partition, F.normalize of q and k,
// q^ k^T (C x C per 4x4 window), softmax, attn . v, un-partition, proj 1x1 conv.  One wave per window, four waves per workgroup
// with wave-private LDS tiles (no barriers); a workgroup covers a strip of 4 rows x 256 columns (16 windows per wave).
//
// Every product is a chain of v_mfma_f32_16x16x16_f16 whose operands are 8-byte LDS reads, arranged so that each intermediate
// is WRITTEN in the layout its consumer reads (lane = (i = lane & 15, g = lane >> 4); D = rows 4g..4g+3 x column i):
//   X  [pixel][c]   staged input (InstanceNorm + ReLU applied on load when in_stats is given)
//   Q^T, K^T [c][pixel]   = X W^T computed pixel-major (A = X, B = W): a lane ends with 4 PIXELS of one channel -> one 8-byte
//                    write into the channel-major tile; the per-pixel L2 norm is a 16-lane row reduction (DPP)
//   V  [pixel][c]   computed channel-major (A = W, B = X): a lane ends with 4 channels of one pixel
//   S^T[c2][c1]     = K^T . Q (contraction over the 16 pixels = one MFMA per 16x16 block); softmax over c2 = over registers
//                    and the four lane groups; scores are bounded by +-16 (unit vectors over 16 pixels): no max subtraction
//   P  [c1][c2]     probabilities, c2 contiguous;   O [pixel][c1] = P V;   Y = Wproj O + b -> 8-byte NHWC stores
// -------------------------------------------------------------------------------------------------------------------------
typedef _Float16 h16x4v __attribute__((ext_vector_type(4)));

template <int C>
struct AttnF16 {
    static constexpr int NB = C / 16;
    static constexpr int LDX = C + 8, LDT = 16 + 8, LDP = C + 8;  // row lengths in halves (+16 bytes: conflict-free 8-byte reads)
    static constexpr int X = 0, QT = X + 16 * LDX, KT = QT + C * LDT, V = KT + C * LDT, P = V + 16 * LDX, END = P + C * LDP;
    static constexpr int NFRAG = 4 * NB * NB;  // q | k | v | proj, each NB x NB fragments of 64 lanes x 4 halves
    static constexpr bool WLDS = C > 32;       // weights in LDS (shared by the workgroup) instead of registers
};
constexpr int ATT_WPW = 16;  // windows per wave

__device__ __forceinline__ f32x4 mfma16h(h16x4v a, h16x4v b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

template <int C>
__global__ __launch_bounds__(256) void attn_f16_kernel(const h16* __restrict__ x, const float* __restrict__ in_stats,
                                                       const h16* __restrict__ wfrag, const float* __restrict__ bias,
                                                       h16* __restrict__ y, int N, int H, int W) {
    typedef AttnF16<C> T;
    constexpr int NB = T::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, i = l & 15, g = l >> 4;
    const int n = blockIdx.z, wy = blockIdx.y;
    h16* wl = reinterpret_cast<h16*>(smem);
    h16* tile = reinterpret_cast<h16*>(smem + (T::WLDS ? (size_t)T::NFRAG * 64 * 8 : 0)) + (size_t)wv * T::END;
    // ---- weights: registers (C <= 32) or LDS (C = 64) --------------------------------------------------------------------------
    h16x4v wr[T::WLDS ? 1 : T::NFRAG];
    if (T::WLDS) {
        for (int e = tid; e < T::NFRAG * 64; e += 256) reinterpret_cast<h16x4v*>(wl)[e] = reinterpret_cast<const h16x4v*>(wfrag)[e];
        __syncthreads();
    } else {
#pragma unroll
        for (int f = 0; f < T::NFRAG; ++f) wr[f] = reinterpret_cast<const h16x4v*>(wfrag)[f * 64 + l];
    }
    auto wget = [&](int part, int f, int ks) -> h16x4v {
        const int idx = (part * NB + f) * NB + ks;
        if (T::WLDS) return reinterpret_cast<const h16x4v*>(wl)[idx * 64 + l];
        return wr[T::WLDS ? 0 : idx];
    };
    float bq[NB], bk[NB];
    f32x4 bv[NB], bp[NB];
#pragma unroll
    for (int f = 0; f < NB; ++f) {
        bq[f] = bias[16 * f + i];
        bk[f] = bias[C + 16 * f + i];
        bv[f] = *reinterpret_cast<const f32x4*>(bias + 2 * C + 16 * f + 4 * g);
        bp[f] = *reinterpret_cast<const f32x4*>(bias + 3 * C + 16 * f + 4 * g);
    }
    // ---- normalise-on-load constants: this lane always stages the same channel octet ------------------------------------------
    constexpr int OCT = C / 8;
    const int o = l % OCT;
    float sc[8], nb[8];  // (x - mean) * rstd = x * sc + nb
    const bool norm = in_stats != nullptr;
    if (norm) {
        const float* st = in_stats + ((size_t)n * C + 8 * o) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) { sc[c] = st[2 * c + 1]; nb[c] = -st[2 * c] * st[2 * c + 1]; }
    }
    h16* Xs = tile + T::X;
    h16* QTs = tile + T::QT;
    h16* KTs = tile + T::KT;
    h16* Vs = tile + T::V;
    h16* Ps = tile + T::P;
    const size_t img = (size_t)n * H * W * C;

    for (int wi = 0; wi < ATT_WPW; ++wi) {
        const int wx = (blockIdx.x * 4 + wv) * ATT_WPW + wi;
        if (4 * wx >= W) break;
        // (0) window -> X tile
#pragma unroll
        for (int e = l; e < 16 * OCT; e += 64) {
            const int p = e / OCT;
            const int yy = 4 * wy + (p >> 2), xx = 4 * wx + (p & 3);
            h16x8 v = *reinterpret_cast<const h16x8*>(x + img + ((size_t)yy * W + xx) * C + 8 * o);
            if (norm) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = (h16)fmaxf(fmaf((float)v[c], sc[c], nb[c]), 0.f);
            }
            *reinterpret_cast<h16x8*>(&Xs[p * T::LDX + 8 * o]) = v;
        }
        h16x4v xa[NB];
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) xa[ks] = *reinterpret_cast<const h16x4v*>(&Xs[i * T::LDX + 16 * ks + 4 * g]);
        // (1) q^T, k^T: D[pixel 4g+r][channel 16nf+i], normalised per pixel over channels, stored channel-major
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            f32x4 acc[NB];
            f32x4 ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nf = 0; nf < NB; ++nf) {
                acc[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NB; ++ks) acc[nf] = mfma16h(xa[ks], wget(part, nf, ks), acc[nf]);
                const float b = part == 0 ? bq[nf] : bk[nf];
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[nf][r] += b; ss[r] += acc[nf][r] * acc[nf][r]; }
            }
            f32x4 inv;
#pragma unroll
            for (int r = 0; r < 4; ++r) inv[r] = fminf(__builtin_amdgcn_rsqf(row16_sum_f(ss[r])), 1e12f);  // 1 / max(||.||, 1e-12)
            h16* dst = part == 0 ? QTs : KTs;
#pragma unroll
            for (int nf = 0; nf < NB; ++nf) {
                h16x4v hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (h16)(acc[nf][r] * inv[r]);
                *reinterpret_cast<h16x4v*>(&dst[(16 * nf + i) * T::LDT + 4 * g]) = hv;
            }
        }
        // (2) v: D[channel 16mf+4g+r][pixel i], stored pixel-major
#pragma unroll
        for (int mf = 0; mf < NB; ++mf) {
            f32x4 acc = bv[mf];
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) acc = mfma16h(wget(2, mf, ks), xa[ks], acc);
            h16x4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (h16)acc[r];
            *reinterpret_cast<h16x4v*>(&Vs[i * T::LDX + 16 * mf + 4 * g]) = hv;
        }
        // (3) S^T[c2][c1] and the softmax over c2, column block by column block
        h16x4v ka[NB];
#pragma unroll
        for (int mf = 0; mf < NB; ++mf) ka[mf] = *reinterpret_cast<const h16x4v*>(&KTs[(16 * mf + i) * T::LDT + 4 * g]);
#pragma unroll
        for (int nf = 0; nf < NB; ++nf) {
            const h16x4v qb = *reinterpret_cast<const h16x4v*>(&QTs[(16 * nf + i) * T::LDT + 4 * g]);
            f32x4 s[NB];
            float sum = 0.f;
#pragma unroll
            for (int mf = 0; mf < NB; ++mf) {
                s[mf] = mfma16h(ka[mf], qb, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[mf][r] = __expf(s[mf][r]); sum += s[mf][r]; }
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int mf = 0; mf < NB; ++mf) {
                h16x4v hv;
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = (h16)(s[mf][r] * inv);
                *reinterpret_cast<h16x4v*>(&Ps[(16 * nf + i) * T::LDP + 16 * mf + 4 * g]) = hv;
            }
        }
        // (4) O[pixel][c1] = sum_c2 P[c1][c2] V[pixel][c2]   (into the X tile: its fragments are in registers)
        h16x4v vb[NB];
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) vb[ks] = *reinterpret_cast<const h16x4v*>(&Vs[i * T::LDX + 16 * ks + 4 * g]);
#pragma unroll
        for (int mf = 0; mf < NB; ++mf) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NB; ++ks)
                acc = mfma16h(*reinterpret_cast<const h16x4v*>(&Ps[(16 * mf + i) * T::LDP + 16 * ks + 4 * g]), vb[ks], acc);
            h16x4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (h16)acc[r];
            *reinterpret_cast<h16x4v*>(&Xs[i * T::LDX + 16 * mf + 4 * g]) = hv;
        }
        // (5) y = Wproj O + b
        h16x4v ob[NB];
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) ob[ks] = *reinterpret_cast<const h16x4v*>(&Xs[i * T::LDX + 16 * ks + 4 * g]);
        const int yy = 4 * wy + (i >> 2), xx = 4 * wx + (i & 3);
        h16* yp = y + img + ((size_t)yy * W + xx) * C + 4 * g;
#pragma unroll
        for (int mf = 0; mf < NB; ++mf) {
            f32x4 acc = bp[mf];
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) acc = mfma16h(wget(3, mf, ks), ob[ks], acc);
            h16x4v hv;
#pragma unroll
            for (int r = 0; r < 4; ++r) hv[r] = (h16)acc[r];
            *reinterpret_cast<h16x4v*>(yp + 16 * mf) = hv;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// The same module as register-resident MFMA chains (round 3; the fp32 training kernels' design, csrc/attention_reg.hip): the
// accumulator layout of v_mfma_f32_16x16x16_f16 -- register r of lane (i, g) is D[4g + r][i] -- is, after rounding the four values to
// fp16, exactly the 4-element operand of a following MFMA that contracts over D's ROW index.  A window fetched with one 8-byte load
// per lane and 16-channel fragment (lane i = pixel, channels 16h + 4g ..) serves as the A operand (rows = pixels) and as the B
// operand (columns = pixels), so q | k -> S^T -> softmax -> O^T -> Y^T run without a single LDS tile: attn_f16_kernel moved seven
// tiles per window through LDS (41 % of its LDS cycles, 44 % of them bank conflicts) and ran at 0.8-3.1 TB/s.  Filters in registers
// at C <= 32 (the packed fragments are already in operand layout), in LDS at C = 64.  Persistent waves over contiguous window
// ranges, the next window's pixels in flight behind the current chain.  Rounding points are attn_f16_kernel's (q^, k^, v, P, O in fp16;
// accumulation, norms and softmax in fp32).
// -------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ h16x4v cvt4(f32x4 v) { return h16x4v{(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]}; }
__device__ __forceinline__ float xg_sum_f(float v) {  // sum over the four lanes sharing i = lane & 15
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int C>
__global__ __launch_bounds__(256, C == 64 ? 2 : 3) void attn_f16r_kernel(const h16* __restrict__ x, const float* __restrict__ in_stats,
                                                                          const h16* __restrict__ wfrag, const float* __restrict__ bias,
                                                                          h16* __restrict__ y, int N, int H, int W) {
    constexpr int NB = C / 16, NFRAG = 4 * NB * NB;
    constexpr bool WLDS = C > 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, l = tid & 63, i = l & 15, g = l >> 4;
    h16x4v wr[WLDS ? 1 : NFRAG];
    if (WLDS) {
        for (int e = tid; e < NFRAG * 64; e += 256) reinterpret_cast<h16x4v*>(smem)[e] = reinterpret_cast<const h16x4v*>(wfrag)[e];
        __syncthreads();
    } else {
#pragma unroll
        for (int f = 0; f < NFRAG; ++f) wr[f] = reinterpret_cast<const h16x4v*>(wfrag)[f * 64 + l];
    }
    auto wget = [&](int part, int f, int ks) -> h16x4v {  // lane (i, g): W[part * C + 16 f + i][16 ks + 4 g + j]
        const int idx = (part * NB + f) * NB + ks;
        if (WLDS) return reinterpret_cast<const h16x4v*>(smem)[idx * 64 + l];
        return wr[WLDS ? 0 : idx];
    };
    const int nwx = W / 4, nwy = H / 4, nwin = N * nwx * nwy;
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (tid >> 6)), nwv = gridDim.x * 4;
    const int per = (nwin + nwv - 1) / nwv, w0 = wv * per, w1 = w0 + per < nwin ? w0 + per : nwin;
    if (w0 >= nwin) return;
    float bq[NB], bk[NB];
    f32x4 bv[NB], bp[NB];
#pragma unroll
    for (int f = 0; f < NB; ++f) {
        bq[f] = bias[16 * f + i];
        bk[f] = bias[C + 16 * f + i];
        bv[f] = *reinterpret_cast<const f32x4*>(bias + 2 * C + 16 * f + 4 * g);
        bp[f] = *reinterpret_cast<const f32x4*>(bias + 3 * C + 16 * f + 4 * g);
    }
    // window coordinates: wave-uniform walkers advanced by carries -- one for the window being fetched, one for the window being computed
    struct Walk {
        int n, wy, wx;
        __device__ __forceinline__ void next(int nwx_, int nwy_) {
            if (++wx == nwx_) {
                wx = 0;
                if (++wy == nwy_) { wy = 0; ++n; }
            }
        }
    };
    Walk fw{w0 / (nwx * nwy), (w0 / nwx) % nwy, w0 % nwx}, pw = fw;
    // address = wave-uniform window offset (scalar registers) + this lane's constant offset inside a window: no vector arithmetic per window
    const unsigned lane_off = (unsigned)(((i >> 2) * W + (i & 3)) * C + 4 * g);
    auto win_off = [&](const Walk& c) -> size_t { return (((size_t)c.n * H + 4 * c.wy) * W + 4 * c.wx) * C; };
    auto fetch = [&](h16x4v (&t)[NB], const Walk& c) {
        const h16* p = x + win_off(c) + lane_off;
#pragma unroll
        for (int h = 0; h < NB; ++h) t[h] = *reinterpret_cast<const h16x4v*>(p + 16 * h);
    };
    // normalise-on-load constants of the current image: (x - mean) * rstd = x * sc + nb for this lane's 4 channels per fragment
    f32x4 sc[NB], nbv[NB];
    int stats_n = -1;
    // DEPTH windows are fetched as a group while the previous group is computed.  Measured: 4 instead of 1 changes nothing at C = 16 and
    // costs 12 % at C = 32 (a wave per SIMD less) -- at 155 / 240 VALU instructions per window against 6 / 24 MFMAs the kernel is bound
    // by the vector pipe (row sums of the two F.normalize, conversions, softmax), not by loads in flight.
    constexpr int DEPTH = 1;
    h16x4v nxt[DEPTH][NB];
    int fwin = w0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (fwin < w1) { fetch(nxt[d], fw); fw.next(nwx, nwy); ++fwin; }
    for (int win = w0; win < w1; win += DEPTH) {
        h16x4v cur[DEPTH][NB];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int h = 0; h < NB; ++h) cur[d][h] = nxt[d][h];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (fwin < w1) { fetch(nxt[d], fw); fw.next(nwx, nwy); ++fwin; }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (win + d >= w1) break;
        const int cn = pw.n, cwy = pw.wy, cwx = pw.wx;
        pw.next(nwx, nwy);
        h16x4v xa[NB];
        if (in_stats) {  // (x - mean) * rstd, ReLU, as attn_f16_kernel applies it while staging: fp32 arithmetic, one rounding to fp16
            if (C == 64 || cn != stats_n) {  // wave-uniform: a wave's windows are contiguous, the image changes rarely (C = 64 has no
                stats_n = cn;                // registers to keep the constants across windows: re-read per window there)
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    const float* st = in_stats + ((size_t)cn * C + 16 * h + 4 * g) * 2;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(st), s1 = *reinterpret_cast<const f32x4*>(st + 4);
                    sc[h] = f32x4{s0[1], s0[3], s1[1], s1[3]};
                    nbv[h] = f32x4{-s0[0] * s0[1], -s0[2] * s0[3], -s1[0] * s1[1], -s1[2] * s1[3]};
                }
            }
#pragma unroll
            for (int h = 0; h < NB; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) xa[h][c] = (h16)fmaxf(fmaf((float)cur[d][h][c], sc[h][c], nbv[h][c]), 0.f);
        } else {
#pragma unroll
            for (int h = 0; h < NB; ++h) xa[h] = cur[d][h];
        }
        // q | k = X W^T: L(p|j); v^T = Wv X^T: L(c|p)
        f32x4 q[NB], k[NB], vt[NB];
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            q[f] = f32x4{bq[f], bq[f], bq[f], bq[f]};
            k[f] = f32x4{bk[f], bk[f], bk[f], bk[f]};
            vt[f] = bv[f];
#pragma unroll
            for (int h = 0; h < NB; ++h) {
                q[f] = mfma16h(xa[h], wget(0, f, h), q[f]);
                k[f] = mfma16h(xa[h], wget(1, f, h), k[f]);
                vt[f] = mfma16h(wget(2, f, h), xa[h], vt[f]);
            }
        }
        // F.normalize over channels: a row (g, r) is a pixel, its channels lie across the 16 lanes and NB fragments
        f32x4 sq = {0.f, 0.f, 0.f, 0.f}, sk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            sq += q[f] * q[f];
            sk += k[f] * k[f];
        }
        f32x4 iq, ik;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            iq[r] = fminf(__builtin_amdgcn_rsqf(row16_sum_f(sq[r])), 1e12f);
            ik[r] = fminf(__builtin_amdgcn_rsqf(row16_sum_f(sk[r])), 1e12f);
        }
        h16x4v qh[NB], kh[NB], vh[NB];
#pragma unroll
        for (int f = 0; f < NB; ++f) {
            qh[f] = cvt4(q[f] * iq);
            kh[f] = cvt4(k[f] * ik);
            vh[f] = cvt4(vt[f]);
        }
        // S^T[c2][c1] = sum_p k^[p][c2] q^[p][c1]; softmax over c2 (rows: registers + the four lane groups); |S| <= 1
        h16x4v pt[NB][NB];
#pragma unroll
        for (int nn = 0; nn < NB; ++nn) {
            f32x4 st[NB];
            float z = 0.f;
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                st[m] = mfma16h(kh[m], qh[nn], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[m][r] = __expf(st[m][r]); z += st[m][r]; }
            }
            const float inv = __builtin_amdgcn_rcpf(xg_sum_f(z));
#pragma unroll
            for (int m = 0; m < NB; ++m) pt[m][nn] = cvt4(st[m] * f32x4{inv, inv, inv, inv});
        }
        // O^T[c1][p] = sum_c2 P^T[c2][c1] v^T[c2][p];  Y^T[co][p] = bp[co] + sum_c1 Wp[co][c1] O^T[c1][p]
        h16x4v oh[NB];
#pragma unroll
        for (int n1 = 0; n1 < NB; ++n1) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NB; ++m) o = mfma16h(pt[m][n1], vh[m], o);
            oh[n1] = cvt4(o);
        }
        h16* yp = y + (((size_t)cn * H + 4 * cwy) * W + 4 * cwx) * C + lane_off;
#pragma unroll
        for (int cf = 0; cf < NB; ++cf) {
            f32x4 acc = bp[cf];
#pragma unroll
            for (int n1 = 0; n1 < NB; ++n1) acc = mfma16h(wget(3, cf, n1), oh[n1], acc);
            *reinterpret_cast<h16x4v*>(yp + 16 * cf) = cvt4(acc);
        }
      }
    }
}

// wqkv (3C, C), wproj (C, C) fp32 -> fragments [part][f][ks][lane][4]: lane (i, g) holds W[part*C + 16f + i][16ks + 4g + j]
__global__ void f16_attn_pack_kernel(const float* __restrict__ wqkv, const float* __restrict__ bqkv, const float* __restrict__ wproj,
                                     const float* __restrict__ bproj, int C, h16* __restrict__ wfrag, float* __restrict__ bias) {
    const int NB = C / 16, total = 4 * NB * NB * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 3, lane = (e >> 2) & 63, idx = e >> 8;
        const int ks = idx % NB, f = (idx / NB) % NB, part = idx / (NB * NB);
        const int row = 16 * f + (lane & 15), k = 16 * ks + 4 * (lane >> 4) + j;
        wfrag[e] = (h16)(part < 3 ? wqkv[(size_t)(part * C + row) * C + k] : wproj[(size_t)row * C + k]);
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 4 * C; c += gridDim.x * blockDim.x)
        bias[c] = c < 3 * C ? (bqkv ? bqkv[c] : 0.f) : (bproj ? bproj[c - 3 * C] : 0.f);
}

extern "C" size_t mstg_f16_attn_plan_bytes(int C) {
    if (C != 16 && C != 32 && C != 64) return 0;
    return (size_t)4 * C * sizeof(float) + (size_t)4 * C * C * sizeof(h16);
}

extern "C" int mstg_f16_attn_pack(const float* wqkv, const float* bqkv, const float* wproj, const float* bproj, int C, void* blob,
                                  size_t blob_bytes, void* stream) {
    if (!wqkv || !wproj || !blob) return mstg::fail_arg(MSTG_E_BADARG, "f16 attn pack: null pointer");
    if (C != 16 && C != 32 && C != 64) return mstg::fail_arg(MSTG_E_UNSUPPORTED, "f16 attn: C must be 16, 32 or 64");
    if (blob_bytes < mstg_f16_attn_plan_bytes(C)) return mstg::fail_arg(MSTG_E_WORKSPACE, "f16 attn pack: blob too small");
    float* bias = (float*)blob;
    h16* wfrag = (h16*)((char*)blob + (size_t)4 * C * sizeof(float));
    MSTG_LAUNCH(f16_attn_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, wqkv, bqkv, wproj, bproj, C, wfrag, bias);
    MSTG_CHECK_LAUNCH("f16_attn_pack_kernel");
    return MSTG_OK;
}

template <int C>
static int launch_attn_f16(const void* x, const float* in_stats, const void* blob, void* y, int N, int H, int W, hipStream_t st) {
    typedef AttnF16<C> T;
    const float* bias = (const float*)blob;
    const h16* wfrag = (const h16*)((const char*)blob + (size_t)4 * C * sizeof(float));
    const size_t lds = (T::WLDS ? (size_t)T::NFRAG * 64 * 8 : 0) + (size_t)4 * T::END * sizeof(h16);
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f16_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const char* e_reg = getenv("MSTG_F16_ATTN_REG");
    if (!(e_reg && e_reg[0] == '0')) {  // register-resident chains (default); MSTG_F16_ATTN_REG=0: the LDS-tile kernel of round 2
        const size_t lds_r = T::WLDS ? (size_t)T::NFRAG * 64 * 8 : 0;
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
            if (cus <= 0) cus = 256;
        }
        const int nwin = N * (H / 4) * (W / 4);
        int nb = cus * (C == 64 ? 2 : (C == 32 ? 4 : 8));  // persistent workgroups: what the kernel's registers allow per SIMD
        if (nb * 4 > nwin) nb = cdiv(nwin, 4);
        MSTG_LAUNCH((attn_f16r_kernel<C>), dim3(nb), dim3(256), lds_r, st, (const h16*)x, in_stats, wfrag, bias, (h16*)y, N, H, W);
        MSTG_CHECK_LAUNCH("attn_f16r_kernel");
        return MSTG_OK;
    }
    dim3 grid(cdiv(W / 4, 4 * ATT_WPW), H / 4, N);
    MSTG_LAUNCH((attn_f16_kernel<C>), grid, dim3(256), lds, st, (const h16*)x, in_stats, wfrag, bias, (h16*)y, N, H, W);
    MSTG_CHECK_LAUNCH("attn_f16_kernel");
    return MSTG_OK;
}

extern "C" int mstg_f16_attn_fwd(const void* x, const float* in_stats, const void* blob, void* y, int N, int H, int W, int C,
                                 void* stream) {
    if (!x || !blob || !y) return fail_arg(MSTG_E_BADARG, "f16 attn: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "f16 attn: empty tensor");
    if (H % 4 || W % 4) return fail_arg(MSTG_E_BADARG, "f16 attn: H and W must be multiples of the 4x4 window");
    if (N > 65535 || H / 4 > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "f16 attn: grid too large");
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 16: return launch_attn_f16<16>(x, in_stats, blob, y, N, H, W, st);
        case 32: return launch_attn_f16<32>(x, in_stats, blob, y, N, H, W, st);
        case 64: return launch_attn_f16<64>(x, in_stats, blob, y, N, H, W, st);
        default: return fail_arg(MSTG_E_UNSUPPORTED, "f16 attn: C must be 16, 32 or 64");
    }
}
