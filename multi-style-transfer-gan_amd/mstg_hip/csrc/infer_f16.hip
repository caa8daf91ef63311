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

#include "common.h"

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

namespace mstg {

constexpr int F16_MAX_STEPS = 64;
constexpr int F16_TW = 16;  // tile width in output pixels of the compute grid (= one MFMA N fragment)

// how the pack kernel fills one half (4 k-elements) of a lane group
enum : int8_t { PK_ZERO = 0, PK_CONV = 1, PK_CONVT = 2, PK_MS_CENTER = 3, PK_MS_RING1 = 4 /* +0,1,2 = branches 2,3,4 */ };
struct PackHalf { int8_t mode, ky, kx; int8_t pad; int16_t cb; };
struct PackTable { PackHalf h[F16_MAX_STEPS][4][2]; };

struct F16Plan {
    int nsteps, ncls;
    int cls_begin[5];
    uint16_t koff[F16_MAX_STEPS][4];  // byte offset of the group's 16 bytes relative to the lane's pixel base in the patch
    uint8_t fmask[F16_MAX_STEPS];     // output-channel fragments the step feeds
    int8_t cls_oy[4], cls_ox[4];      // ConvTranspose: output parity of the class
    int PH, PW, pixstride;            // patch rows / cols / bytes per pixel
    int oy0, ox0;                     // patch origin = tile origin * stride + (oy0, ox0)
    int stride, up;                   // compute-grid stride in the source; up = 1: outputs at 2*y+oy, 2*x+ox
    int NF, TH;
};

struct F16ConvArgs {
    const void* x;          // NHWC fp16, or NCHW fp32 (src = 1)
    h16* y;                 // NHWC fp16, or NCHW fp16 (dst = 1)
    const h16* wpk;         // [step][frag][lane][8]
    const float* bias;      // [16 * NF]
    const float* in_stats;  // nullable [N][Cin][2]: normalise + ReLU while staging
    float* partial;         // nullable [N][tiles][2][16 * NF]
    int N, H, W, Cin, Ho, Wo, Cout;
    int Gh, Gw;             // compute grid per image (= Ho x Wo, or H x W for up)
    int tiles_x, tiles_y;
    int act;
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

__global__ void f16_pack_kernel(PackTable t, PackSrc s, int nsteps, int NF, h16* __restrict__ wpk, float* __restrict__ bias) {
    const int total = nsteps * NF * 64 * 8;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 7, lane = (e >> 3) & 63, f = (e >> 9) % NF, step = (e >> 9) / NF;
        const int m = lane & 15, g = lane >> 4, hf = j >> 2, jj = j & 3;
        const PackHalf ph = t.h[step][g][hf];
        const int co = 16 * f + m, ci = ph.cb + jj;
        float v = 0.f;
        if (ph.mode != PK_ZERO && co < s.Cout && ci < s.Cin && ph.cb >= 0) {
            if (ph.mode == PK_CONV) {
                v = s.w[0][((size_t)(co * s.Cin + ci) * s.KH + ph.ky) * s.KW + ph.kx];
            } else if (ph.mode == PK_CONVT) {
                v = s.w[0][((size_t)(ci * s.Cout + co) * s.KH + ph.ky) * s.KW + ph.kx];
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
// the convolution kernel.  256 threads; wave w computes rows RPW*w .. RPW*w + RPW-1 of a (4*RPW) x 16 tile of the compute grid
// for all 16*NF output channels.
// -------------------------------------------------------------------------------------------------------------------------
template <int RPW, int NF, int SRC, int DST>
__global__ __launch_bounds__(256) void conv_f16_kernel(const F16ConvArgs a, const F16Plan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 4 * RPW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ntile = a.tiles_x * a.tiles_y;
    int t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n = t / ntile;
    t -= n * ntile;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int gy0 = ty * TH, gx0 = tx * F16_TW;                       // tile origin in the compute grid
    const int sy0 = gy0 * p.stride + p.oy0, sx0 = gx0 * p.stride + p.ox0;  // patch origin in the source

    // ---- stage the patch --------------------------------------------------------------------------------------------------
    if (SRC == 0) {
        const h16* src = reinterpret_cast<const h16*>(a.x) + (size_t)n * a.H * a.W * a.Cin;
        const int oct = a.Cin >> 3;                    // 16-byte chunks per pixel (a power of two: 2, 4, 8)
        const int total = p.PH * p.PW * oct;
        const int o = tid & (oct - 1);                  // this thread's channel octet: the same for all its elements
        float mu[8], rs[8];
        const bool norm = a.in_stats != nullptr;
        if (norm) {
            const float* st = a.in_stats + ((size_t)n * a.Cin + 8 * o) * 2;
#pragma unroll
            for (int c = 0; c < 8; ++c) { mu[c] = st[2 * c]; rs[c] = st[2 * c + 1]; }
        }
        for (int e0 = 0; e0 < total; e0 += 1024) {
            h16x8 v[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 256 * k + tid;
                const int pix = e / oct;
                const int r = pix / p.PW, c = pix - r * p.PW;
                const int iy = sy0 + r, ix = sx0 + c;
                ok[k] = e < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const size_t off = ok[k] ? ((size_t)iy * a.W + ix) * a.Cin + 8 * o : 0;
                v[k] = *reinterpret_cast<const h16x8*>(src + off);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 256 * k + tid;
                if (e < total) {
                    h16x8 w = v[k];
                    if (norm) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float f = ((float)w[c] - mu[c]) * rs[c];
                            w[c] = (h16)(f > 0.f ? f : 0.f);
                        }
                    }
                    if (!ok[k]) w = h16x8{0, 0, 0, 0, 0, 0, 0, 0};  // zero padding applies to the NORMALISED activation
                    *reinterpret_cast<h16x8*>(smem + (size_t)(e / oct) * p.pixstride + 16 * o) = w;
                }
            }
        }
    } else {  // NCHW fp32 image with Cin (<= 4) planes -> [pixel][4] fp16
        const float* src = reinterpret_cast<const float*>(a.x) + (size_t)n * a.Cin * a.H * a.W;
        const int total = p.PH * p.PW;
        const size_t plane = (size_t)a.H * a.W;
        for (int e = tid; e < total; e += 256) {
            const int r = e / p.PW, c = e - r * p.PW;
            const int iy = sy0 + r, ix = sx0 + c;
            h16x4 w = {0, 0, 0, 0};
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const size_t off = (size_t)iy * a.W + ix;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < a.Cin) w[ch] = (h16)src[ch * plane + off];
            }
            *reinterpret_cast<h16x4*>(smem + (size_t)e * 8) = w;
        }
    }
    __syncthreads();

    // ---- main loop ----------------------------------------------------------------------------------------------------------
    const int nl = lane & 15, g = lane >> 4;
    unsigned base[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) base[r] = (unsigned)(((RPW * wv + r) * p.stride * p.PW + nl * p.stride) * p.pixstride);
    float ssum[NF][4], ssq[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[f][r] = ssq[f][r] = 0.f;
    const bool want_stats = a.partial != nullptr;
    const h16x8* wp = reinterpret_cast<const h16x8*>(a.wpk) + lane;

    for (int cls = 0; cls < p.ncls; ++cls) {
        f32x4 acc[RPW][NF];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[r][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int s0 = p.cls_begin[cls], s1 = p.cls_begin[cls + 1];
        for (int s = s0; s < s1; ++s) {
            const unsigned k0 = p.koff[s][0], k1 = p.koff[s][1], k2 = p.koff[s][2], k3 = p.koff[s][3];
            const unsigned ko = g == 0 ? k0 : (g == 1 ? k1 : (g == 2 ? k2 : k3));
            const unsigned fm = p.fmask[s];
            h16x8 bf[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) bf[r] = *reinterpret_cast<const h16x8*>(smem + base[r] + ko);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (NF == 1 || ((fm >> f) & 1)) {
                    const h16x8 af = wp[(size_t)(s * NF + f) * 64];
#pragma unroll
                    for (int r = 0; r < RPW; ++r) acc[r][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[r], acc[r][f], 0, 0, 0);
                }
            }
        }
        // ---- epilogue of this class: lane holds output channels 16f + 4g + {0..3} of compute-grid pixel (gy, gx0 + nl) -----------
        const int oyc = p.up ? p.cls_oy[cls] : 0, oxc = p.up ? p.cls_ox[cls] : 0, mul = p.up ? 2 : 1;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int gy = gy0 + RPW * wv + r, gx = gx0 + nl;
            const bool inside = gy < a.Gh && gx < a.Gw;
            const int oy = gy * mul + oyc, ox = gx * mul + oxc;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 16 * f + 4 * g);
                f32x4 v = acc[r][f] + b4;
                if (want_stats && inside) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ssum[f][q] += v[q]; ssq[f][q] += v[q] * v[q]; }
                }
                if (inside) {
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
                                const float o = a.act == MSTG_ACT_TANH ? tanhf(v[q]) : v[q];
                                a.y[(((size_t)n * a.Cout + q) * a.Ho + oy) * a.Wo + ox] = (h16)o;
                            }
                    }
                }
            }
        }
    }
    // ---- statistics of what this tile wrote: per channel, over the tile's pixels ------------------------------------------------
    if (want_stats) {
        __syncthreads();  // the patch is no longer needed: reuse LDS for the cross-wave reduction
        float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][16 * NF]
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = row16_sum_f(ssum[f][q]), s2 = row16_sum_f(ssq[f][q]);
                if (nl == 0) {
                    red[(wv * 2 + 0) * 16 * NF + 16 * f + 4 * g + q] = s1;
                    red[(wv * 2 + 1) * 16 * NF + 16 * f + 4 * g + q] = s2;
                }
            }
        __syncthreads();
        if (tid < 2 * 16 * NF) {
            const float s = red[tid] + red[2 * 16 * NF + tid] + red[4 * 16 * NF + tid] + red[6 * 16 * NF + tid];
            a.partial[((size_t)n * ntile + t) * 2 * 16 * NF + tid] = s;
        }
    }
}

// partial [N][tiles][2][CP] -> stats [N][C][2] = (mean, rstd); one workgroup per image, double accumulation, fixed order
__global__ __launch_bounds__(256) void f16_norm_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                                int tiles, int CP, int C, float count) {
    __shared__ double red[256 * 2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int lanes_per_c = 256 / CP;            // CP in {16, 32, 64} -> 16 / 8 / 4 threads share a channel
    const int c = tid % CP, sub = tid / CP;
    double s1 = 0.0, s2 = 0.0;
    for (int tl = sub; tl < tiles; tl += lanes_per_c) {
        const float* pp = partial + ((size_t)n * tiles + tl) * 2 * CP;
        s1 += (double)pp[c];
        s2 += (double)pp[CP + c];
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
    float mu[8], rs[8];
    const float* st = stats + ((size_t)n * C + 8 * o) * 2;
#pragma unroll
    for (int c = 0; c < 8; ++c) { mu[c] = st[2 * c]; rs[c] = st[2 * c + 1]; }
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
        for (int c = 0; c < 8; ++c) {
            const float f = ((float)a[c] - mu[c]) * rs[c];
            w[c] = (h16)((f > 0.f ? f : 0.f) + (float)r[c]);
        }
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
    p.pixstride = image_src ? 8 : 2 * Cin + 16;
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
        if ((image_src ? K * ((K + 1) / 2) : K * K * ngrp) > F16_MAX_STEPS * 4) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: too many taps x channels"); }
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
    // ---- tile height: the tallest of 16 / 8 rows whose patch (plus room for the statistics scratch) fits 64 KiB ----------------
    const int ext = halo_hi - halo_lo;  // rows / cols beyond (T - 1) * stride + 1
    int TH = 16;
    for (;; TH = 8) {
        p.PH = (TH - 1) * p.stride + 1 + ext;
        p.PW = (F16_TW - 1) * p.stride + 1 + ext + extra_w;
        if ((size_t)p.PH * p.PW * p.pixstride <= 64 * 1024 || TH == 8) break;
    }
    if ((size_t)p.PH * p.PW * p.pixstride > 160 * 1024) { delete[] gl; return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: patch does not fit LDS"); }
    p.TH = TH;
    p.oy0 = p.ox0 = halo_lo;
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
            p.fmask[s] = fm & (uint8_t)((1u << p.NF) - 1);
        }
    }
    p.cls_begin[p.ncls] = s;
    p.nsteps = s;
    delete[] gl;
    return MSTG_OK;
}

static size_t plan_blob_bytes(const F16Plan& p) { return 256 + (size_t)p.nsteps * p.NF * 64 * 16; }

template <int RPW, int NF>
static int launch_conv(const F16ConvArgs& a, const F16Plan& p, int src, int dst, size_t lds, int grid, hipStream_t st) {
    if (src == 0 && dst == 0) hipLaunchKernelGGL((conv_f16_kernel<RPW, NF, 0, 0>), dim3(grid), dim3(256), lds, st, a, p);
    else if (src == 1 && dst == 0) hipLaunchKernelGGL((conv_f16_kernel<RPW, NF, 1, 0>), dim3(grid), dim3(256), lds, st, a, p);
    else if (src == 0 && dst == 1) hipLaunchKernelGGL((conv_f16_kernel<RPW, NF, 0, 1>), dim3(grid), dim3(256), lds, st, a, p);
    else return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: NCHW source and destination in one layer");
    MSTG_CHECK_LAUNCH("conv_f16_kernel");
    return MSTG_OK;
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
    hipLaunchKernelGGL(f16_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, *pt, s, p.nsteps, p.NF, wpk, bias);
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
    const int Gh = p.up ? d->H : d->Ho, Gw = p.up ? d->W : d->Wo;
    return (size_t)d->N * cdiv(Gh, p.TH) * cdiv(Gw, F16_TW) * 2 * 16 * p.NF * sizeof(float);
}

extern "C" int mstg_f16_conv_fwd(const mstg_f16_conv_desc* d, const void* blob, const void* x, const float* in_stats, void* y,
                                 float* out_stats, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !blob || !x || !y) return fail_arg(MSTG_E_BADARG, "f16 conv: null pointer");
    F16Plan p;
    PackTable* pt = new PackTable;
    const int rc = build_plan(d, p, *pt);
    delete pt;
    if (rc) return rc;
    if (in_stats && d->src_nchw_f32) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: no normalise-on-load for the image source");
    F16ConvArgs a;
    a.x = x; a.y = (h16*)y;
    a.bias = (const float*)blob;
    a.wpk = (const h16*)((const char*)blob + 256);
    a.in_stats = in_stats;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.Gh = p.up ? d->H : d->Ho;
    a.Gw = p.up ? d->W : d->Wo;
    a.tiles_y = cdiv(a.Gh, p.TH);
    a.tiles_x = cdiv(a.Gw, F16_TW);
    a.act = d->act;
    const long grid = (long)a.N * a.tiles_x * a.tiles_y;
    if (grid > 0x7fffffffL) return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: grid too large");
    a.partial = nullptr;
    if (out_stats) {
        const size_t need = (size_t)grid * 2 * 16 * p.NF * sizeof(float);
        if (!workspace || workspace_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "f16 conv: workspace too small for the statistics partials");
        a.partial = (float*)workspace;
    }
    size_t lds = (size_t)p.PH * p.PW * p.pixstride;
    if (lds < (size_t)8 * 16 * p.NF * sizeof(float)) lds = (size_t)8 * 16 * p.NF * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const int src = d->src_nchw_f32 ? 1 : 0, dst = d->dst_nchw ? 1 : 0;
    int lrc;
    if (p.TH == 16) {
        switch (p.NF) {
            case 1: lrc = launch_conv<4, 1>(a, p, src, dst, lds, (int)grid, st); break;
            case 2: lrc = launch_conv<4, 2>(a, p, src, dst, lds, (int)grid, st); break;
            case 4: lrc = launch_conv<4, 4>(a, p, src, dst, lds, (int)grid, st); break;
            default: return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: output channels must pad to 16, 32 or 64");
        }
    } else {
        switch (p.NF) {
            case 1: lrc = launch_conv<2, 1>(a, p, src, dst, lds, (int)grid, st); break;
            case 2: lrc = launch_conv<2, 2>(a, p, src, dst, lds, (int)grid, st); break;
            case 4: lrc = launch_conv<2, 4>(a, p, src, dst, lds, (int)grid, st); break;
            default: return fail_arg(MSTG_E_UNSUPPORTED, "f16 conv: output channels must pad to 16, 32 or 64");
        }
    }
    if (lrc) return lrc;
    if (out_stats) {
        const float count = (float)((size_t)d->Ho * d->Wo);
        // ConvTranspose: each compute-grid tile wrote 4 classes; the per-tile sums already cover all of them
        hipLaunchKernelGGL(f16_norm_finalize_kernel, dim3(a.N), dim3(256), 0, st, (const float*)a.partial, out_stats,
                           a.tiles_x * a.tiles_y, 16 * p.NF, d->Cout, count);
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
    hipLaunchKernelGGL(f16_norm_residual_kernel, dim3((unsigned)N * bpi), dim3(256), 0, (hipStream_t)stream, (const h16*)x,
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
    float mu[8], rs[8];
    const bool norm = in_stats != nullptr;
    if (norm) {
        const float* st = in_stats + ((size_t)n * C + 8 * o) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) { mu[c] = st[2 * c]; rs[c] = st[2 * c + 1]; }
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
                for (int c = 0; c < 8; ++c) {
                    const float f = ((float)v[c] - mu[c]) * rs[c];
                    v[c] = (h16)(f > 0.f ? f : 0.f);
                }
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
            for (int r = 0; r < 4; ++r) inv[r] = 1.f / fmaxf(sqrtf(row16_sum_f(ss[r])), 1e-12f);
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
            const float inv = 1.f / sum;
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
    hipLaunchKernelGGL(f16_attn_pack_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, wqkv, bqkv, wproj, bproj, C, wfrag, bias);
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
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f16_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dim3 grid(cdiv(W / 4, 4 * ATT_WPW), H / 4, N);
    hipLaunchKernelGGL((attn_f16_kernel<C>), grid, dim3(256), lds, st, (const h16*)x, in_stats, wfrag, bias, (h16*)y, N, H, W);
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
