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

namespace mstg {
thread_local char g_last_error[256] = "";

struct IGemmArgs {
    const float* x;
    float* y;
    const float* w;
    const float* bias;
    int N;
    int H, W, x_ctot, x_coff, x_nchw;   // source tensor
    int Cr;                             // reduction channels
    int Ho, Wo, y_ctot, y_coff, y_nchw; // destination tensor
    int Co;                             // output channels
    int Gh, Gw;                         // grid walked by the tiles (Ho x Wo, or the source grid in phase mode)
    int tiles_x, tiles_y;
    int KH, KW, stride, pad, dil, flip, phase;
    int w_so, w_sr;                     // weight strides of the output / reduction channel (taps are innermost)
    int PH, PW;                         // LDS patch extent
    int TG;                             // taps per weight-staging group
    int ntaps;
    int act, accumulate;
};

constexpr int TILE_H = 8, TILE_W = 16;
constexpr int W_BUDGET_FLOATS = 6144;  // 24 KiB of LDS for one tap group's filter slice

template <int V> struct Frag;
template <> struct Frag<1> { typedef float T; };
template <> struct Frag<2> { typedef f32x2 T; };
template <> struct Frag<4> { typedef f32x4 T; };
template <int V> __device__ __forceinline__ float frag_get(const typename Frag<V>::T& f, int j);
template <> __device__ __forceinline__ float frag_get<1>(const float& f, int) { return f; }
template <> __device__ __forceinline__ float frag_get<2>(const f32x2& f, int j) { return f[j]; }
template <> __device__ __forceinline__ float frag_get<4>(const f32x4& f, int j) { return f[j]; }

template <int V> __host__ __device__ constexpr int ckp_of() { return V == 4 ? 20 : (V == 2 ? 12 : 4); }

// V   : source channels per MFMA k-slot (a lane reads V consecutive channels; K chunk = 4V channels)
// NFW : 16-channel output fragments per workgroup (BN = 16*NFW)
template <int V, int NFW>
__global__ __launch_bounds__(256) void igemm_kernel(const IGemmArgs a) {
    constexpr int CK = 4 * V, CKP = ckp_of<V>(), BN = 16 * NFW;
    typedef typename Frag<V>::T frag_t;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;
    float* wl = smem + ((a.PH * a.PW * CKP + 3) & ~3);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, g = lane >> 4;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx0 = tile % a.tiles_x, ty0 = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int co0 = blockIdx.y * BN;
    const int pa = blockIdx.z >> 1, pb = blockIdx.z & 1;
    const int s = a.phase ? 1 : a.stride;
    const int y0 = a.phase ? ty0 * TILE_H - 1 : ty0 * TILE_H * s - a.pad;
    const int x0 = a.phase ? tx0 * TILE_W - 1 : tx0 * TILE_W * s - a.pad;
    const int nchunks = (a.Cr + CK - 1) / CK;

    f32x4 acc[NFW][2];
#pragma unroll
    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
        for (int pf = 0; pf < 2; ++pf) acc[wf][pf] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk) __syncthreads();
        // ---- stage the source patch (halo included, zero outside the image) -------------------------------
        if (a.x_nchw) {  // 3-channel image tensor, V == 1: channel 3 of the k-slot group is zero
            for (int pr = wave; pr < a.PH; pr += 4) {
                const int iy = y0 + pr;
                for (int pc = lane; pc < a.PW; pc += 64) {
                    const int ix = x0 + pc;
                    const bool inb = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (inb) {
                        const size_t base = (((size_t)n * a.x_ctot + a.x_coff) * a.H + iy) * a.W + ix;
                        const size_t cs = (size_t)a.H * a.W;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c < a.Cr) v[c] = a.x[base + c * cs];
                    }
                    *reinterpret_cast<f32x4*>(&patch[(pr * a.PW + pc) * CKP]) = v;
                }
            }
        } else {
            const bool al = ((a.x_ctot | a.x_coff) & 3) == 0;
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
                        if (al && c0 + 3 < a.Cr) {
                            v = *reinterpret_cast<const f32x4*>(src);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                if (c0 + c < a.Cr) v[c] = src[c];
                        }
                    }
                    *reinterpret_cast<f32x4*>(&patch[(pr * a.PW + pc) * CKP + 4 * q]) = v;
                }
            }
        }
        for (int t0 = 0; t0 < a.ntaps; t0 += a.TG) {
            __syncthreads();  // patch staged / previous tap group consumed
            const int tn = min(a.TG, a.ntaps - t0);
            // ---- stage this tap group's filter slice as wl[tap][co][ci] ------------------------------------
            for (int idx = tid; idx < tn * BN * CK; idx += 256) {
                const int ci = idx % CK, rest = idx / CK;
                const int col = rest % BN, tl = rest / BN;
                const int t = t0 + tl;
                int widx;
                if (a.phase) {
                    const int u = t >> 1, v = t & 1;
                    widx = ((1 - pa) + 2 * u) * 4 + ((1 - pb) + 2 * v);
                } else {
                    widx = a.flip ? (a.ntaps - 1 - t) : t;
                }
                const int co = co0 + col, cr = chunk * CK + ci;
                float val = 0.f;
                if (co < a.Co && cr < a.Cr) val = a.w[(size_t)co * a.w_so + (size_t)cr * a.w_sr + widx];
                wl[(tl * BN + col) * CKP + ci] = val;
            }
            __syncthreads();
            // ---- MFMA over the group's taps -----------------------------------------------------------------
            int ky = t0 / a.KW, kx = t0 % a.KW;  // running tap coordinates (gather mode)
            for (int tl = 0; tl < tn; ++tl) {
                const int t = t0 + tl;
                int pro, pco;
                if (a.phase) {
                    pro = 1 + pa - (t >> 1);
                    pco = 1 + pb - (t & 1);
                } else {
                    pro = ky * a.dil;
                    pco = kx * a.dil;
                    if (++kx == a.KW) { kx = 0; ++ky; }
                }
                frag_t af[NFW], bf[2];
#pragma unroll
                for (int wf = 0; wf < NFW; ++wf)
                    af[wf] = *reinterpret_cast<const frag_t*>(&wl[(tl * BN + 16 * wf + i) * CKP + V * g]);
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    const int r = 2 * wave + pf;
                    bf[pf] = *reinterpret_cast<const frag_t*>(&patch[((r * s + pro) * a.PW + (i * s + pco)) * CKP + V * g]);
                }
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int wf = 0; wf < NFW; ++wf)
#pragma unroll
                        for (int pf = 0; pf < 2; ++pf)
                            acc[wf][pf] = mfma16(frag_get<V>(af[wf], j), frag_get<V>(bf[pf], j), acc[wf][pf]);
            }
        }
    }

    // ---- epilogue: bias, optional accumulate, activation, store ---------------------------------------------
#pragma unroll
    for (int pf = 0; pf < 2; ++pf) {
        const int gy = ty0 * TILE_H + 2 * wave + pf, gx = tx0 * TILE_W + i;
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
                    float* p = a.y_nchw
                                   ? a.y + (((size_t)n * a.y_ctot + a.y_coff + co + e) * a.Ho + oy) * a.Wo + ox
                                   : a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.y_ctot + a.y_coff + co + e;
                    float val = v[e] + (a.bias ? a.bias[co + e] : 0.f);
                    if (a.accumulate) val += *p;
                    *p = apply_act(val, a.act);
                }
            }
        }
    }
}

template <int V, int NFW>
static int launch_igemm_t(IGemmArgs& a, hipStream_t st) {
    constexpr int CKP = ckp_of<V>(), BN = 16 * NFW;
    a.TG = W_BUDGET_FLOATS / (BN * CKP);
    if (a.TG < 1) a.TG = 1;
    if (a.TG > a.ntaps) a.TG = a.ntaps;
    const size_t lds = ((size_t)((a.PH * a.PW * CKP + 3) & ~3) + (size_t)a.TG * BN * CKP) * sizeof(float);
    if (lds > 160 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "conv: LDS patch too large for this geometry");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<V, NFW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_launch(e, "hipFuncSetAttribute(igemm)");
        attr_set = true;
    }
    dim3 grid(a.N * a.tiles_x * a.tiles_y, cdiv(a.Co, BN), a.phase ? 4 : 1);
    hipLaunchKernelGGL((igemm_kernel<V, NFW>), grid, dim3(256), lds, st, a);
    MSTG_CHECK_LAUNCH("igemm_kernel");
    return MSTG_OK;
}

int launch_igemm(IGemmArgs& a, hipStream_t st) {
    a.tiles_x = cdiv(a.Gw, TILE_W);
    a.tiles_y = cdiv(a.Gh, TILE_H);
    if (a.phase) {
        a.PH = TILE_H + 2;
        a.PW = TILE_W + 2;
        a.ntaps = 4;
    } else {
        a.PH = (TILE_H - 1) * a.stride + (a.KH - 1) * a.dil + 1;
        a.PW = (TILE_W - 1) * a.stride + (a.KW - 1) * a.dil + 1;
        a.ntaps = a.KH * a.KW;
    }
    if (a.N <= 0 || a.Gh <= 0 || a.Gw <= 0 || a.Co <= 0 || a.Cr <= 0) return fail_arg(MSTG_E_BADARG, "conv: empty tensor");
    int V;
    if (a.x_nchw) {
        if (a.Cr > 4) return fail_arg(MSTG_E_UNSUPPORTED, "conv: NCHW source supports at most 4 channels");
        V = 1;
    } else if (a.Cr % 16 == 0) V = 4;
    else if (a.Cr % 8 == 0) V = 2;
    else V = 1;  // any channel count: 4-channel k-slots, the tail quad zero-filled
    const int nfw = a.Co <= 16 ? 1 : (a.Co <= 32 ? 2 : 4);
#define MSTG_DISPATCH(VV, NN) \
    if (V == VV && nfw == NN) return launch_igemm_t<VV, NN>(a, st);
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
    return MSTG_OK;
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_conv2d_fwd(const mstg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                               void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!x || !w || !y) return fail_arg(MSTG_E_BADARG, "conv_fwd: null pointer");
    const int T = d->KH * d->KW;
    IGemmArgs a{};
    a.x = x; a.y = y; a.w = w; a.bias = bias;
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
    return launch_igemm(a, (hipStream_t)stream);
}

extern "C" int mstg_conv2d_dgrad(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, void* stream) {
    if (int rc = check_desc(d)) return rc;
    if (!dy || !w || !dx) return fail_arg(MSTG_E_BADARG, "conv_dgrad: null pointer");
    const int T = d->KH * d->KW;
    IGemmArgs a{};
    a.x = dy; a.y = dx; a.w = w; a.bias = nullptr;
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
    return launch_igemm(a, (hipStream_t)stream);
}

extern "C" const char* mstg_version(void) { return "mstg-hip 0.1.0 gfx950"; }
extern "C" const char* mstg_arch(void) { return "gfx950"; }
extern "C" const char* mstg_last_error(void) { return mstg::g_last_error; }
