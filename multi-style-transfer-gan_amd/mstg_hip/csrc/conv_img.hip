// Input gradient of the discriminator's image-side layer, Conv2d(3, C, 4, 2, 1) on the NCHW image (enhanced_generator.py:237): the path
// the generator's adversarial loss takes back into the fake image.
//
// Three image channels against 16 (or 64) feature channels: on the MFMA kernels (igemm_light_kernel) the 3-channel output is padded to
// 16 rows of the tile -- 8.6 TFLOP/s, 94 us a launch at batch 32 for 58 MB of traffic.  The arithmetic is tiny (768 multiply-adds per
// 2x2 block of image pixels at C = 16), so it runs on the vector pipe here: one thread per 2x2 block, the filter in LDS read as
// wave-uniform (broadcast) 16-byte words, no patch, no barrier after the filter is staged: 39 us.  MSTG_CONV_IMG=0 keeps the MFMA kernel
// (tests compare the two).  The FORWARD of the same layer was tried in the same form (one thread per output pixel x 16 channels): 92 us
// against 34 us on igemm_light_kernel -- 192 broadcast filter reads per pixel keep the LDS pipe busy for longer than the MFMA kernel
// takes, and the stride-2 NCHW gather costs 48 four-byte loads per thread -- so the forward stays on the MFMA kernel.
#include "common.h"
#include "igemm_args.h"

namespace mstg {

constexpr int IMG_T = 16;  // taps of the 4x4 filter

// input gradient: dy NHWC slice (N, Ho, Wo, Cr) of a tensor with x_ctot channels -> dx NCHW (N, CI, 2 Ho, 2 Wo).
// Image row y takes filter rows ky with y = 2 oy - 1 + ky: rows 2q and 2q+1 of block q read dy rows q-1 (ky 3), q (ky 1 | 2), q+1 (ky 0).
template <int CI>
__global__ __launch_bounds__(256) void conv_img_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                             int N, int Ho, int Wo, int Cr, int x_ctot, int x_coff, int w_so, int w_sr) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [tap][ci][Cr]
    const int tid = threadIdx.x;
    for (int e = tid; e < IMG_T * CI * Cr; e += 256) {
        const int cr = e % Cr, r = e / Cr, ci = r % CI, t = r / CI;
        wl[e] = w[(size_t)cr * w_sr + (size_t)ci * w_so + t];
    }
    __syncthreads();
    const long total = (long)N * Ho * Wo;
    const int H = 2 * Ho, W = 2 * Wo;
    const size_t plane = (size_t)H * W;
    for (long idx = (long)blockIdx.x * 256 + tid; idx < total; idx += (long)gridDim.x * 256) {
        const int qx = (int)(idx % Wo), qy = (int)((idx / Wo) % Ho), n = (int)(idx / ((long)Wo * Ho));
        float acc[2][2][CI];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) acc[a][b][ci] = 0.f;
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const int oy = qy + dr - 1;
#pragma unroll
            for (int dc = 0; dc < 3; ++dc) {
                const int ox = qx + dc - 1;
                const bool ok = (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
                if (!ok) continue;
                const float* src = dy + (((size_t)n * Ho + oy) * Wo + ox) * x_ctot + x_coff;
                for (int c4 = 0; c4 < Cr; c4 += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c4);
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int ky = a == 0 ? (dr == 0 ? 3 : (dr == 1 ? 1 : -1)) : (dr == 1 ? 2 : (dr == 2 ? 0 : -1));
                        if (ky < 0) continue;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const int kx = b == 0 ? (dc == 0 ? 3 : (dc == 1 ? 1 : -1)) : (dc == 1 ? 2 : (dc == 2 ? 0 : -1));
                            if (kx < 0) continue;
#pragma unroll
                            for (int ci = 0; ci < CI; ++ci) {
                                const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[((4 * ky + kx) * CI + ci) * Cr + c4]);
                                acc[a][b][ci] += (v[0] * wv[0] + v[1] * wv[1]) + (v[2] * wv[2] + v[3] * wv[3]);
                            }
                        }
                    }
                }
            }
        }
        float* img = dx + (size_t)n * CI * plane + (size_t)(2 * qy) * W + 2 * qx;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float2 o;
                o.x = acc[a][0][ci];
                o.y = acc[a][1][ci];
                *reinterpret_cast<float2*>(img + ci * plane + (size_t)a * W) = o;
            }
    }
}

static bool img_on() {
    const char* e = env_get(ENV_CONV_IMG);
    return !(e && e[0] == '0');
}

static bool img_k4s2(const IGemmArgs& a) {
    return a.KH == 4 && a.KW == 4 && a.stride == 2 && a.pad == 1 && a.dil == 1 && !a.flip && !a.accumulate;
}

bool img_dgrad_eligible(const IGemmArgs& a) {
    return img_on() && img_k4s2(a) && a.phase && !a.x_nchw && a.y_nchw && a.Co >= 1 && a.Co <= 3 && a.y_coff == 0 && a.y_ctot == a.Co &&
           a.Cr % 4 == 0 && a.Cr <= 64 && (a.x_coff & 3) == 0 && (a.x_ctot & 3) == 0 && a.act == MSTG_ACT_NONE && !a.bias &&
           a.Ho == 2 * a.H && a.Wo == 2 * a.W && (a.Wo & 1) == 0;
}

static unsigned img_grid(long total) {
    long g = (total + 255) / 256;
    const long cap = 256L * 16;  // 16 workgroups per CU
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

int launch_img_dgrad(const IGemmArgs& a, hipStream_t st) {
    // source grid = dy (a.H x a.W), destination = the image (a.Ho x a.Wo)
    const long total = (long)a.N * a.H * a.W;
    const size_t lds = (size_t)IMG_T * a.Co * a.Cr * sizeof(float);
#define MSTG_IMG_DGRAD(CI_)                                                                                                   \
    MSTG_LAUNCH(conv_img_dgrad_kernel<CI_>, dim3(img_grid(total)), dim3(256), lds, st, a.x, a.w, a.y, a.N, a.H, a.W, a.Cr, a.x_ctot, \
                a.x_coff, a.w_so, a.w_sr)
    if (a.Co == 3) MSTG_IMG_DGRAD(3);
    else if (a.Co == 2) MSTG_IMG_DGRAD(2);
    else MSTG_IMG_DGRAD(1);
#undef MSTG_IMG_DGRAD
    MSTG_CHECK_LAUNCH("conv_img_dgrad_kernel");
    return MSTG_OK;
}

}  // namespace mstg
