// Device-side image pre/post-processing of the callers either side of the generator:
//   * PIL's two-pass separable resampling of 8-bit RGB images (BILINEAR: torchvision Resize in MonetPhotoDataset, pretrain.py:32-37;
//     LANCZOS: the letterbox and the resize back in batch_process_images.py:183-230), bit-exact: the coefficient tables are built
//     on the HOST in double precision exactly as Pillow's Resample.c does (precompute_coeffs + normalize_coeffs_8bpc, 22-bit fixed
//     point), the passes are integer arithmetic with an 8-bit intermediate image like ImagingResampleHorizontal/Vertical_8bpc;
//   * canvas paste / crop (batch_process_images.py:196-199,221-233);
//   * ToTensor + Normalize(0.5, 0.5) with the optional 8x8-grid mask of MonetPhotoDataset.__getitem__ (pretrain.py:44-57);
//   * the output conversion (y + 1) / 2 -> clamp -> * 255 -> uint8 HWC (batch_process_images.py:213-217), same fp32 operations
//     in the same order as torch / numpy perform them.
// Byte movers: HBM/latency bound, nothing here is shaped into a GEMM.
#include <math.h>

#include "common.h"

namespace mstg {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= RS_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][xx][c] = clip8(2^21 + sum_x src[y0 + y][xmin + x][c] * k[xx][x]);  src rows have `src_w` pixels of 3 bytes
__global__ void resample_h_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int src_w, int y0,
                                     int rows, int out_w, int ksize, const int* __restrict__ kk, const int* __restrict__ bounds) {
    const int xx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xx >= out_w || y >= rows) return;
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const unsigned char* row = src + ((size_t)(y0 + y) * src_w + xmin) * 3;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
        const int w = k[x];
        s0 += row[3 * x] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    unsigned char* o = dst + ((size_t)y * out_w + xx) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// out[yy][x][c] = clip8(2^21 + sum_y src[ymin + y][x][c] * k[yy][y])
__global__ void resample_v_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int w, int out_h,
                                     int ksize, const int* __restrict__ kk, const int* __restrict__ bounds) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (x >= w || yy >= out_h) return;
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
        const unsigned char* p = src + ((size_t)(ymin + y) * w + x) * 3;
        const int wgt = k[y];
        s0 += p[0] * wgt;
        s1 += p[1] * wgt;
        s2 += p[2] * wgt;
    }
    unsigned char* o = dst + ((size_t)yy * w + x) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst (dh, dw) <- fill everywhere, then src (sh, sw) window [sy0.., sx0..) of size (ch, cw) pasted at (dy0, dx0); 3-byte pixels
__global__ void paste_u8_kernel(const unsigned char* __restrict__ src, int sh, int sw, int sy0, int sx0, int ch, int cw,
                                unsigned char* __restrict__ dst, int dh, int dw, int dy0, int dx0, int fill, int do_fill) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int ry = y - dy0, rx = x - dx0;
    unsigned char* o = dst + ((size_t)y * dw + x) * 3;
    if (ry >= 0 && ry < ch && rx >= 0 && rx < cw) {
        const unsigned char* p = src + ((size_t)(sy0 + ry) * sw + sx0 + rx) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    } else if (do_fill) {
        o[0] = o[1] = o[2] = (unsigned char)fill;
    }
}

// ToTensor + Normalize((0.5,)*3, (0.5,)*3) of the window [y0.., x0..) of size (H, W) of an HWC uint8 image, times the 8x8-grid mask
// (bit (i * 8 + j) of `grid` set = cell kept) when use_mask; out (3, H, W) fp32, mask_out (nullable) (3, H, W) fp32 of 0 / 1,
// image_out (nullable) = the unmasked normalised image
__global__ void u8_to_tensor_kernel(const unsigned char* __restrict__ src, int sw, int y0, int x0, int H, int W, float* __restrict__ out,
                                    float* __restrict__ image_out, float* __restrict__ mask_out, unsigned long long grid, int use_mask) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const unsigned char* p = src + ((size_t)(y0 + y) * sw + x0 + x) * 3;
    float m = 1.f;
    if (use_mask) {
        const int ps_y = H / 8, ps_x = W / 8;  // patch_size = img_size // 8 (pretrain.py:46)
        const int i = y / ps_y, j = x / ps_x;
        if (i < 8 && j < 8) m = ((grid >> (i * 8 + j)) & 1ull) ? 1.f : 0.f;
    }
    const size_t plane = (size_t)H * W, o = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = (float)p[c] / 255.f;   // ToTensor: byte -> float32, div(255)
        const float v = (t - 0.5f) / 0.5f;     // Normalize: sub_(mean).div_(std)
        out[c * plane + o] = v * m;
        if (image_out) image_out[c * plane + o] = v;
        if (mask_out) mask_out[c * plane + o] = m;
    }
}

// (y + 1) / 2 -> clamp(0, 1) -> * 255 -> uint8 (truncation), CHW fp32 -> HWC uint8
__global__ void tensor_to_u8_kernel(const float* __restrict__ y, int H, int W, unsigned char* __restrict__ dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y;
    if (x >= W || yy >= H) return;
    const size_t plane = (size_t)H * W, o = (size_t)yy * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = (y[c * plane + o] + 1.0f) / 2.0f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        if (!(t == t)) t = 0.f;  // numpy casts NaN to 0 on this platform; keep the byte defined
        dst[o * 3 + c] = (unsigned char)(t * 255.0f);
    }
}

// result = clip(orig * (1 - w) + styled * w, 0, 255).astype(uint8) as numpy evaluates it on uint8 arrays and a float64 weight
// (batch_process_images.py:308-310 with a scalar strength, :340-342 + :352 with a per-pixel weight map): two float64 products,
// each rounded, one rounded sum (no fused multiply-add -- numpy has none here), clip, truncation.  w0 = 1 - strength is formed on
// the host in the scalar case (Python's own double subtraction), per pixel in float64 with a map.  Byte arithmetic: HBM-bound.
__global__ void blend_u8_kernel(const unsigned char* __restrict__ orig, const unsigned char* __restrict__ styled, double w0, double w1,
                                const double* __restrict__ wmap, unsigned char* __restrict__ out, size_t npix) {
#pragma clang fp contract(off)  // numpy rounds each product and the sum; hipcc would fuse them (it does so even through __dmul_rn / __dadd_rn)
    const size_t px = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= npix) return;
    if (wmap) {
        w1 = wmap[px];
        w0 = 1.0 - w1;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double a = (double)orig[px * 3 + c] * w0;
        const double b = (double)styled[px * 3 + c] * w1;
        double r = a + b;
        r = r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r);
        if (!(r == r)) r = 0.0;  // a NaN weight: numpy's cast is undefined there; keep the byte defined
        out[px * 3 + c] = (unsigned char)r;
    }
}

// ---- Pillow's coefficient tables (host, double precision) ---------------------------------------------------------------------
static double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    if (x < 1.0) return 1.0 - x;
    return 0.0;
}
static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

}  // namespace mstg

using namespace mstg;

extern "C" int mstg_resample_ksize(int in_size, int out_size, int filter) {
    if (in_size <= 0 || out_size <= 0 || (filter != 0 && filter != 1)) return 0;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == 0 ? 1.0 : 3.0) * filterscale;
    return (int)ceil(support) * 2 + 1;
}

extern "C" int mstg_resample_coeffs(int in_size, int out_size, int filter, int* kk /* [out_size][ksize] */, int* bounds /* [out_size][2] */) {
    const int ksize = mstg_resample_ksize(in_size, out_size, filter);
    if (ksize == 0 || !kk || !bounds) return fail_arg(MSTG_E_BADARG, "resample_coeffs: bad size / filter / null pointer");
    double (*f)(double) = filter == 0 ? bilinear_filter : lanczos_filter;
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = (filter == 0 ? 1.0 : 3.0) * filterscale;
    double* k = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x = 0;
        for (; x < xmax; ++x) {
            const double w = f((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        for (x = 0; x < ksize; ++x)
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << RS_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << RS_PRECISION_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    delete[] k;
    return MSTG_OK;
}

extern "C" int mstg_resample_h_u8(const unsigned char* src, unsigned char* dst, int src_w, int y0, int rows, int out_w, int ksize,
                                  const int* kk, const int* bounds, void* stream) {
    if (!src || !dst || !kk || !bounds || src_w <= 0 || rows <= 0 || out_w <= 0 || ksize <= 0 || y0 < 0)
        return fail_arg(MSTG_E_BADARG, "resample_h_u8: bad argument");
    if (rows > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "resample_h_u8: more than 65535 rows");
    MSTG_LAUNCH(resample_h_u8_kernel, dim3(cdiv(out_w, 128), rows), dim3(128), 0, (hipStream_t)stream, src, dst, src_w, y0, rows,
                       out_w, ksize, kk, bounds);
    MSTG_CHECK_LAUNCH("resample_h_u8_kernel");
    return MSTG_OK;
}

extern "C" int mstg_resample_v_u8(const unsigned char* src, unsigned char* dst, int w, int out_h, int ksize, const int* kk,
                                  const int* bounds, void* stream) {
    if (!src || !dst || !kk || !bounds || w <= 0 || out_h <= 0 || ksize <= 0) return fail_arg(MSTG_E_BADARG, "resample_v_u8: bad argument");
    if (out_h > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "resample_v_u8: more than 65535 rows");
    MSTG_LAUNCH(resample_v_u8_kernel, dim3(cdiv(w, 128), out_h), dim3(128), 0, (hipStream_t)stream, src, dst, w, out_h, ksize, kk,
                       bounds);
    MSTG_CHECK_LAUNCH("resample_v_u8_kernel");
    return MSTG_OK;
}

extern "C" int mstg_paste_u8(const unsigned char* src, int sh, int sw, int sy0, int sx0, int ch, int cw, unsigned char* dst, int dh,
                             int dw, int dy0, int dx0, int fill /* < 0: keep what dst holds */, void* stream) {
    if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || ch < 0 || cw < 0) return fail_arg(MSTG_E_BADARG, "paste_u8: bad argument");
    if (sy0 < 0 || sx0 < 0 || sy0 + ch > sh || sx0 + cw > sw) return fail_arg(MSTG_E_BADARG, "paste_u8: source window outside the image");
    if (dh > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "paste_u8: more than 65535 rows");
    MSTG_LAUNCH(paste_u8_kernel, dim3(cdiv(dw, 128), dh), dim3(128), 0, (hipStream_t)stream, src, sh, sw, sy0, sx0, ch, cw, dst, dh,
                       dw, dy0, dx0, fill < 0 ? 0 : fill, fill >= 0 ? 1 : 0);
    MSTG_CHECK_LAUNCH("paste_u8_kernel");
    return MSTG_OK;
}

extern "C" int mstg_u8_to_tensor(const unsigned char* src, int sh, int sw, int y0, int x0, int H, int W, float* out, float* image_out,
                                 float* mask_out, unsigned long long grid, int use_mask, void* stream) {
    if (!src || !out || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "u8_to_tensor: bad argument");
    if (y0 < 0 || x0 < 0 || y0 + H > sh || x0 + W > sw) return fail_arg(MSTG_E_BADARG, "u8_to_tensor: window outside the image");
    if (H > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "u8_to_tensor: more than 65535 rows");
    MSTG_LAUNCH(u8_to_tensor_kernel, dim3(cdiv(W, 128), H), dim3(128), 0, (hipStream_t)stream, src, sw, y0, x0, H, W, out, image_out,
                       mask_out, grid, use_mask);
    MSTG_CHECK_LAUNCH("u8_to_tensor_kernel");
    return MSTG_OK;
}

extern "C" int mstg_tensor_to_u8(const float* y, int H, int W, unsigned char* dst, void* stream) {
    if (!y || !dst || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "tensor_to_u8: bad argument");
    if (H > 65535) return fail_arg(MSTG_E_UNSUPPORTED, "tensor_to_u8: more than 65535 rows");
    MSTG_LAUNCH(tensor_to_u8_kernel, dim3(cdiv(W, 128), H), dim3(128), 0, (hipStream_t)stream, y, H, W, dst);
    MSTG_CHECK_LAUNCH("tensor_to_u8_kernel");
    return MSTG_OK;
}

extern "C" int mstg_blend_u8(const unsigned char* orig, const unsigned char* styled, double one_minus_strength, double strength,
                             const double* weight_map, unsigned char* out, int H, int W, void* stream) {
    if (!orig || !styled || !out || H <= 0 || W <= 0) return fail_arg(MSTG_E_BADARG, "blend_u8: bad argument");
    const size_t npix = (size_t)H * W;
    MSTG_LAUNCH(blend_u8_kernel, dim3((unsigned)cdivz(npix, 256)), dim3(256), 0, (hipStream_t)stream, orig, styled, one_minus_strength,
                strength, weight_map, out, npix);
    MSTG_CHECK_LAUNCH("blend_u8_kernel");
    return MSTG_OK;
}
