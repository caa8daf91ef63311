// Gather description shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_p32.hip): forward passes and input
// gradients of Conv2d / ConvTranspose2d are all "destination pixel = sum over taps and reduction channels of a source pixel".
#pragma once
#include "common.h"

namespace mstg {

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
    int dbg;
    int psz;           // stream kernel: floats reserved for the LDS patch (>= the dpack exchange tiles)
    int wglob;         // light kernel: filter fragments straight from the packed filter in L2 (no LDS filter slice)
    int TH;            // tile height in grid rows: 8, or 16 where the light kernel gives each wave four rows
    int dpack, tapsx;  // <= 4 output channels: rows of the MFMA tile = (pixel shift delta, channel), see igemm_light_kernel
};

// conv_p32.hip: persistent, software-pipelined kernel for the 4x4 stride-2 family (Conv2d k4 s2 p1, ConvTranspose2d k4 s2 p1 and
// their input gradients) at 16 / 32 / 64 channels
bool p32_eligible(const IGemmArgs& a);
size_t p32_workspace_bytes(const IGemmArgs& a);
int launch_p32(const IGemmArgs& a, void* workspace, size_t workspace_bytes, hipStream_t st);
// the same launch with InstanceNorm folded in on either side: in_stats (nullable) = (mean, rstd) of the raw source, normalised +
// ReLU'd while staged; out_stats (nullable) = (mean, rstd) of the output, summed in the epilogue
size_t p32_norm_workspace_bytes(const IGemmArgs& a);
int launch_p32_norm(const IGemmArgs& a, const float* in_stats, float* out_stats, void* workspace, size_t workspace_bytes, hipStream_t st);
// input-gradient launch that also emits the reduction sums of the InstanceNorm + ReLU backward its output feeds (conv_p32.hip)
bool p32_generic(const IGemmArgs& a);
bool p32_bsums_pays(const IGemmArgs& a);
bool p32_stats_pays(const IGemmArgs& a);
int launch_p32_bsums(const IGemmArgs& a, const float* aux, const float* aux_stats, float* sums, void* workspace, size_t workspace_bytes,
                     hipStream_t st);
const char* p32_kernel_name(const IGemmArgs& a);

// conv_img.hip: input gradient of the discriminator's image-side layer, Conv2d(3, C, 4, 2, 1) on the NCHW image, on the vector pipe
bool img_dgrad_eligible(const IGemmArgs& a);
int launch_img_dgrad(const IGemmArgs& a, hipStream_t st);

}  // namespace mstg
