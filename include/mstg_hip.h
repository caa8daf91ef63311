/*
 * mstg_hip.h -- C ABI of the MI355X (gfx950) hot-path library for multi-style-transfer-gan.
 *
 * The reference has no FFI / operator boundary of its own: its hot path is torch.nn layers called from
 * enhanced_generator.py.  This header is the boundary the reference would bind instead of
 * those layers; every entry point names the reference call site it replaces.  Conventions:
 *
 *   - plain C, device pointers + explicit sizes + a hipStream_t passed as void*; no torch types;
 *   - every function returns 0 on success or a negative MSTG_E_* code (the Python shim raises RuntimeError);
 *   - the caller owns every buffer, workspaces included (sizes from the *_workspace_bytes queries);
 *   - the library keeps no mutable global state; calls are asynchronous on `stream`;
 *   - activations are fp32 NHWC ("channels last") unless a descriptor flag says NCHW (only the 3-channel
 *     image tensors at the module boundary are NCHW); weights stay in PyTorch's own layouts
 *     (Conv2d OIHW, ConvTranspose2d IOHW) so reference state_dicts are consumed as they are.
 */
#ifndef MSTG_HIP_H
#define MSTG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSTG_OK 0
#define MSTG_E_BADARG (-1)      /* inconsistent sizes / unsupported geometry */
#define MSTG_E_ALIGN (-2)       /* channel counts / offsets violate the alignment the kernels need */
#define MSTG_E_LAUNCH (-3)      /* hipLaunch failed; see mstg_last_error() */
#define MSTG_E_WORKSPACE (-4)   /* workspace too small */
#define MSTG_E_UNSUPPORTED (-5) /* valid request outside what this build implements */

/* activation codes shared by conv epilogues and the norm kernels */
#define MSTG_ACT_NONE 0
#define MSTG_ACT_RELU 1
#define MSTG_ACT_LEAKY02 2 /* LeakyReLU(0.2), enhanced_generator.py:238-251, pretrain.py:67-76 */
#define MSTG_ACT_TANH 3    /* nn.Tanh, enhanced_generator.py:138 */
#define MSTG_ACT_GELU 4    /* nn.GELU() (erf form): build-defined StructuralTransformerBlock only */

const char* mstg_version(void);    /* "mstg-hip <semver> gfx950" */
const char* mstg_arch(void);       /* "gfx950" */
const char* mstg_last_error(void); /* text of the last failing HIP call on this thread, "" if none */
/* The MSTG_* environment switches are read when the library is loaded; call this after changing one. */
void mstg_env_refresh(void);

/* Per-launch profiler (measurement only; bench.py's `roofline` object).  While enabled, every kernel the library launches is
 * bracketed by two HIP events on the stream it is launched on.  mstg_prof_enable(1) clears earlier records; mstg_prof_get
 * waits for record i and returns its kernel symbol (as rocprofv3 --kernel-trace prints it, without return type, namespace and
 * parameter list) and its duration in milliseconds.  Off by default: no events are created on the product path. */
int mstg_prof_enable(int on);
int mstg_prof_count(void);
int mstg_prof_get(int i, char* name, size_t name_cap, float* ms);

/* ------------------------------------------------------------------------------------------------
 * Convolutions.  One descriptor describes the MODULE (nn.Conv2d or nn.ConvTranspose2d); the three
 * entry points are its forward, its input gradient and its weight/bias gradient.
 * Replaces: nn.Conv2d / nn.ConvTranspose2d at enhanced_generator.py:10-11,53-73,92,99,106,121,128,137,
 * 237-265 and pretrain.py:65-91 (their ATen convolution / convolution_backward dispatch).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mstg_conv_desc {
    int32_t N, H, W, Cin;   /* module input  (N,H,W,Cin)  */
    int32_t Ho, Wo, Cout;   /* module output (N,Ho,Wo,Cout) */
    int32_t KH, KW, stride, pad, dil;
    int32_t transposed;     /* 0: Conv2d (weight OIHW); 1: ConvTranspose2d k4 s2 p1 (weight IOHW) */
    int32_t x_nchw, y_nchw; /* 1: that tensor is NCHW (3-channel image boundary), else NHWC */
    int32_t x_ctot, x_coff; /* module input  = channels [x_coff, x_coff+Cin)  of an NHWC tensor with x_ctot channels */
    int32_t y_ctot, y_coff; /* module output = channels [y_coff, y_coff+Cout) of an NHWC tensor with y_ctot channels */
    int32_t act;            /* forward epilogue activation applied after bias: MSTG_ACT_NONE / _RELU / _LEAKY02 / _TANH */
    int32_t accumulate;     /* fwd: y += result; dgrad: dx += result (branches that share an input) */
} mstg_conv_desc;

/* Input gradient of a convolution whose INPUT was y = ReLU(InstanceNorm2d(x_raw)) (enhanced_generator.py:93-94, 72-75 + 84: the
 * stem norm in front of down1, a MultiScaleBlock's concat norm in front of its fusion convolution, a block's fusion norm in front
 * of the next stage's convolution): besides dx = dL/dy the launch emits sums[n][0][c] = sum_p dx [x^ > 0] and sums[n][1][c] =
 * sum_p dx [x^ > 0] x^ (x^ = (x_raw - mean) * rstd from x_stats [N][Cin][2]) from its epilogue -- the two reductions of that norm's
 * backward, which mstg_norm_bwd_apply(..., sums, 1, ...) consumes; the statistics pass over (x_raw, dx) is not needed.  Supported
 * where the persistent kernel runs the input gradient (mstg_conv2d_dgrad_bsums_supported); workspace_packed as for the *_cached
 * entry points below. */
int mstg_conv2d_dgrad_bsums_supported(const mstg_conv_desc* d);
size_t mstg_conv2d_dgrad_bsums_workspace_bytes(const mstg_conv_desc* d);
int mstg_conv2d_dgrad_bsums(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, const float* x_raw, const float* x_stats,
                            float* sums, void* workspace, size_t workspace_bytes, int workspace_packed, void* stream);

/* Filter-pack caching.  mstg_conv2d_fwd / _fwd_norm / _dgrad and mstg_msblock_fwd / _dgrad begin by re-packing the filter into the
 * caller's workspace (a ~5 us launch, ~170 of them per CycleGAN step).  The *_cached twins take workspace_packed: non-zero = "this
 * workspace still holds what the SAME call (same descriptor, same pass, same weight VALUES) packed into it", and the pack launch is
 * skipped.  Whether that is true is the caller's knowledge (weights change at optimizer.step(); mstg_hip/ops.py keeps one workspace
 * per (layer, pass, stream) and a stamp of the weights' version); with workspace_packed = 0 they are the plain entry points. */
int mstg_conv2d_fwd_cached(const mstg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace,
                           size_t workspace_bytes, int workspace_packed, void* stream);
int mstg_conv2d_fwd_norm_cached(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* w, const float* bias,
                                float* y, float* out_stats, void* workspace, size_t workspace_bytes, int workspace_packed, void* stream);
int mstg_conv2d_dgrad_cached(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                             int workspace_packed, void* stream);
int mstg_msblock_fwd_cached(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                            const float* b3, const float* w4, const float* b4, float* y, int N, int H, int W, int CH, void* workspace,
                            size_t workspace_bytes, int workspace_packed, void* stream);
int mstg_msblock_dgrad_cached(const float* dy, const float* w1, const float* w2, const float* w3, const float* w4, const float* dres,
                              float* dx, int N, int H, int W, int CH, void* workspace, size_t workspace_bytes, int workspace_packed,
                              void* stream);

/* name of the kernel a pass (0 forward, 1 dgrad, 2 wgrad) of this layer launches, as a profiler prints it (for reports) */
const char* mstg_conv2d_kernel_name(const mstg_conv_desc* d, int pass);
/* workspace of fwd and dgrad: room for the filter re-packed into the order the kernel stages it (a few 100 KB at most) */
size_t mstg_conv2d_workspace_bytes(const mstg_conv_desc* d);
int mstg_conv2d_fwd(const mstg_conv_desc* d, const float* x, const float* w, const float* bias /*nullable*/,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);
/* dx = d(loss)/d(module input), from dy = d(loss)/d(module output) */
/* Weight gradient of a 1x1 Conv2d whose input is ReLU(InstanceNorm2d(x_raw)) (the MultiScaleBlock's fusion convolution behind the
 * branch concat's norm, enhanced_generator.py:72-75, 83): x_raw is normalised with in_stats[n][c] = (mean, rstd) while the kernel
 * stages it, so the normalised tensor need not exist.  Supported where a staged pixel run stays inside one image. */
int mstg_conv2d_wgrad_norm_supported(const mstg_conv_desc* d);
int mstg_conv2d_wgrad_norm(const mstg_conv_desc* d, const float* x_raw, const float* in_stats, const float* dy, float* dw, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream);
/* Forward with InstanceNorm folded in on either side, for the layers the persistent kernel covers (4x4 stride-2 Conv2d /
 * ConvTranspose2d and 1x1 Conv2d at 16 / 32 / 64 channels, unsliced NHWC, no fused activation): in_stats (nullable) = (mean, rstd)
 * [N][Cin][2] of the RAW source -- it is normalised and ReLU'd while staged, i.e. x is what stands in front of
 * nn.InstanceNorm2d + nn.ReLU (enhanced_generator.py:54-75) and the normalised tensor is never written; out_stats (nullable) =
 * (mean, rstd) [N][Cout][2] of y, summed in the epilogue (same layout as mstg_norm_stats; sums of squares instead of pivoted
 * sums, combined in double).  Backward: mstg_conv2d_dgrad / _wgrad as for mstg_conv2d_fwd. */
int mstg_conv2d_fwd_norm_supported(const mstg_conv_desc* d);
/* 1 where the epilogue statistics also cost less than a statistics pass over the output (the callers' default routing) */
int mstg_conv2d_fwd_stats_pays(const mstg_conv_desc* d);
size_t mstg_conv2d_fwd_norm_workspace_bytes(const mstg_conv_desc* d);
int mstg_conv2d_fwd_norm(const mstg_conv_desc* d, const float* x, const float* in_stats, const float* w, const float* bias, float* y,
                         float* out_stats, void* workspace, size_t workspace_bytes, void* stream);
int mstg_conv2d_dgrad(const mstg_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace,
                      size_t workspace_bytes, void* stream);
/* dw (same layout as w) and dbias (nullable) ; workspace holds per-split partial sums (deterministic, no atomics);
 * d->accumulate != 0: dw += and dbias += (gradients accumulated straight into an optimizer's flat gradient buffer) */
size_t mstg_conv2d_wgrad_workspace_bytes(const mstg_conv_desc* d);
int mstg_conv2d_wgrad(const mstg_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias /*nullable*/,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm2d(affine=False, eps=1e-5, biased variance) fused with the activation that follows it and an
 * optional residual add.  Replaces nn.InstanceNorm2d + nn.ReLU / nn.LeakyReLU(0.2) pairs at
 * enhanced_generator.py:54-75,93-94,100-101,107-108,122-123,129-130,242-251,263-264 and the `+ x` of :84.
 *   y = act((x - mean) * rstd) [+ residual]      stats[n][c] = {mean, rstd} is saved for the backward.
 * batch_stats=1 turns it into BatchNorm2d (training: statistics over N,H,W; pretrain.py:69-89) with affine
 * gamma/beta and running-stat update (momentum 0.1, unbiased running_var); batch_stats=2 = BatchNorm eval.
 * ---------------------------------------------------------------------------------------------- */
size_t mstg_norm_workspace_bytes(int N, int HW, int C);
int mstg_norm_act_fwd(const float* x, const float* residual /*nullable*/, float* y, float* stats /* [N][C][2] */,
                      int N, int HW, int C, int act, int batch_stats, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, void* workspace, size_t workspace_bytes, void* stream);
/* dx from dy (gradient w.r.t. y); the residual's gradient is dy itself.  dgamma/dbeta only for batch_stats=1. */
int mstg_norm_act_bwd(const float* x, const float* stats, const float* dy, float* dx, int N, int HW, int C, int act,
                      int batch_stats, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The two halves on their own, for a norm whose neighbours do the other half (mstg_window_attn_norm_*): statistics only
 * (stats[n][c] = {mean, rstd}, bit-identical to what mstg_norm_act_fwd stores), and the backward's apply pass given
 * sums[n][s][2][C]: rows that add up, per (image, channel), to sum(dy * act') and sum(dy * act' * x^) (InstanceNorm only). */
int mstg_norm_apply_fwd(const float* x, const float* stats, const float* residual /*nullable*/, float* y, int N, int HW, int C, int act,
                        void* stream);  /* y = act((x - mean) * rstd) [+ residual] with the statistics given */
int mstg_norm_stats(const float* x, float* stats, int N, int HW, int C, void* workspace, size_t workspace_bytes, void* stream);
int mstg_norm_bwd_apply(const float* x, const float* stats, const float* dy, const float* sums /* [N][sums_split][2][C] */,
                        int sums_split, float* dx, int N, int HW, int C, int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LocalAttention core (enhanced_generator.py:22-35,39-42), window 4x4.  The qkv and proj 1x1 convolutions
 * (:28, :36) run through mstg_conv2d_*; this is everything between them, per window, with the reference's
 * window partition / un-partition permutes reduced to index arithmetic on NHWC:
 *   q,k L2-normalised per pixel over channels (F.normalize, eps 1e-12) ; attn = softmax_c2(sum_p q^[p,c1] k^[p,c2])
 *   (C x C per window, no scale) ; o[p,c1] = sum_c2 attn[c1,c2] v[p,c2].
 * qkv: NHWC (N,H,W,3C) with q|k|v channel blocks (= qkv.chunk(3, dim=1)); o: NHWC (N,H,W,C).
 * H, W multiples of 4 (anything else raises in the reference too); C a multiple of 4, <= 256 (row-blocked above 64).
 * ---------------------------------------------------------------------------------------------- */
int mstg_window_attn_core_fwd(const float* qkv, float* o, int N, int H, int W, int C, void* stream);
/* dqkv from d_o; the attention matrix is recomputed from qkv, nothing but qkv is saved by the forward */
int mstg_window_attn_core_bwd(const float* qkv, const float* d_o, float* dqkv, int N, int H, int W, int C, void* stream);

/* The same core for any window size (enhanced_generator.py:7: the constructor default is window_size=8; every caller passes 4,
 * which the entry points above serve): qkv NHWC (N,H,W,3C) -> o NHWC (N,H,W,C), one workgroup per ws x ws window, plain fp32 loops.
 * Supported while a window's q|k|v (+ dO, dS in the backward) fits one CU's LDS (mstg_window_attn_ws_supported). */
int mstg_window_attn_ws_supported(int C, int ws);
int mstg_window_attn_ws_fwd(const float* qkv, float* o, int N, int H, int W, int C, int ws, void* stream);
int mstg_window_attn_ws_bwd(const float* qkv, const float* d_o, float* dqkv, int N, int H, int W, int C, int ws, void* stream);

/* Fully fused LocalAttention for C = 16 and 32 (enhanced_generator.py:13-47 in one kernel per direction): the qkv and proj
 * 1x1 convolutions (:28, :36) and the window attention between them; x is read once, y written once.  wqkv (3C,C) and
 * wproj (C,C) are the 1x1 conv weights exactly as stored (OIHW with 1x1 taps).  The backward returns dx and ONE flat
 * gradient vector dparams = [dWqkv (3C*C) | dWproj (C*C) | dbqkv (3C) | dbproj (C)]. */
int mstg_window_attn_fused_supported(int C);
int mstg_window_attn_fwd(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj,
                         float* y, int N, int H, int W, int C, void* stream);
/* The stage's first InstanceNorm + ReLU (enhanced_generator.py:93-94, 100-101, 122-123, 129-130) folded into LocalAttention, its
 * only consumer: x_raw is the convolution output in front of the norm, in_stats[n][c] = (mean, rstd) from mstg_norm_stats; the
 * normalised tensor never exists in memory.  The backward also returns norm_sums[n][S][2][C], S = mstg_window_attn_norm_sums_split():
 * rows adding up to the per (image, channel) sums of dz * [z > 0] and dz * [z > 0] * z for mstg_norm_bwd_apply (the norm backward's
 * reduction pass, done in this kernel's epilogue). */
/* mstg_window_attn_bwd / mstg_window_attn_norm_bwd with the four parameter gradients written -- or, accumulate != 0, added -- straight
 * into the caller's gradient tensors (dwqkv (3C,C), dbqkv (3C), dwproj (C,C), dbproj (C)) by the fixed-order slab reduce, instead of
 * into one flat vector: the caller's framework then launches no per-parameter accumulation kernels. */
int mstg_window_attn_bwd_direct(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj, const float* dy,
                                float* dx, float* dwqkv, float* dbqkv, float* dwproj, float* dbproj, int accumulate, int N, int H, int W, int C,
                                void* workspace, size_t workspace_bytes, void* stream);
int mstg_window_attn_norm_bwd_direct(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv, const float* wproj,
                                     const float* bproj, const float* dy, float* dz, float* dwqkv, float* dbqkv, float* dwproj,
                                     float* dbproj, int accumulate, float* norm_sums, int N, int H, int W, int C, void* workspace,
                                     size_t workspace_bytes, void* stream);
int mstg_window_attn_norm_sums_split(void);
int mstg_window_attn_norm_fwd(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv, const float* wproj,
                              const float* bproj, float* y, int N, int H, int W, int C, void* stream);
size_t mstg_window_attn_norm_bwd_workspace_bytes(int N, int H, int W, int C);
int mstg_window_attn_norm_bwd(const float* x_raw, const float* in_stats, const float* wqkv, const float* bqkv, const float* wproj,
                              const float* bproj, const float* dy, float* dz, float* dparams, float* norm_sums, int N, int H, int W,
                              int C, void* workspace, size_t workspace_bytes, void* stream);
size_t mstg_window_attn_bwd_workspace_bytes(int N, int H, int W, int C);
int mstg_window_attn_bwd(const float* x, const float* wqkv, const float* bqkv, const float* wproj, const float* bproj,
                         const float* dy, float* dx, float* dparams, int N, int H, int W, int C, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise / reduction helpers of the training step (enhanced_train.py:49-52,72-115,36-43).
 * ---------------------------------------------------------------------------------------------- */
/* y = act(x) ; dx = dy * act'(x)  (LeakyReLU(0.2) of the discriminator stem; Tanh backward uses y) */
int mstg_act_fwd(const float* x, float* y, size_t n, int act, void* stream);
int mstg_act_bwd(const float* x_or_y, const float* dy, float* dx, size_t n, int act, void* stream);
/* mean losses: out[0] = mean(|a-b|) (kind 0, nn.L1Loss) or mean((a-b)^2) (kind 1, nn.MSELoss);
 * b == NULL means the constant `bconst`.  Deterministic two-stage reduction; workspace >= 4 KiB + 8 B per 4096 elems */
size_t mstg_loss_workspace_bytes(size_t n);
int mstg_loss_mean_fwd(const float* a, const float* b, float bconst, size_t n, int kind, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);
/* da = gscale[0]*scale * d(mean loss)/da ; db (nullable) = -da */
int mstg_loss_mean_bwd(const float* a, const float* b, float bconst, size_t n, int kind, const float* gscale, float scale,
                       float* da, float* db, void* stream);
/* y = a + b, n floats (residual connections; 16-byte aligned pointers) */
int mstg_add(const float* a, const float* b, float* y, size_t n, void* stream);
/* out[j] = sum_i weights[j * n + i] * terms[i][0] for 1..8 scalar loss terms and 1..6 outputs (terms: HOST array of device pointers,
 * weights: host matrix, row 0 = the differentiable total), and the backward of output 0, dterms[i] = g[0] * weights[i]: the weighted
 * loss sums of enhanced_train.py:72-81, 95-131 (total + the reported components) in one launch each way */
int mstg_weighted_sum_fwd(const float* const* terms, const float* weights, int n, int nout, float* out, void* stream);
int mstg_weighted_sum_bwd(const float* g, const float* weights, int n, float* dterms, void* stream);
/* masked-image pre-training loss (pretrain.py:160-162): out[0] = mean(|a * (1 - m) - b * (1 - m)|) and da = gscale[0] * d/da;
 * a, b, m same shape (m = the 0/1 mask of MonetPhotoDataset).  Workspace as for mstg_loss_mean_fwd. */
int mstg_masked_l1_mean_fwd(const float* a, const float* b, const float* m, size_t n, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);
int mstg_masked_l1_mean_bwd(const float* a, const float* b, const float* m, size_t n, const float* gscale, float* da, void* stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) on ONE flat gradient buffer (pretrain.py:165): g *= min(1, max_norm /
 * (||g||_2 + 1e-6)); norm_out (nullable) receives ||g||_2 before clipping.  Workspace as for mstg_loss_mean_fwd. */
int mstg_clip_grad_norm(float* g, size_t n, float max_norm, float* norm_out, void* workspace, size_t workspace_bytes, void* stream);
/* per-channel sum over pixels of an NHWC tensor slice: out[c] = sum_p x[p][coff+c]  (bias gradients, pooling) */
size_t mstg_channel_sum_workspace_bytes(size_t P, int C);
int mstg_channel_sum(const float* x, size_t P, int ctot, int coff, int C, float scale, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* same for a planar NCHW tensor (the 3-channel image tensors): out[c] = scale * sum_{n,i} x[n][c][i] */
size_t mstg_plane_sum_workspace_bytes(int N, int C, size_t HW);
int mstg_plane_sum(const float* x, int N, int C, size_t HW, float scale, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);
/* nn.AdaptiveAvgPool2d(1) on NHWC (enhanced_generator.py:143,257): x (S,P,C) -> out (S,C) = mean over P; and its
 * gradient dx[s][p][c] = dy[s][c] / P */
int mstg_segment_mean_fwd(const float* x, int S, size_t P, int C, float* out, void* stream);
int mstg_segment_mean_bwd(const float* dy, int S, size_t P, int C, float* dx, void* stream);
/* torch.optim.Adam step over one flat fp32 buffer (enhanced_train.py:36-43: betas (0.5,0.999), eps 1e-8) */
int mstg_adam_step_flat(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                        float eps, int step, const unsigned char* mask /*nullable: 0 = skip element*/, void* stream);

/* MultiScaleBlock (enhanced_generator.py:52-71,79-83): weight and bias gradients of the four branch convolutions
 * (1x1, 3x3 d1, 3x3 d2, 3x3 d4; each CH -> CH/4) in ONE pass over x (N,H,W,CH) and dy (N,H,W,CH = the four branch outputs
 * concatenated).  dw1 (CH/4,CH,1,1), dw2..4 (CH/4,CH,3,3), db1..4 (CH/4), PyTorch layouts.  CH in {16, 32, 64}
 * (mstg_msblock_fused_supported); other widths use mstg_conv2d_wgrad per branch. */
int mstg_msblock_fused_supported(int CH);
/* forward of the four branches: x (N,H,W,CH) -> y (N,H,W,CH) = [1x1 | 3x3 d1 | 3x3 d2 | 3x3 d4] + biases, one staging of x */
size_t mstg_msblock_fwd_workspace_bytes(int CH);
int mstg_msblock_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                     const float* b3, const float* w4, const float* b4, float* y, int N, int H, int W, int CH, void* workspace,
                     size_t workspace_bytes, void* stream);
/* input gradient of the four branches: dy (N,H,W,CH) -> dx (N,H,W,CH), written once (no accumulation passes);
 * dres (nullable, (N,H,W,CH)) = gradient arriving over the block's residual connection (`+ x`, :84), added in the epilogue */
size_t mstg_msblock_dgrad_workspace_bytes(int CH);
int mstg_msblock_dgrad(const float* dy, const float* w1, const float* w2, const float* w3, const float* w4, const float* dres,
                       float* dx, int N, int H, int W, int CH, void* workspace, size_t workspace_bytes, void* stream);
size_t mstg_msblock_wgrad_workspace_bytes(int N, int H, int W, int CH);
int mstg_msblock_wgrad(const float* x, const float* dy, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                       float* dw4, float* db4, int accumulate /* != 0: add to what the eight buffers hold */, int N, int H, int W,
                       int CH, void* workspace, size_t workspace_bytes, void* stream);

/* torch.nn.utils.spectral_norm on a conv weight seen as an (M = Cout, K = Cin*kh*kw) matrix (enhanced_generator.py:269-271).
 * fwd: training != 0 runs the one power iteration in place on u (M) and v (K); always sigma = u.(W v), w_out = w / sigma.
 * bwd: dw = dwn / sigma - (sum(dwn * w) / sigma^2) u v^T with the u, v, sigma of that forward (the caller keeps copies:
 * the next forward moves u and v on). */
size_t mstg_spectral_norm_workspace_bytes(int M, int K); /* scratch of the multi-workgroup path (matrices >= 32768 elements) */
/* u_save / v_save (nullable): copies of the u, v this call ends with, for that call's backward */
int mstg_spectral_norm_fwd(const float* w, float* u, float* v, float* w_out, float* sigma, float* u_save, float* v_save, int M,
                           int K, float eps, int training, void* workspace, size_t workspace_bytes, void* stream);
int mstg_spectral_norm_bwd(const float* dwn, const float* w, const float* u, const float* v, const float* sigma, float* dw,
                           int accumulate /* != 0: dw += */, int M, int K, void* workspace, size_t workspace_bytes, void* stream);
/* The same for up to mstg_spectral_norm_group_max() weights at once -- every convolution of one EnhancedDiscriminator forward
 * (enhanced_generator.py:236-271: seven weights, each normalised by its own hook before its convolution runs) in three launches
 * (two in the backward) instead of thirteen.  Arrays hold `count` entries; per weight the semantics are those of the calls above.
 * u_save / v_save may be null as a whole; sigma (fwd) is `count` floats, sigma (bwd) `count` pointers to them. */
int mstg_spectral_norm_group_max(void);
size_t mstg_spectral_norm_group_workspace_bytes(int count, const int* M, const int* K);
int mstg_spectral_norm_group_fwd(int count, const float* const* w, float* const* u, float* const* v, float* const* w_out, float* sigma,
                                 float* const* u_save, float* const* v_save, const int* M, const int* K, float eps, int training,
                                 void* workspace, size_t workspace_bytes, void* stream);
int mstg_spectral_norm_group_bwd(int count, const float* const* dwn, const float* const* w, const float* const* u,
                                 const float* const* v, const float* const* sigma, float* const* dw, const int* accumulate, const int* M,
                                 const int* K, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Build-defined multi-style perceptual loss pieces.  The reference has NO implementation of them (the
 * north star's "VGG-feature Gram-matrix / perceptual style loss" is a README bullet only) -- parity unpinned.
 * The VGG-topology 3x3 convolutions go through mstg_conv2d_* (ReLU = epilogue activation MSTG_ACT_RELU).
 *   max-pool 2x2/2 on NHWC with the arg-max slot (0..3, row-major window order, first maximum wins like torch) per element;
 *   Gram: g[n] = scale * F[n]^T F[n], F[n] = NHWC features viewed (HW, C), C a multiple of 16;
 *   Gram backward: df[n] = scale * F[n] (dg[n] + dg[n]^T).
 * ---------------------------------------------------------------------------------------------- */
int mstg_maxpool2x2_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, void* stream);
int mstg_maxpool2x2_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C, void* stream);
size_t mstg_gram_workspace_bytes(int N, int HW, int C);
int mstg_gram_fwd(const float* f, float* g, int N, int HW, int C, float scale, void* workspace, size_t workspace_bytes,
                  void* stream);
int mstg_gram_bwd(const float* f, const float* dg, float* df, int N, int HW, int C, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inference-only forward in fp16 storage / fp16 MFMA / fp32 accumulation (BASELINE config #5).  Replaces
 * EnhancedGenerator.forward as the reference's inference scripts call it under torch.no_grad()
 * (direct_transform.py:62-63, batch_process_images.py:210-211, advanced_transform.py:99-100).
 * Activations between calls are NHWC fp16; statistics, softmax and every accumulation are fp32.
 * InstanceNorm is split: a convolution that feeds a norm also returns that norm's (mean, rstd) per
 * (image, channel) in `out_stats` [N][Cout][2]; the consumer of the normalised tensor takes them as
 * `in_stats` and applies (x - mean) * rstd -> ReLU while staging its input (nn.InstanceNorm2d + nn.ReLU
 * at enhanced_generator.py:54-75,93-94,100-101,107-108,122-123,129-130 never touch HBM on this path).
 * Filters are packed ONCE per weight set (inference: weights are frozen) into an opaque device blob.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mstg_f16_conv_desc {
    int32_t kind;         /* 0: nn.Conv2d; 1: nn.ConvTranspose2d(k4,s2,p1); 2: the four MultiScaleBlock branch convs (:52-71) as one */
    int32_t N, H, W, Cin; /* source (N,H,W,Cin) NHWC fp16 with Cin in {16,32,64}, or (N,Cin,H,W) NCHW fp32 with Cin <= 4 */
    int32_t Ho, Wo, Cout; /* destination (N,Ho,Wo,Cout) NHWC fp16 (Cout <= 64, multiple of 4) or (N,Cout,Ho,Wo) NCHW fp16 (Cout <= 4) */
    int32_t K, stride, pad, dil;
    int32_t src_nchw_f32; /* 1: the 3-channel fp32 image at the module boundary (stem) */
    int32_t dst_nchw;     /* 1: the 3-channel output image (head) */
    int32_t act;          /* MSTG_ACT_NONE or MSTG_ACT_TANH (dst_nchw only) */
} mstg_f16_conv_desc;
size_t mstg_f16_conv_plan_bytes(const mstg_f16_conv_desc* d); /* size of the packed-filter blob; 0 = unsupported geometry */
/* w0/b0: the layer's fp32 weight (OIHW, or IOHW for kind 1) and bias (nullable); kind 2: w0..w3 / b0..b3 = branch1..branch4 */
int mstg_f16_conv_pack(const mstg_f16_conv_desc* d, const float* w0, const float* b0, const float* w1, const float* b1,
                       const float* w2, const float* b2, const float* w3, const float* b3, void* blob, size_t blob_bytes,
                       void* stream);
size_t mstg_f16_conv_partial_bytes(const mstg_f16_conv_desc* d); /* workspace needed when out_stats != NULL */
int mstg_f16_conv_fwd(const mstg_f16_conv_desc* d, const void* blob, const void* x, const float* in_stats /*nullable*/, void* y,
                      float* out_stats /*nullable*/, void* workspace, size_t workspace_bytes, void* stream);
/* The same layer reading relu((x - mean) * rstd) + residual instead of x: a MultiScaleBlock's closing norm + ReLU + `+ x`
 * (enhanced_generator.py:84) formed while the NEXT layer (:106, :121, :128, :137) stages its input, bit for bit the values
 * mstg_f16_norm_residual would have written.  residual (nullable: then exactly mstg_f16_conv_fwd) needs in_stats and an NHWC source. */
int mstg_f16_conv_fwd_res(const mstg_f16_conv_desc* d, const void* blob, const void* x, const float* in_stats,
                          const void* residual /*nullable*/, void* y, float* out_stats /*nullable*/, void* workspace,
                          size_t workspace_bytes, void* stream);
/* y = relu((x - mean) * rstd) + residual (nullable): the fusion conv's norm and the block's `+ x` (:84); NHWC fp16 */
int mstg_f16_norm_residual(const void* x, const void* residual, const float* stats, void* y, int N, int HW, int C, void* stream);
/* whole LocalAttention module (:13-47) on NHWC fp16, C in {16,32,64}; in_stats (nullable) = normalise + ReLU on load of x;
 * params = blob from mstg_f16_attn_pack (qkv and proj 1x1 filters as fp16 MFMA fragments + fp32 biases) */
size_t mstg_f16_attn_plan_bytes(int C);
int mstg_f16_attn_pack(const float* wqkv, const float* bqkv, const float* wproj, const float* bproj, int C, void* blob,
                       size_t blob_bytes, void* stream);
int mstg_f16_attn_fwd(const void* x, const float* in_stats /*nullable*/, const void* blob, void* y, int N, int H, int W, int C,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Image pre/post-processing of the callers either side of the generator, on the device (8-bit RGB, HWC, 3 bytes per pixel).
 * Replaces PIL / torchvision / numpy work in MonetPhotoDataset (pretrain.py:32-57) and process_cyclegan
 * (batch_process_images.py:183-233).  Integer work: bit-exact against Pillow.
 *   filter 0 = BILINEAR (torchvision Resize on a PIL image), 1 = LANCZOS; ksize / coefficient tables follow Pillow's
 *   Resample.c (precompute_coeffs + normalize_coeffs_8bpc) and are computed on the HOST: kk and bounds are HOST pointers,
 *   the caller copies the tables to the device and passes the device copies to the two passes below.
 * ---------------------------------------------------------------------------------------------- */
int mstg_resample_ksize(int in_size, int out_size, int filter);
int mstg_resample_coeffs(int in_size, int out_size, int filter, int* kk /* host [out_size][ksize] */, int* bounds /* host [out_size][2] */);
/* horizontal pass over source rows [y0, y0 + rows): dst (rows, out_w, 3); vertical pass: dst (out_h, w, 3) */
int mstg_resample_h_u8(const unsigned char* src, unsigned char* dst, int src_w, int y0, int rows, int out_w, int ksize,
                       const int* kk, const int* bounds, void* stream);
int mstg_resample_v_u8(const unsigned char* src, unsigned char* dst, int w, int out_h, int ksize, const int* kk, const int* bounds,
                       void* stream);
/* dst (dh, dw): filled with `fill` (if >= 0), then the (ch, cw) window of src at (sy0, sx0) pasted at (dy0, dx0): Image.new +
 * Image.paste (batch_process_images.py:196-199) and Image.crop (:221-233) */
int mstg_paste_u8(const unsigned char* src, int sh, int sw, int sy0, int sx0, int ch, int cw, unsigned char* dst, int dh, int dw,
                  int dy0, int dx0, int fill, void* stream);
/* ToTensor + Normalize(0.5, 0.5) of the (H, W) window at (y0, x0) -> out (3, H, W) fp32, times the 8x8-grid mask (bit i*8+j of
 * `grid` set = cell kept) when use_mask (pretrain.py:44-57); image_out / mask_out (nullable): unmasked image / the mask */
int mstg_u8_to_tensor(const unsigned char* src, int sh, int sw, int y0, int x0, int H, int W, float* out, float* image_out,
                      float* mask_out, unsigned long long grid, int use_mask, void* stream);
/* (y + 1) / 2 -> clamp(0, 1) -> * 255 -> uint8: y (3, H, W) fp32 -> dst (H, W, 3) (batch_process_images.py:213-217) */
int mstg_tensor_to_u8(const float* y, int H, int W, unsigned char* dst, void* stream);
/* Per-pixel blend of the letterboxed original with the styled image (uint8 HWC both), bit-exact with numpy's float64 evaluation:
 * out = clip(orig * (1 - w) + styled * w, 0, 255).astype(uint8).  weight_map == NULL: w = strength everywhere and
 * one_minus_strength is the caller's double 1 - strength (process_local_style mode 'simple', batch_process_images.py:304-312);
 * else w = weight_map[y][x] (float64, H x W: the 'enhanced' / 'advanced' weight-map blend of :340-342 / :386-387 followed by the
 * clip + cast of :352 -- the masks themselves come from cv2 / scipy on the host and are not part of this library). */
int mstg_blend_u8(const unsigned char* orig, const unsigned char* styled, double one_minus_strength, double strength,
                  const double* weight_map /*nullable*/, unsigned char* out, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BUILD-DEFINED StructuralTransformerBlock pieces.  The reference imports the class from a file its snapshot does not contain
 * (enhanced_generator.py:4; call shape :115, :218-225) -- parity unpinned; definition in structural_transformer.py.
 * The block's Linear layers go through mstg_conv2d_* (a Linear over tokens (N, L, dim) is a 1x1 convolution on NHWC).
 * ---------------------------------------------------------------------------------------------- */
/* img (N,3,H,W) fp32 -> out (N, H/4, W/4, 4): per 4x4 cell mean R, G, B and mean |dx| + |dy| of the luminance */
int mstg_structure_map(const float* img, float* out, int N, int H, int W, void* stream);
/* y = (LayerNorm(x; gamma, beta, eps) ) * (1 + gmod[n]) + bmod[n] over tokens x (N, L, dim); gmod / bmod (N, dim) nullable (plain
 * LayerNorm); stats (N*L, 2) = (mean, rstd) for the backward.  dim multiple of 4, <= 256. */
int mstg_ln_mod_fwd(const float* x, const float* gamma, const float* beta, const float* gmod, const float* bmod, float* y,
                    float* stats, int N, int L, int dim, float eps, void* stream);
size_t mstg_ln_mod_bwd_workspace_bytes(int N, int L, int dim);
/* dx, dgamma / dbeta (dim; accumulate != 0: +=), dgmod / dbmod (N, dim; nullable with gmod) */
int mstg_ln_mod_bwd(const float* x, const float* stats, const float* gamma, const float* beta, const float* gmod, const float* dy,
                    float* dx, float* dgamma, float* dbeta, float* dgmod, float* dbmod, int accumulate, int N, int L, int dim,
                    void* workspace, size_t workspace_bytes, void* stream);
/* softmax(q k^T / sqrt(D)) v over all L tokens per image and head: qkv (N, L, 3*heads*D) with q | k | v channel blocks,
 * out (N, L, heads*D), lse (N, heads, L) = log-sum-exp of the scaled scores (kept for the backward).  D in {8, 16, 32, 64}. */
int mstg_flash_attn_fwd(const float* qkv, float* out, float* lse, int N, int L, int heads, int D, void* stream);
/* dqkv from d_out; delta_ws: scratch of N*heads*L floats */
int mstg_flash_attn_bwd(const float* qkv, const float* out, const float* lse, const float* d_out, float* dqkv, float* delta_ws, int N,
                        int L, int heads, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSTG_HIP_H */
