// Persistent, software-pipelined implicit-GEMM kernel for the 4x4 stride-2 family on the fp32 MFMA (v_mfma_f32_16x16x4_f32):
// Conv2d(k4,s2,p1) forward, ConvTranspose2d(k4,s2,p1) forward (four output-parity classes, each a 2x2 stride-1 gather) and
// their input gradients (each is the other's shape with the weight read through swapped strides) at 16 / 32 / 64 channels on
// both sides -- the down / up convolutions of every generator stage and the discriminator body (enhanced_generator.py:92, 99,
// 121, 128, 240-247), forward and dgrad.
//
// Why a second kernel: igemm_light_kernel gives a workgroup ONE tile and relies on 3-5 co-resident workgroups to hide the
// patch round trip behind somebody else's MFMAs; PMC counters put its matrix pipe at 24-46 % busy on these layers.  Here a
// workgroup is persistent and overlaps the phases itself (the design of the fp16 inference kernel, infer_f16.hip):
//     registers (prefetched patch of the NEXT tile) -> LDS | barrier | issue the loads of the tile after |
//     K-steps: LDS offset of step s+2 | operand fragments of step s+1 | 4 * RPW * NF MFMAs of step s | epilogue | barrier
// A K-step is one tap x 16 source channels: lane (n = pixel column, g) reads the 16 bytes of channels 4g..4g+3 of its pixel at
// that tap (one ds_read_b128 per tile row) and the matching filter quads; MFMA j of the step multiplies element j of both.
// The per-step tap offsets live in an LDS table (a dynamically indexed kernel argument would be a scalar load whose wait also
// drains the LDS reads in flight).  LDS pixel stride = 4 Cin + 16 bytes for the stride-2 gather and 4 Cin + 32 for the
// stride-1 classes: conflict-free for ds_read_b128's lane grouping.  The packed filter [step][fragment][lane][4] sits in LDS
// when it fits beside the patch at two workgroups per CU, else its fragments are read lane-linear from L2.
#include <stdlib.h>

#include "igemm_args.h"

namespace mstg {

struct P32True { static constexpr bool value = true; };
struct P32False { static constexpr bool value = false; };

constexpr int P32_MAX_STEPS = 64;
constexpr int P32_MAX_SEG = 4;
constexpr int P32_TW = 16;
constexpr int P32_TABLE_BYTES = P32_MAX_STEPS * 4 + P32_MAX_SEG * 16;

struct P32Plan {
    int nsteps, nseg;
    struct Seg { int s0, s1, oy, ox; } seg[P32_MAX_SEG];  // one segment per output-parity class (one in all for the stride-2 gather)
    unsigned koff[P32_MAX_STEPS];                          // byte offset of the step's tap / channel chunk from the lane's pixel base
    int8_t tky[P32_MAX_STEPS], tkx[P32_MAX_STEPS];         // filter tap of the step
    int16_t tcb[P32_MAX_STEPS];                            // first source channel of the step
    int PH, PW, pixstride, oy0, ox0, stride, up, NF, TH, npf, wlds, KW;
    unsigned m_pw, m_ntile, m_tx;
};

struct P32Args {
    const float* x;
    float* y;
    const float* wpk;   // [step][frag][lane][4]
    const float* bias;  // [16 * NF], zeros for an input gradient
    const float* in_stats;  // nullable [N][Cin][2] (mean, rstd): the source is normalised + ReLU'd while it is staged
    float* partial;         // STATS kernels: [N][workgroups][2][16 * NF] sums / sums of squares of what the workgroup wrote; only the
                            // (image, workgroup) pairs that meet are written -- and read
    const float* aux;       // STATS == 2 (input-gradient launch in front of an InstanceNorm + ReLU): the RAW tensor that norm
    const float* aux_stats; // normalised, same shape as this launch's output, and its (mean, rstd) [N][Cout][2]
    int N, H, W, Cin, Ho, Wo, Cout, Gh, Gw, tiles_x, tiles_y, dbg;
};

// ---- filter pack: PyTorch layout (through the gather's strides) -> MFMA A-fragment order ---------------------------------------
__global__ void p32_pack_kernel(const P32Plan p, const float* __restrict__ w, const float* __restrict__ bias, int w_so, int w_sr,
                                int Cout, int Cin, float* __restrict__ wpk, float* __restrict__ bpk) {
    const int total = p.nsteps * p.NF * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 3, lane = (e >> 2) & 63, sf = e >> 8, f = sf % p.NF, s = sf / p.NF;
        const int co = 16 * f + (lane & 15), ci = p.tcb[s] + 4 * (lane >> 4) + j;
        float v = 0.f;
        if (co < Cout && ci < Cin) v = w[(size_t)co * w_so + (size_t)ci * w_sr + p.tky[s] * p.KW + p.tkx[s]];
        wpk[e] = v;
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 16 * p.NF; c += gridDim.x * blockDim.x) bpk[c] = (bias && c < Cout) ? bias[c] : 0.f;
}

template <int NPF>
struct P32Regs {
    f32x4 v[NPF];
    unsigned okmask;
};

__device__ __forceinline__ int p32_tile(int it, int b, int G) { return it * G + (b & 7) * (G >> 3) + (b >> 3); }  // XCD-contiguous

template <int NPF>
__device__ __forceinline__ void p32_fetch(const P32Args& a, const P32Plan& p, int t, int TH, int tid, const unsigned (&rel)[NPF],
                                          unsigned vmask, P32Regs<NPF>& R) {
    const int ntile = a.tiles_x * a.tiles_y;
    const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
    const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
    const int sy0 = ty * TH * p.stride + p.oy0, sx0 = tx * P32_TW * p.stride + p.ox0;
    const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + p.PH <= a.H && sx0 + p.PW <= a.W;  // uniform
    const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * a.H * a.W * a.Cin * 4;
    if (interior) {
        const char* org = img + ((size_t)sy0 * a.W + sx0) * a.Cin * 4;
        R.okmask = vmask;
#pragma unroll
        for (int k = 0; k < NPF; ++k) R.v[k] = *reinterpret_cast<const f32x4*>(org + (((vmask >> k) & 1) ? rel[k] : 0u));
    } else {
        R.okmask = 0;
        const long org = ((long)sy0 * a.W + sx0) * a.Cin * 4;
        const int quads = a.Cin >> 2, sh = quads == 4 ? 2 : (quads == 8 ? 3 : 4);
#pragma unroll
        for (int k = 0; k < NPF; ++k) {  // border tile: recompute the pixel instead of keeping (r, c) in registers
            const int pix = (256 * k + tid) >> sh;
            const int r = (int)__umulhi((unsigned)pix, p.m_pw), c = pix - r * p.PW;
            const int iy = sy0 + r, ix = sx0 + c;
            const bool ok = ((vmask >> k) & 1) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            R.okmask |= (unsigned)ok << k;
            R.v[k] = *reinterpret_cast<const f32x4*>(img + (ok ? org + (long)rel[k] : 0L));
        }
    }
}

template <int CTRL>
__device__ __forceinline__ float p32_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float p32_row16_sum(float v) {
    v += p32_dpp<0xB1>(v);
    v += p32_dpp<0x4E>(v);
    v += p32_dpp<0x141>(v);
    v += p32_dpp<0x140>(v);
    return v;
}

// STATS: the epilogue also sums what it stores (InstanceNorm statistics of the output without another pass over it); a.in_stats:
// the source is the RAW tensor in front of an InstanceNorm + ReLU, normalised while the patch is committed to LDS (zero padding
// applies to the normalised tensor, as in the reference where the convolution pads what the norm produced).
// STATS == 2: the launch computes an input gradient dz whose consumer is the backward of y = ReLU(InstanceNorm(f)) (dz = dL/dy);
// the epilogue re-reads f (a.aux) at the pixels it stores and sums, per (image, channel), g = dz [f^ > 0] and g f^ -- the two
// reductions that norm backward needs (norm_partial_kernel<true> read f and dz once more for them).
template <int RPW, int NF, int NPF, bool WLDS, int STATS>
__global__ __launch_bounds__(256) void conv_p32_kernel(const P32Args a, const P32Plan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 4 * RPW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nl = lane & 15, g = lane >> 4;
    const int ntile = a.tiles_x * a.tiles_y, total_tiles = a.N * ntile, G = gridDim.x;
    unsigned char* wl = smem + P32_TABLE_BYTES;
    unsigned char* patch = wl + (WLDS ? (size_t)p.nsteps * NF * 1024 : 0);
    if (tid < P32_MAX_STEPS) {
        reinterpret_cast<unsigned*>(smem)[tid] = p.koff[tid];
    } else if (tid < P32_MAX_STEPS + P32_MAX_SEG) {
        const int c = tid - P32_MAX_STEPS;
        int* e = reinterpret_cast<int*>(smem + 4 * P32_MAX_STEPS + 16 * c);
        e[0] = p.seg[c].s0; e[1] = p.seg[c].s1; e[2] = p.seg[c].oy; e[3] = p.seg[c].ox;
    }
    if (WLDS) {
        const int nchunk = p.nsteps * NF * 64;
        for (int e = tid; e < nchunk; e += 256) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(a.wpk)[e];
    }
    const f32x4* wglob = reinterpret_cast<const f32x4*>(a.wpk) + lane;  // static address spaces: no flat loads
    const unsigned char* wlds_lane = wl + 16 * lane;
    const int nseg = p.nseg, up = p.up;
    unsigned base[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) base[r] = (unsigned)(((RPW * wv + r) * p.stride * p.PW + nl * p.stride) * p.pixstride + 16 * g);
    f32x4 b4[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) b4[f] = *reinterpret_cast<const f32x4*>(a.bias + 16 * f + 4 * g);

    // what a thread stages is the same for every tile: element k of thread tid = channel quad o of patch pixel (r, c)
    const int quads = a.Cin >> 2, o = tid & (quads - 1), sh = quads == 4 ? 2 : (quads == 8 ? 3 : 4);
    unsigned rel[NPF], vmask = 0;
    {
        const int total = p.PH * p.PW * quads;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int e = 256 * k + tid, pix = e >> sh;
            const int r = (int)__umulhi((unsigned)pix, p.m_pw), c = pix - r * p.PW;
            if (e < total) vmask |= 1u << k;
            rel[k] = (unsigned)(((r * a.W + c) * a.Cin + 4 * o) * 4);
        }
    }
    float ssum[STATS ? NF : 1][4], ssq[STATS ? NF : 1][4];
#pragma unroll
    for (int f = 0; f < (STATS ? NF : 1); ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) ssum[f][q] = ssq[f][q] = 0.f;
    const int lb = STATS ? xcd_swizzle((int)blockIdx.x, G) : (int)blockIdx.x;  // logical index of this workgroup's tile range (below)
    auto flush_stats = [&](int n_img) {  // between tiles only (the scratch aliases the patch); ends with a barrier
        float* red = reinterpret_cast<float*>(patch);  // [4 waves][2][16 * NF]
#pragma unroll
        for (int f = 0; f < (STATS ? NF : 1); ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float s1 = p32_row16_sum(ssum[f][q]), s2 = p32_row16_sum(ssq[f][q]);
                if (nl == 0) {
                    red[(wv * 2 + 0) * 16 * NF + 16 * f + 4 * g + q] = s1;
                    red[(wv * 2 + 1) * 16 * NF + 16 * f + 4 * g + q] = s2;
                }
                ssum[f][q] = ssq[f][q] = 0.f;
            }
        __syncthreads();
        if (tid < 2 * 16 * NF)
            a.partial[((size_t)n_img * G + lb) * 2 * 16 * NF + tid] = red[tid] + red[2 * 16 * NF + tid] + red[4 * 16 * NF + tid] + red[6 * 16 * NF + tid];
        __syncthreads();
    };
    int cur_n = -1;
    f32x4 amu[STATS == 2 ? NF : 1], ars[STATS == 2 ? NF : 1];  // STATS == 2: (mean, rstd) of the lane's output channels, image cur_n
    float nsc[4], nnb[4];  // in_stats: (rstd, mean) of this thread's channel quad of image cur_in
    int cur_in = -1;
    const size_t out_row = (size_t)(up ? 2 : 1) * a.Wo * a.Cout * 4;  // bytes between this wave's consecutive rows
    P32Regs<NPF> R;
    int it = 0;
    // STATS: a workgroup takes a CONTIGUOUS range of tiles, so that an image is covered by a few workgroups and the statistics
    // finalize reads a few rows per image (with the strided order every workgroup touches every image: a row per pair, a memset and a
    // finalize that cost what the statistics pass they replace costs); otherwise the strided, XCD-contiguous order
    // ... and the ranges are dealt XCD-aware: workgroups b and b + 8 share an XCD (and its L2), so the logical range index is the
    // XCD-contiguous renumbering of the block index -- the ranges one L2 serves at a time are then neighbours (shared halo rows)
    const int chunk = STATS ? (total_tiles + G - 1) / G : 0;
    const int t_end = STATS ? min(total_tiles, (lb + 1) * chunk) : total_tiles;
    auto tile_of = [&](int i) { return STATS ? lb * chunk + i : p32_tile(i, blockIdx.x, G); };
    int t = tile_of(it);
    if (t < t_end) p32_fetch<NPF>(a, p, t, TH, tid, rel, vmask, R);
    while (t < t_end) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
        const int gy0 = ty * TH, gx0 = tx * P32_TW;
        if (STATS && n != cur_n) {
            if (cur_n >= 0) flush_stats(cur_n);
            cur_n = n;
            if (STATS == 2) {
#pragma unroll
                for (int f = 0; f < (STATS == 2 ? NF : 1); ++f) {
                    const float* st = a.aux_stats + ((size_t)n * a.Cout + 16 * f + 4 * g) * 2;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(st), s1_ = *reinterpret_cast<const f32x4*>(st + 4);
                    amu[f] = f32x4{s0[0], s0[2], s1_[0], s1_[2]};
                    ars[f] = f32x4{s0[1], s0[3], s1_[1], s1_[3]};
                }
            }
        }
        if (a.in_stats && n != cur_in) {
            const float* st = a.in_stats + ((size_t)n * a.Cin + 4 * o) * 2;
#pragma unroll
            for (int c = 0; c < 4; ++c) { nsc[c] = st[2 * c + 1]; nnb[c] = st[2 * c]; }
            cur_in = n;
        }
#pragma unroll
        for (int k = 0; k < NPF; ++k)
            if (((vmask >> k) & 1) && !(a.dbg & 8)) {
                f32x4 w = R.v[k];
                if (a.in_stats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = fmaxf((w[c] - nnb[c]) * nsc[c], 0.f);  // norm_apply_kernel's arithmetic, bit for bit
                }
                if (!((R.okmask >> k) & 1)) w = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(patch + (unsigned)(((256 * k + tid) >> sh) * p.pixstride + 16 * o)) = w;
            }
        __syncthreads();
        const int tnext = tile_of(it + 1);
        if (tnext < t_end && !(a.dbg & 4)) p32_fetch<NPF>(a, p, tnext, TH, tid, rel, vmask, R);

        for (int sg = 0; sg < nseg; ++sg) {
            const int4 ci = *reinterpret_cast<const int4*>(smem + 4 * P32_MAX_STEPS + 16 * sg);
            const int s0 = __builtin_amdgcn_readfirstlane(ci.x), s1 = __builtin_amdgcn_readfirstlane(ci.y);
            const int oyc = __builtin_amdgcn_readfirstlane(ci.z), oxc = __builtin_amdgcn_readfirstlane(ci.w);
            f32x4 acc[RPW][NF];
            // STATS == 2: the raw tensor at the pixels this segment will store, in flight behind the whole K loop
            const int mul = up ? 2 : 1;
            const bool full_tile = gy0 + TH <= a.Gh && gx0 + P32_TW <= a.Gw && (a.Cout & 15) == 0;
            const size_t ybyte = ((((size_t)n * a.Ho + gy0 * mul + oyc) * a.Wo + gx0 * mul + oxc) * a.Cout) * 4;
            const unsigned out_off0 = (unsigned)(((RPW * wv * mul) * a.Wo + nl * mul) * a.Cout + 4 * g) * 4u;
            f32x4 av[STATS == 2 ? RPW : 1][STATS == 2 ? NF : 1];
            if (STATS == 2 && full_tile && !(a.dbg & 2)) {
                const char* abase = reinterpret_cast<const char*>(a.aux) + ybyte;
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int f = 0; f < NF; ++f) av[STATS == 2 ? r : 0][STATS == 2 ? f : 0] = *reinterpret_cast<const f32x4*>(abase + r * out_row + out_off0 + 64 * f);
            }
            auto load_ko = [&](int s) -> unsigned { return reinterpret_cast<const unsigned*>(smem)[s]; };
            auto load_ops = [&](unsigned ko, int s, f32x4 (&bf)[RPW], f32x4 (&af)[NF]) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) bf[r] = *reinterpret_cast<const f32x4*>(patch + base[r] + ko);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (WLDS) af[f] = *reinterpret_cast<const f32x4*>(wlds_lane + (size_t)(s * NF + f) * 1024);
                    else af[f] = wglob[(size_t)(s * NF + f) * 64];
                }
            };
            auto mma_step = [&](const f32x4 (&bf)[RPW], const f32x4 (&af)[NF], auto from_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
#pragma unroll
                        for (int r = 0; r < RPW; ++r)
                            acc[r][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[f][j], bf[r][j],
                                                                             (decltype(from_bias)::value && j == 0) ? b4[f] : acc[r][f], 0, 0, 0);
            };
            auto clip = [&](int s) { return s < s1 ? s : s1 - 1; };  // past the end: re-read the last step (loaded, never used)
            f32x4 bA[RPW], aA[NF], bB[RPW], aB[NF];
            unsigned k0 = load_ko(s0), k1 = load_ko(clip(s0 + 1));
            load_ops(k0, s0, bA, aA);
            k0 = load_ko(clip(s0 + 2));
            load_ops(k1, clip(s0 + 1), bB, aB);
            mma_step(bA, aA, P32True{});  // accumulators start from the bias (the MFMA's C operand)
            int s = s0 + 1;  // invariant: B holds the operands of step s (if s < s1), k0 the offset of step s + 1
            if (a.dbg & 1) s = s1;  // experiments (MSTG_P32_DBG): 1 one K-step only, 2 no stores, 4 no patch fetch, 8 no patch commit
            for (; s + 1 < s1; s += 2) {
                k1 = load_ko(clip(s + 2));
                load_ops(k0, s + 1, bA, aA);
                mma_step(bB, aB, P32False{});
                k0 = load_ko(clip(s + 3));
                load_ops(k1, clip(s + 2), bB, aB);
                mma_step(bA, aA, P32False{});
            }
            if (s < s1) mma_step(bB, aB, P32False{});
            // ---- epilogue of this class: lane holds output channels 16f + 4g + {0..3} of compute-grid pixel (gy, gx0 + nl) --------
            if (a.dbg & 2) continue;
            if (full_tile) {
                char* ybase = reinterpret_cast<char*>(a.y) + ybyte;
#pragma unroll
                for (int r = 0; r < RPW; ++r)
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (STATS == 1) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { const float dv = acc[r][f][q] - b4[f][q]; ssum[f][q] += dv; ssq[f][q] += dv * dv; }  // pivot = the channel's bias
                        }
                        *reinterpret_cast<f32x4*>(ybase + r * out_row + out_off0 + 64 * f) = acc[r][f];
                    }
                if (STATS == 2) {
#pragma unroll
                    for (int r = 0; r < RPW; ++r)
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const f32x4 xh = (av[STATS == 2 ? r : 0][STATS == 2 ? f : 0] - amu[STATS == 2 ? f : 0]) * ars[STATS == 2 ? f : 0];  // norm_partial_kernel<true>'s arithmetic
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float gg = xh[q] > 0.f ? acc[r][f][q] : 0.f;
                                ssum[f][q] += gg;
                                ssq[f][q] += gg * xh[q];
                            }
                        }
                }
            } else {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const int gy = gy0 + RPW * wv + r, gx = gx0 + nl;
                    if (gy < a.Gh && gx < a.Gw) {
                        const int oy = gy * mul + oyc, ox = gx * mul + oxc;
#pragma unroll
                        for (int f = 0; f < NF; ++f)
                            if (16 * f + 4 * g < a.Cout) {
                                const size_t oe = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + 16 * f + 4 * g;
                                if (STATS == 1) {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) { const float dv = acc[r][f][q] - b4[f][q]; ssum[f][q] += dv; ssq[f][q] += dv * dv; }  // pivot = the channel's bias
                                }
                                if (STATS == 2) {
                                    const f32x4 xh = (*reinterpret_cast<const f32x4*>(a.aux + oe) - amu[STATS == 2 ? f : 0]) * ars[STATS == 2 ? f : 0];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        const float gg = xh[q] > 0.f ? acc[r][f][q] : 0.f;
                                        ssum[f][q] += gg;
                                        ssq[f][q] += gg * xh[q];
                                    }
                                }
                                *reinterpret_cast<f32x4*>(a.y + oe) = acc[r][f];
                            }
                    }
                }
            }
        }
        __syncthreads();  // every wave is done with the patch
        ++it;
        t = tnext;
    }
    if (STATS && cur_n >= 0) flush_stats(cur_n);
}

// partial [N][G][2][CP] -> stats [N][C][2] = (mean, rstd); one workgroup per image, double accumulation, fixed order
// The sums are of (y - bias[c]): the epilogue pivots on the channel's bias (as norm.hip pivots on the first pixel), so that a
// bias-dominated or near-constant channel (|mean| / std of 100 or more) does not lose its variance to the cancellation in
// E[y^2] - E[y]^2 while the partials are still fp32; mean = bias + E[y - bias].
__global__ __launch_bounds__(256) void p32_norm_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int G, int CP,
                                                                int C, float count, int ntile, int chunk, const float* __restrict__ pivot) {
    __shared__ double red[256 * 2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int per_c = 256 / CP, c = tid % CP, sub = tid / CP;  // CP in {16, 32, 64} -> 16 / 8 / 4 threads share a channel
    // rows written for image n: the workgroups whose contiguous tile range [b * chunk, (b + 1) * chunk) meets the image's tiles
    const int b_lo = (int)(((long)n * ntile) / chunk), b_hi = (int)((((long)n + 1) * ntile - 1) / chunk);
    double s1 = 0.0, s2 = 0.0;
    {   // eight rows' loads in flight per trip (at batch 1 an image has hundreds of rows); the additions keep their order
        const int b_end = (b_hi < G - 1 ? b_hi : G - 1) + 1;
        int b = b_lo + sub;
        for (; b + 7 * per_c < b_end; b += 8 * per_c) {
            float u[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* pp = partial + ((size_t)n * G + b + k * per_c) * 2 * CP;
                u[k] = pp[c];
                w[k] = pp[CP + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1 += (double)u[k];
                s2 += (double)w[k];
            }
        }
        for (; b < b_end; b += per_c) {
            const float* pp = partial + ((size_t)n * G + b) * 2 * CP;
            s1 += (double)pp[c];
            s2 += (double)pp[CP + c];
        }
    }
    red[tid * 2] = s1;
    red[tid * 2 + 1] = s2;
    __syncthreads();
    if (sub == 0 && c < C) {
        for (int k = 1; k < per_c; ++k) { s1 += red[(k * CP + c) * 2]; s2 += red[(k * CP + c) * 2 + 1]; }
        const double e1 = s1 / count;
        double var = s2 / count - e1 * e1;
        if (var < 0.0) var = 0.0;
        stats[((size_t)n * C + c) * 2] = (float)((double)pivot[c] + e1);
        stats[((size_t)n * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// STATS == 2: partial [N][G][2][CP] -> sums [N][2][C] (what norm_apply_kernel<true> reads with one row per image); same row
// selection and fixed order as above
__global__ __launch_bounds__(256) void p32_bsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums, int G, int CP,
                                                                int C, int ntile, int chunk) {
    __shared__ double red[256 * 2];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int per_c = 256 / CP, c = tid % CP, sub = tid / CP;
    const int b_lo = (int)(((long)n * ntile) / chunk), b_hi = (int)((((long)n + 1) * ntile - 1) / chunk);
    double s1 = 0.0, s2 = 0.0;
    {   // eight rows' loads in flight per trip (at batch 1 an image has hundreds of rows); the additions keep their order
        const int b_end = (b_hi < G - 1 ? b_hi : G - 1) + 1;
        int b = b_lo + sub;
        for (; b + 7 * per_c < b_end; b += 8 * per_c) {
            float u[8], w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* pp = partial + ((size_t)n * G + b + k * per_c) * 2 * CP;
                u[k] = pp[c];
                w[k] = pp[CP + c];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1 += (double)u[k];
                s2 += (double)w[k];
            }
        }
        for (; b < b_end; b += per_c) {
            const float* pp = partial + ((size_t)n * G + b) * 2 * CP;
            s1 += (double)pp[c];
            s2 += (double)pp[CP + c];
        }
    }
    red[tid * 2] = s1;
    red[tid * 2 + 1] = s2;
    __syncthreads();
    if (sub == 0 && c < C) {
        for (int k = 1; k < per_c; ++k) { s1 += red[(k * CP + c) * 2]; s2 += red[(k * CP + c) * 2 + 1]; }
        sums[(size_t)n * 2 * C + c] = (float)s1;
        sums[(size_t)n * 2 * C + C + c] = (float)s2;
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// Image-source variant: stride-1 K x K convolution from an NCHW tensor with <= 4 channels (the RGB image at the stem, the RGB
// gradient at the head's input gradient) to 16 NHWC channels.  The patch is staged as [pixel][4 floats]; a K-step covers FOUR taps
// (lane group g reads the 16 bytes of tap 4s + g at its pixel) and issues three MFMAs (channel 3 is padding): a 7x7 filter costs
// 13 x 3 = 39 MFMAs per 16 pixels instead of the 49 of the one-tap-per-MFMA mapping.  Everything else is conv_p32_kernel's scheme.
// -------------------------------------------------------------------------------------------------------------------------
constexpr int P32I_MAX_STEPS = 16, P32I_TH = 16, P32I_NPX = 2;  // 22 x 22 patch pixels of a 7x7 filter = 484 <= 2 per thread

struct P32iPlan {
    int nsteps, K, pad, PH, PW, flip;
    unsigned koff[P32I_MAX_STEPS][4];
    unsigned m_pw, m_ntile, m_tx;
};

__global__ void p32i_pack_kernel(const P32iPlan p, const float* __restrict__ w, const float* __restrict__ bias, int w_so, int w_sr, int Cin,
                                 float* __restrict__ wpk, float* __restrict__ bpk) {
    const int total = p.nsteps * 64 * 4, T = p.K * p.K;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int j = e & 3, lane = (e >> 2) & 63, s = e >> 8;
        const int co = lane & 15, t = 4 * s + (lane >> 4);
        float v = 0.f;
        if (t < T && j < Cin) v = w[(size_t)co * w_so + (size_t)j * w_sr + (p.flip ? T - 1 - t : t)];
        wpk[e] = v;
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 16; c += gridDim.x * blockDim.x) bpk[c] = bias ? bias[c] : 0.f;
}

struct P32iRegs {
    float v[P32I_NPX][3];
    unsigned okmask;
};

__device__ __forceinline__ void p32i_fetch(const P32Args& a, const P32iPlan& p, int t, int tid, P32iRegs& R) {
    const int ntile = a.tiles_x * a.tiles_y;
    const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
    const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
    const int sy0 = ty * P32I_TH - p.pad, sx0 = tx * P32_TW - p.pad;
    const size_t plane = (size_t)a.H * a.W;
    const float* img = a.x + (size_t)n * a.Cin * plane;
    R.okmask = 0;
#pragma unroll
    for (int k = 0; k < P32I_NPX; ++k) {
        const int e = 256 * k + tid;
        const int r = (int)__umulhi((unsigned)e, p.m_pw), c = e - r * p.PW;
        const int iy = sy0 + r, ix = sx0 + c;
        const bool ok = e < p.PH * p.PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        R.okmask |= (unsigned)ok << k;
        const unsigned off = ok ? (unsigned)(iy * a.W + ix) : 0u;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) R.v[k][ch] = img[off + (ch < a.Cin ? ch * plane : 0)];
    }
}

__global__ __launch_bounds__(256) void conv_p32i_kernel(const P32Args a, const P32iPlan p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPW = P32I_TH / 4, TH = P32I_TH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nl = lane & 15, g = lane >> 4;
    const int ntile = a.tiles_x * a.tiles_y, total_tiles = a.N * ntile, G = gridDim.x;
    unsigned char* wl = smem + P32I_MAX_STEPS * 16;            // filter [step][lane][4]
    unsigned char* patch = wl + (size_t)p.nsteps * 1024;        // [PH][PW][4 floats]
    if (tid < P32I_MAX_STEPS * 4) reinterpret_cast<unsigned*>(smem)[tid] = p.koff[tid >> 2][tid & 3];
    for (int e = tid; e < p.nsteps * 64; e += 256) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(a.wpk)[e];
    const unsigned char* wlds_lane = wl + 16 * lane;
    unsigned base[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) base[r] = (unsigned)(((RPW * wv + r) * p.PW + nl) * 16);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * g);
    const size_t out_row = (size_t)a.Wo * 16 * 4;
    P32iRegs R;
    int it = 0;
    int t = p32_tile(it, blockIdx.x, G);
    if (t < total_tiles) p32i_fetch(a, p, t, tid, R);
    const int nsteps = p.nsteps;
    while (t < total_tiles) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
        const int gy0 = ty * TH, gx0 = tx * P32_TW;
#pragma unroll
        for (int k = 0; k < P32I_NPX; ++k) {
            const int e = 256 * k + tid;
            if (e < p.PH * p.PW) {
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if ((R.okmask >> k) & 1) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) w[ch] = ch < a.Cin ? R.v[k][ch] : 0.f;
                }
                *reinterpret_cast<f32x4*>(patch + 16 * e) = w;
            }
        }
        __syncthreads();
        const int tnext = p32_tile(it + 1, blockIdx.x, G);
        if (tnext < total_tiles) p32i_fetch(a, p, tnext, tid, R);

        f32x4 acc[RPW];
        auto load_ko = [&](int s) -> unsigned { return reinterpret_cast<const unsigned*>(smem)[4 * s + g]; };
        auto load_ops = [&](unsigned ko, int s, f32x4 (&bf)[RPW], f32x4& af) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) bf[r] = *reinterpret_cast<const f32x4*>(patch + base[r] + ko);
            af = *reinterpret_cast<const f32x4*>(wlds_lane + (size_t)s * 1024);
        };
        auto mma_step = [&](const f32x4 (&bf)[RPW], const f32x4& af, auto from_bias) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < RPW; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[r][j], (decltype(from_bias)::value && j == 0) ? b4 : acc[r], 0, 0, 0);
        };
        auto clip = [&](int s) { return s < nsteps ? s : nsteps - 1; };
        f32x4 bA[RPW], bB[RPW], aA, aB;
        unsigned k0 = load_ko(0), k1 = load_ko(clip(1));
        load_ops(k0, 0, bA, aA);
        k0 = load_ko(clip(2));
        load_ops(k1, clip(1), bB, aB);
        mma_step(bA, aA, P32True{});
        int s = 1;
        for (; s + 1 < nsteps; s += 2) {
            k1 = load_ko(clip(s + 2));
            load_ops(k0, s + 1, bA, aA);
            mma_step(bB, aB, P32False{});
            k0 = load_ko(clip(s + 3));
            load_ops(k1, clip(s + 2), bB, aB);
            mma_step(bA, aA, P32False{});
        }
        if (s < nsteps) mma_step(bB, aB, P32False{});
        if (gy0 + TH <= a.Gh && gx0 + P32_TW <= a.Gw) {
            char* ybase = reinterpret_cast<char*>(a.y) + (((size_t)n * a.Ho + gy0 + RPW * wv) * a.Wo + gx0 + nl) * 64 + 16 * g;
#pragma unroll
            for (int r = 0; r < RPW; ++r) *reinterpret_cast<f32x4*>(ybase + r * out_row) = acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int gy = gy0 + RPW * wv + r, gx = gx0 + nl;
                if (gy < a.Gh && gx < a.Gw) *reinterpret_cast<f32x4*>(a.y + (((size_t)n * a.Ho + gy) * a.Wo + gx) * 16 + 4 * g) = acc[r];
            }
        }
        __syncthreads();
        ++it;
        t = tnext;
    }
}

// -------------------------------------------------------------------------------------------------------------------------
// <= 4 output channels to an NCHW tensor (7x7 head forward with tanh, stem input gradient): the 16 rows of the MFMA tile are
// (column shift delta = 0..3) x (4 channels), igemm_light's "dpack" mapping -- packed tap (ky, j) reads patch column p + 4j + 3
// and row 4 delta + c carries the real tap kx = 4j + 3 - delta, so accumulator column p holds a partial sum of output column
// p + delta; tiles advance 13 columns and an LDS exchange adds the four shifted partials.  4x fewer MFMAs than padding 3
// channels to 16 rows; here in the persistent, pipelined form of conv_p32_kernel.
// -------------------------------------------------------------------------------------------------------------------------
struct P32dPlan {
    int nsteps, K, pad, tapsx, PH, PW, pixstride, flip, npf;
    unsigned koff[P32_MAX_STEPS];
    int8_t tky[P32_MAX_STEPS], tj[P32_MAX_STEPS];
    int16_t tcb[P32_MAX_STEPS];
    unsigned m_pw, m_ntile, m_tx;
};

__global__ void p32d_pack_kernel(const P32dPlan p, const float* __restrict__ w, const float* __restrict__ bias, int w_so, int w_sr, int Co,
                                 int Cin, float* __restrict__ wpk, float* __restrict__ bpk) {
    const int total = p.nsteps * 64 * 4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int jj = e & 3, lane = (e >> 2) & 63, s = e >> 8;
        const int m = lane & 15, delta = m >> 2, c = m & 3, ci = p.tcb[s] + 4 * (lane >> 4) + jj;
        const int kx = 4 * p.tj[s] + 3 - delta, real = p.tky[s] * p.K + kx;
        float v = 0.f;
        if (c < Co && ci < Cin && kx >= 0 && kx < p.K) v = w[(size_t)c * w_so + (size_t)ci * w_sr + (p.flip ? p.K * p.K - 1 - real : real)];
        wpk[e] = v;
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 4; c += gridDim.x * blockDim.x) bpk[c] = (bias && c < Co) ? bias[c] : 0.f;
}

template <int NPF>
__device__ __forceinline__ void p32d_fetch(const P32Args& a, const P32dPlan& p, int t, int TH, int tid, P32Regs<NPF>& R) {
    const int ntile = a.tiles_x * a.tiles_y;
    const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
    const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
    const int sy0 = ty * TH - p.pad, sx0 = tx * 13 - 3 - p.pad;
    const char* img = reinterpret_cast<const char*>(a.x) + (size_t)n * a.H * a.W * a.Cin * 4;
    const int quads = a.Cin >> 2, sh = quads == 4 ? 2 : (quads == 8 ? 3 : 4), total = p.PH * p.PW * quads;
    R.okmask = 0;
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
        const int e = 256 * k + tid, pix = e >> sh;
        const int r = (int)__umulhi((unsigned)pix, p.m_pw), c = pix - r * p.PW;
        const int iy = sy0 + r, ix = sx0 + c;
        const bool ok = e < total && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        R.okmask |= (unsigned)ok << k;
        R.v[k] = *reinterpret_cast<const f32x4*>(img + (ok ? (unsigned)(((iy * a.W + ix) * a.Cin + 4 * (e & (quads - 1))) * 4) : 0u));
    }
}

template <int RPW, int NPF>
__global__ __launch_bounds__(256) void conv_p32d_kernel(const P32Args a, const P32dPlan p, const int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TH = 4 * RPW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nl = lane & 15, g = lane >> 4;
    const int ntile = a.tiles_x * a.tiles_y, total_tiles = a.N * ntile, G = gridDim.x;
    unsigned char* wl = smem + 4 * P32_MAX_STEPS;
    unsigned char* patch = wl + (size_t)p.nsteps * 1024;
    if (tid < P32_MAX_STEPS) reinterpret_cast<unsigned*>(smem)[tid] = p.koff[tid];
    for (int e = tid; e < p.nsteps * 64; e += 256) reinterpret_cast<f32x4*>(wl)[e] = reinterpret_cast<const f32x4*>(a.wpk)[e];
    const unsigned char* wlds_lane = wl + 16 * lane;
    unsigned base[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) base[r] = (unsigned)(((RPW * wv + r) * p.PW + nl) * p.pixstride + 16 * g);
    const int quads = a.Cin >> 2, o = tid & (quads - 1), sh = quads == 4 ? 2 : (quads == 8 ? 3 : 4), total = p.PH * p.PW * quads;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias);
    const int nsteps = p.nsteps;
    P32Regs<NPF> R;
    int it = 0;
    int t = p32_tile(it, blockIdx.x, G);
    if (t < total_tiles) p32d_fetch<NPF>(a, p, t, TH, tid, R);
    while (t < total_tiles) {
        const int n = ntile == 1 ? t : (int)__umulhi((unsigned)t, p.m_ntile), tt = t - n * ntile;
        const int ty = a.tiles_x == 1 ? tt : (int)__umulhi((unsigned)tt, p.m_tx), tx = tt - ty * a.tiles_x;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int e = 256 * k + tid;
            if (e < total) *reinterpret_cast<f32x4*>(patch + (unsigned)((e >> sh) * p.pixstride + 16 * o)) = ((R.okmask >> k) & 1) ? R.v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        const int tnext = p32_tile(it + 1, blockIdx.x, G);
        if (tnext < total_tiles) p32d_fetch<NPF>(a, p, tnext, TH, tid, R);

        f32x4 acc[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto load_ko = [&](int s) -> unsigned { return reinterpret_cast<const unsigned*>(smem)[s]; };
        auto load_ops = [&](unsigned ko, int s, f32x4 (&bf)[RPW], f32x4& af) {
#pragma unroll
            for (int r = 0; r < RPW; ++r) bf[r] = *reinterpret_cast<const f32x4*>(patch + base[r] + ko);
            af = *reinterpret_cast<const f32x4*>(wlds_lane + (size_t)s * 1024);
        };
        auto mma_step = [&](const f32x4 (&bf)[RPW], const f32x4& af) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < RPW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[r][j], acc[r], 0, 0, 0);
        };
        auto clip = [&](int s) { return s < nsteps ? s : nsteps - 1; };
        f32x4 bA[RPW], bB[RPW], aA, aB;
        unsigned k0 = load_ko(0), k1 = load_ko(clip(1));
        load_ops(k0, 0, bA, aA);
        k0 = load_ko(clip(2));
        load_ops(k1, clip(1), bB, aB);
        mma_step(bA, aA);
        int s = 1;
        for (; s + 1 < nsteps; s += 2) {
            k1 = load_ko(clip(s + 2));
            load_ops(k0, s + 1, bA, aA);
            mma_step(bB, aB);
            k0 = load_ko(clip(s + 3));
            load_ops(k1, clip(s + 2), bB, aB);
            mma_step(bA, aA);
        }
        if (s < nsteps) mma_step(bB, aB);
        // ---- y[q][c] = sum_delta D[delta][q - delta]: exchange through LDS (the patch is dead), lanes (row g, column nl < 13) finish -------
        __syncthreads();
        float* comb = reinterpret_cast<float*>(patch) + wv * (256 * RPW);  // [row][delta][p][4]
#pragma unroll
        for (int r = 0; r < RPW; ++r) *reinterpret_cast<f32x4*>(&comb[((r * 4 + g) * 16 + nl) * 4]) = acc[r];
        __syncthreads();
        if (g < RPW && nl < 13) {
            f32x4 v = b4;
#pragma unroll
            for (int d = 0; d < 4; ++d) v += *reinterpret_cast<const f32x4*>(&comb[((g * 4 + d) * 16 + nl + 3 - d) * 4]);
            const int oy = ty * TH + RPW * wv + g, ox = tx * 13 + nl;
            if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < a.Cout) a.y[(((size_t)n * a.Cout + e) * a.Ho + oy) * a.Wo + ox] = apply_act(v[e], act);
            }
        }
        __syncthreads();
        ++it;
        t = tnext;
    }
}

static bool p32d_eligible(const IGemmArgs& a) {
    if (!(a.y_nchw && !a.x_nchw && a.Co <= 4 && a.y_coff == 0 && a.y_ctot == a.Co && a.x_coff == 0 && a.x_ctot == a.Cr && a.Cr == 16 &&
          a.KH == a.KW && a.KH >= 3 && a.KH <= 7 && a.stride == 1 && a.dil == 1 && !a.phase && !a.accumulate &&
          (a.act == MSTG_ACT_NONE || a.act == MSTG_ACT_TANH)))
        return false;
    const char* e = env_get(ENV_NO_DPACK);
    return !(e && e[0] == '1') && (size_t)a.H * a.W * a.Cr * 4 < ((size_t)1 << 32);
}

static void p32d_plan(const IGemmArgs& a, P32dPlan& p, int TH) {
    memset(&p, 0, sizeof(p));
    p.K = a.KH; p.pad = a.pad; p.flip = a.flip;
    p.tapsx = cdiv(p.K, 4);
    p.PH = TH + p.K - 1;
    p.PW = 15 + 4 * p.tapsx;
    p.pixstride = 4 * a.Cr + 32;
    int s = 0;
    for (int ky = 0; ky < p.K; ++ky)
        for (int j = 0; j < p.tapsx; ++j, ++s) {
            p.koff[s] = (unsigned)((ky * p.PW + 4 * j + 3) * p.pixstride);
            p.tky[s] = (int8_t)ky; p.tj[s] = (int8_t)j; p.tcb[s] = 0;
        }
    p.nsteps = s;
    p.m_pw = magic_u32((unsigned)p.PW);
    p.npf = cdiv(p.PH * p.PW * (a.Cr / 4), 256);
}

template <int RPW, int NPF>
static int p32d_launch_t(const P32Args& a, const P32dPlan& p, int act, size_t lds, long tiles, hipStream_t st) {
    auto kern = conv_p32d_kernel<RPW, NPF>;
    static int occ = 0;
    static size_t occ_lds = 0;
    if (!occ || occ_lds != lds) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int nb = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, lds) != hipSuccess || nb < 1) nb = 1;
        occ = nb > 4 ? 4 : nb;
        occ_lds = lds;
    }
    long g_ = 256L * occ;
    if (g_ > tiles) g_ = (tiles + 7) & ~7L;
    MSTG_LAUNCH(kern, dim3((unsigned)g_), dim3(256), lds, st, a, p, act);
    MSTG_CHECK_LAUNCH("conv_p32d_kernel");
    return MSTG_OK;
}

static int launch_p32d(const IGemmArgs& g, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int TH = g.Ho >= 16 ? 16 : 8;
    P32dPlan p;
    p32d_plan(g, p, TH);
    const size_t need = 256 + (size_t)p.nsteps * 1024;
    if (!workspace || workspace_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "conv_p32d: workspace too small for the packed filter");
    P32Args a{};
    a.x = g.x; a.y = g.y;
    a.bias = (const float*)workspace;
    a.wpk = (const float*)((const char*)workspace + 256);
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = g.Cr; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Co;
    a.tiles_y = cdiv(g.Ho, TH);
    a.tiles_x = cdiv(g.Wo, 13);
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    if ((unsigned long long)(tiles + 4096) * (unsigned long long)(a.tiles_x * a.tiles_y) >= (1ull << 32))
        return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32d: tensor too large for the 32-bit tile arithmetic");
    p.m_ntile = magic_u32((unsigned)(a.tiles_x * a.tiles_y));
    p.m_tx = magic_u32((unsigned)a.tiles_x);
    MSTG_PACK_LAUNCH(p32d_pack_kernel, dim3(16), dim3(256), 0, st, p, g.w, g.bias, g.w_so, g.w_sr, g.Co, g.Cr, (float*)a.wpk, (float*)a.bias);
    MSTG_CHECK_LAUNCH("p32d_pack_kernel");
    size_t patch = (size_t)p.PH * p.PW * p.pixstride;
    if (patch < (size_t)4 * 256 * (TH / 4) * sizeof(float)) patch = (size_t)4 * 256 * (TH / 4) * sizeof(float);  // the exchange tiles live there too
    const size_t lds = 4 * P32_MAX_STEPS + (size_t)p.nsteps * 1024 + patch;
    if (TH == 16) return p.npf <= 8 ? p32d_launch_t<4, 8>(a, p, g.act, lds, tiles, st) : fail_arg(MSTG_E_UNSUPPORTED, "conv_p32d: patch too large");
    return p.npf <= 6 ? p32d_launch_t<2, 6>(a, p, g.act, lds, tiles, st) : fail_arg(MSTG_E_UNSUPPORTED, "conv_p32d: patch too large");
}

// -------------------------------------------------------------------------------------------------------------------------
// One output channel (the discriminator's score head, enhanced_generator.py:253-254: Conv2d(8C, 1, 4, 1, 1) on a 16 x 16 map): a
// 16 x 16 MFMA tile would carry one useful row and a 32 x 15 x 15-pixel layer fills an eighth of the chip for 30 us.  Here a
// workgroup owns one output row, thread (ox, tap) takes the dot product over the channels of its tap (filter transposed into
// LDS), and the taps are summed in a fixed order: ~2000 multiply-adds per output, a few microseconds.
// -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_co1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int H, int W, int Cr, int Ho, int Wo, int KH, int KW, int pad,
                                                       int w_sr) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = KH * KW, tid = threadIdx.x;
    float* wl = sm;            // [T][Cr]
    float* red = sm + T * Cr;  // [256]
    for (int e = tid; e < T * Cr; e += 256) {
        const int t = e / Cr, ci = e - t * Cr;
        wl[e] = w[(size_t)ci * w_sr + t];
    }
    __syncthreads();
    const int oy = blockIdx.x, n = blockIdx.y;
    const int per = 256 / T, t = tid % T, slot = tid / T, ky = t / KW, kx = t - ky * KW;
    const float b = bias ? bias[0] : 0.f;
    for (int ox0 = 0; ox0 < Wo; ox0 += per) {
        const int ox = ox0 + slot, iy = oy - pad + ky, ix = ox - pad + kx;
        float acc = 0.f;
        if (slot < per && ox < Wo && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            const f32x4* xp = reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * Cr);
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl + t * Cr);
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int c = 0; c < (Cr >> 2); ++c) a4 += xp[c] * wp[c];
            acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        }
        red[tid] = acc;
        __syncthreads();
        if (t == 0 && slot < per && ox < Wo) {
            float sum = b;
            for (int k = 0; k < T; ++k) sum += red[tid + k];
            y[((size_t)n * Ho + oy) * Wo + ox] = sum;
        }
        __syncthreads();
    }
}

static bool co1_eligible(const IGemmArgs& a) {
    return a.Co == 1 && a.y_ctot == 1 && a.y_coff == 0 && !a.x_nchw && a.x_coff == 0 && a.x_ctot == a.Cr && (a.Cr & 3) == 0 && a.Cr >= 16 &&
           a.stride == 1 && a.dil == 1 && !a.phase && !a.flip && !a.accumulate && a.act == MSTG_ACT_NONE && a.KH * a.KW <= 16 &&
           (size_t)(a.KH * a.KW * a.Cr + 256) * sizeof(float) <= 64 * 1024 && a.Ho <= 65535 && a.N <= 65535;
}

static int launch_co1(const IGemmArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(a.KH * a.KW * a.Cr + 256) * sizeof(float);
    MSTG_LAUNCH(conv_co1_kernel, dim3(a.Ho, a.N), dim3(256), lds, st, a.x, a.w, a.bias, a.y, a.H, a.W, a.Cr, a.Ho, a.Wo, a.KH, a.KW,
                       a.pad, a.w_sr);
    MSTG_CHECK_LAUNCH("conv_co1_kernel");
    return MSTG_OK;
}

// ---- host -------------------------------------------------------------------------------------------------------------------
static int p32_plan(const IGemmArgs& a, P32Plan& p);

static bool p32i_eligible(const IGemmArgs& a) {
    return a.x_nchw && !a.y_nchw && a.Cr <= 3 && a.x_coff == 0 && a.x_ctot == a.Cr && a.Co == 16 && a.y_ctot == 16 && a.y_coff == 0 &&
           a.KH == a.KW && a.KH >= 3 && a.KH * a.KW <= 4 * P32I_MAX_STEPS && a.stride == 1 && a.dil == 1 && !a.phase && !a.accumulate &&
           a.act == MSTG_ACT_NONE && (a.KH + P32I_TH - 1) * (a.KW + P32_TW - 1) <= 256 * P32I_NPX && (size_t)a.H * a.W < ((size_t)1 << 30);
}

static void p32i_plan(const IGemmArgs& a, P32iPlan& p) {
    memset(&p, 0, sizeof(p));
    p.K = a.KH; p.pad = a.pad; p.flip = a.flip;
    p.PH = P32I_TH + p.K - 1; p.PW = P32_TW + p.K - 1;
    p.nsteps = cdiv(p.K * p.K, 4);
    for (int s = 0; s < p.nsteps; ++s)
        for (int g = 0; g < 4; ++g) {
            const int t = 4 * s + g < p.K * p.K ? 4 * s + g : 0;  // beyond the filter: any valid pixel, the packed weights are zero
            p.koff[s][g] = (unsigned)(((t / p.K) * p.PW + t % p.K) * 16);
        }
    p.m_pw = magic_u32((unsigned)p.PW);
}

static int launch_p32i(const IGemmArgs& g, void* workspace, size_t workspace_bytes, hipStream_t st) {
    P32iPlan p;
    p32i_plan(g, p);
    const size_t need = 256 + (size_t)p.nsteps * 1024;
    if (!workspace || workspace_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "conv_p32i: workspace too small for the packed filter");
    P32Args a{};
    a.x = g.x; a.y = g.y;
    a.bias = (const float*)workspace;
    a.wpk = (const float*)((const char*)workspace + 256);
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = g.Cr; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Co;
    a.Gh = g.Ho; a.Gw = g.Wo;
    a.tiles_y = cdiv(a.Gh, P32I_TH);
    a.tiles_x = cdiv(a.Gw, P32_TW);
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    if ((unsigned long long)(tiles + 4096) * (unsigned long long)(a.tiles_x * a.tiles_y) >= (1ull << 32))
        return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32i: tensor too large for the 32-bit tile arithmetic");
    p.m_ntile = magic_u32((unsigned)(a.tiles_x * a.tiles_y));
    p.m_tx = magic_u32((unsigned)a.tiles_x);
    MSTG_PACK_LAUNCH(p32i_pack_kernel, dim3(16), dim3(256), 0, st, p, g.w, g.bias, g.w_so, g.w_sr, g.Cr, (float*)a.wpk, (float*)a.bias);
    MSTG_CHECK_LAUNCH("p32i_pack_kernel");
    const size_t lds = (size_t)P32I_MAX_STEPS * 16 + (size_t)p.nsteps * 1024 + (size_t)p.PH * p.PW * 16;
    static int occ = 0;
    static size_t occ_lds = 0;
    if (!occ || occ_lds != lds) {
        int nb = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(conv_p32i_kernel), 256, lds) != hipSuccess || nb < 1) nb = 1;
        occ = nb > 4 ? 4 : nb;
        occ_lds = lds;
    }
    int occ_ = occ;
    { const char* e = env_get(ENV_P32_OCC); if (e && atoi(e) >= 1 && atoi(e) < occ_) occ_ = atoi(e); }
    long g_ = 256L * occ_;
    if (g_ > tiles) g_ = (tiles + 7) & ~7L;
    MSTG_LAUNCH(conv_p32i_kernel, dim3((unsigned)g_), dim3(256), lds, st, a, p);
    MSTG_CHECK_LAUNCH("conv_p32i_kernel");
    return MSTG_OK;
}

bool p32_eligible(const IGemmArgs& a) {
    const char* e = env_get(ENV_P32);
    if (e && e[0] == '0') return false;
    if (co1_eligible(a)) return true;
    if (p32i_eligible(a)) return true;
    if (p32d_eligible(a)) return true;
    auto ch_ok = [](int c) { return c == 16 || c == 32 || c == 64; };
    if (a.x_nchw || a.y_nchw || a.x_coff || a.y_coff || a.x_ctot != a.Cr || a.y_ctot != a.Co) return false;
    if (!ch_ok(a.Cr) || !ch_ok(a.Co)) return false;
    if (a.accumulate || a.act != MSTG_ACT_NONE) return false;
    if (a.KH == 1 && a.KW == 1) {  // 1x1: a streaming GEMM over the pixels (forward and, through swapped strides, the input gradient)
        if (a.phase || a.stride != 1 || a.pad != 0 || a.H != a.Ho || a.W != a.Wo) return false;
    } else {
        if (a.KH != 4 || a.KW != 4 || a.dil != 1 || a.flip) return false;
        if (a.phase ? !(a.Ho == 2 * a.H && a.Wo == 2 * a.W) : !(a.stride == 2 && a.pad == 1 && a.H == 2 * a.Ho && a.W == 2 * a.Wo)) return false;
    }
    P32Plan p;
    return p32_plan(a, p) == MSTG_OK;  // e.g. a 64-channel stride-2 patch does not fit the prefetch registers: igemm_light takes it
}

static int p32_plan(const IGemmArgs& a, P32Plan& p) {
    memset(&p, 0, sizeof(p));
    const int Cin = a.Cr, nchunk = Cin / 16;
    p.NF = a.Co / 16;
    const bool one = a.KH == 1;
    p.KW = a.KW;
    p.up = a.phase ? 1 : 0;
    p.stride = (p.up || one) ? 1 : 2;
    p.pixstride = 4 * Cin + (p.stride == 1 ? 32 : 16);
    const int halo_lo = one ? 0 : -1, halo_hi = one ? 0 : (p.up ? 1 : 2);  // gather rows y*s - 1 ... y*s + 2 (stride 2) / y - 1 ... y + 1 (classes)
    const int ext = halo_hi - halo_lo;
    p.PW = (P32_TW - 1) * p.stride + 1 + ext;
    p.oy0 = p.ox0 = halo_lo;
    int s = 0;
    if (one) {
        p.nseg = 1;
        p.seg[0].s0 = 0;
        for (int c = 0; c < nchunk; ++c, ++s) { p.koff[s] = (unsigned)(64 * c); p.tky[s] = p.tkx[s] = 0; p.tcb[s] = (int16_t)(16 * c); }
        p.seg[0].s1 = s;
    } else if (!p.up) {
        p.nseg = 1;
        p.seg[0].s0 = 0;
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx)
                for (int c = 0; c < nchunk; ++c, ++s) {
                    p.koff[s] = (unsigned)((ky * p.PW + kx) * p.pixstride + 64 * c);
                    p.tky[s] = (int8_t)ky; p.tkx[s] = (int8_t)kx; p.tcb[s] = (int16_t)(16 * c);
                }
        p.seg[0].s1 = s;
    } else {
        p.nseg = 4;
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            p.seg[cls].s0 = s; p.seg[cls].oy = py; p.seg[cls].ox = px;
            for (int aa = 0; aa < 2; ++aa)
                for (int bb = 0; bb < 2; ++bb) {
                    // output row 2y + py gathers source rows y - 1 (ky = 3), y (ky = 1) for py = 0 and y (ky = 2), y + 1 (ky = 0) for py = 1
                    const int dyy = py == 0 ? aa - 1 : aa, ky = py == 0 ? (aa == 0 ? 3 : 1) : (aa == 0 ? 2 : 0);
                    const int dxx = px == 0 ? bb - 1 : bb, kx = px == 0 ? (bb == 0 ? 3 : 1) : (bb == 0 ? 2 : 0);
                    for (int c = 0; c < nchunk; ++c, ++s) {
                        p.koff[s] = (unsigned)(((dyy + 1) * p.PW + dxx + 1) * p.pixstride + 64 * c);
                        p.tky[s] = (int8_t)ky; p.tkx[s] = (int8_t)kx; p.tcb[s] = (int16_t)(16 * c);
                    }
                }
            p.seg[cls].s1 = s;
        }
    }
    if (s > P32_MAX_STEPS) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32: too many K-steps");
    p.nsteps = s;
    // tile height and where the filter lives: two workgroups per CU (<= 78 KiB each) with the filter in LDS if possible
    const size_t wb = (size_t)p.nsteps * p.NF * 1024;
    auto patch_bytes_of = [&](int th) {
        const int ph = (th - 1) * p.stride + 1 + ext;
        return cdiv(ph * p.PW * (Cin / 4), 256) > 12 ? (size_t)1 << 30 : (size_t)ph * p.PW * p.pixstride;
    };
    int TH = 0, wlds = 0;
    const int cand[3] = {16, 8, 4};
    for (int pass = 0; pass < 2 && !TH; ++pass)
        for (int k = 0; k < 3 && !TH; ++k) {
            const size_t need = P32_TABLE_BYTES + patch_bytes_of(cand[k]) + (pass == 0 ? wb : 0);
            if (need <= 78 * 1024 && !(p.NF == 4 && cand[k] == 16)) { TH = cand[k]; wlds = pass == 0; }
        }
    if (!TH) { TH = 4; wlds = 0; }
    // 1x1 convolutions on 32 channels are HBM-bound and carry the statistics / backward-sums epilogues of the folded norms: at four rows
    // per wave those variants need 219-256 VGPRs (one or two workgroups per CU, 32-64 KB in flight per CU: 3.1-3.9 TB/s); at two rows
    // per wave they reach 3.8-4.7 TB/s.  (The 16-channel ones are faster at four rows: measured both ways.)
    if (a.KH == 1 && a.KW == 1 && p.NF == 2 && TH == 16) TH = 8;
    { const char* e = env_get(ENV_P32_TH); if (e && (atoi(e) == 4 || atoi(e) == 8 || atoi(e) == 16) && !(p.NF == 4 && atoi(e) == 16)) TH = atoi(e); }
    { const char* e = env_get(ENV_P32_WLDS); if (e) wlds = atoi(e) != 0; }
    if (patch_bytes_of(TH) >= ((size_t)1 << 30)) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32: patch needs more than 12 prefetch registers");
    if (P32_TABLE_BYTES + patch_bytes_of(TH) + (wlds ? wb : 0) > 156 * 1024) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32: LDS budget");
    p.TH = TH; p.wlds = wlds;
    p.PH = (TH - 1) * p.stride + 1 + ext;
    p.m_pw = magic_u32((unsigned)p.PW);
    p.npf = cdiv(p.PH * p.PW * (Cin / 4), 256);
    return MSTG_OK;
}

size_t p32_workspace_bytes(const IGemmArgs& a) {
    if (p32_eligible(a) && co1_eligible(a)) return 16;  // no packed filter
    if (p32_eligible(a) && p32i_eligible(a)) return 256 + (size_t)cdiv(a.KH * a.KW, 4) * 1024;
    if (p32_eligible(a) && p32d_eligible(a)) return 256 + (size_t)a.KH * cdiv(a.KW, 4) * 1024;
    P32Plan p;
    if (!p32_eligible(a) || p32_plan(a, p)) return 0;  // (eligible implies a plan)
    return 256 + (size_t)p.nsteps * p.NF * 1024;
}

template <int RPW, int NF, int NPF>
static int p32_launch_t(P32Args& a, const P32Plan& p, size_t lds, long tiles, hipStream_t st, float* out_stats) {
    const int mode = !out_stats ? 0 : (a.aux ? 2 : 1);
    auto kern = mode == 2 ? (p.wlds ? conv_p32_kernel<RPW, NF, NPF, true, 2> : conv_p32_kernel<RPW, NF, NPF, false, 2>)
              : mode == 1 ? (p.wlds ? conv_p32_kernel<RPW, NF, NPF, true, 1> : conv_p32_kernel<RPW, NF, NPF, false, 1>)
                          : (p.wlds ? conv_p32_kernel<RPW, NF, NPF, true, 0> : conv_p32_kernel<RPW, NF, NPF, false, 0>);
    const void* kptr = reinterpret_cast<const void*>(kern);
    // persistent workgroups: what a CU really holds of this kernel at this LDS size (registers, LDS), at most 4.  One slot per
    // (filter in LDS, statistics) variant; a race between threads recomputes the same value.
    static size_t c_lds_tab[6] = {0, 0, 0, 0, 0, 0};
    static int c_occ_tab[6] = {0, 0, 0, 0, 0, 0};
    const int slot = (p.wlds ? 1 : 0) + 2 * mode;
    if (!c_occ_tab[slot] || c_lds_tab[slot] != lds) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(kptr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int nb = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kptr, 256, lds) != hipSuccess || nb < 1) nb = 1;
        c_lds_tab[slot] = lds;
        c_occ_tab[slot] = nb > 4 ? 4 : nb;
    }
    int c_occ = c_occ_tab[slot];
    { const char* e = env_get(ENV_P32_OCC); if (e && atoi(e) >= 1 && atoi(e) < c_occ) c_occ = atoi(e); }  // experiments: leave room for the other stream
    long g_ = 256L * c_occ;
    if (g_ > tiles) g_ = (tiles + 7) & ~7L;
    MSTG_LAUNCH(kern, dim3((unsigned)g_), dim3(256), lds, st, a, p);
    MSTG_CHECK_LAUNCH("conv_p32_kernel");
    if (mode == 2) {
        const int ntile = a.tiles_x * a.tiles_y;
        MSTG_LAUNCH(p32_bsum_finalize_kernel, dim3(a.N), dim3(256), 0, st, (const float*)a.partial, out_stats, (int)g_, 16 * NF, a.Cout, ntile,
                    (int)((tiles + g_ - 1) / g_));
        MSTG_CHECK_LAUNCH("p32_bsum_finalize_kernel");
    } else if (out_stats) {
        const int ntile = a.tiles_x * a.tiles_y;
        MSTG_LAUNCH(p32_norm_finalize_kernel, dim3(a.N), dim3(256), 0, st, (const float*)a.partial, out_stats, (int)g_, 16 * NF, a.Cout,
                           (float)((size_t)a.Ho * a.Wo), ntile, (int)((tiles + g_ - 1) / g_), a.bias);
        MSTG_CHECK_LAUNCH("p32_norm_finalize_kernel");
    }
    return MSTG_OK;
}

template <int RPW, int NF>
static int p32_launch_npf(P32Args& a, const P32Plan& p, size_t lds, long tiles, hipStream_t st, float* out_stats) {
    // prefetch registers per thread: the layers of the path need 6 (16-channel stride-2 / 32-channel class patches), 8 (1x1) or 12
    if (p.npf <= 6) return p32_launch_t<RPW, NF, 6>(a, p, lds, tiles, st, out_stats);
    if (p.npf <= 8) return p32_launch_t<RPW, NF, 8>(a, p, lds, tiles, st, out_stats);
    return p32_launch_t<RPW, NF, 12>(a, p, lds, tiles, st, out_stats);
}

size_t p32_norm_workspace_bytes(const IGemmArgs& g) {  // packed filter + statistics partials [N][<= 1024 workgroups][2][Cout]
    const size_t base = p32_workspace_bytes(g);
    return base ? ((base + 255) & ~(size_t)255) + (size_t)g.N * 1024 * 2 * g.Co * sizeof(float) : 0;
}

int launch_p32(const IGemmArgs& g, void* workspace, size_t workspace_bytes, hipStream_t st) { return launch_p32_norm(g, nullptr, nullptr, workspace, workspace_bytes, st); }

// the generic (non image-source, non packed-head) persistent kernel runs this launch: what the folded variants need
bool p32_generic(const IGemmArgs& g) {
    return p32_eligible(g) && !co1_eligible(g) && !p32i_eligible(g) && !p32d_eligible(g) && !g.x_nchw && !g.y_nchw;
}
// Where the backward-sums epilogue is worth its price.  Measured per layer (profiles/r03_bench_b32_256_kernel_table.txt): the epilogue
// costs the launch 7-30 % where the output has <= 32 channels and the patch <= 8 prefetch registers (then it is cheaper than the
// two-read statistics pass it replaces: 1x1 fusion convolutions, the 16 <-> 32 channel 4x4 layers), but 45-55 % on the 64-channel /
// 12-register variants, whose 256 VGPRs it fills -- more than norm_partial_kernel<true> takes on their (small) tensors.
bool p32_bsums_pays(const IGemmArgs& g) {
    P32Plan p;
    if (!p32_generic(g) || p32_plan(g, p)) return false;
    return g.Co <= 32 && p.npf <= 8;
}

// ... and the statistics epilogue of a forward launch: 20-25 % on the 12-register variants (32 -> 64 and 64 -> 32 channel 4x4 layers:
// 88 and 120 us a launch at batch 64, against 35 and 60 us for the statistics pass over their output), a few per cent elsewhere.
bool p32_stats_pays(const IGemmArgs& g) {
    P32Plan p;
    if (!p32_eligible(g) || g.x_nchw || g.y_nchw || g.Co == 1 || p32_plan(g, p)) return false;
    return p.npf <= 8;
}

static int launch_p32_full(const IGemmArgs& g, const float* in_stats, float* out_stats, const float* aux, const float* aux_stats,
                           void* workspace, size_t workspace_bytes, hipStream_t st);

int launch_p32_norm(const IGemmArgs& g, const float* in_stats, float* out_stats, void* workspace, size_t workspace_bytes, hipStream_t st) {
    return launch_p32_full(g, in_stats, out_stats, nullptr, nullptr, workspace, workspace_bytes, st);
}

// input-gradient launch whose output dz feeds the backward of ReLU(InstanceNorm(aux)): sums [N][2][Cout] = per (image, channel)
// sum of dz [aux^ > 0] and of dz [aux^ > 0] aux^ (workspace as for the statistics-emitting forward)
int launch_p32_bsums(const IGemmArgs& g, const float* aux, const float* aux_stats, float* sums, void* workspace, size_t workspace_bytes,
                     hipStream_t st) {
    if (!aux || !aux_stats || !sums) return fail_arg(MSTG_E_BADARG, "conv_p32 bsums: null pointer");
    if (!p32_generic(g)) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32 bsums: only the layers the generic persistent kernel runs");
    return launch_p32_full(g, nullptr, sums, aux, aux_stats, workspace, workspace_bytes, st);
}

static int launch_p32_full(const IGemmArgs& g, const float* in_stats, float* out_stats, const float* aux, const float* aux_stats,
                           void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (co1_eligible(g)) {
        if (in_stats || out_stats) return fail_arg(MSTG_E_UNSUPPORTED, "conv_co1: no InstanceNorm folding");
        return launch_co1(g, st);
    }
    if (p32i_eligible(g)) {
        if (in_stats || out_stats) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32i: no InstanceNorm folding for the image-source variant");
        return launch_p32i(g, workspace, workspace_bytes, st);
    }
    if (p32d_eligible(g)) {
        if (in_stats || out_stats) return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32d: no InstanceNorm folding");
        return launch_p32d(g, workspace, workspace_bytes, st);
    }
    P32Plan p;
    if (int rc = p32_plan(g, p)) return rc;
    const size_t need = out_stats ? p32_norm_workspace_bytes(g) : 256 + (size_t)p.nsteps * p.NF * 1024;
    if (!workspace || workspace_bytes < need) return fail_arg(MSTG_E_WORKSPACE, "conv_p32: workspace too small for the packed filter");
    P32Args a;
    a.in_stats = in_stats;
    a.aux = aux;
    a.aux_stats = aux_stats;
    a.partial = out_stats ? (float*)((char*)workspace + ((256 + (size_t)p.nsteps * p.NF * 1024 + 255) & ~(size_t)255)) : nullptr;
    a.x = g.x; a.y = g.y;
    a.bias = (const float*)workspace;
    a.wpk = (const float*)((const char*)workspace + 256);
    a.N = g.N; a.H = g.H; a.W = g.W; a.Cin = g.Cr; a.Ho = g.Ho; a.Wo = g.Wo; a.Cout = g.Co;
    a.Gh = p.up ? g.H : g.Ho;
    a.Gw = p.up ? g.W : g.Wo;
    a.tiles_y = cdiv(a.Gh, p.TH);
    a.tiles_x = cdiv(a.Gw, P32_TW);
    { const char* e = env_get(ENV_P32_DBG); a.dbg = e ? atoi(e) : 0; }
    const long tiles = (long)a.N * a.tiles_x * a.tiles_y;
    if ((unsigned long long)(tiles + 4096) * (unsigned long long)(a.tiles_x * a.tiles_y) >= (1ull << 32) || (size_t)g.H * g.W * g.Cr * 4 >= (1ull << 32))
        return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32: tensor too large for the 32-bit tile / offset arithmetic");
    p.m_ntile = magic_u32((unsigned)(a.tiles_x * a.tiles_y));
    p.m_tx = magic_u32((unsigned)a.tiles_x);
    MSTG_PACK_LAUNCH(p32_pack_kernel, dim3(32), dim3(256), 0, st, p, g.w, g.bias, g.w_so, g.w_sr, g.Co, g.Cr, (float*)a.wpk, (float*)a.bias);
    MSTG_CHECK_LAUNCH("p32_pack_kernel");
    size_t lds = P32_TABLE_BYTES + (size_t)p.PH * p.PW * p.pixstride + (p.wlds ? (size_t)p.nsteps * p.NF * 1024 : 0);
    lds = (lds + 15) & ~(size_t)15;
    if (out_stats && lds < (size_t)P32_TABLE_BYTES + (p.wlds ? (size_t)p.nsteps * p.NF * 1024 : 0) + 8 * 16 * p.NF * sizeof(float)) lds += 8 * 16 * p.NF * sizeof(float);
#define MSTG_P32_CASE(R_, F_) if (p.TH == 4 * R_ && p.NF == F_) return p32_launch_npf<R_, F_>(a, p, lds, tiles, st, out_stats);
    MSTG_P32_CASE(1, 1) MSTG_P32_CASE(2, 1) MSTG_P32_CASE(4, 1)
    MSTG_P32_CASE(1, 2) MSTG_P32_CASE(2, 2) MSTG_P32_CASE(4, 2)
    MSTG_P32_CASE(1, 4) MSTG_P32_CASE(2, 4)
#undef MSTG_P32_CASE
    return fail_arg(MSTG_E_UNSUPPORTED, "conv_p32: no kernel variant");
}

const char* p32_kernel_name(const IGemmArgs& a) {
    static thread_local char name[64];
    if (co1_eligible(a)) return "conv_co1_kernel";
    if (p32i_eligible(a)) return "conv_p32i_kernel";
    if (p32d_eligible(a)) return a.Ho >= 16 ? "conv_p32d_kernel<4, 8>" : "conv_p32d_kernel<2, 6>";
    P32Plan p;
    if (p32_plan(a, p)) return "";
    snprintf(name, sizeof(name), "conv_p32_kernel<%d, %d, %d, %s, false>", p.TH / 4, p.NF, p.npf <= 6 ? 6 : (p.npf <= 8 ? 8 : 12),
             p.wlds ? "true" : "false");  // as rocprofv3 prints it; mstg_conv2d_fwd_norm with out_stats runs the <..., true> instantiation
    return name;
}

}  // namespace mstg
