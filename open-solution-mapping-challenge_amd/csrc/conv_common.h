// Shared by the convolution translation units (igemm.hip, conv1x1.hip): the kernel-side descriptor and the common epilogue.
#pragma once
#include "common.h"
#include "dma.h"

namespace msc_conv {

struct ConvK {
    const char* in; const char* wt; char* out; const char* res;
    const float* scale; const float* shift; double* stats;
    const char* sz; long sz_ld;                     // stats_kind 1: the activation the ReLU mask is taken from (0: from scale*sy + shift, or none)
    int sz_bits;                                    // 1: sz is the byte mask msc_bn_apply wrote (one byte per 16-byte channel vector, sz_ld bytes per pixel)
    const char* sy; long sy_ld; int stats_kind;     // stats_kind 1: BatchNorm-backward sums against the tensor sy;
                                                    // 2: ReLU backward (mask [sy > 0] applied to the output) + bias-gradient sums
    long in_ld, out_ld, res_ld;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, flip, relu;
    int M, Hq, Wq;
    int span_bytes;                  // > 0: a K row of Cin*ES bytes spans several consecutive input pixels of span_bytes each (the KW taps of
                                     // a compact narrow tensor merged into one tap, conv_fill); the lanes of the later pixels are bounds-checked on their own
    unsigned in_bytes, wt_bytes;     // extents for the buffer descriptors of the DMA kernel
    int ntc;                         // channel tiles (DMA kernel: 1-D grid of ntm*ntc blocks, XCD-aware order)
    const float* fin_w; const float* fin_b; float* fin_logits; float* fin_probs; int fin_skip;      // fused final 1x1 + softmax (conv3x3_c32_halo_kernel)
    int ksplit; float* kws;          // split-K: slices of the reduction, fp32 partial sums (conv_igemm_dma_kernel, mode 0)
    int mode;                        // 0 gather, 1 transposed
    int xcd_order;                   // 1: XCD-aware tile order, 0: pixel tile fastest (for A/B measurements)
    float rcp_hw, rcp_w;             // 1/(Hq*Wq), 1/Wq for the pixel decode of the DMA kernel (a launch has fewer than 2^24 pixels)
    // msc_conv_desc.in_bn (ABI v9; appended, so that the layout the other kernels read is untouched): `in` is the RAW output of a training-mode
    // BatchNorm'd conv -- the kernel finalises that layer's coefficients from bnl.slots and applies relu(scale * y + shift) to the operand tile
    // in LDS; the blocks of channel tile 0 also store the activation to bnl_out (bnl_out_ld elements per pixel; may be null)
    BnFwdFin bnl; char* bnl_out; long bnl_out_ld; unsigned bnl_out_bytes;
};

// floor(m / d) through the float reciprocal, exact for m < 2^24 (one correction step either way)
__device__ __forceinline__ unsigned udiv_rcp(unsigned m, unsigned d, float rcp) {
    unsigned q = (unsigned)((float)m * rcp);
    int r = (int)(m - q * d);
    if (r < 0) { --q; r += (int)d; }
    if (r >= (int)d) ++q;
    return q;
}

// floor(a / d) for a, d < 2^24 through v_rcp_f32 (1 ulp: the correction steps of udiv_rcp absorb it)
__device__ __forceinline__ unsigned udiv24(unsigned a, unsigned d) { return udiv_rcp(a, d, __builtin_amdgcn_rcpf((float)d)); }

// shared epilogue: lane holds NV consecutive channels cb.. of pixel rows (b*16+pl), b < FN
// PATCH (halo-tile kernels): fragment b of pixel wave wp is row wp*FN + b of a 16-pixel-wide image patch whose first pixel
// is m0, i.e. pixel m0 + (wp*FN + b)*Wo + pl -- otherwise the tile is M-linear
// The tile part (values, stores, per-lane partial sums s1/s2 -- accumulated, so a persistent kernel can carry them over its tiles)
// and the statistics part (fold over the block, one atomic per channel and block) are separate; conv_epilogue is the two in a row.
//
// KIND = what the launch reduces besides storing (wave-uniform, from the descriptor):
//   0 nothing: out = relu?(acc * scale + shift (+ res))
//   3 forward BatchNorm statistics (sum, sum of squares) of the raw accumulators, stored as they are (scale / shift / res / relu as in 0)
//   1 stats_kind 1 (a data-gradient conv that also produces the BatchNorm-backward sums of the layer whose output gradient it
//     writes): (sum dh, sum dh*y), dh = acc * [scale*y + shift > 0] (no mask without scale); scale/shift are that layer's forward
//     coefficients, used for the mask only; the stored value is the raw accumulator.  With sz (ABI v6) the mask is [sz > 0] -- the ReLU
//     of a residual join sits after the add -- and a residual is added first: the conv that accumulates the last addend of the
//     join's gradient (out = acc + res) reduces what msc_bn_bwd_reduce would read the three tensors back for
//   2 stats_kind 2 (a data-gradient conv whose output is the gradient w.r.t. a bias+ReLU layer's activation sy): the stored value is
//     dh = acc * [sy > 0] and the sum is that layer's bias gradient -- the separate ReLU-backward / bias-gradient pass over the
//     tensor (msc_relu_bias_grad) is not launched
// One body per KIND behind a uniform switch, the optional steps behind uniform branches: written as one body with the options as
// selects (the first form) the compiler flattened everything into predicated code -- 21 VALU instructions per output element whatever
// the launch needed, which bounded the HBM-bound 1x1 layers (84 MB in 25-35 us) and sat at the end of every block of every conv.
struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };      // four coefficients: a bias is a view into the flat parameter buffer, 4-byte aligned only

template <typename T, int FM, int FN, int WTP, int WP, int MODE, bool PATCH, int KIND>
__device__ __forceinline__ void conv_epilogue_body(const ConvK& p, f32x4 (&acc)[FM][FN], int m0, int wp, int cb, int pl, int py, int px,
                                                   float (&s1)[FM * 4], float (&s2)[FM * 4]) {
    constexpr int NV = FM * 4;
    constexpr int CE = 16 / (int)sizeof(T);
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;      // both (folded BatchNorm, the mask of KIND 1), shift only (a bias) or none
    float sc[NV], sh[NV];
    if (has_sc) {
#pragma unroll
        for (int j = 0; j < NV; j += 4) {
            const F4U t = *reinterpret_cast<const F4U*>(p.scale + cb + j);
            sc[j] = t.x; sc[j + 1] = t.y; sc[j + 2] = t.z; sc[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) sc[j] = 1.f;
    }
    if (has_sh) {
#pragma unroll
        for (int j = 0; j < NV; j += 4) {
            const F4U t = *reinterpret_cast<const F4U*>(p.shift + cb + j);
            sh[j] = t.x; sh[j + 1] = t.y; sh[j + 2] = t.z; sh[j + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) sh[j] = 0.f;
    }
    T* out = reinterpret_cast<T*>(p.out);
    // the tensor the epilogue reads (residual / BatchNorm-backward y) is fetched for all fragments before the first use --
    // nothing else is left to hide its latency behind -- where the wave tile is small enough to afford the registers
    constexpr bool PRE = FN * (NV / CE) <= 8;
    const T* side = (KIND == 1 || KIND == 2) ? reinterpret_cast<const T*>(p.sy) : reinterpret_cast<const T*>(p.res);
    const long side_ld = (KIND == 1 || KIND == 2) ? p.sy_ld : p.res_ld;
    long opixs[FN];
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int m = PATCH ? m0 + (wp * FN + b) * p.Wo + pl : m0 + wp * WTP + b * 16 + pl;
        const int mc = m < p.M ? m : p.M - 1;            // rows past the end: computed on a valid address, not stored (their accumulators are zero)
        long opix = mc;
        if (MODE) {
            const int n = mc / (p.Hq * p.Wq);
            const int rem = mc - n * (p.Hq * p.Wq);
            const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
            opix = ((long)n * p.Ho + 2 * qy + py) * p.Wo + 2 * qx + px;
        }
        opixs[b] = opix;
    }
    uint4 pre[PRE ? FN : 1][PRE ? NV / CE : 1];
    if (PRE && side) {
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; j += CE) pre[PRE ? b : 0][PRE ? j / CE : 0] = *reinterpret_cast<const uint4*>(side + opixs[b] * side_ld + cb + j);
    }
    // KIND 1 at a residual join reads two more tensors (the addend and the mask's activation): requested up front as well -- fetched
    // fragment by fragment inside the loop they were FN dependent round trips to HBM at the end of every block (+8-16 us per launch)
    constexpr bool JOIN = KIND == 1 && FM * FN <= 16;      // the residual-join form (sz / res) is compiled for wave tiles of up to 16 fragments (msc_conv_cfg_ok)
    constexpr bool PRE2 = JOIN && PRE;
    uint4 prez[PRE2 ? FN : 1][PRE2 ? NV / CE : 1], prer[PRE2 ? FN : 1][PRE2 ? NV / CE : 1];
    unsigned zbits[JOIN ? FN : 1][JOIN ? NV / CE : 1];      // sz_bits: the mask bytes of this lane's channel vectors (one byte where the activation is 16)
    if (JOIN && p.sz && p.sz_bits) {
        const uint8_t* zb = reinterpret_cast<const uint8_t*>(p.sz);
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; j += CE) zbits[JOIN ? b : 0][JOIN ? j / CE : 0] = zb[opixs[b] * p.sz_ld + (cb + j) / CE];
    } else if (PRE2 && p.sz) {
        const T* sz = reinterpret_cast<const T*>(p.sz);
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; j += CE) prez[PRE2 ? b : 0][PRE2 ? j / CE : 0] = *reinterpret_cast<const uint4*>(sz + opixs[b] * p.sz_ld + cb + j);
    }
    if (PRE2 && p.res) {
        const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
        for (int b = 0; b < FN; ++b)
#pragma unroll
            for (int j = 0; j < NV; j += CE) prer[PRE2 ? b : 0][PRE2 ? j / CE : 0] = *reinterpret_cast<const uint4*>(res + opixs[b] * p.res_ld + cb + j);
    }
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int m = PATCH ? m0 + (wp * FN + b) * p.Wo + pl : m0 + wp * WTP + b * 16 + pl;
        float v[NV];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a * 4 + r] = acc[a][b][r];
        if (KIND == 3) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
        }
        if (KIND == 1 || KIND == 2) {
            float yv[NV];
#pragma unroll
            for (int j = 0; j < NV; j += CE) {
                if (PRE) Vec16<T>::unpack(pre[PRE ? b : 0][PRE ? j / CE : 0], yv + j);
                else Vec16<T>::load(side + opixs[b] * side_ld + cb + j, yv + j);
            }
            if (JOIN && p.res) {                   // the accumulating writer of a gradient: out = acc + res, reduced and stored
                asm volatile("" ::: "memory");
                const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    float rv[CE];
                    if (PRE2) Vec16<T>::unpack(prer[PRE2 ? b : 0][PRE2 ? j / CE : 0], rv);
                    else Vec16<T>::load(res + opixs[b] * p.res_ld + cb + j, rv);
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[j + e] += m < p.M ? rv[e] : 0.f;      // rows past the end stay zero (they are summed, not stored)
                }
            }
            if (KIND == 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    v[j] = yv[j] > 0.f ? v[j] : 0.f;
                    s1[j] += v[j];
                }
            } else if (JOIN && p.sz && p.sz_bits) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    const unsigned mb = zbits[JOIN ? b : 0][JOIN ? j / CE : 0];
#pragma unroll
                    for (int e = 0; e < CE; ++e) {
                        const float dh = ((mb >> e) & 1u) ? v[j + e] : 0.f;
                        s1[j + e] += dh;
                        s2[j + e] = fmaf(dh, yv[j + e], s2[j + e]);
                    }
                }
            } else if (JOIN && p.sz) {
                asm volatile("" ::: "memory");
                const T* sz = reinterpret_cast<const T*>(p.sz);
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    float zv[CE];
                    if (PRE2) Vec16<T>::unpack(prez[PRE2 ? b : 0][PRE2 ? j / CE : 0], zv);
                    else Vec16<T>::load(sz + opixs[b] * p.sz_ld + cb + j, zv);
#pragma unroll
                    for (int e = 0; e < CE; ++e) {
                        const float dh = zv[e] > 0.f ? v[j + e] : 0.f;
                        s1[j + e] += dh;
                        s2[j + e] = fmaf(dh, yv[j + e], s2[j + e]);
                    }
                }
            } else if (has_sc) {
                asm volatile("" ::: "memory");           // a real branch (see above)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float dh = fmaf(yv[j], sc[j], sh[j]) > 0.f ? v[j] : 0.f;
                    s1[j] += dh;
                    s2[j] = fmaf(dh, yv[j], s2[j]);
                }
            } else {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    s1[j] += v[j];
                    s2[j] = fmaf(v[j], yv[j], s2[j]);
                }
            }
        }
        if (KIND == 0 || KIND == 3) {
            if (has_sc) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
            } else if (has_sh) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] += sh[j];
            }
            if (side) {
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    float rv[CE];
                    if (PRE) Vec16<T>::unpack(pre[PRE ? b : 0][PRE ? j / CE : 0], rv);
                    else Vec16<T>::load(side + opixs[b] * side_ld + cb + j, rv);
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[j + e] += rv[e];
                }
            }
            if (p.relu) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = fmaxf(v[j], 0.f);
            }
        }
        if (m < p.M) {
#pragma unroll
            for (int j = 0; j < NV; j += CE) Vec16<T>::store(out + opixs[b] * p.out_ld + cb + j, v + j);
        }
    }
}

template <typename T, int FM, int FN, int WTP, int WP, int MODE, bool PATCH = false>
__device__ __forceinline__ void conv_epilogue_tile(const ConvK& p, f32x4 (&acc)[FM][FN], int m0, int wp, int cb, int pl, int py, int px,
                                                   float (&s1)[FM * 4], float (&s2)[FM * 4]) {
    const int kind = !p.stats ? 0 : p.stats_kind == 1 ? 1 : p.stats_kind == 2 ? 2 : 3;
    switch (kind) {
        case 0: conv_epilogue_body<T, FM, FN, WTP, WP, MODE, PATCH, 0>(p, acc, m0, wp, cb, pl, py, px, s1, s2); break;
        case 1: conv_epilogue_body<T, FM, FN, WTP, WP, MODE, PATCH, 1>(p, acc, m0, wp, cb, pl, py, px, s1, s2); break;
        case 2: conv_epilogue_body<T, FM, FN, WTP, WP, MODE, PATCH, 2>(p, acc, m0, wp, cb, pl, py, px, s1, s2); break;
        default: conv_epilogue_body<T, FM, FN, WTP, WP, MODE, PATCH, 3>(p, acc, m0, wp, cb, pl, py, px, s1, s2); break;
    }
}

template <typename T, int FM, int WP, int WC>
__device__ __forceinline__ void conv_epilogue_stats(const ConvK& p, float (&s1)[FM * 4], float (&s2)[FM * 4], int wp, int wc, int pl, int c0, float* red) {
    constexpr int NV = FM * 4;
    const bool rlb = p.stats_kind == 2;
    {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            s1[j] = row16_sum(s1[j]);
            s2[j] = row16_sum(s2[j]);
        }
        // One slot per XCD, layout [MSC_BN_SLOTS][Cout][2] (common.h); the consumer (msc_bn_apply / msc_bn_bwd_apply) sums the
        // slots in its prologue.  The block's waves fold their sums through LDS and ONE coalesced atomic instruction per
        // 64 consecutive floats goes out: an atomic costs the L2 per touched line, not per lane (4 scattered lanes per
        // instruction cost 5 ms per train step), and ops on one line serialise, so the fewer per block the better.
        constexpr int WTC = FM * 16, TC = WTC * WC;
        raw_barrier();                               // every wave is done reading the operand ring
        if (pl == 0) {
            float* mine = red + ((wp * WC + wc) * WTC + (lane_id() >> 4) * NV) * 2;
#pragma unroll
            for (int j = 0; j < NV; ++j) *reinterpret_cast<float2*>(mine + 2 * j) = make_float2(s1[j], s2[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();
        // double accumulation: the fp32 partial of a block is exact enough, a running fp32 total over all blocks is not -- the
        // consumer forms sum(dh*y) - mean*sum(dh) and E[y^2] - mean^2, which cancel by orders of magnitude
        double* slot = p.stats + ((long)msc_xcc_id() * p.Cout + c0) * 2;
        for (int f = threadIdx.x; f < TC * 2; f += WP * WC * 64) {
            const int ch = f >> 1, k = f & 1;
            if (rlb && k) continue;                  // only the first sum exists
            const int wcs = ch / WTC, chw = ch - wcs * WTC;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < WP; ++w) a += red[((w * WC + wcs) * WTC + chw) * 2 + k];
            atomicAdd(slot + f, (double)a);
        }
    }
}

template <typename T, int FM, int FN, int WTP, int WP, int MODE, int WC = 1, bool PATCH = false>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x4 (&acc)[FM][FN], int m0, int wp, int cb, int pl, int py, int px,
                                              int mtile, int ntm, float* red = nullptr, int wc = 0, int c0 = 0) {
    float s1[FM * 4], s2[FM * 4];
#pragma unroll
    for (int j = 0; j < FM * 4; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    conv_epilogue_tile<T, FM, FN, WTP, WP, MODE, PATCH>(p, acc, m0, wp, cb, pl, py, px, s1, s2);
    if (p.stats) conv_epilogue_stats<T, FM, WP, WC>(p, s1, s2, wp, wc, pl, c0, red);
}


bool xcd_order_enabled();              // igemm.hip: MSC_XCD_ORDER != 0 (XCD-aware block order; 0 for A/B measurements)

// conv1x1.hip: the streaming kernel of the 1x1 / stride 1 layers (configuration 57 of msc_conv_igemm)
constexpr int CFG_STREAM = 57;        // conv1x1_stream_kernel (pixel tiles through LDS by DMA, weights in registers)
bool conv1x1_cfg_ok(const ConvK& k, int es);
int conv1x1_launch(const ConvK& k, int dtype, hipStream_t st);

// halo32.hip: the halo-tile kernels of the 32-channel full-resolution layers (configurations 27 / 28)
int halo32_conv_launch(const ConvK& k, int dtype, hipStream_t st);
int halo32_deconv_launch(const ConvK& k, int dtype, hipStream_t st);
int halo32_stem_launch(const ConvK& k, int dtype, hipStream_t st);       // configuration 58: the 7x7 / stride 2 stem on the prepared input
int halo32_down_launch(const ConvK& k, int dtype, hipStream_t st);       // configuration 59: Conv2d(k4, s2, p1) 32 -> 128, the data gradient of dec1's ConvTranspose2d

}  // namespace msc_conv
