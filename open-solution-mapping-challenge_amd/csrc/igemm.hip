// Implicit-GEMM convolution family for gfx950 (MI355X): forward conv, data-gradient and
// transposed conv (one kernel, two addressing modes) + weight-gradient kernel.
//
// Replaces the cuDNN/MKL-DNN calls behind the reference's nn.Conv2d / nn.ConvTranspose2d
// (src/unet_models.py:21-34,136-141,360-383 and the torchvision ResNet blocks used at :345-371).
//
// Layout: activations NHWC with an explicit per-pixel channel stride (`ld`) so a tensor can be a
// channel slice of a wider buffer (this is how the skip-concats of src/unet_models.py:395-399 are
// never materialised).  Weights [Cout][KH][KW][Cin] (k-contiguous per tap).
//
// GEMM mapping (MFMA 16x16, wave64):   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]
//   A operand = weight rows, B operand = im2col pixel rows, both read k-contiguous (16 B per lane)
//   from LDS.  The weight-row -> fragment-row assignment is permuted so that every lane ends up
//   holding FM*4 CONSECUTIVE output channels of one pixel: the epilogue (scale/shift = folded BN or
//   bias, residual add, ReLU, BN partial statistics) stores 16-byte vectors, 128 B per pixel per wave.
//   One K step = 64 bytes of K per row (32 bf16 / 16 f32) = one v_mfma_f32_16x16x32_bf16 or four
//   v_mfma_f32_16x16x4_f32 per fragment pair; fp32 mode is the exact-f32 parity path.
//
// Modes:  0 = gather   out[q] = sum_t in[q*stride + off(t)] W[t]      (conv fwd, stride-1 dgrad with
//                                                                      flip=1, convT dgrad)
//         1 = transposed, stride 2, by output parity phase (blockIdx.z):
//                      fine[q] = sum_{t: (q+pad-t) even} coarse[(q+pad-t)/2] W[t]
//                                                                     (convT fwd, stride-2 dgrad)
//
// Operands go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) through an LDS ring with
// counted s_waitcnt vmcnt(N) and one raw s_barrier per k-step; per-lane byte offsets are 32-bit, computed once per filter
// tap, the k-chunk advance is the instruction's scalar offset; out-of-image taps are out-of-range buffer offsets, which the
// hardware returns as zeros (no zero-fill code, no branches).  The LDS image is linear (DMA destination = wave base +
// lane*16), so bank conflicts are removed by permuting 16-byte k-chunks on the SOURCE side and applying the same involution
// on the fragment reads.  (The first generation staged HBM -> VGPR -> ds_write_b128 and recomputed all addressing every
// k-step: 0.23-0.25 PFLOP/s, the 64-bit address arithmetic sat in front of the MFMAs of an in-order wave; removed in round 3.
// Operands beyond the 31-bit offsets of a buffer descriptor are run as image ranges, one launch after the other.)
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.h"
#include "dma.h"
#include "msc_internal.h"
#include "conv_common.h"

namespace {

using namespace msc_conv;

// ------------------------------------------------------------------------------------------------ implicit GEMM (DMA)
// K step = KB bytes per row: 128 B (one full cache line per row) whenever Cin*sizeof(T) is a multiple of 128,
// else 64 B.  Measured LDS-DMA fill rate with the operand's 512-byte row pitch (profiles/r1_dma_fill_probe.txt):
// 64-byte row segments 9-16 TB/s, 128-byte segments 28-31 TB/s -- the 64-byte form bounded the whole kernel.
// LDS image per stage: [TP pixel rows][KB] then [TC weight rows][KB], linear (DMA destination = wave base +
// lane*16).  Bank conflicts of the ds_read_b128 fragment reads are removed by permuting the 16-byte chunks of a
// row on the SOURCE side and applying the same involution on the read:
//   KB = 64 : slot = chunk ^ S[q],  S = {0,2,3,1},  q = (row>>2)&3 (pixel rows) | (row/NV)&3 (weight rows)
//   KB = 128: slot = chunk ^ q,     q = (row>>1)&7 (pixel rows) | ((row/NV)&3)<<1 | (row>>1)&1 (weight rows)
//   KB = 256: slot = chunk ^ q,     q = row&15     (pixel rows) | ((row/NV)&3)<<2 | row&3      (weight rows)
//             (a row is exactly the 64 banks, so the key alone has to spread the 16 lanes of a read)
// both keys reduce to a function of the fragment lane (pl) only, so the four lane groups of ds_read_b128
// ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots.
template <int KB> __device__ __forceinline__ int swz_x(int row) {
    return KB == 64 ? ((0x1320 >> (4 * ((row >> 2) & 3))) & 3) : KB == 128 ? ((row >> 1) & 7) : (row & 15);
}
template <int KB, int NV> __device__ __forceinline__ int swz_w(int row) {
    return KB == 64 ? ((0x1320 >> (4 * ((row / NV) & 3))) & 3)
         : KB == 128 ? ((((row / NV) & 3) << 1) | ((row >> 1) & 1)) : ((((row / NV) & 3) << 2) | (row & 3));
}
template <int KB> __device__ __forceinline__ int swz_frag(int pl) {      // key of both fragment kinds for lane pl
    return KB == 64 ? ((0x1320 >> (4 * ((pl >> 2) & 3))) & 3) : KB == 128 ? ((pl >> 1) & 7) : pl;
}

// ABL (probes/conv_ablate.hip only; 0 in the product): bit 0 = no MFMA (fragments still read), bit 1 = no DMA after the
// prologue (compute runs on whatever the ring holds), bit 2 = no fragment reads and no MFMA, bit 3 = no barrier,
// bit 4 = scheduling barriers around every DMA piece (pins its place between the MFMAs)
// BNL (msc_conv_desc.in_bn, 1x1 / stride 1, 16-bit): the input is the raw output of a training-mode BatchNorm'd conv.  Each wave rewrites the
// pixel-tile pieces IT fetched with relu(scale * y + shift), in LDS, right after its own counted wait and before the k-step's barrier (its
// own DMA has landed by then and nobody reads the stage before the barrier: no second barrier).  The coefficients are finalised from the
// producer's statistics slots into an LDS table behind the prologue's fills (block 0 publishes them and updates the running statistics: what
// msc_bn_apply's prologue does); the blocks of channel tile 0 store what they transformed -- the activation the weight gradient reads --
// from the registers of the pass.  Measured before it was built (probes/bn_on_load_probe.hip, profiles/r4_run28_bn_on_load_probe.txt): +1.3-2.5 us
// per launch on the 256x128 / 128x256 tiles against 3.8-9 us of the msc_bn_apply launch it replaces; same bits.
constexpr int BNL_CMAX = 512;        // input channels the coefficient table holds (2 x 2 KB of LDS next to the ring)
template <typename T, int TP, int TC, int WP, int WC, int MODE, int NST, int KB, int ABL = 0, int BNL = 0>
__global__ __launch_bounds__(WP * WC * 64) void conv_igemm_dma_kernel(ConvK p) {
    constexpr int ES = sizeof(T);
    constexpr int NW = WP * WC;                  // waves per block
    constexpr int KE = KB / ES;                  // K elements per step
    constexpr int KSUB = KB / 64;                // MFMA sub-steps per K step
    constexpr int LPR = KB / 16;                 // lanes (16-byte chunks) per row
    constexpr int RPI = 64 / LPR;                // rows per DMA wave-instruction
    constexpr int WTP = TP / WP, WTC = TC / WC;
    constexpr int FM = WTC / 16, FN = WTP / 16;
    constexpr int NV = FM * 4;
    constexpr int NIX = TP / RPI, NIW = TC / RPI;                 // DMA wave-instructions per tile
    constexpr int XI = (NIX + NW - 1) / NW, WI = (NIW + NW - 1) / NW;   // ... per wave (short tiles are fetched redundantly)
    constexpr int STAGE = (TP + TC) * KB;
    constexpr int LPW = XI + WI;                 // DMA instructions per wave per stage, uniform over waves
    static_assert(NIX % NW == 0 || NIX < NW, "pixel tile / wave count");
    static_assert(NST * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE + (BNL ? 8 * BNL_CMAX : 0)];
    float* const bnl_tab = reinterpret_cast<float*>(smem + NST * STAGE);      // BNL: scale[BNL_CMAX], shift[BNL_CMAX] of the input channels, behind the ring
    static_assert(!BNL || (NST * STAGE + 8 * BNL_CMAX <= 160 * 1024 && NIX >= NW && MODE == 0 && ES == 2),
                  "BNL: LDS with the table; every pixel-tile piece has ONE fetching wave; gather mode; 16-bit");

    // (round 6) The kernel arguments the prologue needs are fetched in ONE round trip, here: left to the compiler, the 328-byte descriptor arrived in five
    // dependent s_load rounds spread over the prologue, the last of them (the tensor pointers) right in front of the first fill
    // (probes/conv_timeline.hip: 2960 clk from block entry to the first fill)
    asm volatile("" ::"s"(p.in), "s"(p.wt), "s"(p.out), "s"(p.in_ld), "s"(p.N), "s"(p.Hi), "s"(p.Wi), "s"(p.Cin), "s"(p.Ho), "s"(p.Wo), "s"(p.Cout), "s"(p.KH), "s"(p.KW),
                 "s"(p.stride), "s"(p.pad), "s"(p.flip), "s"(p.M), "s"(p.Hq), "s"(p.Wq), "s"(p.span_bytes), "s"(p.in_bytes), "s"(p.wt_bytes), "s"(p.ntc), "s"(p.ksplit),
                 "s"(p.xcd_order), "s"(p.rcp_hw), "s"(p.rcp_w));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    // XCD-aware tile order: workgroup b runs on XCD b%8 (observed dispatch order; only speed depends on it).  Each
    // XCD gets a contiguous run of tiles with the channel tile varying fastest, so the blocks that re-read one pixel
    // tile (one per channel tile) and one weight tile share that XCD's L2 close in time.
    // split-K: the grid holds ksplit copies of the tile grid, slice slowest
    // (the block-index arithmetic divides small uniform numbers: through v_rcp_f32, udiv24 -- a 32-bit integer division is a 25-instruction
    // sequence, and nine of them stood at the head of every block)
    const int nwg = p.ksplit > 1 ? (int)udiv24(gridDim.x, (unsigned)p.ksplit) : (int)gridDim.x;
    const int kslice = p.ksplit > 1 ? (int)udiv24(blockIdx.x, (unsigned)nwg) : 0, orig = (int)blockIdx.x - kslice * nwg;
    const int xcd = orig & 7, wq = nwg >> 3, wr = nwg & 7;
    const int wgid = (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (orig >> 3);
    int mtile, ctile;
    if (p.xcd_order) {
        mtile = (int)udiv24((unsigned)wgid, (unsigned)p.ntc);
        ctile = wgid - mtile * p.ntc;
    } else {
        const int ntm_ = (int)udiv24((unsigned)nwg, (unsigned)p.ntc);
        ctile = (int)udiv24((unsigned)orig, (unsigned)ntm_);
        mtile = orig - ctile * ntm_;
    }
    const int m0 = mtile * TP;
    const int c0 = ctile * TC;
    const int ph = MODE ? (int)blockIdx.z : 0;
    const int py = ph >> 1, px = ph & 1;

    int kh0 = 0, kw0 = 0, nkh = p.KH, nkw = p.KW;
    if (MODE) {
        kh0 = (py + p.pad) & 1; kw0 = (px + p.pad) & 1;
        nkh = p.KH > kh0 ? (p.KH - kh0 + 1) / 2 : 0;
        nkw = p.KW > kw0 ? (p.KW - kw0 + 1) / 2 : 0;
    }
    const int cps = p.Cin / KE;
    const int nsteps_all = nkh * nkw * cps;
    // this block's slice of the k-steps (all of them without split-K)
    const int sper = p.ksplit > 1 ? (int)udiv24((unsigned)(nsteps_all + p.ksplit - 1), (unsigned)p.ksplit) : nsteps_all;
    const int sbeg = kslice * sper;
    const int nsteps = max(0, min(nsteps_all, sbeg + sper) - sbeg);

    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const u32x4_t rw = make_srd(p.wt, p.wt_bytes);

    // ---- DMA lanes: instruction j (of this wave: j = i*NW + wid) covers tile rows j*RPI .. +RPI-1
    const int lr = lane / LPR, slot = lane % LPR;
    const unsigned pix_bytes = (unsigned)p.in_ld * ES;
    const unsigned tap_bytes = (unsigned)p.Cin * ES;
    // 1x1 / stride 1 / pad 0 (half of the network's launches): output pixel m reads input pixel m -- no decode, no bounds but m < M,
    // one tap; everything else decodes (image, row, column) through float reciprocals (a launch covers fewer than 2^24 pixels:
    // conv_image_chunk).  The prologue stands in front of the first fill: the first form (both division flavours, a division per
    // tap, the decode also for 1x1) was 1200 instructions deep before the first DMA of a 13 us kernel.
    const bool lin = !MODE && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.span_bytes;
    int xn[XI], xby[XI], xbx[XI];
    unsigned xkc[XI];
    bool xv[XI];
    unsigned xoff[XI], woff[WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int j = NIX >= NW ? i * NW + wid : wid % NIX;
        const int row = j * RPI + lr;
        const int m = m0 + row;
        xv[i] = m < p.M;
        xkc[i] = (unsigned)(slot ^ swz_x<KB>(row)) * 16u;
        xn[i] = 0; xby[i] = 0; xbx[i] = 0;
        if (lin) {
            xoff[i] = xv[i] ? (unsigned)m * pix_bytes + xkc[i] : OOB_OFF;
        } else {
            const int mm = xv[i] ? m : 0;
            const int n = (int)udiv_rcp((unsigned)mm, (unsigned)(p.Hq * p.Wq), p.rcp_hw);
            const int qy = (int)udiv_rcp((unsigned)(mm - n * (p.Hq * p.Wq)), (unsigned)p.Wq, p.rcp_w);
            const int qx = mm - n * (p.Hq * p.Wq) - qy * p.Wq;
            xn[i] = n * p.Hi;
            xby[i] = MODE ? qy : qy * p.stride;
            xbx[i] = MODE ? qx : qx * p.stride;
            if (p.span_bytes) xbx[i] += (int)(xkc[i] / (unsigned)p.span_bytes);      // merged taps (one k-step per tap row): this lane's own pixel
        }
    }
    unsigned wrow[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int j = NIW >= NW ? i * NW + wid : wid % NIW;
        const int row = j * RPI + lr;
        const int co = c0 + row;
        wrow[i] = co < p.Cout ? (unsigned)co * (unsigned)(p.KH * p.KW) * tap_bytes + (unsigned)(slot ^ swz_w<KB, NV>(row)) * 16u : OOB_OFF;
    }

    if (lin) {
#pragma unroll
        for (int i = 0; i < WI; ++i) woff[i] = wrow[i];
    }
    // ---- issue iterator: (tap, k-chunk) of the next stage to fetch; per-lane offsets refreshed once per tap.  (khi, kwi) of the next
    // set_tap are carried along (taps come in order, from the slice's first one): no division per tap
    int itap = sbeg ? (int)udiv24((unsigned)sbeg, (unsigned)cps) : 0, icch = sbeg - itap * cps, istage = 0;
    int tkh = itap ? (int)udiv24((unsigned)itap, (unsigned)nkw) : 0, tkw = itap - tkh * nkw;
    auto set_tap = [&](int tap) {
        if (lin) return;
        const int khi = tkh, kwi = tkw;
        if (++tkw == nkw) { tkw = 0; ++tkh; }
        const int kh = MODE ? kh0 + 2 * khi : khi;
        const int kw = MODE ? kw0 + 2 * kwi : kwi;
        int dy, dx;
        if (MODE) { dy = (py + p.pad - kh) / 2; dx = (px + p.pad - kw) / 2; }
        else if (p.flip) { dy = p.pad - kh; dx = p.pad - kw; }
        else { dy = kh - p.pad; dx = kw - p.pad; }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int iy = xby[i] + dy, ix = xbx[i] + dx;
            const bool ok = xv[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            // merged taps: the row starts at the first pixel; xkc already is the lane's byte offset within the whole row
            const int ix0 = p.span_bytes ? ix - (int)(xkc[i] / (unsigned)p.span_bytes) : ix;
            xoff[i] = ok ? (unsigned)((xn[i] + iy) * p.Wi + ix0) * pix_bytes + xkc[i] : OOB_OFF;
        }
        const unsigned toff = (unsigned)(kh * p.KW + kw) * tap_bytes;
#pragma unroll
        for (int i = 0; i < WI; ++i) woff[i] = wrow[i] == OOB_OFF ? OOB_OFF : wrow[i] + toff;
    };
    // One stage = LPW DMA wave-instructions per wave.  The prologue issues whole stages; in the main loop the LPW pieces
    // of the stage being fetched are spread between the MFMAs of the k-step (piece i right before MFMA i*NM/LPW): an
    // in-order wave that issues all its DMA instructions at once sits in the memory pipeline's queue until the CU's
    // texture addresser (64 B/clk, shared by all waves that just passed the same barrier) has taken them, and only then
    // starts its MFMAs -- measured with probes/conv_ablate.hip (profiles/r2_run4_conv_ablation_probe.txt):
    // time(full) = time(DMA only) + time(MFMA only), no overlap.  Interleaved, the addresser works while the matrix pipes do.
    auto piece = [&](int i, char* sx, char* sw, int soff) {
        if (i < XI) dma16(rx, sx + (NIX >= NW ? i * NW + wid : wid % NIX) * 1024, xoff[i < XI ? i : 0], soff);
        else dma16(rw, sw + (NIW >= NW ? (i - XI) * NW + wid : wid % NIW) * 1024, woff[i >= XI ? i - XI : 0], soff);
    };
    auto advance = [&]() {
        if (++icch == cps) { icch = 0; ++itap; }
        if (++istage == NST) istage = 0;
    };
    auto issue = [&]() {
        if (icch == 0) set_tap(itap);
        const int soff = icch * KB;
        char* sx = smem + istage * STAGE;
        char* sw = sx + TP * KB;
#pragma unroll
        for (int i = 0; i < LPW; ++i) piece(i, sx, sw, soff);
        advance();
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets within a stage for MFMA sub-step 0 (sub-step kk: chunk index + 4*kk before the swizzle)
    const int key = swz_frag<KB>(pl);
    int aoff[FM], boff[FN];
#pragma unroll
    for (int a = 0; a < FM; ++a) aoff[a] = TP * KB + (wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3)) * KB;
#pragma unroll
    for (int b = 0; b < FN; ++b) boff[b] = (wp * WTP + b * 16 + pl) * KB;

    constexpr int NM = KSUB * FM * FN;           // MFMAs (fragment pairs) per k-step and wave
    // one k-step on the landed stage `cstage`; ISSUE: also fetch the stage NST-1 steps ahead, piecewise
    auto kstep = [&](auto issue_tag, int cstage) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        const bool live = (ABL & 2) ? p.N < 0 : true;       // ABL bit 1: never true, but not provably so (the code path stays)
        int soff = 0;
        char* sx = smem;
        char* sw = smem;
        if (ISSUE) {
            if (icch == 0) set_tap(itap);
            soff = icch * KB;
            sx = smem + istage * STAGE;
            sw = sx + TP * KB;
        }
        const char* sb = smem + cstage * STAGE;
        if (!(ABL & 4)) {
#pragma unroll
            for (int kk = 0; kk < KSUB; ++kk) {
                const int so = ((kk * 4 + g) ^ key) * 16;
                uint4 af[FM], bf[FN];
#pragma unroll
                for (int a = 0; a < FM; ++a) af[a] = *reinterpret_cast<const uint4*>(sb + aoff[a] + so);
#pragma unroll
                for (int b = 0; b < FN; ++b) bf[b] = *reinterpret_cast<const uint4*>(sb + boff[b] + so);
                if (ABL & 1) {               // keep the reads alive without the matrix pipe
#pragma unroll
                    for (int a = 0; a < FM; ++a) asm volatile("" ::"v"(af[a].x), "v"(af[a].y), "v"(af[a].z), "v"(af[a].w));
#pragma unroll
                    for (int b = 0; b < FN; ++b) asm volatile("" ::"v"(bf[b].x), "v"(bf[b].y), "v"(bf[b].z), "v"(bf[b].w));
                }
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        const int m = (kk * FM + a) * FN + b;
                        if (ISSUE) {
#pragma unroll
                            for (int i = 0; i < LPW; ++i)
                                if ((i * NM) / LPW == m && live) {
                                    if (ABL & 16) __builtin_amdgcn_sched_barrier(0);       // probe: pin the placement
                                    piece(i, sx, sw, soff);
                                    if (ABL & 16) __builtin_amdgcn_sched_barrier(0);
                                }
                        }
                        if (!(ABL & 1)) Mma<T>::run(af[a], bf[b], acc[a][b]);
                    }
            }
        } else if (ISSUE && live) {
#pragma unroll
            for (int i = 0; i < LPW; ++i) piece(i, sx, sw, soff);
        }
        if (ISSUE) advance();
    };

    // BNL: the pieces this wave fetched of the landed stage `cst` (k-step s = input channels s * KE ..), rewritten in place; lane's 16 bytes =
    // source chunk xkc / 16 of pixel row (i * NW + wid) * RPI + lr
    auto bnl_fixup = [&](int s, int cst) __attribute__((always_inline)) {
        if constexpr (BNL != 0) {
            char* sx = smem + cst * STAGE;
            const int cb = s * KE;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                char* ptr = sx + (i * NW + wid) * 1024 + lane * 16;
                uint4 v = *reinterpret_cast<uint4*>(ptr);
                const int ch = cb + (int)(xkc[i] / ES);
                float f[8];
                Vec16<T>::unpack(v, f);
                const float4 s0 = *reinterpret_cast<const float4*>(&bnl_tab[ch]), s1 = *reinterpret_cast<const float4*>(&bnl_tab[ch + 4]);
                const float4 h0 = *reinterpret_cast<const float4*>(&bnl_tab[BNL_CMAX + ch]), h1 = *reinterpret_cast<const float4*>(&bnl_tab[BNL_CMAX + ch + 4]);
                f[0] = fmaxf(fmaf(f[0], s0.x, h0.x), 0.f); f[1] = fmaxf(fmaf(f[1], s0.y, h0.y), 0.f);
                f[2] = fmaxf(fmaf(f[2], s0.z, h0.z), 0.f); f[3] = fmaxf(fmaf(f[3], s0.w, h0.w), 0.f);
                f[4] = fmaxf(fmaf(f[4], s1.x, h1.x), 0.f); f[5] = fmaxf(fmaf(f[5], s1.y, h1.y), 0.f);
                f[6] = fmaxf(fmaf(f[6], s1.z, h1.z), 0.f); f[7] = fmaxf(fmaf(f[7], s1.w, h1.w), 0.f);
                // rows past the last pixel keep the zeros the DMA wrote: the statistics epilogue sums every row of the tile and relies on
                // their accumulators being zero (conv_epilogue_body, KIND 3)
                const uint4 t = Vec16<T>::pack(f);
                v = make_uint4(xv[i] ? t.x : v.x, xv[i] ? t.y : v.y, xv[i] ? t.z : v.z, xv[i] ? t.w : v.w);
                *reinterpret_cast<uint4*>(ptr) = v;
                if (ctile == 0 && p.bnl_out && xv[i])      // the activation itself, once per pixel tile: pixel m0 + row, channels ch .. ch + 7
                    store16(p.bnl_out + ((long)(m0 + (i * NW + wid) * RPI + lr) * p.bnl_out_ld + ch) * ES, v);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rewritten pieces are in LDS before the barrier lets the others read them
        }
    };
    if (nsteps > 0) {
        if (icch != 0) set_tap(itap);            // a slice that starts inside a tap (issue() refreshes the offsets at chunk 0 only)
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nsteps) issue();
        if constexpr (BNL != 0) {
            // the coefficient table, behind the prologue's fills (the compiler's vmcnt(0) for these loads waits for those as well); the first
            // block publishes (every block computes the same values)
            for (int c = tid; c < p.Cin; c += NW * 64) {
                float sc, sh;
                bn_fwd_coeffs(p.bnl, p.Cin, c, blockIdx.x == 0, sc, sh);
                bnl_tab[c] = sc; bnl_tab[BNL_CMAX + c] = sh;
            }
            __syncthreads();
        }
        int cstage = 0;
        const int nmain = nsteps - (NST - 1);    // k-steps that still have a stage to fetch
        int s = 0;
        for (; s < nmain; ++s) {
            // stage s must have landed; stages s+1 .. s+NST-2 may stay in flight
            wait_vmcnt<(NST - 2) * LPW>();
            bnl_fixup(s, cstage);
            if (!(ABL & 8)) raw_barrier();       // everyone's DMA of stage s is in LDS, everyone is done with stage s-1
            kstep(std::true_type{}, cstage);
            if (++cstage == NST) cstage = 0;
        }
        for (; s < nsteps; ++s) {
            if (s + NST - 2 <= nsteps - 1) wait_vmcnt<(NST - 2) * LPW>();
            else wait_vmcnt<0>();
            bnl_fixup(s, cstage);
            if (!(ABL & 8)) raw_barrier();
            kstep(std::false_type{}, cstage);
            if (++cstage == NST) cstage = 0;
        }
    }
    if (p.ksplit > 1) {
        // split-K: the fp32 partial tile goes to this slice's plane of the workspace [slice][pixel][Cout] with plain 16-byte stores (a
        // lane's NV channels are consecutive); splitk_finish_kernel adds the planes in slice order -- deterministic -- and applies the
        // epilogue.  (fp32 atomics into one plane were 2x SLOWER than the unsplit launch: 16 slices hammering the same lines.)
        if (MODE == 0) {
            // no channel bound on the store: msc_conv_cfg_ok admits a configuration only when Cout % TC == 0 (every lane's NV channels exist)
            const int cb = c0 + wc * WTC + g * NV;
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int m = m0 + wp * WTP + b * 16 + pl;
                if (m < p.M) {
                    float* dst = p.kws + ((long)kslice * p.M + m) * p.Cout + cb;
#pragma unroll
                    for (int a = 0; a < FM; ++a)
                        *reinterpret_cast<float4*>(dst + a * 4) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
                }
            }
        }
        return;
    }
    conv_epilogue<T, FM, FN, WTP, WP, MODE, WC>(p, acc, m0, wp, c0 + wc * WTC + g * NV, pl, py, px, mtile, 0,
                                                reinterpret_cast<float*>(smem), wc, c0);
}

// second pass of a split-K convolution: out = relu?(sum over slices of ws[slice] * scale + shift (+ res)), 8 channels per lane
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int slices, T* __restrict__ out, long out_ld,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, const T* __restrict__ res,
                                                            long res_ld, int relu, long M, int Cout) {
    const int cpp = Cout / 8;                         // 8-channel groups per pixel
    const long total = M * cpp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cpp;
        const int c = (int)(i - m * cpp) * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < slices; ++sl) {
            const float4* w4 = reinterpret_cast<const float4*>(ws + ((long)sl * M + m) * Cout + c);
            const float4 a = w4[0], b = w4[1];
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * (scale ? scale[c + j] : 1.f) + (shift ? shift[c + j] : 0.f);
        if (res) {
            constexpr int CE = 16 / (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < 8; j += CE) {
                float rv[CE];
                Vec16<T>::load(res + m * res_ld + c + j, rv);
#pragma unroll
                for (int e = 0; e < CE; ++e) v[j + e] += rv[e];
            }
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        constexpr int CE = 16 / (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < 8; j += CE) Vec16<T>::store(out + m * out_ld + c + j, v + j);
    }
}

// ------------------------------------------------------------------------------------------------ halo tile, any width
// 3x3 / stride 1 / pad 1 convolutions (forward and, with flip, data gradient) of any channel count that is a multiple of
// 64: the implicit GEMM above re-fetches every pixel row nine times (once per tap) into LDS, and the L2 -> LDS fill -- not
// the matrix pipes -- is what bounds it (64x128 tile: 42 FLOP per filled byte against 300+ for the MFMA peak at the
// measured 14-34 TB/s of fill).  Here a block owns a PH x 16 pixel patch of one image and TC output channels.  The
// reduction runs chunk-major: for each 64-channel chunk (128-byte rows) the (PH+2) x 18 halo of input rows is fetched ONCE
// and the nine taps read shifted windows of it, while the nine [TC][128 B] weight slices stream through a ring.  Per
// chunk and block that is 9 x TC x 128 B of weights + ~1.3 x the patch instead of 9 x (patch + TC rows): 2.3x fewer
// bytes for a 128-pixel x 64-channel tile, 2.1x for 256 x 128.
//   LDS: two halo buffers (chunk c+1 arrives during the nine k-steps of chunk c, one DMA piece per wave and k-step) +
//   NWS weight stages.  Every wave issues the same number of DMA instructions per k-step -- pieces past the end of the
//   reduction or of the halo go out with an out-of-range offset (no memory traffic) -- so the counted vmcnt of a k-step is
//   a compile-time constant per tap.
//   TPS taps per k-step (1 or 3): on the 16x16 maps a k-step of one tap is 257 MFMA cycles per SIMD between two barriers;
//   a kernel row per k-step (weight stage = 3 slices) has 12 barriers per chunk-loop pass instead of 36.
//   MINB = blocks per CU the register allocation has to allow (HIP's second launch bound counts waves per SIMD).
//   BNL (msc_conv_desc.in_bn, see conv_igemm_dma_kernel): the halo of a chunk is rewritten with relu(scale * y + shift) by the waves that
//   fetched its pieces, at the k-step whose counted wait covers them (NWS - 1 k-steps after they were issued; the last ones at k-step 0 of the
//   chunk that reads them, before its barrier).  Lanes whose pixel lies outside the image keep the zeros the DMA wrote (the padding is of the
//   ACTIVATION); the blocks of channel tile 0 store the interior of what they transformed.
template <typename T, int PH, int TC, int WP, int WC, int NWS, int TPS = 1, int MINB = 1, int BNL = 0>
__global__ __launch_bounds__(WP * WC * 64, MINB * WP * WC / 4) void conv3x3_halo_dma_kernel(ConvK p) {
    static_assert(sizeof(T) == 2, "16-bit types");
    static_assert(TPS == 1 || TPS == 3, "taps per k-step");
    constexpr int ES = 2, KB = 128;
    constexpr int NW = WP * WC;
    constexpr int SPC = 9 / TPS;                              // k-steps per chunk
    constexpr int HCOLS = 18, HPIX = (PH + 2) * HCOLS;
    constexpr int NHI = (HPIX + 7) / 8;                       // DMA wave-instructions per halo chunk (8 pixels x 128 B)
    constexpr int XH = (NHI + NW - 1) / NW;                   // ... per wave
    constexpr int HBUF = XH * NW * 1024;                      // a halo buffer (padded to whole instructions of every wave)
    constexpr int NIW = TC / 8;                               // DMA wave-instructions per weight slice (one tap)
    constexpr int WI1 = (NIW + NW - 1) / NW;                  // ... per wave
    constexpr int WI = WI1 * TPS;                             // weight pieces per wave and k-step
    constexpr int WSLICE = TC * KB, WSTAGE = TPS * WSLICE;
    // halo pieces of the next chunk: HPS per k-step, all of them out by k-step SPC - (NWS - 1) (the k-step that issues the
    // first weight stage of the next chunk -- the wait that covers that stage then covers them too)
    constexpr int HSTEPS = SPC - (NWS - 1) + 1 > 0 ? SPC - (NWS - 1) + 1 : 1;
    constexpr int HPS = (XH + HSTEPS - 1) / HSTEPS;
    constexpr int FN = PH / WP, WTC = TC / WC, FM = WTC / 16, NV = FM * 4;
    static_assert(PH % WP == 0 && TC % (WC * 16) == 0, "wave tiling");
    static_assert(NWS - 1 <= SPC, "ring deeper than a chunk");
    static_assert(NIW % NW == 0 || (NIW < NW && TPS == 1), "weight tile / wave count");
    static_assert(2 * HBUF + NWS * WSTAGE + (BNL ? 8 * BNL_CMAX : 0) <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * HBUF + NWS * WSTAGE + (BNL ? 8 * BNL_CMAX : 0)];
    float* const bnl_tab = reinterpret_cast<float*>(smem + 2 * HBUF + NWS * WSTAGE);      // BNL: scale[BNL_CMAX], shift[BNL_CMAX], behind the rings

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    // XCD-aware order (see conv_igemm_dma_kernel): consecutive tiles on one XCD, channel tile fastest
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, wq = nwg >> 3, wr = nwg & 7;
    const int wgid = p.xcd_order ? (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (orig >> 3) : orig;
    const int patch = (int)udiv24((unsigned)wgid, (unsigned)p.ntc), ctile = wgid - patch * p.ntc;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / PH;
    const int prow = (int)udiv24((unsigned)patch, (unsigned)tiles_x), bx = patch - prow * tiles_x;       // (small uniform numbers: udiv24, conv_common.h)
    const int n = (int)udiv24((unsigned)prow, (unsigned)tiles_y), by = prow - n * tiles_y;
    const int y0 = by * PH, x0 = bx * 16;
    const int c0 = ctile * TC;
    const int nchunks = (p.Cin * ES) / KB;

    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const u32x4_t rw = make_srd(p.wt, p.wt_bytes);
    const u32x4_t ro = make_srd(p.bnl_out, p.bnl_out_bytes);      // BNL: where the blocks of channel tile 0 store the activation
    // (ctile comes out of a float reciprocal, i.e. a vector register: made a scalar explicitly, or the branch on it counts as divergent and the
    // SRD operand of the store inside it is no longer accepted as wave-uniform)
    const bool bnl_wb = BNL != 0 && __builtin_amdgcn_readfirstlane((int)(ctile == 0 && p.bnl_out != nullptr)) != 0;
    const int lr = lane >> 3, slot = lane & 7;
    const unsigned pix_bytes = (unsigned)p.in_ld * ES;
    const unsigned tap_bytes = (unsigned)p.Cin * ES;

    // ---- per-lane source offsets, chunk/tap independent (the chunk and tap advance through the scalar offset)
    unsigned hoff[XH];
#pragma unroll
    for (int i = 0; i < XH; ++i) {
        const int hp = (i * NW + wid) * 8 + lr;
        const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        hoff[i] = ok ? (unsigned)((n * p.Hi + iy) * p.Wi + ix) * pix_bytes + (unsigned)(slot ^ (hx & 7)) * 16u : OOB_OFF;      // halo_key: the COLUMN, see boff
    }
    unsigned woff[WI1];
#pragma unroll
    for (int i = 0; i < WI1; ++i) {
        const int j = NIW >= NW ? i * NW + wid : wid % NIW;
        const int row = j * 8 + lr;
        const int co = c0 + row;
        woff[i] = co < p.Cout ? (unsigned)co * 9u * tap_bytes + (unsigned)(slot ^ swz_w<KB, NV>(row)) * 16u : OOB_OFF;
    }
    // ---- fragment read offsets: weights as in the implicit-GEMM kernel; pixels = halo position of (patch row, pl) shifted by the tap
    const int key = swz_frag<KB>(pl);
    int aoff[FM];
#pragma unroll
    for (int a = 0; a < FM; ++a) aoff[a] = (wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3)) * KB + ((g ^ key) * 16);
    int boff[9][FN];                      // byte offset within a halo buffer of sub-step 0 (sub-step 1: ^ 64)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t - kh * 3;
        const int dy = p.flip ? 1 - kh : kh - 1, dx = p.flip ? 1 - kw : kw - 1;
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            // round 6: the swizzle key is the halo COLUMN (hx & 7), not (pixel >> 1) & 7.  A tap's fragment is 16 consecutive halo pixels starting at an
            // arbitrary column: with the pixel-pair key two of the three column shifts put two lanes of a ds_read_b128 lane group on one 16-byte slot
            // (checked exhaustively; SQ_LDS_BANK_CONFLICT = 21 % of SQ_LDS_IDX_ACTIVE on the decoder layers, profiles/r5_final_sq_summary.txt, against 0
            // for the implicit-GEMM kernel, whose fragments start on 16-pixel boundaries).  Rows are 18 x 128 B = 9 x 256 B apart, so the row does not
            // enter the bank: hx & 7 is conflict-free for every shift.
            const int hx = pl + 1 + dx;
            const int hp = (wp * FN + b + 1 + dy) * HCOLS + hx;
            boff[t][b] = hp * KB + ((g ^ (hx & 7)) * 16);
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    char* const hbase = smem;
    char* const wbase = smem + 2 * HBUF;
    // ---- issue state: the next weight stage to fetch is (chunk iwc, first tap iwt) into ring slot iws
    int iwc = 0, iwt = 0, iws = 0;
    auto w_piece = [&](int i) __attribute__((always_inline)) {      // piece i of the stage: slice i / WI1 (a tap), rows of piece i % WI1
        const int ts = i / WI1, ii = i % WI1;
        const bool live = iwc < nchunks;
        const int soff = (iwt + ts) * (int)tap_bytes + iwc * KB;
        dma16(rw, wbase + iws * WSTAGE + ts * WSLICE + (NIW >= NW ? ii * NW + wid : wid % NIW) * 1024, live ? woff[ii] : OOB_OFF, live ? soff : 0);
    };
    auto w_advance = [&]() __attribute__((always_inline)) {
        iwt += TPS;
        if (iwt == 9) { iwt = 0; ++iwc; }
        if (++iws == NWS) iws = 0;
    };
    auto h_piece = [&](int i, int chunk) __attribute__((always_inline)) {      // piece i of the halo of `chunk` into buffer chunk & 1
        const bool live = chunk < nchunks;
        dma16(rx, hbase + (chunk & 1) * HBUF + (i * NW + wid) * 1024, live ? hoff[i] : OOB_OFF, live ? chunk * KB : 0);
    };

    // ---- prologue: halo of chunk 0, then NWS-1 weight stages
#pragma unroll
    for (int i = 0; i < XH; ++i) h_piece(i, 0);
#pragma unroll
    for (int st = 0; st < NWS - 1; ++st) {
#pragma unroll
        for (int i = 0; i < WI; ++i) w_piece(i);
        w_advance();
    }
    // BNL: pieces [i0, i1) of the halo of `chunk` (landed: the caller's counted wait covers them), rewritten in place by this wave
    auto bnl_fixup = [&](int i0, int i1, int chunk) __attribute__((always_inline)) {
        if constexpr (BNL != 0) {
            if (chunk < nchunks) {
#pragma unroll
                for (int i = 0; i < XH; ++i) {
                    if (i >= i0 && i < i1) {
                        const bool ok = hoff[i] != OOB_OFF;        // outside the image (or past the halo): the zeros stay
                        char* ptr = hbase + (chunk & 1) * HBUF + (i * NW + wid) * 1024 + lane * 16;
                        const int hp = (i * NW + wid) * 8 + lr;
                        const int ch = chunk * 64 + ((slot ^ ((hp % HCOLS) & 7)) << 3);
                        const uint4 v = *reinterpret_cast<uint4*>(ptr);
                        float f[8];
                        Vec16<T>::unpack(v, f);
                        const float4 s0 = *reinterpret_cast<const float4*>(&bnl_tab[ch]), s1 = *reinterpret_cast<const float4*>(&bnl_tab[ch + 4]);
                        const float4 h0 = *reinterpret_cast<const float4*>(&bnl_tab[BNL_CMAX + ch]), h1 = *reinterpret_cast<const float4*>(&bnl_tab[BNL_CMAX + ch + 4]);
                        f[0] = fmaxf(fmaf(f[0], s0.x, h0.x), 0.f); f[1] = fmaxf(fmaf(f[1], s0.y, h0.y), 0.f);
                        f[2] = fmaxf(fmaf(f[2], s0.z, h0.z), 0.f); f[3] = fmaxf(fmaf(f[3], s0.w, h0.w), 0.f);
                        f[4] = fmaxf(fmaf(f[4], s1.x, h1.x), 0.f); f[5] = fmaxf(fmaf(f[5], s1.y, h1.y), 0.f);
                        f[6] = fmaxf(fmaf(f[6], s1.z, h1.z), 0.f); f[7] = fmaxf(fmaf(f[7], s1.w, h1.w), 0.f);
                        const uint4 t = Vec16<T>::pack(f);
                        const uint4 o = make_uint4(ok ? t.x : v.x, ok ? t.y : v.y, ok ? t.z : v.z, ok ? t.w : v.w);
                        *reinterpret_cast<uint4*>(ptr) = o;
                        // the activation, once per pixel: the interior of the patch, from the blocks of channel tile 0 (the per-lane part of
                        // the condition goes into the offset: bstore16)
                        if (bnl_wb) {
                            const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
                            const bool st = ok && hy >= 1 && hy <= PH && hx >= 1 && hx <= 16;
                            bstore16(ro, st ? ((unsigned)((n * p.Hi + y0 - 1 + hy) * p.Wi + x0 - 1 + hx) * (unsigned)p.bnl_out_ld + (unsigned)ch) * ES : OOB_OFF, o);
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    };
    if constexpr (BNL != 0) {
        // the coefficient table, behind the prologue's fills (the compiler's vmcnt(0) for these loads waits for those as well)
        for (int c = tid; c < p.Cin; c += NW * 64) {
            float sc, sh;
            bn_fwd_coeffs(p.bnl, p.Cin, c, blockIdx.x == 0, sc, sh);
            bnl_tab[c] = sc; bnl_tab[BNL_CMAX + c] = sh;
        }
        __syncthreads();
    }

    int cst = 0;                                      // ring slot of the weight stage being consumed
    constexpr int NM = 2 * FM * FN * TPS;             // MFMAs per k-step and wave
    for (int c = 0; c < nchunks; ++c) {
        const char* hb = hbase + (c & 1) * HBUF;
        // (always_inline: with the BNL pass the body outgrew the inliner's budget, and a lambda left as a CALL gets its captures through memory --
        // the LDS-DMA statements inside need their wave-uniform operands in scalar registers)
        auto step = [&](auto jj) __attribute__((always_inline)) {
            constexpr int j = decltype(jj)::value;
            // halo pieces this k-step issues, and those the NWS-2 k-steps before it issued: the operations younger than the
            // last piece of this k-step's weight stage are the NWS-2 later stages and those halo pieces
            constexpr int H0 = j * HPS < XH ? j * HPS : XH, H1 = (j + 1) * HPS < XH ? (j + 1) * HPS : XH;
            constexpr int NH = [] {
                int h = 0;
                for (int d = 1; d <= NWS - 2; ++d) {
                    const int jp = (j - d + SPC) % SPC;
                    const int a0 = jp * HPS < XH ? jp * HPS : XH, a1 = (jp + 1) * HPS < XH ? (jp + 1) * HPS : XH;
                    h += a1 - a0;
                }
                return h;
            }();
            wait_vmcnt<(NWS - 2) * WI + NH>();
            if constexpr (BNL != 0) {
                // this wait covers the halo pieces issued NWS - 1 k-steps ago: at k-step 0 the last pieces of THIS chunk's halo (all of them for
                // the first chunk, which the prologue fetched), from k-step NWS - 1 on pieces of the next chunk's
                constexpr int jp = (j - (NWS - 1) + SPC) % SPC;
                constexpr int F0 = jp * HPS < XH ? jp * HPS : XH, F1 = (jp + 1) * HPS < XH ? (jp + 1) * HPS : XH;
                if (j == 0) bnl_fixup(c == 0 ? 0 : F0, c == 0 ? XH : F1, c);
                else if (j >= NWS - 1) bnl_fixup(F0, F1, c + 1);
            }
            raw_barrier();
            const char* wb = wbase + cst * WSTAGE;
#pragma unroll
            for (int i = H0; i < H1; ++i) h_piece(i, c + 1);          // before this k-step's weight pieces (the count above relies on it)
#pragma unroll
            for (int ts = 0; ts < TPS; ++ts) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 af[FM], bf[FN];
#pragma unroll
                    for (int a = 0; a < FM; ++a) af[a] = *reinterpret_cast<const uint4*>(wb + ts * WSLICE + (aoff[a] ^ (kk * 64)));
#pragma unroll
                    for (int b = 0; b < FN; ++b) bf[b] = *reinterpret_cast<const uint4*>(hb + (boff[j * TPS + ts][b] ^ (kk * 64)));
#pragma unroll
                    for (int a = 0; a < FM; ++a)
#pragma unroll
                        for (int b = 0; b < FN; ++b) {
                            const int m = ((ts * 2 + kk) * FM + a) * FN + b;
#pragma unroll
                            for (int i = 0; i < WI; ++i)
                                if ((i * NM) / WI == m) w_piece(i);
                            Mma<T>::run(af[a], bf[b], acc[a][b]);
                        }
                }
            }
            w_advance();
            if (++cst == NWS) cst = 0;
        };
        if constexpr (TPS == 1) {
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        } else {
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        }
    }
    wait_vmcnt<0>();                                  // the trailing out-of-range pieces still write zeros into the ring
    const int m0 = (n * p.Ho + y0) * p.Wo + x0;
    conv_epilogue<T, FM, FN, FN * 16, WP, 0, WC, true>(p, acc, m0, wp, c0 + wc * WTC + g * NV, pl, 0, 0, patch, 0,
                                                        reinterpret_cast<float*>(smem), wc, c0);
}

}  // namespace
namespace msc_conv {
bool xcd_order_enabled() { static int v = -1; if (v < 0) { const char* e = getenv("MSC_XCD_ORDER"); v = (e && e[0] == '0') ? 0 : 1; } return v == 1; }
}  // namespace msc_conv
namespace {

// ---- kernel configurations.  A configuration = (pixel rows, channels, waves along pixels, waves along channels,
// bytes of K per row and step, ring depth); LDS = NST*(TP+TC)*KB.  Which one is fastest depends on the layer
// (fill-rate-bound small layers want occupancy, large ones want big tiles and full-line K steps), so besides the
// heuristic (cfg 0) the caller may pick one explicitly -- UNetResNet times the valid ones per layer when it builds
// a program (msc_conv_cfg_ok enumerates them).
struct ConvCfg { int tp, tc, wp, wc, kb, nst; };
constexpr int N_CONV_CFG = 59;
static inline bool cfg_is_halo3(int cfg) { return (cfg >= 42 && cfg <= 46) || (cfg >= 51 && cfg <= 56); }      // conv3x3_halo_dma_kernel
constexpr int CFG_HALO = 27;          // conv3x3_c32_halo_kernel (not a tile of the DMA kernel)
constexpr int CFG_HALO_T = 28;        // deconv4_c128_c32_halo_kernel
constexpr int CFG_STEM = 58;          // stem7_halo_kernel (halo32.hip)
constexpr int CFG_DOWN4 = 59;         // down4_c32_halo_kernel (halo32.hip)
static const ConvCfg CONV_CFGS[N_CONV_CFG + 1] = {
    {0, 0, 0, 0, 0, 0},
    {256, 128, 4, 2, 128, 3},   //  1: 144 KB, 8 waves, 1 block/CU
    {256, 128, 4, 2, 64, 4},    //  2:  96 KB, 8 waves
    {128, 128, 2, 2, 128, 4},   //  3: 128 KB
    {128, 128, 2, 2, 64, 4},    //  4:  64 KB, 2 blocks/CU
    {128, 64, 4, 1, 128, 3},    //  5:  72 KB, 2 blocks/CU
    {128, 64, 4, 1, 64, 4},     //  6:  48 KB, 3 blocks/CU
    {64, 64, 2, 2, 128, 4},     //  7:  64 KB
    {64, 64, 2, 2, 64, 4},      //  8:  32 KB
    {256, 32, 4, 1, 128, 3},    //  9: 108 KB
    {256, 32, 4, 1, 64, 4},     // 10:  72 KB
    // 8-wave variants of the mid-size tiles for the 16x16 / 32x32 feature maps of layer3/layer2, where one block per
    // CU is all the layer offers: twice the waves per block to hide the L2 -> LDS latency
    {128, 64, 4, 2, 128, 4},    // 11:  96 KB, 8 waves
    {128, 128, 4, 2, 128, 3},   // 12:  96 KB, 8 waves
    {64, 128, 2, 2, 128, 4},    // 13:  96 KB
    {64, 128, 2, 4, 128, 4},    // 14:  96 KB, 8 waves
    // deeper rings: the small layers are bound by the latency of the L2 -> LDS fill (SQ counters: waves parked 44 % of
    // the time at 13 TB/s aggregate fill), i.e. by the bytes in flight per CU
    {64, 128, 2, 4, 128, 6},    // 15: 144 KB, 8 waves
    {128, 64, 4, 2, 128, 6},    // 16: 144 KB, 8 waves
    {64, 64, 2, 2, 128, 5},     // 17:  80 KB, 2 blocks/CU
    {128, 128, 4, 2, 128, 4},   // 18: 128 KB, 8 waves
    {256, 128, 4, 2, 64, 6},    // 19: 144 KB, 8 waves
    // 256-byte k-steps: 4 MFMA sub-steps (instead of 2) between two barriers
    {64, 64, 2, 2, 256, 4},     // 20: 128 KB
    {64, 128, 2, 4, 256, 3},    // 21: 144 KB, 8 waves
    {128, 64, 4, 2, 256, 3},    // 22: 144 KB, 8 waves
    {128, 128, 4, 2, 256, 2},   // 23: 128 KB, 8 waves
    // two resident 8-wave blocks per CU (4 waves per SIMD) to fill the time the waves of one block are parked
    {64, 128, 2, 4, 128, 3},    // 24:  72 KB, 8 waves
    {128, 64, 4, 2, 128, 3},    // 25:  72 KB, 8 waves
    {64, 64, 2, 2, 256, 2},     // 26:  64 KB
    {16, 32, 1, 1, 64, 1},      // 27: halo-tile kernel for the 32-channel 3x3 layers
    {16, 32, 1, 1, 256, 1},     // 28: halo-tile kernel for ConvTranspose2d(k4, s2, p1) 128 -> 32
    // wide wave tiles: LDS traffic per MFMA is (WTP + WTC) / (WTP * WTC) fragment reads -- a 32x32 wave tile reads one
    // 16-byte fragment per MFMA and is LDS-bound (reads + DMA writes > MFMA cycles), 128x64 reads 0.375
    {256, 128, 2, 2, 128, 3},   // 29: 144 KB, 4 waves of 128 px x 64 ch
    {256, 128, 4, 1, 128, 3},   // 30: 144 KB, 4 waves of 64 px x 128 ch
    {256, 256, 2, 4, 128, 2},   // 31: 128 KB, 8 waves of 128 px x 64 ch (the 256x256 tile of the large-K decoder layers)
    {256, 256, 2, 4, 64, 4},    // 32: 128 KB, 8 waves of 128 px x 64 ch, 64-byte k-steps in a 4-deep ring
    {128, 256, 2, 4, 128, 3},   // 33: 144 KB, 8 waves of 64 px x 64 ch
    {128, 128, 1, 4, 128, 4},   // 34: 128 KB, 4 waves of 128 px x 32 ch
    {128, 64, 2, 1, 128, 4},    // 35:  96 KB, 2 waves of 64 px x 64 ch
    {64, 128, 1, 2, 256, 3},    // 36: 144 KB, 2 waves of 64 px x 64 ch, 256-byte k-steps
    // two blocks per CU with full-line (128-byte) k-steps: a block's prologue (first fill from HBM) and epilogue (stores) are
    // not overlapped with anything of its own -- 40 % of the time of the multi-tile layers (probes/conv_ablate.hip) -- so
    // the second resident block computes meanwhile
    {128, 128, 2, 2, 128, 2},   // 37:  64 KB, 4 waves
    {128, 128, 4, 2, 128, 2},   // 38:  64 KB, 8 waves
    {256, 64, 4, 1, 128, 2},    // 39:  80 KB, 4 waves of 64 px x 64 ch
    {256, 64, 4, 2, 128, 2},    // 40:  80 KB, 8 waves
    {128, 64, 2, 2, 128, 3},    // 41:  72 KB, 4 waves
    // halo-tile kernel for 3x3 / stride 1 convs of any width (conv3x3_halo_dma_kernel): tp = patch rows x 16 pixels, kb = 128,
    // nst = weight ring depth; LDS = 2 halo buffers + nst weight stages
    {256, 128, 4, 2, 128, 3},   // 42: 16x16 patch x 128 ch, 8 waves of 64 px x 64 ch, 144 KB
    {128, 128, 2, 4, 128, 4},   // 43:  8x16 patch x 128 ch, 8 waves of 64 px x 32 ch, 112 KB
    {128, 64, 2, 2, 128, 4},    // 44:  8x16 patch x  64 ch, 4 waves of 64 px x 32 ch,  80 KB, 2 blocks/CU
    {256, 64, 4, 2, 128, 4},    // 45: 16x16 patch x  64 ch, 8 waves of 64 px x 32 ch, 128 KB
    {128, 64, 4, 2, 128, 4},    // 46:  8x16 patch x  64 ch, 8 waves of 32 px x 32 ch,  80 KB, 2 blocks/CU
    // many resident blocks for the HBM-bound 1x1 layers with one or two k-steps (64 / 128 input channels, >= 32 k pixels): the
    // ring hides nothing there, the bytes in flight per CU are what counts
    {64, 64, 2, 2, 128, 2},     // 47:  32 KB, 5 blocks/CU
    {128, 64, 2, 2, 128, 2},    // 48:  48 KB, 3 blocks/CU
    {64, 128, 2, 2, 128, 2},    // 49:  48 KB, 3 blocks/CU
    {128, 128, 2, 2, 64, 2},    // 50:  32 KB, 5 blocks/CU (64-byte k-steps)
    // more halo-tile variants: a kernel row (three taps) per k-step = a third of the barriers; two blocks per CU
    {128, 64, 4, 2, 128, 3},    // 51:  8x16 patch x  64 ch, 8 waves, 3 taps per k-step, 120 KB
    {128, 64, 2, 2, 128, 3},    // 52:  8x16 patch x  64 ch, 4 waves, 3 taps per k-step, 120 KB
    {256, 64, 4, 2, 128, 2},    // 53: 16x16 patch x  64 ch, 8 waves, 3 taps per k-step, 144 KB
    {128, 128, 2, 4, 128, 2},   // 54:  8x16 patch x 128 ch, 8 waves, 3 taps per k-step, 144 KB
    {128, 128, 2, 4, 128, 2},   // 55:  8x16 patch x 128 ch, 8 waves, 80 KB, 2 blocks/CU
    {256, 128, 4, 2, 128, 2},   // 56: 16x16 patch x 128 ch, 8 waves, 128 KB
    {64, 256, 1, 8, 128, 2},    // 57: persistent streaming kernel for 1x1 / stride 1 layers of 64..512 input channels (tile shape per layer: STREAM_VARS)
    {128, 64, 4, 2, 64, 2},     // 58: halo-tile kernel for the stem (7x7 / stride 2 on the prepared 4-channel input)
    {128, 128, 2, 4, 64, 2},    // 59: halo-tile kernel for Conv2d(k4, s2, p1) 32 -> 128 (the data gradient of dec1's ConvTranspose2d), 96 KB
};

template <typename T, int TP, int TC, int WP, int WC, int KB, int NST, int ABL = 0, int BNL = 0>
int launch_dma(const ConvK& k0, int mode, hipStream_t st) {
    ConvK k = k0;
    k.ntc = ceil_div(k.Cout, TC);
    k.xcd_order = xcd_order_enabled() ? 1 : 0;
    dim3 grid(ceil_div(k.M, TP) * k.ntc * k.ksplit, 1, mode ? 4 : 1);
    constexpr int NT = WP * WC * 64;
    if constexpr (BNL != 0) {      // in_bn: gather mode, 16-bit (conv_cfg_ok)
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, TP, TC, WP, WC, 0, NST, KB, ABL, 1>), grid, dim3(NT), 0, st, k);
        return msc_check_launch("conv_igemm_dma (in_bn)");
    }
    if (mode) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, TP, TC, WP, WC, 1, NST, KB, ABL>), grid, dim3(NT), 0, st, k);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<T, TP, TC, WP, WC, 0, NST, KB, ABL>), grid, dim3(NT), 0, st, k);
    if (k.ksplit > 1) {
        const long units = (long)k.M * (k.Cout / 8);
        const int blocks = (int)(units / 256 < 2048 ? (units + 255) / 256 : 2048);
        hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float*)k.kws, k.ksplit, (T*)k.out, k.out_ld, k.scale, k.shift, (const T*)k.res, k.res_ld,
                           k.relu, (long)k.M, k.Cout);
    }
    return msc_check_launch("conv_igemm_dma");
}

template <typename T, int PH, int TC, int WP, int WC, int NWS, int TPS = 1, int MINB = 1, int BNL = 0>
int launch_halo3(const ConvK& k0, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        ConvK k = k0;
        k.ntc = k.Cout / TC;
        k.xcd_order = xcd_order_enabled() ? 1 : 0;
        const int blocks = k.N * (k.Ho / PH) * (k.Wo / 16) * k.ntc;
        hipLaunchKernelGGL((conv3x3_halo_dma_kernel<T, PH, TC, WP, WC, NWS, TPS, MINB, BNL>), dim3(blocks), dim3(WP * WC * 64), 0, st, k);
    }
    return msc_check_launch("conv3x3_halo_dma");
}

// in_bn (BatchNorm + ReLU of the input on load): the implicit-GEMM kernel's 1x1 form on the two tiles it was measured on (1, 33) and three
// tiles of the 3x3 halo kernel (42, 51, 53)
static inline bool cfg_has_bnl(int cfg, int kh) { return kh == 1 ? (cfg == 1 || cfg == 33) : (cfg == 42 || cfg == 51 || cfg == 53); }
bool conv_cfg_ok(const ConvK& k, int es, int cfg) {
    if (cfg < 1 || cfg > N_CONV_CFG) return false;
    if (k.bnl.slots && !(cfg_has_bnl(cfg, k.KH) && es == 2 && k.mode == 0 && k.KH == k.KW && (k.KH == 1 || k.KH == 3) && k.stride == 1 && k.pad == k.KH / 2 &&
                         !k.flip && !k.span_bytes && k.ksplit == 1 && k.Cin <= BNL_CMAX && k.Cin % 64 == 0)) return false;
    if (k.fin_w && cfg != CFG_HALO) return false;           // the fused final 1x1 lives in the 32-channel halo kernel's epilogue only
    if (k.sz && cfg >= 29 && cfg <= 32) return false;       // the residual-join epilogue (stats_z) is not compiled for the 32-fragment wave tiles
    if (k.ksplit > 1 && (cfg == CFG_HALO || cfg == CFG_HALO_T || cfg == CFG_STREAM || cfg == CFG_STEM || cfg == CFG_DOWN4 || cfg_is_halo3(cfg))) return false;      // split-K: the implicit-GEMM kernel only
    if (cfg == CFG_STREAM) return conv1x1_cfg_ok(k, es);
    if (cfg == CFG_STEM)
        return es == 2 && k.mode == 0 && k.KH == 7 && k.KW == 1 && k.stride == 2 && k.pad == 0 && k.Cin == 32 && k.in_ld == 4 && k.Cout == 64 && !k.flip &&
               !k.span_bytes && k.ksplit == 1 && k.Ho % 8 == 0 && k.Wo % 16 == 0 && k.Hi >= 2 * k.Ho + 5 && k.Wi >= 2 * k.Wo + 6 &&
               (!k.stats || k.stats_kind == 0) && k.out_ld % 8 == 0 && !k.res && (long)k.M * k.out_ld * 2 < 0x7fffffffL;      // (round 6: its own epilogue, buffer stores)
    if (cfg == CFG_DOWN4) {
        static const bool off = [] { const char* e = getenv("MSC_DOWN4"); return e && e[0] == '0'; }();      // A/B: the implicit-GEMM tiles for this layer
        if (off) return false;
    }
    if (cfg == CFG_DOWN4)
        return es == 2 && k.mode == 0 && k.KH == 4 && k.KW == 4 && k.stride == 2 && k.pad == 1 && k.Cin == 32 && k.Cout == 128 && !k.flip && !k.span_bytes &&
               k.Ho % 8 == 0 && k.Wo % 16 == 0 && k.Hi == 2 * k.Ho && k.Wi == 2 * k.Wo && (!k.stats || (k.stats_kind == 2 && k.sy && k.sy_ld % 8 == 0)) &&
               k.in_ld % 8 == 0 && k.out_ld % 8 == 0 && !k.res && !k.scale && !k.shift && !k.relu &&       // its own epilogue: plain, or ReLU backward + bias sums
               (long)k.M * k.out_ld * 2 < 0x7fffffffL && (long)k.M * k.sy_ld * 2 < 0x7fffffffL;
    if (cfg == CFG_HALO)
        return es == 2 && k.mode == 0 && k.KH == 3 && k.KW == 3 && k.stride == 1 && k.pad == 1 && k.Cin == 32 && k.Cout == 32 &&
               k.Ho % 16 == 0 && k.Wo % 16 == 0 && k.Hi == k.Ho && k.Wi == k.Wo && (!k.stats || k.stats_kind == 2) && k.res_ld % 8 == 0 && k.out_ld % 8 == 0;
    if (cfg == CFG_HALO_T) {
        static const bool wide_off = [] { const char* e = getenv("MSC_DECONV_WIDE"); return e && e[0] == '0'; }();      // A/B: 32 output channels only (round 4)
        if (wide_off && k.Cout != 32) return false;
        return es == 2 && k.mode == 1 && k.KH == 4 && k.KW == 4 && k.stride == 2 && k.pad == 1 && k.Cin == 128 && k.Cout % 32 == 0 && k.Cout <= 256 &&
               k.Hi % 8 == 0 && k.Wi % 16 == 0 && k.Ho == 2 * k.Hi && k.Wo == 2 * k.Wi && !k.stats && k.res_ld % 8 == 0 && k.out_ld % 8 == 0;
    }
    const ConvCfg& c = CONV_CFGS[cfg];
    if (cfg_is_halo3(cfg))
        return es == 2 && k.mode == 0 && k.KH == 3 && k.KW == 3 && k.stride == 1 && k.pad == 1 && k.Hi == k.Ho && k.Wi == k.Wo &&
               k.Wo % 16 == 0 && k.Ho % (c.tp / 16) == 0 && (k.Cin * es) % 128 == 0 && k.Cout % c.tc == 0;
    if (k.Cout % c.tc) return false;
    if (k.span_bytes && (long)k.Cin * es != c.kb) return false;      // merged taps: the row is one k-step
    if (c.tc == 32 && k.Cout % 64 == 0) return false;        // a 32-channel tile only for the 32-channel layers
    if (((long)k.Cin * es) % c.kb) return false;
    return true;
}

// heuristic: largest tile that still gives every CU a block; 64-byte K steps (best on average over the network)
int pick_cfg(const ConvK& k) {
    const int M = k.M, Cout = k.Cout;
    if (k.ksplit > 1 && Cout % 128 == 0 && ((long)k.Cin * 2) % 128 == 0) return 3;      // split-K: 128x128 tiles, 128-byte k-steps
    if (k.span_bytes) return Cout % 128 == 0 ? 23 : 20;      // merged taps: 256-byte k-steps only
    if (Cout % 128 == 0) {
        if ((long)ceil_div(M, 256) * (Cout / 128) >= 512) return 2;
        if ((long)ceil_div(M, 128) * (Cout / 128) >= 512) return 4;
        return 8;
    }
    if (Cout % 64 == 0) return ((long)ceil_div(M, 128) * (Cout / 64) >= 512) ? 6 : 8;
    return 10;
}

template <typename T>
int conv_dispatch(const ConvK& k, int mode, int cfg, hipStream_t st) {
    if (cfg == 0) {
        if (k.fin_w) cfg = CFG_HALO;
        else if (k.bnl.slots) {          // in_bn: the first of the configurations that apply BatchNorm on load and take this layer
            static const int cand3[] = {42, 51, 53}, cand1[] = {33, 1};
            const int* cand = k.KH == 3 ? cand3 : cand1;
            const int nc = k.KH == 3 ? 3 : 2;
            cfg = cand[0];
            for (int i = 0; i < nc; ++i)
                if (conv_cfg_ok(k, (int)sizeof(T), cand[i])) { cfg = cand[i]; break; }
        } else cfg = pick_cfg(k);
    }
    if (!conv_cfg_ok(k, (int)sizeof(T), cfg)) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: configuration %d is not valid for this layer", cfg);
    if (cfg == CFG_HALO) return halo32_conv_launch(k, std::is_same<T, f16_t>::value ? MSC_F16 : MSC_BF16, st);
    if (cfg == CFG_HALO_T) return halo32_deconv_launch(k, std::is_same<T, f16_t>::value ? MSC_F16 : MSC_BF16, st);
    if (cfg == CFG_STEM) return halo32_stem_launch(k, std::is_same<T, f16_t>::value ? MSC_F16 : MSC_BF16, st);
    if (cfg == CFG_DOWN4) return halo32_down_launch(k, std::is_same<T, f16_t>::value ? MSC_F16 : MSC_BF16, st);
    if (cfg == CFG_STREAM) return conv1x1_launch(k, std::is_same<T, f16_t>::value ? MSC_F16 : MSC_BF16, st);
    if (k.bnl.slots) {
        if (cfg == 1) return launch_dma<T, 256, 128, 4, 2, 128, 3, 0, 1>(k, mode, st);
        if (cfg == 33) return launch_dma<T, 128, 256, 2, 4, 128, 3, 0, 1>(k, mode, st);
        if (cfg == 42) return launch_halo3<T, 16, 128, 4, 2, 3, 1, 1, 1>(k, st);
        if (cfg == 51) return launch_halo3<T, 8, 64, 4, 2, 3, 3, 1, 1>(k, st);
        return launch_halo3<T, 16, 64, 4, 2, 2, 3, 1, 1>(k, st);
    }
    switch (cfg) {
        case 1: return launch_dma<T, 256, 128, 4, 2, 128, 3>(k, mode, st);
        case 2: return launch_dma<T, 256, 128, 4, 2, 64, 4>(k, mode, st);
        case 3: return launch_dma<T, 128, 128, 2, 2, 128, 4>(k, mode, st);
        case 4: return launch_dma<T, 128, 128, 2, 2, 64, 4>(k, mode, st);
        case 5: return launch_dma<T, 128, 64, 4, 1, 128, 3>(k, mode, st);
        case 6: return launch_dma<T, 128, 64, 4, 1, 64, 4>(k, mode, st);
        case 7: return launch_dma<T, 64, 64, 2, 2, 128, 4>(k, mode, st);
        case 8: return launch_dma<T, 64, 64, 2, 2, 64, 4>(k, mode, st);
        case 9: return launch_dma<T, 256, 32, 4, 1, 128, 3>(k, mode, st);
        case 10: return launch_dma<T, 256, 32, 4, 1, 64, 4>(k, mode, st);
        case 11: return launch_dma<T, 128, 64, 4, 2, 128, 4>(k, mode, st);
        case 12: return launch_dma<T, 128, 128, 4, 2, 128, 3>(k, mode, st);
        case 13: return launch_dma<T, 64, 128, 2, 2, 128, 4>(k, mode, st);
        case 14: return launch_dma<T, 64, 128, 2, 4, 128, 4>(k, mode, st);
        case 15: return launch_dma<T, 64, 128, 2, 4, 128, 6>(k, mode, st);
        case 16: return launch_dma<T, 128, 64, 4, 2, 128, 6>(k, mode, st);
        case 17: return launch_dma<T, 64, 64, 2, 2, 128, 5>(k, mode, st);
        case 18: return launch_dma<T, 128, 128, 4, 2, 128, 4>(k, mode, st);
        case 19: return launch_dma<T, 256, 128, 4, 2, 64, 6>(k, mode, st);
        case 20: return launch_dma<T, 64, 64, 2, 2, 256, 4>(k, mode, st);
        case 21: return launch_dma<T, 64, 128, 2, 4, 256, 3>(k, mode, st);
        case 22: return launch_dma<T, 128, 64, 4, 2, 256, 3>(k, mode, st);
        case 23: return launch_dma<T, 128, 128, 4, 2, 256, 2>(k, mode, st);
        case 24: return launch_dma<T, 64, 128, 2, 4, 128, 3>(k, mode, st);
        case 25: return launch_dma<T, 128, 64, 4, 2, 128, 3>(k, mode, st);
        case 26: return launch_dma<T, 64, 64, 2, 2, 256, 2>(k, mode, st);
        case 29: return launch_dma<T, 256, 128, 2, 2, 128, 3>(k, mode, st);
        case 30: return launch_dma<T, 256, 128, 4, 1, 128, 3>(k, mode, st);
        case 31: return launch_dma<T, 256, 256, 2, 4, 128, 2>(k, mode, st);
        case 32: return launch_dma<T, 256, 256, 2, 4, 64, 4>(k, mode, st);
        case 33: return launch_dma<T, 128, 256, 2, 4, 128, 3>(k, mode, st);
        case 34: return launch_dma<T, 128, 128, 1, 4, 128, 4>(k, mode, st);
        case 35: return launch_dma<T, 128, 64, 2, 1, 128, 4>(k, mode, st);
        case 36: return launch_dma<T, 64, 128, 1, 2, 256, 3>(k, mode, st);
        case 37: return launch_dma<T, 128, 128, 2, 2, 128, 2>(k, mode, st);
        case 38: return launch_dma<T, 128, 128, 4, 2, 128, 2>(k, mode, st);
        case 39: return launch_dma<T, 256, 64, 4, 1, 128, 2>(k, mode, st);
        case 40: return launch_dma<T, 256, 64, 4, 2, 128, 2>(k, mode, st);
        case 42: return launch_halo3<T, 16, 128, 4, 2, 3>(k, st);
        case 43: return launch_halo3<T, 8, 128, 2, 4, 4>(k, st);
        case 44: return launch_halo3<T, 8, 64, 2, 2, 4>(k, st);
        case 45: return launch_halo3<T, 16, 64, 4, 2, 4>(k, st);
        case 46: return launch_halo3<T, 8, 64, 4, 2, 4>(k, st);
        case 47: return launch_dma<T, 64, 64, 2, 2, 128, 2>(k, mode, st);
        case 48: return launch_dma<T, 128, 64, 2, 2, 128, 2>(k, mode, st);
        case 49: return launch_dma<T, 64, 128, 2, 2, 128, 2>(k, mode, st);
        case 50: return launch_dma<T, 128, 128, 2, 2, 64, 2>(k, mode, st);
        case 51: return launch_halo3<T, 8, 64, 4, 2, 3, 3>(k, st);
        case 52: return launch_halo3<T, 8, 64, 2, 2, 3, 3>(k, st);
        case 53: return launch_halo3<T, 16, 64, 4, 2, 2, 3>(k, st);
        case 54: return launch_halo3<T, 8, 128, 2, 4, 2, 3>(k, st);
        case 55: return launch_halo3<T, 8, 128, 2, 4, 2, 1, 2>(k, st);
        case 56: return launch_halo3<T, 16, 128, 4, 2, 2, 1>(k, st);
        default: return launch_dma<T, 128, 64, 2, 2, 128, 3>(k, mode, st);
    }
}


}  // namespace

static int conv_fill(const msc_conv_desc* d, ConvK* k) {
    if (!d || !d->in || !d->wt || !d->out) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: null pointer");
    if (!msc_dtype_ok(d->dtype)) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: dtype %d", d->dtype);
    const int es = msc_dtype_size(d->dtype);
    if ((d->Cin * es) % 64) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: Cin*%d must be a multiple of 64 bytes (Cin=%d)", es, d->Cin);
    if (d->Cout % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: Cout must be a multiple of 32 (Cout=%d)", d->Cout);
    // 16-byte vector loads: every pixel start must be 16-byte aligned, or (stem form: KW == 1, pad == 0) every pixel
    // the kernel can address (x*stride) must be
    const bool in_ok = (d->in_ld * es) % 16 == 0 ||
                       (d->mode == 0 && d->KW == 1 && d->pad == 0 && (d->stride * d->in_ld * es) % 16 == 0 && ((int64_t)d->Wi * d->in_ld * es) % 16 == 0);
    if (!in_ok || (d->out_ld * es) % 16 || (d->res && (d->res_ld * es) % 16))
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: channel strides must keep 16-byte alignment");
    if (((uintptr_t)d->in | (uintptr_t)d->wt | (uintptr_t)d->out | (uintptr_t)d->res) & 15)
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: pointers must be 16-byte aligned");
    k->in = (const char*)d->in; k->wt = (const char*)d->wt; k->out = (char*)d->out; k->res = (const char*)d->res;
    k->scale = d->scale; k->shift = d->shift; k->stats = d->stats;
    k->sy = (const char*)d->stats_y; k->sy_ld = d->stats_y_ld; k->stats_kind = d->stats ? d->stats_kind : 0;
    if (k->stats_kind < 0 || k->stats_kind > 2) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: stats_kind %d", d->stats_kind);
    if (k->stats_kind == 2 && (!d->stats_y || d->res || d->relu || d->scale || d->shift || (d->stats_y_ld * es) % 16 || ((uintptr_t)d->stats_y & 15)))
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: ReLU-backward statistics need stats_y (16-byte aligned), no residual, no ReLU, no scale/shift");
    k->sz = k->stats_kind == 1 ? (const char*)d->stats_z : nullptr; k->sz_ld = d->stats_z_ld;
    k->sz_bits = (k->sz && d->stats_z_bits) ? 1 : 0;      // the byte mask of msc_bn_apply instead of the activation (byte stride, no alignment)
    if (k->sz_bits && d->stats_z_ld * (16 / es) < d->Cout) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: stats_z_bits needs Cout / %d mask bytes per pixel", 16 / es);
    if (k->stats_kind == 1 && (!d->stats_y || d->relu || (d->stats_y_ld * es) % 16 || ((uintptr_t)d->stats_y & 15) || !d->scale != !d->shift ||
                               (d->res && !d->stats_z) ||
                               (d->stats_z && (d->scale || (!d->stats_z_bits && ((d->stats_z_ld * es) % 16 || ((uintptr_t)d->stats_z & 15)))))))
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: BatchNorm-backward statistics need stats_y (16-byte aligned), no ReLU, and either the forward coefficients "
                                     "(no residual) or stats_z (no coefficients) for the mask");
    k->in_ld = d->in_ld; k->out_ld = d->out_ld; k->res_ld = d->res_ld;
    k->N = d->N; k->Hi = d->Hi; k->Wi = d->Wi; k->Cin = d->Cin; k->Ho = d->Ho; k->Wo = d->Wo; k->Cout = d->Cout;
    k->KH = d->KH; k->KW = d->KW; k->stride = d->stride; k->pad = d->pad; k->flip = d->flip; k->relu = d->relu;
    k->mode = d->mode;
    k->fin_w = d->final_w; k->fin_b = d->final_b; k->fin_logits = d->final_logits; k->fin_probs = d->final_probs;
    k->fin_skip = d->final_w ? d->final_skip_store : 0;
    k->ksplit = d->splitk > 1 ? d->splitk : 1;
    k->kws = d->splitk_ws;
    if (k->ksplit > 1 && (d->mode != 0 || d->stats || d->final_w || !d->splitk_ws || ((uintptr_t)d->splitk_ws & 15) || d->Cout % 8 || k->ksplit > 64))
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: split-K needs mode 0, no statistics, a 16-byte aligned fp32 workspace and at most 64 slices");
    if (d->final_w && (d->res || d->stats || (!d->final_logits && !d->final_probs)))
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: the fused final 1x1 takes no residual / statistics and needs a logits or probabilities output");
    k->bnl = BnFwdFin{}; k->bnl_out = nullptr; k->bnl_out_ld = 0; k->bnl_out_bytes = 0;
    if (d->in_bn) {
        const msc_bn_input* b = d->in_bn;
        if (!b->slots || !b->scale || !b->shift || b->count <= 0 || es != 2 || (b->out && ((b->out_ld * es) % 16 || ((uintptr_t)b->out & 15))))
            return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: in_bn needs slots, scale / shift outputs, a pixel count, a 16-bit dtype and a 16-byte aligned activation");
        k->bnl = BnFwdFin{b->slots, (double)b->count, b->gamma, b->beta, b->eps, b->momentum, b->running_mean, b->running_var, b->scale, b->shift,
                          b->save_mean, b->save_invstd};
        k->bnl_out = (char*)b->out; k->bnl_out_ld = b->out_ld;
        const long out_b = (((long)d->N * d->Hi * d->Wi - 1) * b->out_ld + d->Cin) * es;
        if (b->out && out_b >= 0x7fffffffL) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: in_bn activation beyond 2 GiB (%ld bytes)", out_b);
        k->bnl_out_bytes = b->out ? (unsigned)out_b : 0;
    }
    k->span_bytes = 0;
    if (d->mode == 1) {
        if (d->stride != 2 || (d->Ho & 1) || (d->Wo & 1)) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: transposed mode needs stride 2 and even output size");
        // statistics in transposed mode (round 4): the epilogue is the same code in both modes (side tensors are addressed by the OUTPUT
        // pixel); every output pixel belongs to exactly one parity phase, so the phases' blocks add disjoint contributions to the slots
        k->Hq = d->Ho / 2; k->Wq = d->Wo / 2;
    } else if (d->mode == 0) {
        if (d->flip && d->stride != 1) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: flip needs stride 1");
        k->Hq = d->Ho; k->Wq = d->Wo;
    } else {
        return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: mode %d", d->mode);
    }
    const long m = (long)d->N * k->Hq * k->Wq;
    if (m <= 0 || m > 0x7fffffffL) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: bad pixel count %ld", m);
    k->M = (int)m;
    // the kernels decode pixel -> (image, row, column) through float reciprocals, exact below 2^24 pixels: msc_conv_igemm hands over image
    // ranges that stay below (conv_image_chunk)
    if (m >= (1L << 24)) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: one image has %ld output pixels (>= 2^24)", m / d->N);
    k->rcp_hw = 1.0f / (float)(k->Hq * k->Wq);
    k->rcp_w = 1.0f / (float)k->Wq;
    // extents of the buffer descriptors: 31-bit offsets (msc_conv_igemm hands over image ranges that fit, conv_image_chunk)
    const long in_b = (((long)d->N * d->Hi * d->Wi - 1) * d->in_ld + d->Cin) * es;
    const long wt_b = (long)d->Cout * d->KH * d->KW * d->Cin * es;
    const bool fits = in_b < 0x7fffffffL && wt_b < 0x7fffffffL;
    if (!fits) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: one image (%ld bytes) or the weights (%ld bytes) exceed 2 GiB", in_b / d->N, wt_b);
    k->in_bytes = (unsigned)in_b;
    k->wt_bytes = (unsigned)wt_b;
    // Narrow compact inputs (32 channels = 64-byte rows, the slow LDS-DMA case): the KW taps of a kernel row read KW consecutive
    // pixels = KW*Cin contiguous elements, and the weights of those taps are contiguous too ([Cout][KH][KW][Cin]) -- run the layer
    // as KW' = 1 with Cin' = KW*Cin (one 256-byte k-step per kernel row instead of four 64-byte ones); the DMA kernel bounds-checks
    // every lane against the pixel its 16 bytes belong to (span_bytes).  Gather mode without flip only (the 4x4 / stride-2 data
    // gradient of dec1's ConvTranspose2d: 32 -> 128 channels at full resolution).
    // Measured on the one layer it applies to (32 -> 128, 524288 output pixels): 250 us merged vs 174 us with four 64-byte k-steps per
    // kernel row -- the merged rows start 64 bytes off the 128-byte lines (pixel 2x-1) and straddle three of them; OFF by default
    // (MSC_CONV_MERGE_KW=2 enables it).  The same merge in the weight gradient (wgrad_plan) is a gain and on by default.
    static const bool merge_on = [] { const char* e = getenv("MSC_CONV_MERGE_KW"); return e && e[0] == '2'; }();
    if (merge_on && fits && d->mode == 0 && !d->flip && d->KW > 1 && d->in_ld == d->Cin && (long)d->KW * d->Cin * es == 256 &&
        (d->Cin * es) % 16 == 0 && d->Cout % 64 == 0) {
        k->span_bytes = d->Cin * es;
        k->Cin = d->KW * d->Cin;
        k->KW = 1;
    }
    return MSC_OK;
}

// Images per launch: the kernels address their input with 31-bit byte offsets (buffer descriptors) and decode pixels through float
// reciprocals (exact below 2^24 pixels), so a tensor beyond 2 GiB
// (fp32 mode, 512x512, batch 64: 2.1 GB at full resolution) runs as consecutive image ranges -- images are independent, the
// statistics epilogues accumulate atomically.  Returns the number of images a launch may cover (>= 1; d->N when all fit).
static int conv_image_chunk(const msc_conv_desc* d) {
    if (!d || d->N <= 0 || !msc_dtype_ok(d->dtype)) return 1;
    const long es = msc_dtype_size(d->dtype);
    const long per_image = (long)d->Hi * d->Wi * d->in_ld * es;
    const long out_pixels = d->mode == 1 ? (long)(d->Ho / 2) * (d->Wo / 2) : (long)d->Ho * d->Wo;      // per image and launch grid (transposed: per parity phase)
    long n = d->N;
    if (per_image > 0 && per_image * n >= 0x7fff0000L) n = 0x7fff0000L / per_image;
    if (out_pixels > 0 && out_pixels * n >= (1L << 24)) n = ((1L << 24) - 1) / out_pixels;
    return n < 1 ? 1 : (int)n;
}

// descriptor of the image range [n0, n0 + n) of *d
static msc_conv_desc conv_image_range(const msc_conv_desc* d, int n0, int n) {
    msc_conv_desc c = *d;
    const long es = msc_dtype_size(d->dtype);
    c.N = n;
    c.in = (const char*)d->in + (long)n0 * d->Hi * d->Wi * d->in_ld * es;
    c.out = (char*)d->out + (long)n0 * d->Ho * d->Wo * d->out_ld * es;
    if (d->res) c.res = (const char*)d->res + (long)n0 * d->Ho * d->Wo * d->res_ld * es;
    if (d->stats_y) c.stats_y = (const char*)d->stats_y + (long)n0 * d->Ho * d->Wo * d->stats_y_ld * es;
    if (d->stats_z) c.stats_z = (const char*)d->stats_z + (long)n0 * d->Ho * d->Wo * d->stats_z_ld * (d->stats_z_bits ? 1 : es);
    if (d->final_logits) c.final_logits = d->final_logits + (long)n0 * 2 * d->Ho * d->Wo;
    if (d->final_probs) c.final_probs = d->final_probs + (long)n0 * 2 * d->Ho * d->Wo;
    return c;
}

extern "C" int msc_conv_stats_slices(const msc_conv_desc* d) {
    ConvK probe;
    if (!d) return -1;
    const msc_conv_desc first = conv_image_range(d, 0, conv_image_chunk(d));
    return conv_fill(&first, &probe) == MSC_OK ? MSC_BN_SLOTS : -1;      // one accumulation slot per XCD, whatever the tile
}

extern "C" int msc_conv_num_cfgs(void) { return N_CONV_CFG; }

extern "C" int msc_conv_cfg_ok(const msc_conv_desc* d, int cfg) {
    ConvK k;
    if (!d) return 0;
    const msc_conv_desc first = conv_image_range(d, 0, conv_image_chunk(d));
    if (d->in_bn && first.N < d->N) return 0;      // in_bn: one image range per launch (msc_conv_igemm)
    if (conv_fill(&first, &k) != MSC_OK) return 0;
    return conv_cfg_ok(k, msc_dtype_size(d->dtype), cfg) ? 1 : 0;
}

extern "C" int msc_conv_igemm(const msc_conv_desc* d, void* stream) {
    if (!d) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: null descriptor");
    hipStream_t st = (hipStream_t)stream;
    const int chunk = conv_image_chunk(d);
    // in_bn: the first block of the launch publishes the coefficients and moves the running statistics -- once per layer, so one image range
    if (d->in_bn && chunk < d->N) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: in_bn needs the whole batch in one image range (%d of %d images fit)", chunk, d->N);
    for (int n0 = 0; n0 < (d->N > 0 ? d->N : 1); n0 += chunk) {
        const msc_conv_desc part = conv_image_range(d, n0, d->N - n0 < chunk ? d->N - n0 : chunk);
        ConvK k;
        int rc = conv_fill(&part, &k);
        if (rc != MSC_OK) return rc;
        if (d->dtype == MSC_F16) rc = conv_dispatch<f16_t>(k, d->mode, d->cfg, st);
        else if (d->dtype == MSC_BF16) rc = conv_dispatch<bf16_t>(k, d->mode, d->cfg, st);
        else rc = conv_dispatch<float>(k, d->mode, d->cfg, st);
        if (rc != MSC_OK) return rc;
    }
    return MSC_OK;
}
