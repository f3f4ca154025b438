// Weight gradients of the convolution family for gfx950 (MI355X): one launch per layer (msc_conv_wgrad) or many layers of one tile
// shape per launch (msc_wgrad_group_*).  Replaces the weight-gradient half of the cuDNN / MKL-DNN backward behind the reference's
// nn.Conv2d / nn.ConvTranspose2d (src/unet_models.py:21-34,136-141,360-383; loss.backward(), src/steps/pytorch/models.py:110).
// Split from igemm.hip in round 4 (the two halves compile in parallel); layouts and the LDS-DMA ring are described there.
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

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[a][kh][kw][b] += sum_m P[m][a] * Q[pix(m)*stride - pad + (kh,kw)][b]
//   conv  wgrad: P = dY (a = cout), Q = X  (b = cin)
//   convT wgrad: P = X  (a = cin, coarse grid), Q = dOut (b = cout, fine grid), stride 2
// GEMM K = pixels, the strided dimension of NHWC: both operands are staged pixel-major in LDS and the
// k-contiguous MFMA fragments are produced by a transposing LDS read.
constexpr int WGRAD_NS = 5, WGRAD_NCFG = 3 * WGRAD_NS;

struct WgK {
    const char* p; const char* q; float* dw;
    long p_ld, q_ld;
    int N, Hp, Wp, A, Hq, Wq, B, KH, KW, stride, pad;
    int M, mchunk, tiles_b;
    int ntiles, ntaps, xcd_order;    // 1-D grid of ntiles*ntaps*splits blocks; split slowest, channel tile fastest
    int nblocks;                     // ntiles*ntaps*splits
    unsigned p_bytes, q_bytes;
    float rcp_hw, rcp_w;
    int kw3, sw;                     // kw3: taps of a kernel row a block covers (wgrad3_dma_body; ntaps = KH): 0 one, 3 (3x3 / stride 1), 4 (4x4 / stride 2: ConvTranspose2d); sw = min(Wp, 32)
    int span_bytes;                  // > 0: a Q row of B*ES bytes spans several consecutive pixels of span_bytes each (KW taps merged, wgrad_plan)
    int no_direct;                   // 1: keep the general pixel decode also for 1x1 / stride 1 (A/B measurements, MSC_WGRAD_DIRECT=0)
    int pad_;
    float* part;                     // MSC_WGRAD_ORDERED groups: split s of this problem stores its tile into plane s (part + s * plane) instead of
    long plane;                      //   adding into dw atomically; wgrad_finish_kernel sums the planes in split order.  nullptr: atomics, or
                                     //   (plane < 0: an ordered layer that is not split -- every element has ONE writer) a plain dw += v
};
static_assert(sizeof(WgK) % 8 == 0, "WgK is copied in dwords and holds pointers");

// one accumulator out of a block: an fp32 atomic onto the gradient, or (ordered groups) a plain store into the split's plane
__device__ __forceinline__ void wgrad_emit(const WgK& p, int split, long idx, float v) {
    if (p.part) p.part[(long)split * p.plane + idx] = v;
    else if (p.plane < 0) p.dw[idx] += v;
    else atomicAdd(p.dw + idx, v);
}

// Blocks that share a pixel range (one split) share P/Q: give each XCD (block b runs on XCD b % 8) a contiguous
// run of the split-major order so that range stays in one L2 instead of all eight.
__device__ __forceinline__ void wgrad_block(const WgK& p, int orig, int nwg, int& tile, int& tap, int& split) {
    int wgid = orig;
    if (p.xcd_order) {
        const int xcd = orig & 7, wq = nwg >> 3, wr = nwg & 7;
        wgid = (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (orig >> 3);
    }
    const int per = p.ntiles * p.ntaps;
    split = wgid / per;
    const int r = wgid - split * per;
    tap = r / p.ntiles;
    tile = r - tap * p.ntiles;
}

// both operands HBM -> LDS by DMA (pixel rows of TA*ES / TB*ES contiguous bytes, 4-stage ring, counted
// vmcnt), k-contiguous fragments by ds_read_b64_tr_b16 (bf16: the hardware transposes a [4 pixels][16 channels]
// block per 16-lane group; probe: profiles/r1_tr_b16_probe.txt) or by ds_read_b32 (f32, one k element per lane).
// Bank conflicts: 16-byte units of pixel row r are permuted with
//   bf16: unit pair (32 B = the 16 channels one lane group reads) index ^= (r&3) | ((r>>3)&1)<<2
//   f32 : 64-byte granule index ^= (r>>2)&1
// applied on the DMA source side and on the read side alike.

// QS = 2 (the four-tap ConvTranspose2d form reads every other row of its Q stage: rows 2 j + kw): keyed on row >> 1, so that the four
// rows a lane group reads still fall into four different unit pairs (keyed on the row itself they alternate between two: 2-way conflicts
// on 32 of the 36 transposing reads of a k-step)
template <typename T, int QS = 1> __device__ __forceinline__ int wg_swz(int unit, int row, int upr) {
    if (sizeof(T) == 2) {
        if (QS == 2) row >>= 1;
        int key = ((row & 3) | (((row >> 3) & 1) << 2)) & (upr / 2 - 1);
        // round 6: rows of 128 / 64 bytes (the 64- and 32-channel tiles).  A 32-lane group of ds_read_b64_tr_b16 reads rows {r .. r+3, r+8 .. r+11}, 32 bytes of
        // each; the LDS bank window is 256 bytes, so with 128-byte rows the row's parity already selects the half and the key has to tell r, r+2, r+8, r+10
        // apart (bits 1 and 3 of the row), with 64-byte rows r&3 selects the quarter and only r and r+8 collide (bit 3).  The masked 256-byte key above used
        // bits 0-1 there: rows r and r+8 got the same unit pair -- SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE on the 64x64 / 64x32 / 32x32 grouped
        // kernels (profiles/r5_final_sq_summary.txt), 4 % on the 128x128 ones.
        if (QS == 1 && upr / 2 == 4) key = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
        if (QS == 1 && upr / 2 == 2) key = (row >> 3) & 1;
        return (((unit >> 1) ^ key) << 1) | (unit & 1);
    }
    return unit ^ ((((row >> 2) & 1) << 2) & (upr - 1));
}

typedef short v4i16_t __attribute__((ext_vector_type(4)));

// ABL (probes/wgrad_ablate.hip only; 0 in the product): bit 0 = no MFMA, bit 1 = no DMA after the prologue, bit 2 = no fragment
// reads and no MFMA, bit 5 = no atomics (the epilogue keeps the accumulators alive only)
template <typename T, int TA, int TB, int NST, int ABL = 0, int NWV = 4>
__device__ __forceinline__ void wgrad_dma_body(const WgK& p, const int orig, const int nwg) {
    constexpr int ES = sizeof(T);
    constexpr int KP = 64 / ES;                  // pixels per k-step (32 bf16 / 16 f32)
    constexpr int RBA = TA * ES, RBB = TB * ES;  // bytes per pixel row of each tile
    constexpr int UA = RBA / 16, UB = RBB / 16;  // 16-byte units per row
    constexpr int NIA = KP * RBA / 1024, NIB = KP * RBB / 1024;   // DMA wave-instructions per tile
    constexpr int IA = (NIA + NWV - 1) / NWV, IB = (NIB + NWV - 1) / NWV;   // ... per wave
    constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;            // pixel rows per wave-instruction
    constexpr int STAGE = KP * (RBA + RBB);
    constexpr int LPW = IA + IB;
    constexpr int NWA = NWV / 2;                                 // waves along A x 2 along B
    constexpr int WTA = TA / NWA, WTB = TB / 2;
    constexpr int FM = WTA / 16, FN = WTB / 16;
    static_assert(NIA >= 2 && NIB >= 2, "tile too small");
    static_assert(NWV == 4 || (NIA >= NWV && NIB >= NWV), "8-wave form: every wave fetches");
    static_assert(NST * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wid >> 1, wb = wid & 1;
    const int g = lane >> 4, pl = lane & 15;
    int tile, tap, split;
    wgrad_block(p, orig, nwg, tile, tap, split);
    const int ta = tile / p.tiles_b, tb = tile - ta * p.tiles_b;
    const int a0 = ta * TA, b0 = tb * TB;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + KP - 1) / KP : 0;

    const u32x4_t rp = make_srd(p.p, p.p_bytes);
    const u32x4_t rq = make_srd(p.q, p.q_bytes);
    const unsigned ppix = (unsigned)p.p_ld * ES, qpix = (unsigned)p.q_ld * ES;
    const unsigned hw = (unsigned)(p.Hp * p.Wp);

    // per-lane (row, unit) of each DMA instruction this wave issues
    int prow[IA], qrow[IB];
    unsigned pcol[IA], qcol[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int j = NIA >= NWV ? i * NWV + wid : (wid & (NIA - 1));
        prow[i] = j * RPA + lane / UA;
        pcol[i] = (unsigned)a0 * ES + (unsigned)wg_swz<T>(lane % UA, prow[i], UA) * 16u;
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int j = NIB >= NWV ? i * NWV + wid : (wid & (NIB - 1));
        qrow[i] = j * RPB + lane / UB;
        qcol[i] = (unsigned)b0 * ES + (unsigned)wg_swz<T>(lane % UB, qrow[i], UB) * 16u;
    }
    // (n, y, x) of the P-grid pixel each Q piece fetches next.  Stages are issued in k-step order, KP pixels apart, so the
    // decode advances incrementally (one conditional wrap per axis) instead of two divisions per DMA instruction and
    // k-step -- the address arithmetic was what bounded this kernel (probes/wgrad_ablate.hip: "DMA only" = 80 % of the full
    // time at 20 B/clk/CU).  Images smaller than a k-step's pixel run keep the division path.
    const bool incr = KP / p.Wp + 1 <= p.Hp;
    const bool direct = !p.no_direct && p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0 && p.Hq == p.Hp && p.Wq == p.Wp && !p.span_bytes;
    const int dxs = KP % p.Wp, dys = KP / p.Wp;
    int qn_[IB], qy_[IB], qx_[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const unsigned m = (unsigned)(mbeg + qrow[i]);
        const unsigned n = udiv_rcp(m, hw, p.rcp_hw);
        const unsigned rem = m - n * hw;
        const unsigned y = udiv_rcp(rem, (unsigned)p.Wp, p.rcp_w);
        qn_[i] = (int)n; qy_[i] = (int)y; qx_[i] = (int)(rem - y * (unsigned)p.Wp);
    }
    // one DMA wave-instruction of the stage holding k-step s (pieces 0..IA-1: P rows, IA..IA+IB-1: Q rows); every piece
    // is issued exactly once per k-step, in k-step order
    auto piece = [&](int i, int s, int stage) {
        const int mb = mbeg + s * KP;
        char* sp = smem + stage * STAGE;
        char* sq = sp + KP * RBA;
        if (i < IA) {
            const int ii = i < IA ? i : 0;
            const int m = mb + prow[ii];
            const unsigned off = m < mend ? (unsigned)m * ppix + pcol[ii] : OOB_OFF;
            dma16(rp, sp + (NIA >= NWV ? ii * NWV + wid : (wid & (NIA - 1))) * 1024, off, 0);
        } else if (direct) {          // 1x1 / stride 1: the Q pixel IS the P pixel -- no decode, no bounds beyond the pixel range
            const int ii = i >= IA ? i - IA : 0;
            const int m = mb + qrow[ii];
            const unsigned off = m < mend ? (unsigned)m * qpix + qcol[ii] : OOB_OFF;
            dma16(rq, sq + (NIB >= NWV ? ii * NWV + wid : (wid & (NIB - 1))) * 1024, off, 0);
        } else {
            const int ii = i >= IA ? i - IA : 0;
            const int m = mb + qrow[ii];
            int n = qn_[ii], y = qy_[ii], x = qx_[ii];
            if (!incr) {
                const unsigned nn = udiv_rcp((unsigned)m, hw, p.rcp_hw);
                const unsigned rem = (unsigned)m - nn * hw;
                const unsigned yy = udiv_rcp(rem, (unsigned)p.Wp, p.rcp_w);
                n = (int)nn; y = (int)yy; x = (int)(rem - yy * (unsigned)p.Wp);
            }
            unsigned off = OOB_OFF;
            // merged taps (span_bytes): the Q row covers several pixels, this lane's 16 bytes belong to pixel ix0 + qsp and are
            // bounds-checked as such; qcol already is the byte offset within the whole row
            const int qsp = p.span_bytes ? (int)(qcol[ii] / (unsigned)p.span_bytes) : 0;
            const int iy = y * p.stride - p.pad + kh, ix = x * p.stride - p.pad + kw + qsp;
            if (m < mend && (unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq)
                off = (unsigned)((n * p.Hq + iy) * p.Wq + ix - qsp) * qpix + qcol[ii];
            dma16(rq, sq + (NIB >= NWV ? ii * NWV + wid : (wid & (NIB - 1))) * 1024, off, 0);
            // advance this piece's pixel by one k-step
            x += dxs;
            if (x >= p.Wp) { x -= p.Wp; ++y; }
            y += dys;
            if (y >= p.Hp) { y -= p.Hp; ++n; }
            qn_[ii] = n; qy_[ii] = y; qx_[ii] = x;
        }
    };
    auto issue = [&](int s, int stage) {
#pragma unroll
        for (int i = 0; i < LPW; ++i) piece(i, s, stage);
    };

    // fragment read offsets (bytes within a stage)
    // bf16: lane t=pl of a 16-lane group addresses pixel row 8g + (pl>>2) (+4 for the second half), channels 4*(pl&3)..+3
    // f32 : lane reads pixel rows 4g + s (s < 4), channel pl
    int aoff[FM][ES == 2 ? 2 : 4], boff[FN][ES == 2 ? 2 : 4];
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        if (ES == 2) {
            const int c = wa * WTA + a * 16 + 4 * (pl & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * g + (pl >> 2) + 4 * h;
                aoff[a][h] = r * RBA + wg_swz<T>(c >> 3, r, UA) * 16 + (c & 7) * 2;
            }
        } else {
            const int c = wa * WTA + a * 16 + pl;
#pragma unroll
            for (int s = 0; s < (ES == 2 ? 2 : 4); ++s) {
                const int r = 4 * g + s;
                aoff[a][s] = r * RBA + wg_swz<T>(c >> 2, r, UA) * 16 + (c & 3) * 4;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        if (ES == 2) {
            const int c = wb * WTB + b * 16 + 4 * (pl & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * g + (pl >> 2) + 4 * h;
                boff[b][h] = KP * RBA + r * RBB + wg_swz<T>(c >> 3, r, UB) * 16 + (c & 7) * 2;
            }
        } else {
            const int c = wb * WTB + b * 16 + pl;
#pragma unroll
            for (int s = 0; s < (ES == 2 ? 2 : 4); ++s) {
                const int r = 4 * g + s;
                boff[b][s] = KP * RBA + r * RBB + wg_swz<T>(c >> 2, r, UB) * 16 + (c & 3) * 4;
            }
        }
    }
    auto frag = [&](const char* sb, const int* off) -> uint4 {
        if (ES == 2) {
            typedef __attribute__((address_space(3))) v4i16_t* lp_t;
            const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(sb + off[0]));
            const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(sb + off[1]));
            const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
            return make_uint4(l.x, l.y, h.x, h.y);
        }
        return make_uint4(*reinterpret_cast<const uint32_t*>(sb + off[0]), *reinterpret_cast<const uint32_t*>(sb + off[1]),
                          *reinterpret_cast<const uint32_t*>(sb + off[ES == 2 ? 0 : 2]), *reinterpret_cast<const uint32_t*>(sb + off[ES == 2 ? 1 : 3]));
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nsteps) issue(st, st);
        constexpr int NM = FM * FN;
        // the DMA pieces of the stage NST-1 steps ahead go out between the MFMAs (see conv_igemm_dma_kernel)
        auto kstep = [&](auto issue_tag, int s) {
            constexpr bool ISSUE = decltype(issue_tag)::value;
            const bool live = (ABL & 2) ? p.N < 0 : true;
            const char* sb = smem + (s & (NST - 1)) * STAGE;
            uint4 af[FM], bf[FN];
            if (!(ABL & 4)) {
#pragma unroll
                for (int a = 0; a < FM; ++a) af[a] = frag(sb, aoff[a]);
#pragma unroll
                for (int b = 0; b < FN; ++b) bf[b] = frag(sb, boff[b]);
                if (ABL & 1) {
#pragma unroll
                    for (int a = 0; a < FM; ++a) asm volatile("" ::"v"(af[a].x), "v"(af[a].y), "v"(af[a].z), "v"(af[a].w));
#pragma unroll
                    for (int b = 0; b < FN; ++b) asm volatile("" ::"v"(bf[b].x), "v"(bf[b].y), "v"(bf[b].z), "v"(bf[b].w));
                }
            }
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    if (ISSUE) {
#pragma unroll
                        for (int i = 0; i < LPW; ++i)
                            if ((i * NM) / LPW == a * FN + b && live) piece(i, s + NST - 1, (s + NST - 1) & (NST - 1));
                    }
                    if (!(ABL & 5)) Mma<T>::run(af[a], bf[b], acc[a][b]);
                }
        };
        const int nmain = nsteps - (NST - 1);
        int s = 0;
        for (; s < nmain; ++s) {
            wait_vmcnt<(NST - 2) * LPW>();
            raw_barrier();
            kstep(std::true_type{}, s);
        }
        for (; s < nsteps; ++s) {
            if (s + NST - 2 <= nsteps - 1) wait_vmcnt<(NST - 2) * LPW>();
            else wait_vmcnt<0>();
            raw_barrier();
            kstep(std::false_type{}, s);
        }
        const long taps = (long)p.KH * p.KW;
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ia = a0 + wa * WTA + a * 16 + 4 * g + r;
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    const int ib = b0 + wb * WTB + b * 16 + pl;
                    if (ABL & 32) asm volatile("" ::"v"(acc[a][b][r]));
                    else if (ia < p.A && ib < p.B) wgrad_emit(p, split, (ia * taps + tap) * p.B + ib, acc[a][b][r]);
                }
            }
    }
}

// Three taps per block (3x3, stride 1, pad 1, 16-bit): the P rows (the output gradient) of a k-step do not depend on the tap,
// and the Q rows of the taps (kh, 0..2) are the same image-row segment shifted by one pixel -- so the segment is staged ONCE
// with a one-pixel halo on both sides (row q = k + kw + 2*(k / sw), sw = pixels per image-row segment of a k-step) and the
// three taps read it at row offsets 0, 1, 2: per k-step 32 P rows + 34..40 Q rows feed three tile products instead of
// 3 x (32 + 32) rows feeding them one by one (2.9x fewer bytes through the L2 -> LDS fill that bounds the single-tap kernel,
// and the P fragments are read from LDS once for the three).  A block accumulates 3 tiles; the kernel rows (kh) stay separate
// blocks.  Geometry: Wp a multiple of 32, or 16, or 8 (a k-step is 32 consecutive pixels = one segment, 2 or 4 image rows).
// Round 4, the same for ConvTranspose2d(k4, s2, p1) (NT = 4, QS = 2): P is the coarse input, Q the fine output gradient, tap (kh, kw) of
// coarse pixel (y, x) reads fine pixel (2y - 1 + kh, 2x - 1 + kw) -- the four kw taps of a run of sw coarse pixels read ONE fine row
// segment of 2 sw + 2 pixels at rows 2 j + kw.  Staged once it feeds four products: 32 P rows + 66-72 Q rows per k-step instead of 4 x
// (32 + 32), and the single-tap form these layers ran in is bound by exactly those bytes (fabric reads, section 3 of DESIGN.md).
//   general: NT taps, Q row stride QS; a segment holds QS (sw - 1) + NT rows; LDS row of (pixel r, tap kw) = seg SEGROWS + QS (r % sw) + kw
template <typename T, int TA, int TB, int NST, int NWV = 4, int NT = 3, int QS = 1>
__device__ __forceinline__ void wgrad3_dma_body(const WgK& p, const int orig, const int nwg) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int ES = 2, KP = 32;
    constexpr int RBA = TA * ES, RBB = TB * ES;
    constexpr int UA = RBA / 16, UB = RBB / 16;
    constexpr int RPA = 1024 / RBA, RPB = 1024 / RBB;            // pixel rows per DMA wave-instruction
    constexpr int NIA = KP * RBA / 1024;
    constexpr int NWA = NWV / 2;                                    // waves along the A (output-gradient channel) dimension x 2 along B
    constexpr int IA = (NIA + NWV - 1) / NWV;
    constexpr int QRMAX = (KP / 8) * (QS * 7 + NT);                // sw = 8: four segments of QS (sw - 1) + NT rows
    constexpr int IB = ((QRMAX + RPB - 1) / RPB + NWV - 1) / NWV;  // Q wave-instructions per wave (uniform; rows past the image: out of range)
    constexpr int QROWS = IB * NWV * RPB;                          // rows the Q part of a stage holds
    constexpr int STAGE = KP * RBA + QROWS * RBB;
    constexpr int LPW = IA + IB;
    constexpr int WTA = TA / NWA, WTB = TB / 2;
    constexpr int FM = WTA / 16, FN = WTB / 16;
    static_assert(NIA >= 2 && (NIA >= NWV || NWV == 4), "tile too small");
    static_assert(NST * STAGE <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wid >> 1, wb = wid & 1;
    const int g = lane >> 4, pl = lane & 15;
    int tile, kh, split;
    wgrad_block(p, orig, nwg, tile, kh, split);
    const int ta = tile / p.tiles_b, tb = tile - ta * p.tiles_b;
    const int a0 = ta * TA, b0 = tb * TB;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + KP - 1) / KP : 0;
    const int sw = p.sw, seg_rows = QS * (sw - 1) + NT;
    const int qr_used = (KP / sw) * seg_rows;

    const u32x4_t rp = make_srd(p.p, p.p_bytes);
    const u32x4_t rq = make_srd(p.q, p.q_bytes);
    const unsigned ppix = (unsigned)p.p_ld * ES, qpix = (unsigned)p.q_ld * ES;
    const unsigned hw = (unsigned)(p.Hp * p.Wp);

    int prow[IA];
    unsigned pcol[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int j = NIA >= NWV ? i * NWV + wid : (wid & (NIA - 1));
        prow[i] = j * RPA + lane / UA;
        pcol[i] = (unsigned)a0 * ES + (unsigned)wg_swz<T>(lane % UA, prow[i], UA) * 16u;
    }
    // Q rows: LDS row q of the stage = segment q / (sw+2), position q % (sw+2) - 1 in [-1, sw] relative to the segment's first
    // pixel; (n, y, x) of that first pixel advance by one k-step (32 pixels) per stage
    unsigned qcol[IB];
    int qxi[IB], qn_[IB], qy_[IB], qx_[IB];
    bool qlive[IB];
    const int dxs = KP % p.Wp, dys = KP / p.Wp;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int q = (i * NWV + wid) * RPB + lane / UB;
        const int sg = q / seg_rows;
        qlive[i] = q < qr_used;
        qxi[i] = q - sg * seg_rows - 1;
        qcol[i] = (unsigned)b0 * ES + (unsigned)wg_swz<T, QS>(lane % UB, q, UB) * 16u;
        const unsigned m = (unsigned)(mbeg + (qlive[i] ? sg * sw : 0));
        const unsigned n = udiv_rcp(m, hw, p.rcp_hw);
        const unsigned rem = m - n * hw;
        const unsigned y = udiv_rcp(rem, (unsigned)p.Wp, p.rcp_w);
        qn_[i] = (int)n; qy_[i] = (int)y; qx_[i] = (int)(rem - y * (unsigned)p.Wp);
    }
    auto piece = [&](int i, int s, int stage) {
        const int mb = mbeg + s * KP;
        char* sp = smem + stage * STAGE;
        char* sq = sp + KP * RBA;
        if (i < IA) {
            const int ii = i < IA ? i : 0;
            const int m = mb + prow[ii];
            const unsigned off = m < mend ? (unsigned)m * ppix + pcol[ii] : OOB_OFF;
            dma16(rp, sp + (NIA >= NWV ? ii * NWV + wid : (wid & (NIA - 1))) * 1024, off, 0);
        } else {
            const int ii = i >= IA ? i - IA : 0;
            int n = qn_[ii], y = qy_[ii], x = qx_[ii];
            unsigned off = OOB_OFF;
            const int iy = QS * y - 1 + kh, ix = QS * x + qxi[ii];
            if (qlive[ii] && mb < mend && (unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq)
                off = (unsigned)((n * p.Hq + iy) * p.Wq + ix) * qpix + qcol[ii];
            dma16(rq, sq + (ii * NWV + wid) * 1024, off, 0);
            x += dxs;
            if (x >= p.Wp) { x -= p.Wp; ++y; }
            y += dys;
            if (y >= p.Hp) { y -= p.Hp; ++n; }
            qn_[ii] = n; qy_[ii] = y; qx_[ii] = x;
        }
    };
    auto issue = [&](int s, int stage) {
#pragma unroll
        for (int i = 0; i < LPW; ++i) piece(i, s, stage);
    };

    // fragment read offsets (bytes within a stage): lane t=pl of a 16-lane group addresses pixel row 8g + (pl>>2) (+4 for the
    // second half), channels 4*(pl&3)..+3; Q rows additionally shifted by the tap and the halo rows of the segments before
    int aoff[FM][2], boff[NT][FN][2];
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        const int c = wa * WTA + a * 16 + 4 * (pl & 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 8 * g + (pl >> 2) + 4 * h;
            aoff[a][h] = r * RBA + wg_swz<T>(c >> 3, r, UA) * 16 + (c & 7) * 2;
        }
    }
#pragma unroll
    for (int kw = 0; kw < NT; ++kw)
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int c = wb * WTB + b * 16 + 4 * (pl & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * g + (pl >> 2) + 4 * h;
                const int q = (r / sw) * seg_rows + QS * (r % sw) + kw;
                boff[kw][b][h] = KP * RBA + q * RBB + wg_swz<T, QS>(c >> 3, q, UB) * 16 + (c & 7) * 2;
            }
        }
    auto frag = [&](const char* sb, const int* off) -> uint4 {
        typedef __attribute__((address_space(3))) v4i16_t* lp_t;
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(sb + off[0]));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(sb + off[1]));
        const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
        return make_uint4(l.x, l.y, h.x, h.y);
    };

    f32x4 acc[NT][FM][FN];
#pragma unroll
    for (int kw = 0; kw < NT; ++kw)
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b) acc[kw][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st)
            if (st < nsteps) issue(st, st);
        constexpr int NM = NT * FM * FN;
        auto kstep = [&](auto issue_tag, int s) {
            constexpr bool ISSUE = decltype(issue_tag)::value;
            const char* sb = smem + (s & (NST - 1)) * STAGE;
            uint4 af[FM];
#pragma unroll
            for (int a = 0; a < FM; ++a) af[a] = frag(sb, aoff[a]);
#pragma unroll
            for (int kw = 0; kw < NT; ++kw) {
                uint4 bf[FN];
#pragma unroll
                for (int b = 0; b < FN; ++b) bf[b] = frag(sb, boff[kw][b]);
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        if (ISSUE) {
#pragma unroll
                            for (int i = 0; i < LPW; ++i)
                                if ((i * NM) / LPW == (kw * FM + a) * FN + b) piece(i, s + NST - 1, (s + NST - 1) & (NST - 1));
                        }
                        Mma<T>::run(af[a], bf[b], acc[kw][a][b]);
                    }
            }
        };
        const int nmain = nsteps - (NST - 1);
        int s = 0;
        for (; s < nmain; ++s) {
            wait_vmcnt<(NST - 2) * LPW>();
            raw_barrier();
            kstep(std::true_type{}, s);
        }
        for (; s < nsteps; ++s) {
            if (s + NST - 2 <= nsteps - 1) wait_vmcnt<(NST - 2) * LPW>();
            else wait_vmcnt<0>();
            raw_barrier();
            kstep(std::false_type{}, s);
        }
#pragma unroll
        for (int kw = 0; kw < NT; ++kw)
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ia = a0 + wa * WTA + a * 16 + 4 * g + r;
#pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        const int ib = b0 + wb * WTB + b * 16 + pl;
                        if (ia < p.A && ib < p.B) wgrad_emit(p, split, (ia * (long)(p.KH * NT) + kh * NT + kw) * p.B + ib, acc[kw][a][b][r]);
                    }
                }
    }
}

template <typename T, int TA, int TB, int NST, int NWV = 4, int NT = 3, int QS = 1>
__global__ __launch_bounds__(NWV * 64) void conv_wgrad3_dma_kernel(WgK p) {
    if constexpr (sizeof(T) == 2) wgrad3_dma_body<T, TA, TB, NST, NWV, NT, QS>(p, blockIdx.x, gridDim.x);
}

template <typename T, int TA, int TB, int NST, int ABL = 0>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(WgK p) {
    wgrad_dma_body<T, TA, TB, NST, ABL>(p, blockIdx.x, gridDim.x);
}

// Several weight-gradient problems of one tile shape in a single launch (msc_wgrad_group_*): the layers of a
// ResNet stage are too small to fill 256 CUs one at a time, together they do.  `blk` holds one (problem, block of that problem)
// pair per workgroup (msc_wgrad_group_create / wgrad_place); problem -1 marks padding.
__device__ __forceinline__ bool wgrad_group_fetch(const WgK* __restrict__ tab, const int2* __restrict__ blk, WgK& p, int& orig) {
    const int2 e = blk[blockIdx.x];
    const int i = __builtin_amdgcn_readfirstlane(e.x);
    orig = __builtin_amdgcn_readfirstlane(e.y);
    if (i < 0) return false;
    const int* src = reinterpret_cast<const int*>(tab + i);
    int* dst = reinterpret_cast<int*>(&p);
#pragma unroll
    for (unsigned j = 0; j < sizeof(WgK) / 4; ++j) dst[j] = __builtin_amdgcn_readfirstlane(src[j]);
    return orig < p.nblocks;
}

template <typename T, int TA, int TB, int NST, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void conv_wgrad_group_kernel(const WgK* __restrict__ tab, const int2* __restrict__ blk) {
    if constexpr (TA <= 128 || sizeof(T) == 2) {      // the 256x256 tile exists for the 16-bit types
        WgK p;
        int orig;
        if (!wgrad_group_fetch(tab, blk, p, orig)) return;
        wgrad_dma_body<T, TA, TB, NST, 0, NWV>(p, orig, p.nblocks);
    }
}

template <typename T, int TA, int TB, int NST, int NWV = 4, int NT = 3, int QS = 1>
__global__ __launch_bounds__(NWV * 64) void conv_wgrad3_group_kernel(const WgK* __restrict__ tab, const int2* __restrict__ blk) {
    if constexpr (sizeof(T) == 2) {
        WgK p;
        int orig;
        if (!wgrad_group_fetch(tab, blk, p, orig)) return;
        wgrad3_dma_body<T, TA, TB, NST, NWV, NT, QS>(p, orig, p.nblocks);
    }
}

// Ordered groups: dw += plane 0 + plane 1 + ... in that order, one thread per four consecutive gradient elements.  `items` lists the
// layers, `blk` (item, first element) per workgroup of 1024 elements.  The sum no longer depends on which block finished first:
// with it (and msc_final_bwd's ordered workspace) a training step is reproducible bit for bit.
struct WgFin { float* dw; const float* part; long plane; int splits, pad_; };

__global__ __launch_bounds__(256) void wgrad_finish_kernel(const WgFin* __restrict__ items, const int2* __restrict__ blk) {
    const int2 e = blk[blockIdx.x];
    const WgFin it = items[e.x];
    const long i = (long)e.y * 1024 + threadIdx.x * 4;
    if (i >= it.plane) return;
    float4 s = *reinterpret_cast<const float4*>(it.part + i);
    int k = 1;
    for (; k + 4 <= it.splits; k += 4) {      // four planes in flight, added in plane order (a 512-split layer is a chain of loads otherwise)
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(it.part + (long)(k + j) * it.plane + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; k < it.splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(it.part + (long)k * it.plane + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* d = it.dw + i;
    if (((uintptr_t)it.dw & 15) == 0) {
        float4 o = *reinterpret_cast<const float4*>(d);
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *reinterpret_cast<float4*>(d) = o;
    } else {      // a gradient range that does not start on 16 bytes
        d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w;
    }
}

}  // namespace

extern "C" int msc_conv_wgrad_num_cfgs(void) { return WGRAD_NCFG; }

namespace {

struct WgPlan { WgK k; int dtype, ta, tb; int kw3; };

// Validates a descriptor and fixes tile shape and split-K.  steps_per_block > 0 (grouped launches: other problems
// fill the chip, so a block just runs that many k-steps) overrides the per-launch policy selected by d->cfg.
int wgrad_plan(const msc_wgrad_desc* d, int steps_per_block, int tile_cap, WgPlan* out) {
    if (!d || !d->p || !d->q || !d->dw) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: null pointer");
    if (!msc_dtype_ok(d->dtype)) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: dtype %d", d->dtype);
    const int es = msc_dtype_size(d->dtype);
    if (d->A % 32 || d->B % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_wgrad: channel counts must be multiples of 32 (A=%d B=%d)", d->A, d->B);
    const bool q_ok = (d->q_ld * es) % 16 == 0 ||
                      (d->KW == 1 && d->pad == 0 && (d->stride * d->q_ld * es) % 16 == 0 && ((int64_t)d->Wq * d->q_ld * es) % 16 == 0);
    if ((d->p_ld * es) % 16 || !q_ok || (((uintptr_t)d->p | (uintptr_t)d->q) & 15))
        return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: operands must keep 16-byte alignment");
    // Narrow compact Q operand (32 channels = 64-byte rows): the KW taps of a kernel row read KW consecutive pixels = KW*B contiguous
    // elements and write KW*B contiguous gradient columns ([A][KH][KW][B]) -- run it as KW' = 1, B' = KW*B (256-byte Q rows, the
    // 128x128 tile instead of 64x32); the kernel bounds-checks every lane against the pixel its 16 bytes belong to.
    msc_wgrad_desc merged = *d;
    int span_bytes = 0;
    static const bool merge_on = [] { const char* e = getenv("MSC_CONV_MERGE_KW"); return !(e && e[0] == '0'); }();
    const long m_all = (long)d->N * d->Hp * d->Wp;
    const bool dma_ok = m_all > 0 && m_all < (1L << 24) && ((m_all - 1) * d->p_ld + d->A) * es < 0x7fffffffL &&
                        (((long)d->N * d->Hq * d->Wq - 1) * d->q_ld + (long)d->KW * d->B) * es < 0x7fffffffL;      // the merged row still fits (see `fits` below)
    if (merge_on && dma_ok && d->KW > 1 && d->q_ld == d->B && (long)d->KW * d->B * es == 256 && (d->B * es) % 16 == 0) {
        span_bytes = d->B * es;
        merged.B = d->KW * d->B;
        merged.KW = 1;
        d = &merged;
    }
    WgK& k = out->k;
    k.span_bytes = span_bytes;
    static const bool direct_off = [] { const char* e = getenv("MSC_WGRAD_DIRECT"); return e && e[0] == '0'; }();
    k.no_direct = direct_off ? 1 : 0;
    k.pad_ = 0; k.part = nullptr; k.plane = 0;
    k.p = (const char*)d->p; k.q = (const char*)d->q; k.dw = d->dw; k.p_ld = d->p_ld; k.q_ld = d->q_ld;
    k.N = d->N; k.Hp = d->Hp; k.Wp = d->Wp; k.A = d->A; k.Hq = d->Hq; k.Wq = d->Wq; k.B = d->B;
    k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad;
    const long m = (long)d->N * d->Hp * d->Wp;
    if (m <= 0 || m > 0x7fffffffL) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: bad pixel count %ld", m);
    k.M = (int)m;
    const long p_b = ((m - 1) * d->p_ld + d->A) * es;
    const long q_b = (((long)d->N * d->Hq * d->Wq - 1) * d->q_ld + d->B) * es;
    // 31-bit buffer offsets; the float-reciprocal pixel decode is exact below 2^24 pixels (callers hand over image ranges that
    // fit: wgrad_image_chunk)
    const bool fits = p_b < 0x7fffffffL && q_b < 0x7fffffffL && m < (1L << 24);
    if (!fits) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_wgrad: one image is beyond 2 GiB / 2^24 pixels (%ld pixels, %ld / %ld bytes)", m, p_b, q_b);
    k.p_bytes = (unsigned)p_b;
    k.q_bytes = (unsigned)q_b;
    k.rcp_hw = 1.0f / (float)(d->Hp * d->Wp);
    k.rcp_w = 1.0f / (float)d->Wp;
    const int kp = 64 / es;
    const int ksteps = ceil_div(k.M, kp);
    // three taps of a kernel row per block (wgrad3_dma_body): 3x3 / stride 1 / pad 1 in a 16-bit type, image rows that a
    // 32-pixel k-step covers in whole segments
    static const bool kw3_on = [] { const char* e = getenv("MSC_WGRAD_KW3"); return !(e && e[0] == '0'); }();
    const bool kw3 = kw3_on && fits && es == 2 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Hq == d->Hp &&
                     d->Wq == d->Wp && (d->Wp % 32 == 0 || d->Wp == 16 || d->Wp == 8) && d->Hp >= 8 && k.M % 32 == 0;
    // ... and the four kw taps of a ConvTranspose2d(k4, s2, p1) kernel row (round 4).  MEASURED SLOWER and therefore opt-in (MSC_WGRAD_KW4=1;
    // profiles/r4_run12_wgrad_kw4_ab.txt): grouped launches 2.04-2.08 -> 2.12-2.16 ms, also with the stride-2 swizzle key -- at 128x128
    // four accumulator sets leave one block of 228 registers and 128 KB of LDS per CU, where the single-tap form runs two blocks.
    static const bool kw4_on = [] { const char* e = getenv("MSC_WGRAD_KW4"); return e && e[0] == '1'; }();
    const bool kw4 = kw4_on && fits && es == 2 && !span_bytes && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad == 1 && d->Hq == 2 * d->Hp &&
                     d->Wq == 2 * d->Wp && (d->Wp % 32 == 0 || d->Wp == 16 || d->Wp == 8) && d->Hp >= 8 && k.M % 32 == 0;
    const int ntaps = (kw3 || kw4) ? d->KH : d->KH * d->KW;
    k.kw3 = kw3 ? 3 : kw4 ? 4 : 0;
    k.sw = d->Wp < 32 ? d->Wp : 32;
    bool big = (d->A % 128 == 0) && (d->B % 128 == 0);
    // 256x256 tiles (grouped launches, 16-bit, single-tap problems whose channel counts allow it -- ResNet101's layer3 / layer4 1x1
    // convolutions, the 512 -> 256 ConvTranspose2d layers): half the operand bytes per MFMA of the 128x128 tile; eight waves, 128
    // accumulator registers each, one block per CU, half the k-steps per block.  MEASURED NEUTRAL (round 4, profiles/
    // r4_run4_wgrad_t256_ab.txt): the 284 GFLOP that move into this bucket take 499 us against ~470 us in the 128x128 one -- the
    // single-tap kernels are bound by what they fetch from BEYOND the L2 (5 TB/s of fabric reads, 2.7 us of queueing latency against
    // 48 KB in flight per block), and with 4 tiles x 2-4 splits a problem has one block per XCD: every slab is still fetched once per
    // block.  Opt-in: MSC_WGRAD_T256=1.
    static const bool t256_on = [] { const char* e = getenv("MSC_WGRAD_T256"); return e && e[0] == '1'; }();
    bool huge = false;
    int tsel = 0, splits;
    if (steps_per_block > 0) {
        if (tile_cap < 128) big = false;
        if (tile_cap < 64) tsel = 2;
        huge = t256_on && big && fits && es == 2 && !kw3 && !kw4 && d->A % 256 == 0 && d->B % 256 == 0;
        splits = ceil_div(ksteps, huge ? (steps_per_block + 1) / 2 : steps_per_block);
    } else {
        // Tile and split-K choice.  Every split adds one fp32-atomic pass over dW and a pipeline fill, so a block
        // should run >= 24 k-steps; within that, prefer the largest tile that still gives >= 384 blocks.
        const int max_splits = ksteps / 24 > 1 ? ksteps / 24 : 1;
        // cfg 0: heuristic; else 1 + tsel*5 + ssel with tsel 0 = 128x128 tiles when possible, 1 = 64x64 at most,
        // 2 = 32 output channels x 64 at most; ssel = index into the target block counts below (last: no split-K)
        static const int TARGETS[WGRAD_NS] = {256, 512, 1024, 2048, 1};
        if (d->cfg < 0 || d->cfg > WGRAD_NCFG) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: cfg %d", d->cfg);
        tsel = d->cfg ? (d->cfg - 1) / WGRAD_NS : 0;
        const int target = d->cfg ? TARGETS[(d->cfg - 1) % WGRAD_NS] : 768;
        if (d->cfg) {
            if (tsel) big = false;
        } else if (big) {
            const int tiles128 = (d->A / 128) * (d->B / 128) * ntaps;
            int sp = ceil_div(768, tiles128);
            if (sp > max_splits) sp = max_splits;
            if ((long)tiles128 * sp < 384) big = false;     // not enough parallelism: 64x64 tiles instead
        }
        const int ta0 = big ? 128 : (d->A % 64 == 0 && tsel != 2 ? 64 : 32), tb0 = big ? 128 : (d->B % 64 == 0 ? 64 : 32);
        splits = ceil_div(target, (d->A / ta0) * (d->B / tb0) * ntaps);
        if (splits > max_splits) splits = max_splits;
    }
    const int ta = huge ? 256 : big ? 128 : (d->A % 64 == 0 && tsel != 2 ? 64 : 32), tbs = huge ? 256 : big ? 128 : (d->B % 64 == 0 ? 64 : 32);
    if (splits < 1) splits = 1;
    int mchunk = ceil_div(k.M, splits);
    mchunk = ceil_div(mchunk, kp) * kp;
    splits = ceil_div(k.M, mchunk);
    k.mchunk = mchunk; k.tiles_b = d->B / tbs;
    k.ntiles = (d->A / ta) * (d->B / tbs); k.ntaps = ntaps; k.xcd_order = xcd_order_enabled() ? 1 : 0;
    k.nblocks = k.ntiles * k.ntaps * splits;
    out->dtype = d->dtype; out->ta = ta; out->tb = tbs;
    out->kw3 = k.kw3;
    // 128x128 tiles: with 4 waves three accumulator sets leave one wave per SIMD and one block per CU -- measured 15 % slower
    // than the single-tap blocks at two blocks per CU (859 vs 745 us for the 3x3 layers of the ResNet101 step); the 8-wave form
    // (two waves per SIMD, 96 accumulator registers) was neutral in round 2 and is 4 % ahead since the epilogue / prologue work of
    // round 3 (grouped launches 2.14-2.16 -> 2.05-2.06 ms, two A/B pairs on one box): the default.  MSC_WGRAD_KW3=1 keeps the
    // 128x128 tiles single-tap.
    static const bool kw3_big = [] { const char* e = getenv("MSC_WGRAD_KW3"); return !(e && e[0] == '1'); }();
    if (kw3 && ta == 128 && !kw3_big) {
        out->kw3 = 0;
        k.kw3 = 0;
        k.ntaps = d->KH * d->KW;
        k.nblocks = k.ntiles * k.ntaps * splits;
    }
    return MSC_OK;
}

// calls f.template operator()<T, TA, TB>() for the plan's (dtype, tile)
template <typename F>
void wgrad_tile_dispatch(int dtype, int ta, int tb, F&& f) {
#define MSC_WG_CASE(T) \
    if (ta == 256) f.template operator()<T, 256, 256>(); \
    else if (ta == 128) f.template operator()<T, 128, 128>(); \
    else if (ta == 64 && tb == 64) f.template operator()<T, 64, 64>(); \
    else if (ta == 64) f.template operator()<T, 64, 32>(); \
    else if (tb == 64) f.template operator()<T, 32, 64>(); \
    else f.template operator()<T, 32, 32>();
    if (dtype == MSC_F16) { MSC_WG_CASE(f16_t) } else if (dtype == MSC_BF16) { MSC_WG_CASE(bf16_t) } else { MSC_WG_CASE(float) }
#undef MSC_WG_CASE
}

struct WgLaunchOne {
    const WgK& k; hipStream_t st;
    template <typename T, int TA, int TB> void operator()() const {
        if constexpr (TA <= 128) {      // (256x256 tiles are planned for grouped launches only)
            if (k.kw3 == 4) {
                if constexpr (TA == 128) hipLaunchKernelGGL((conv_wgrad3_dma_kernel<T, TA, TB, 4, 8, 4, 2>), dim3(k.nblocks), dim3(512), 0, st, k);
                else hipLaunchKernelGGL((conv_wgrad3_dma_kernel<T, TA, TB, 4, 4, 4, 2>), dim3(k.nblocks), dim3(256), 0, st, k);
            } else if (k.kw3) {
                if constexpr (TA == 128) hipLaunchKernelGGL((conv_wgrad3_dma_kernel<T, TA, TB, 4, 8>), dim3(k.nblocks), dim3(512), 0, st, k);
                else hipLaunchKernelGGL((conv_wgrad3_dma_kernel<T, TA, TB, 4>), dim3(k.nblocks), dim3(256), 0, st, k);
            }
            else hipLaunchKernelGGL((conv_wgrad_dma_kernel<T, TA, TB, 4>), dim3(k.nblocks), dim3(256), 0, st, k);
        }
    }
};

struct WgLaunchGroup {
    const WgK* tab; const int2* blk; int blocks; hipStream_t st; int kw3;
    template <typename T, int TA, int TB> void operator()() const {
        if (kw3 == 4) {
            if constexpr (TA == 128) hipLaunchKernelGGL((conv_wgrad3_group_kernel<T, TA, TB, 4, 8, 4, 2>), dim3(blocks), dim3(512), 0, st, tab, blk);
            else if constexpr (TA < 128) hipLaunchKernelGGL((conv_wgrad3_group_kernel<T, TA, TB, 4, 4, 4, 2>), dim3(blocks), dim3(256), 0, st, tab, blk);
        } else if (kw3) {
            if constexpr (TA == 128) hipLaunchKernelGGL((conv_wgrad3_group_kernel<T, TA, TB, 4, 8>), dim3(blocks), dim3(512), 0, st, tab, blk);
            else if constexpr (TA < 128) hipLaunchKernelGGL((conv_wgrad3_group_kernel<T, TA, TB, 4>), dim3(blocks), dim3(256), 0, st, tab, blk);
        }
        else if constexpr (TA == 256) hipLaunchKernelGGL((conv_wgrad_group_kernel<T, TA, TB, 4, 8>), dim3(blocks), dim3(512), 0, st, tab, blk);
        else hipLaunchKernelGGL((conv_wgrad_group_kernel<T, TA, TB, 4>), dim3(blocks), dim3(256), 0, st, tab, blk);
    }
};

// Images per problem (see conv_image_chunk): both operands within 31-bit byte offsets and fewer than 2^24 pixels; the weight
// gradient is a sum over images, accumulated atomically, so image ranges are independent problems.
int wgrad_image_chunk(const msc_wgrad_desc* d) {
    if (!d || d->N <= 0 || !msc_dtype_ok(d->dtype)) return 1;
    const long es = msc_dtype_size(d->dtype);
    const long pp = (long)d->Hp * d->Wp, per_p = pp * d->p_ld * es, per_q = (long)d->Hq * d->Wq * d->q_ld * es;
    if (pp <= 0 || per_p <= 0 || per_q <= 0) return d->N;
    long n = d->N;
    if (pp * n >= (1L << 24)) n = ((1L << 24) - 1) / pp;
    if (per_p * n >= 0x7fff0000L) n = 0x7fff0000L / per_p;
    if (per_q * n >= 0x7fff0000L) n = 0x7fff0000L / per_q;
    return n < 1 ? 1 : (int)n;
}

msc_wgrad_desc wgrad_image_range(const msc_wgrad_desc* d, int n0, int n) {
    msc_wgrad_desc c = *d;
    const long es = msc_dtype_size(d->dtype);
    c.N = n;
    c.p = (const char*)d->p + (long)n0 * d->Hp * d->Wp * d->p_ld * es;
    c.q = (const char*)d->q + (long)n0 * d->Hq * d->Wq * d->q_ld * es;
    return c;
}

}  // namespace

extern "C" int msc_conv_wgrad(const msc_wgrad_desc* d, void* stream) {
    if (!d) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: null descriptor");
    const int chunk = wgrad_image_chunk(d);
    for (int n0 = 0; n0 < (d->N > 0 ? d->N : 1); n0 += chunk) {
        const msc_wgrad_desc part = wgrad_image_range(d, n0, d->N - n0 < chunk ? d->N - n0 : chunk);
        WgPlan pl;
        int rc = wgrad_plan(&part, 0, 128, &pl);
        if (rc != MSC_OK) return rc;
        wgrad_tile_dispatch(pl.dtype, pl.ta, pl.tb, WgLaunchOne{pl.k, (hipStream_t)stream});
    }
    return msc_check_launch("conv_wgrad");
}

// ---- grouped weight gradients ---------------------------------------------------------------------
struct msc_wgrad_group {
    struct Bucket { int dtype, ta, tb, n, blocks; WgK* tab; int2* blk; int kw3; };
    std::vector<Bucket> buckets;      // problems by (dtype, tile): one launch each
    void* dev = nullptr;              // one allocation behind every table
    // MSC_WGRAD_ORDERED: the split planes, and the finish launch that sums them into the gradients
    void* planes = nullptr;
    const void* fin_items = nullptr;
    const int2* fin_blk = nullptr;
    int fin_blocks = 0;
};

namespace {

// Which workgroup of a grouped launch runs which block: problem after problem (longest blocks first), each padded to a multiple
// of 8 workgroups so that its XCD-local order (wgrad_block: workgroup b runs on XCD b % 8, every XCD gets a contiguous run of the
// problem's split-major block order) is the one a launch of its own would have.
// Measured and NOT kept (round 4, profiles/r4_run1_wgrad_place_ab.txt): placing all blocks that read one pixel range of a problem on
// ONE XCD (whole small problems per XCD instead of four blocks on each of the eight).  The PMC traffic of these launches is 2.8x
// their operands (7.4 GB for 2.6 GB) because every XCD's L2 fetches the slabs its four blocks need, and the placement removes
// exactly those re-fetches -- the grouped launches got SLOWER, 2.06 -> 2.28 ms (queues balanced by cost) and 2.47 ms (queues
// position-balanced by block length): the re-fetches are served by the 256 MB Infinity Cache at a rate that is not the bound, while
// sixteen to forty-eight blocks hammering the same lines of ONE L2 are.  FETCH_SIZE counts fabric requests, not DRAM reads.
void wgrad_place(const std::vector<WgPlan>& plans, const std::vector<int>& members, std::vector<int2>& blk) {
    blk.clear();
    const bool xo = xcd_order_enabled();
    for (size_t j = 0; j < members.size(); ++j) {
        const WgK& k = plans[members[j]].k;
        const int nb8 = (k.nblocks + 7) & ~7;
        for (int o = 0; o < nb8; ++o) {
            int wgid = o;
            if (xo && o < k.nblocks) {
                const int xcd = o & 7, wq = k.nblocks >> 3, wr = k.nblocks & 7;
                wgid = (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (o >> 3);
            }
            blk.push_back(o < k.nblocks ? make_int2((int)j, wgid) : make_int2(-1, 0));
        }
    }
}

}  // namespace

extern "C" int msc_wgrad_group_create(const msc_wgrad_desc* descs, int n, int steps_per_block, int tile_cap, int flags, msc_wgrad_group** out) {
    if (!descs || n <= 0 || !out) return msc_fail(MSC_ERR_ARG, "msc_wgrad_group_create: bad argument");
    if (flags & ~MSC_WGRAD_ORDERED) return msc_fail(MSC_ERR_ARG, "msc_wgrad_group_create: flags 0x%x", flags);
    const bool ordered = (flags & MSC_WGRAD_ORDERED) != 0;
    std::vector<WgPlan> plans;
    plans.reserve(n);
    std::vector<WgFin> fin;            // ordered: one entry per layer ...
    std::vector<long> fin_at, plan_at; // ... and where its planes / the planes of every image range start (floats into the allocation made below)
    long plane_floats = 0;
    for (int i = 0; i < n; ++i) {      // a layer beyond the 31-bit offsets enters the table as several image ranges
        const int chunk = wgrad_image_chunk(&descs[i]);
        if (ordered) {
            for (int j = 0; j < i; ++j)
                if (descs[j].dw == descs[i].dw) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_wgrad_group_create: ordered sums need one descriptor per gradient (%d and %d share one)", j, i);
            fin.push_back(WgFin{descs[i].dw, nullptr, (long)descs[i].A * descs[i].KH * descs[i].KW * descs[i].B, 0, 0});
        }
        for (int n0 = 0; n0 < (descs[i].N > 0 ? descs[i].N : 1); n0 += chunk) {
            const msc_wgrad_desc part = wgrad_image_range(&descs[i], n0, descs[i].N - n0 < chunk ? descs[i].N - n0 : chunk);
            WgPlan pl;
            int rc = wgrad_plan(&part, steps_per_block, tile_cap > 0 ? tile_cap : 128, &pl);
            if (rc != MSC_OK) return rc;
            pl.k.xcd_order = 0;        // the block table below carries the placement: a block's index within its problem is used as it is
            if (ordered) {             // the image ranges of a layer continue its run of planes
                WgFin& f = fin.back();
                pl.k.plane = f.plane;
                plan_at.push_back(plane_floats + (long)f.splits * f.plane);
                f.splits += pl.k.nblocks / (pl.k.ntiles * pl.k.ntaps);
            }
            plans.push_back(pl);
        }
        if (ordered) {
            if (fin.back().splits == 1) {      // one block per gradient tile: it adds to dw itself, no plane
                fin.pop_back();
                plan_at.back() = -1;
            } else {
                fin_at.push_back(plane_floats);
                plane_floats += (long)fin.back().splits * fin.back().plane;
            }
        }
    }
    n = (int)plans.size();
    msc_wgrad_group* g = new msc_wgrad_group;
    if (ordered) {
        // every split writes its whole plane on every run; the memset only covers a split the plan could leave without pixels
        const size_t pb = (size_t)(plane_floats > 0 ? plane_floats : 1) * sizeof(float);
        if (hipMalloc(&g->planes, pb) != hipSuccess || hipMemset(g->planes, 0, pb) != hipSuccess) {
            if (g->planes) (void)hipFree(g->planes);
            delete g;
            return msc_fail(MSC_ERR_HIP, "msc_wgrad_group_create: %ld bytes of split planes", plane_floats * (long)sizeof(float));
        }
        for (size_t i = 0; i < plans.size(); ++i) {
            if (plan_at[i] < 0) plans[i].k.plane = -1;
            else plans[i].k.part = (float*)g->planes + plan_at[i];
        }
        for (size_t i = 0; i < fin.size(); ++i) fin[i].part = (const float*)g->planes + fin_at[i];
    }
    // the longest-running blocks first: the launch ends when its slowest block does
    std::stable_sort(plans.begin(), plans.end(), [](const WgPlan& a, const WgPlan& b) { return a.k.mchunk > b.k.mchunk; });
    std::vector<std::vector<int>> members;
    for (int i = 0; i < n; ++i) {
        const WgPlan& p = plans[i];
        size_t b = 0;
        for (; b < g->buckets.size(); ++b)
            if (g->buckets[b].dtype == p.dtype && g->buckets[b].ta == p.ta && g->buckets[b].tb == p.tb && g->buckets[b].kw3 == p.kw3) break;
        if (b == g->buckets.size()) { g->buckets.push_back({p.dtype, p.ta, p.tb, 0, 0, nullptr, nullptr, p.kw3}); members.emplace_back(); }
        members[b].push_back(i);
    }
    std::vector<std::vector<int2>> tables(g->buckets.size());
    size_t bytes = 0;
    for (size_t b = 0; b < g->buckets.size(); ++b) {
        wgrad_place(plans, members[b], tables[b]);
        bytes += ((members[b].size() * sizeof(WgK) + 255) & ~(size_t)255) + ((tables[b].size() * sizeof(int2) + 255) & ~(size_t)255);
    }
    std::vector<int2> fin_blk;
    for (size_t i = 0; i < fin.size(); ++i)
        for (long e = 0; e < fin[i].plane; e += 1024) fin_blk.push_back(make_int2((int)i, (int)(e / 1024)));
    const size_t fin_off = bytes;
    bytes += ((fin.size() * sizeof(WgFin) + 255) & ~(size_t)255) + ((fin_blk.size() * sizeof(int2) + 255) & ~(size_t)255);
    if (bytes) {
        if (hipMalloc(&g->dev, bytes) != hipSuccess) {
            if (g->planes) (void)hipFree(g->planes);
            delete g;
            return msc_fail(MSC_ERR_HIP, "msc_wgrad_group_create: hipMalloc(%zu)", bytes);
        }
        std::vector<char> host(bytes, 0);
        if (!fin.empty()) {
            memcpy(host.data() + fin_off, fin.data(), fin.size() * sizeof(WgFin));
            const size_t boff = fin_off + ((fin.size() * sizeof(WgFin) + 255) & ~(size_t)255);
            memcpy(host.data() + boff, fin_blk.data(), fin_blk.size() * sizeof(int2));
            g->fin_items = (const char*)g->dev + fin_off;
            g->fin_blk = reinterpret_cast<const int2*>((const char*)g->dev + boff);
            g->fin_blocks = (int)fin_blk.size();
        }
        size_t off = 0;
        for (size_t b = 0; b < g->buckets.size(); ++b) {
            auto& bk = g->buckets[b];
            bk.n = (int)members[b].size();
            bk.tab = reinterpret_cast<WgK*>((char*)g->dev + off);
            WgK* ht = reinterpret_cast<WgK*>(host.data() + off);
            for (int j = 0; j < bk.n; ++j) ht[j] = plans[members[b][j]].k;
            off += ((size_t)bk.n * sizeof(WgK) + 255) & ~(size_t)255;
            bk.blk = reinterpret_cast<int2*>((char*)g->dev + off);
            memcpy(host.data() + off, tables[b].data(), tables[b].size() * sizeof(int2));
            off += (tables[b].size() * sizeof(int2) + 255) & ~(size_t)255;
            bk.blocks = (int)tables[b].size();
        }
        if (hipMemcpy(g->dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(g->dev);
            if (g->planes) (void)hipFree(g->planes);
            delete g;
            return msc_fail(MSC_ERR_HIP, "msc_wgrad_group_create: hipMemcpy");
        }
    }
    *out = g;
    return MSC_OK;
}

extern "C" int msc_wgrad_group_run(const msc_wgrad_group* g, void* stream) {
    if (!g) return msc_fail(MSC_ERR_ARG, "msc_wgrad_group_run: null group");
    hipStream_t st = (hipStream_t)stream;
    for (const auto& bk : g->buckets)
        if (bk.blocks > 0) wgrad_tile_dispatch(bk.dtype, bk.ta, bk.tb, WgLaunchGroup{bk.tab, bk.blk, bk.blocks, st, bk.kw3});
    if (g->fin_blocks > 0)
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(g->fin_blocks), dim3(256), 0, st, reinterpret_cast<const WgFin*>(g->fin_items), g->fin_blk);
    return msc_check_launch("wgrad_group");
}

extern "C" int msc_wgrad_group_run_part(const msc_wgrad_group* g, int part, void* stream) {
    if (!g) return msc_fail(MSC_ERR_ARG, "msc_wgrad_group_run_part: null group");
    const int nb = (int)g->buckets.size();
    if (part < 0 || part >= nb + (g->fin_blocks > 0 ? 1 : 0)) return msc_fail(MSC_ERR_ARG, "msc_wgrad_group_run_part: part %d of %d", part, nb);
    hipStream_t st = (hipStream_t)stream;
    if (part < nb) {
        const auto& bk = g->buckets[part];
        if (bk.blocks > 0) wgrad_tile_dispatch(bk.dtype, bk.ta, bk.tb, WgLaunchGroup{bk.tab, bk.blk, bk.blocks, st, bk.kw3});
    } else {
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3(g->fin_blocks), dim3(256), 0, st, reinterpret_cast<const WgFin*>(g->fin_items), g->fin_blk);
    }
    return msc_check_launch("wgrad_group_part");
}

extern "C" int msc_wgrad_group_launches(const msc_wgrad_group* g) {
    return g ? (int)g->buckets.size() + (g->fin_blocks > 0 ? 1 : 0) : -1;
}

extern "C" void msc_wgrad_group_destroy(msc_wgrad_group* g) {
    if (!g) return;
    if (g->dev) (void)hipFree(g->dev);
    if (g->planes) (void)hipFree(g->planes);
    delete g;
}
