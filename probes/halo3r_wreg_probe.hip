// PROBE (round 6), not part of the library: the 3x3 halo-tile convolution with its weights streamed L2 -> VGPR instead of through an LDS ring.
// Built as configurations 59 / 60 of msc_conv_igemm for one measurement (the dispatch glue was: two constants + declarations in conv_common.h,
// two lines in conv_cfg_ok / conv_dispatch of igemm.hip, this file in the Makefile) and passed the halo kernel's parity test
// (tests/test_gpu_kernels.py::test_halo_tile_kernel_for_3x3_convs_of_any_width with 59 / 60 added to its list: 12 / 12).
// RESULT (profiles/r6_run7_wreg_cfg_table.txt, same box, bf16, batch 32): SLOWER than configuration 42 on every decoder layer --
//   dec1 128->128 @128: 195 vs 189 us;  dec2 320->128 @64: 109 vs 81;  dec3 768->256 @32: 122 vs 87;  dec4 1280->512 @16: 172 vs 110 (cfg 53: 85);
//   the 64-channel form with a 9-deep ring (cfg 60; every weight byte loaded by four waves) 1.7-2.2x slower still.
// One barrier per nine k-steps did not pay for what replaced the ring: 8 waves x 4 KB of weights per k-step as 64-byte row segments through the
// CU's texture addresser (32 KB per k-step against 16 KB by LDS-DMA), two k-steps of prefetch where the L2 answers in more than that, and no less
// LDS traffic (a wave of 128 px x 32 ch reads as many pixel fragments as 64 x 64 read of both kinds).
// Two things found on the way went INTO the product: the conflict-free halo swizzle key (hx & 7; igemm.hip, bottleneck.hip) and the rule that
// asm-loaded registers must stay allocated until the wait that covers the LAST load issued into them (see the end of the kernel).
//
// 3x3 / stride 1 / pad 1 convolution (forward and, with flip, data gradient) for gfx950 with the WEIGHTS STREAMED THROUGH REGISTERS
// (round 6; configurations 59 / 60 of msc_conv_igemm).  Replaces the same nn.Conv2d calls as conv3x3_halo_dma_kernel (igemm.hip):
// the decoder's ConvRelu layers and their data gradients (src/unet_models.py:21-34,373-383) and the 3x3 of the ResNet blocks (:345-371).
//
// conv3x3_halo_dma_kernel keeps the (PH+2) x 18 halo of a 64-channel chunk in LDS for the nine taps, but its weights go through an LDS ring
// that all eight waves share: one s_barrier per k-step (tap), 32 MFMAs apart, and after every barrier all waves read their first fragments at
// once while the matrix pipes idle.  Measured on the decoder layers (tools/conv_cfg_table.py, profiles/r6_run3_conv_cfg_table.txt): every
// tile / ring / taps-per-barrier variant of that kernel ends within 15 % of 1.0-1.4 PFLOP/s; a linear fit over the layers gives 0.77 us per
// k-step against 0.43-0.5 us of MFMA issue and 4.9 us of prologue + epilogue per tile.
//
// Here a block owns the same 16 x 16 pixel patch x TC output channels, but a wave owns WTC = 32 output channels for 256 / WP pixels:
// a weight byte is needed by WP waves only, so the A fragments are loaded L2 -> VGPR directly from the [Cout][3][3][Cin] tensor (64-byte row
// segments, RW - 1 k-steps ahead, inline asm with counted vmcnt), nothing about the weights is shared between waves, and the ONLY barrier
// left is the one per 64-channel chunk that hands over the halo buffer (one per nine k-steps).  LDS holds two halo buffers and nothing else
// (96 KB); LDS traffic per MFMA is the pixel fragments only.  The halo swizzle key is a function of the halo COLUMN, hx & 7 (rows are 9 x 256 bytes
// apart, so they all see the same banks; checked exhaustively: the four lane groups of ds_read_b128 hit 16 distinct 16-byte slots for all three
// column shifts, where the (pixel >> 1) & 7 key of conv3x3_halo_dma_kernel is two-way conflicted for two of them): a fragment's address is
// base(column shift, sub-step) + row x 2304, the row term an immediate.
#include <type_traits>
#include <utility>

#include "common.h"
#include "dma.h"
#include "msc_internal.h"
#include "conv_common.h"

namespace {

using namespace msc_conv;

// one 16 x 32 weight fragment: lane (g, pl) takes 16 bytes (8 k-elements) of its row.  Inline asm for the same reason as dma16: the loads
// stay where they are written and are waited for by OUR counted s_waitcnt, which also counts the LDS-DMA instructions the compiler cannot see.
template <int IMM>
__device__ __forceinline__ void wload(u32x4_t& r, u32x4_t srd, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(srd), "s"(soff), "n"(IMM) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_frags(u32x4_t (&w)[4]) {
    asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : [cnt] "n"(N) : "memory");
}
__device__ __forceinline__ uint4 as_u4(const u32x4_t& v) { return make_uint4(v.x, v.y, v.z, v.w); }

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// PH x 16 pixel patch, TC output channels, WP x WC waves (a wave: PH / WP patch rows x TC / WC = 32 channels), RW = depth of the weight ring
// in registers (k-steps; 9 % RW == 0 so that a tap always uses the same ring slot and no ring register changes its slot across the chunk loop)
template <typename T, int PH, int TC, int WP, int WC, int RW>
__global__ __launch_bounds__(WP * WC * 64) void conv3x3_wreg_kernel(ConvK p) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int ES = 2, KB = 128;
    constexpr int NW = WP * WC;
    constexpr int HCOLS = 18, HPIX = (PH + 2) * HCOLS, ROWB = HCOLS * KB;      // a halo row: 2304 bytes = 9 x 256
    constexpr int NHI = (HPIX + 7) / 8;                       // DMA wave-instructions per halo chunk (8 pixels x 128 B)
    constexpr int XH = (NHI + NW - 1) / NW;                   // ... per wave
    constexpr int HBUF = XH * NW * 1024;
    constexpr int FN = PH / WP, WTC = TC / WC, FM = WTC / 16, NV = FM * 4;
    constexpr int NF = 2 * FM;                                // weight fragments per wave and k-step (FM x two 32-channel sub-steps)
    static_assert(PH % WP == 0 && WTC == 32 && FM == 2, "a wave owns 32 output channels");
    static_assert(9 % RW == 0 && RW >= 2, "ring slots are fixed per tap");
    static_assert(XH <= 9 - 1, "the halo pieces of the next chunk go out one per k-step");
    static_assert(2 * HBUF <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * HBUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    // XCD-aware order (conv_igemm_dma_kernel): consecutive tiles on one XCD, channel tile fastest
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, wq = nwg >> 3, wr = nwg & 7;
    const int wgid = p.xcd_order ? (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (orig >> 3) : orig;
    const int patch = (int)udiv24((unsigned)wgid, (unsigned)p.ntc), ctile = wgid - patch * p.ntc;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / PH;
    const int prow = (int)udiv24((unsigned)patch, (unsigned)tiles_x), bx = patch - prow * tiles_x;
    const int n = (int)udiv24((unsigned)prow, (unsigned)tiles_y), by = prow - n * tiles_y;
    const int y0 = by * PH, x0 = bx * 16;
    const int c0 = ctile * TC;
    const int nchunks = (p.Cin * ES) / KB;

    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const u32x4_t rw = make_srd(p.wt, p.wt_bytes);
    const unsigned pix_bytes = (unsigned)p.in_ld * ES;
    const int tap_bytes = p.Cin * ES;

    // ---- halo DMA: piece i of a chunk = 8 halo pixels x 128 B; the lane's 16-byte chunk is permuted by the COLUMN key on the source side
    const int lr = lane >> 3, slot = lane & 7;
    unsigned hoff[XH];
#pragma unroll
    for (int i = 0; i < XH; ++i) {
        const int hp = (i * NW + wid) * 8 + lr;
        const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        hoff[i] = ok ? (unsigned)((n * p.Hi + iy) * p.Wi + ix) * pix_bytes + (unsigned)(slot ^ (hx & 7)) * 16u : OOB_OFF;
    }
    auto h_piece = [&](int i, int chunk) __attribute__((always_inline)) {
        const bool live = chunk < nchunks;
        dma16(rx, smem + (chunk & 1) * HBUF + (i * NW + wid) * 1024, live ? hoff[i] : OOB_OFF, live ? chunk * KB : 0);
    };
    // ---- weight fragments: row (output channel) of lane pl in fragment a = the permutation that leaves a lane with NV consecutive channels
    unsigned wvo[FM];
#pragma unroll
    for (int a = 0; a < FM; ++a) {
        const int co = c0 + wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3);
        wvo[a] = co < p.Cout ? (unsigned)co * 9u * (unsigned)tap_bytes + (unsigned)g * 16u : OOB_OFF;
    }
    u32x4_t wreg[RW][NF];
    // the NF loads of the k-step (chunk ch, window w; with flip the window of tap w pairs with the weights of tap 8 - w) into ring slot SLOT
    auto w_issue = [&](auto slot_tag, int ch, int w) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const bool live = ch < nchunks;
        const int soff = live ? (p.flip ? 8 - w : w) * tap_bytes + ch * KB : 0;
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            wload<0>(wreg[SLOT][a * 2 + 0], rw, live ? wvo[a] : OOB_OFF, soff);
            wload<64>(wreg[SLOT][a * 2 + 1], rw, live ? wvo[a] : OOB_OFF, soff);
        }
    };
    // ---- pixel fragments: halo pixel (patch row wp * FN + b + 1 + dy, column pl + 1 + dx); per-lane bases for the three column shifts and the
    // two sub-steps, the row as an immediate
    int cbase[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int hx = pl + d;                               // pl + 1 + dx, dx = d - 1
        const int o = (wp * FN) * ROWB + hx * KB + ((g ^ (hx & 7)) * 16);
        cbase[d][0] = o;
        cbase[d][1] = o ^ 64;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: halo of chunk 0, weights of k-steps 0 .. RW-2
#pragma unroll
    for (int i = 0; i < XH; ++i) h_piece(i, 0);
    static_for<RW - 1>([&](auto st) { w_issue(st, 0, decltype(st)::value); });

    uint4 bf[2][FN];
    for (int c = 0; c < nchunks; ++c) {
        const char* hb = smem + (c & 1) * HBUF;
        static_for<9>([&](auto tt) {
            constexpr int t = decltype(tt)::value;            // window index: (dy, dx) = (t / 3 - 1, t % 3 - 1)
            constexpr int SL = t % RW;
            constexpr int dyr = t / 3, dxi = t % 3;           // row offset dy + 1, column-shift index dx + 1
            // instructions younger than this k-step's weights: the halo piece of the previous k-step (pieces go out at windows 0 .. XH-1) and
            // the RW - 2 weight stages issued since
            constexpr int N = (RW - 2) * NF + [] {
                int h = 0;
                for (int d = 1; d <= RW - 2; ++d) h += ((t - d + 9) % 9) < XH ? 1 : 0;
                return h;
            }();
            wait_frags<N>(wreg[SL]);
            if (t == 0) {
                raw_barrier();                                // the halo of chunk c is in LDS for everyone, everyone is done with chunk c - 1
#pragma unroll
                for (int b = 0; b < FN; ++b) bf[0][b] = *reinterpret_cast<const uint4*>(hb + cbase[dxi][0] + (b + dyr) * ROWB);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // the pixel fragments of the NEXT sub-step (the other half of the register pair) are requested before this sub-step's MFMAs
                if (kk == 0) {
#pragma unroll
                    for (int b = 0; b < FN; ++b) bf[1][b] = *reinterpret_cast<const uint4*>(hb + cbase[dxi][1] + (b + dyr) * ROWB);
                } else if (t < 8) {
                    constexpr int t1 = t < 8 ? t + 1 : 0;
#pragma unroll
                    for (int b = 0; b < FN; ++b) bf[0][b] = *reinterpret_cast<const uint4*>(hb + cbase[t1 % 3][0] + (b + t1 / 3) * ROWB);
                }
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        const int m = (kk * FM + a) * FN + b;
                        if (m == 2 && t < XH) h_piece(t, c + 1);                         // into the buffer chunk c - 1 was read from
                        if (m == 6) w_issue(std::integral_constant<int, (t + RW - 1) % RW>{}, c + (t + RW - 1) / 9, (t + RW - 1) % 9);
                        Mma<T>::run(as_u4(wreg[SL][a * 2 + kk]), bf[kk][b], acc[a][b]);
                    }
            }
        });
    }
    // The trailing out-of-range loads (issued to keep the counts uniform) still WRITE their registers when they land: every ring register stays
    // allocated until this wait -- without the operands below the compiler hands the dead registers to the epilogue (the channel offset landed in
    // one and was zeroed by a late load: all channel groups stored to channels 0-7)
    static_for<RW>([&](auto sl) { wait_frags<0>(wreg[decltype(sl)::value]); });
    const int m0 = (n * p.Ho + y0) * p.Wo + x0;
    conv_epilogue<T, FM, FN, FN * 16, WP, 0, WC, true>(p, acc, m0, wp, c0 + wc * WTC + g * NV, pl, 0, 0, patch, 0,
                                                        reinterpret_cast<float*>(smem), wc, c0);
}

template <typename T, int PH, int TC, int WP, int WC, int RWD>
int launch_wreg(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    k.ntc = k.Cout / TC;
    k.xcd_order = xcd_order_enabled() ? 1 : 0;
    const int blocks = k.N * (k.Ho / PH) * (k.Wo / 16) * k.ntc;
    hipLaunchKernelGGL((conv3x3_wreg_kernel<T, PH, TC, WP, WC, RWD>), dim3(blocks), dim3(WP * WC * 64), 0, st, k);
    return msc_check_launch("conv3x3_wreg");
}

}  // namespace

namespace msc_conv {

bool halo3r_cfg_ok(const ConvK& k, int es, int cfg) {
    const int tc = cfg == CFG_WREG128 ? 128 : 64;
    return es == 2 && k.mode == 0 && k.KH == 3 && k.KW == 3 && k.stride == 1 && k.pad == 1 && k.Hi == k.Ho && k.Wi == k.Wo && k.Wo % 16 == 0 &&
           k.Ho % 16 == 0 && (k.Cin * es) % 128 == 0 && k.Cout % tc == 0 && !k.span_bytes && k.ksplit == 1 && !k.bnl.slots && !k.fin_w;
}

int halo3r_launch(const ConvK& k, int dtype, int cfg, hipStream_t st) {
    if (cfg == CFG_WREG128)
        return dtype == MSC_F16 ? launch_wreg<f16_t, 16, 128, 2, 4, 3>(k, st) : launch_wreg<bf16_t, 16, 128, 2, 4, 3>(k, st);
    return dtype == MSC_F16 ? launch_wreg<f16_t, 16, 64, 4, 2, 9>(k, st) : launch_wreg<bf16_t, 16, 64, 4, 2, 9>(k, st);
}

}  // namespace msc_conv
