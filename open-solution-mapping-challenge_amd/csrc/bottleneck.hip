// Eval-mode fused ResNet Bottleneck for gfx950 (MI355X): ONE launch per identity block
//     out = ReLU( bn3(conv1x1( ReLU(bn2(conv3x3( ReLU(bn1(conv1x1(x))) ))) )) + x )
// i.e. the torchvision Bottleneck (stride 1, no downsample branch) the reference's encoder stages are made of
// (src/unet_models.py:345-351,365-371; ResNet101: 29 of its 33 blocks, ResNet152: 46 of 50), with BatchNorm folded into
// per-channel (scale, shift) -- in eval mode there is no statistic between the three convolutions, so nothing forces the
// two CMID-channel intermediates through HBM.  As three launches these layers were 72 of ResNet101's 118 convolutions and sat
// at 0.1-0.6 PFLOP/s: HBM-bound at a third of the HBM rate on the 64x64 maps, and at a 11-15 us floor per launch on the
// 16x16 maps whatever the tile.
//
// A block owns a PH x 16 pixel patch of one image and ALL channels:
//   phase 1  conv1 (1x1, 4*CMID -> CMID) over the (PH+2) x 18 halo of the patch: the input rows stream HBM -> LDS by DMA in
//            64-channel k-steps through a ring; the result (bf16/fp16, zero outside the image = conv2's padding) stays in
//            LDS as CMID/64 planes of [halo pixel][128 B]
//   phase 2  conv2 (3x3, CMID -> CMID): the nine taps read shifted windows of that halo (as conv3x3_halo_dma_kernel does);
//            result -> LDS (over the dead input ring)
//   phase 3  conv3 (1x1, CMID -> 4*CMID) in four passes of CMID output channels: + shift, + residual (the block input, read
//            back from L2), ReLU, 16-byte stores.
// Weights never touch LDS: every wave owns 32 output channels for all of its pixels, so a weight byte is needed by exactly
// one wave (WP > 1: by WP waves) -- they are pre-packed in MFMA-fragment order (msc_bottleneck_pack: one fully coalesced
// 1 KiB load per 16x32 fragment) and stream L2 -> VGPR through a register ring, three k-steps ahead, as ONE sequence across
// the three phases (the ring keeps running over the phase boundaries; only phase 1 has a barrier per k-step, for the pixel
// rows).  LDS carries pixel operands only: no weight stage, no LDS bandwidth spent on data a single wave consumes.
// What bounds it: the L1 fill (64 B/clk/CU) of the weight stream on the 16x16 maps (2.2 MB per block and CU for CMID = 256),
// HBM on the 64x64 maps (the block input is read once and the output written once: 134 MB per ResNet101 layer1 block).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "dma.h"
#include "msc_internal.h"

namespace {

struct BnkK {
    unsigned long long* dbg;      // probe only (ABL bit 3): per-phase shader-clock stamps of wave 0 of every block
    const char* x; char* out; const char* wpk;
    const float* sc1; const float* sh1; const float* sc2; const float* sh2; const float* sc3; const float* sh3;
    long x_ld, out_ld;
    int N, H, W;
    unsigned x_bytes, w_bytes;
    int xcd_order;
};

// one 16x32 weight fragment (64 lanes x 16 B, contiguous) L2 -> VGPR.  Inline asm for the same reason as dma16: the loads of the
// ring must stay where they are written (between the MFMAs, k-steps ahead of their use) and be waited for by OUR counted
// s_waitcnt, which also counts the LDS-DMA instructions the compiler cannot see.
template <int IMM>
__device__ __forceinline__ void wload(u32x4_t& r, u32x4_t srd, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(srd), "s"(soff), "n"(IMM) : "memory");
}
// s_waitcnt vmcnt(N) that the four fragments of a ring slot depend on: the compiler may not touch them before it
template <int N>
__device__ __forceinline__ void wait_frags(u32x4_t (&w)[4]) {
    asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : [cnt] "n"(N) : "memory");
}
__device__ __forceinline__ uint4 as_u4(const u32x4_t& v) { return make_uint4(v.x, v.y, v.z, v.w); }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): the k-steps of a block are straight-line code, so that ring
// slots, scalar offsets and the counted waits are compile-time constants and no register that a load in flight is about to
// write ever crosses a loop back-edge
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The memory instructions of a block in program order: prologue = weight stages 0 .. RW-2 (NF fragment loads each), then
// pixel-row stages 0 .. RX-2 (XH DMA instructions each); k-step s then issues -- between its MFMAs, in this order -- the XH
// pixel-row instructions of stage s+RX-1 (phase 1 only, s < K1S; past the end of the reduction they go out with an
// out-of-range offset) and the NF fragment loads of weight stage s+RW-1 (while the stream lasts); the FR residual rows of an
// output pass of phase 3 are requested right before the first k-step of the pass and waited for after its last one.
// vmcnt counts instructions in issue order, so a wait is "at most N instructions younger than the last one I need":
//   what = 0: k-step `idx` waits for its weight stage and (phase 1) its pixel-row stage;  what = 1: the epilogue of pass `idx` of
//   phase 3 waits for its residual rows.
template <int S, int K1S, int S2, int NCH, int NF, int XH, int FR, int RX, int RW>
constexpr int pending_at(int what, int idx) {
    int pos = 0, last = -1;
    for (int t = 0; t < RW - 1; ++t)
        for (int n = 0; n < NF; ++n) { if (what == 0 && t == idx) last = pos; ++pos; }
    for (int t = 0; t < RX - 1; ++t)
        for (int n = 0; n < XH; ++n) { if (what == 0 && t == idx && idx < K1S) last = pos; ++pos; }
    for (int s = 0; s < S; ++s) {
        const int q = s - K1S - S2;                        // k-step of phase 3
        if (q >= 0 && q % NCH == 0)
            for (int n = 0; n < FR; ++n) { if (what == 1 && q / NCH == idx) last = pos; ++pos; }
        if (what == 0 && s == idx) return pos - 1 - last;
        if (s < K1S)
            for (int n = 0; n < XH; ++n) { if (what == 0 && s + RX - 1 == idx && idx < K1S) last = pos; ++pos; }
        if (s + RW - 1 < S)
            for (int n = 0; n < NF; ++n) { if (what == 0 && s + RW - 1 == idx) last = pos; ++pos; }
        if (what == 1 && q >= 0 && q % NCH == NCH - 1 && q / NCH == idx) return pos - 1 - last;
    }
    return 0;
}

// s_waitcnt vmcnt(N) that FN asm-loaded registers depend on
template <int N>
__device__ __forceinline__ void wait_regs(u32x4_t (&r)[1]) { asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(r[0]) : [cnt] "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_regs(u32x4_t (&r)[2]) { asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(r[0]), "+v"(r[1]) : [cnt] "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_regs(u32x4_t (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [cnt] "n"(N) : "memory");
}

template <typename T> __device__ __forceinline__ uint4 pack8(const float* v);
template <> __device__ __forceinline__ uint4 pack8<bf16_t>(const float* v) { return Vec16<bf16_t>::pack(v); }
template <> __device__ __forceinline__ uint4 pack8<f16_t>(const float* v) { return Vec16<f16_t>::pack(v); }

// CMID: bottleneck width (64 / 128 / 256), PH: patch rows, RX: depth of the pixel-row ring of phase 1 (LDS), RW: depth of the weight
// ring (registers: 16 VGPRs per stage) -- RW-1 k-steps of weights are in flight per wave: the stream comes out of L2 (or, for
// the first block of an XCD to touch a line, out of HBM) with a microsecond or more of latency, and 8 waves x 4 KB x (RW-1) per
// CU is what has to cover it; MINB: resident blocks per CU the register allocation has to allow
// ABL (tools/bneck_probe.py only, MSC_BNECK_ABL; 0 in the product): bit 0 = no MFMA (operands still fetched), bit 1 = weight loads go out
// with an out-of-range offset (same instruction stream, no L2 traffic), bit 2 = no pixel-fragment reads from LDS, bit 3 = per-phase
// shader-clock stamps into the debug buffer.  Only 7 and 8 are instantiated in the library (results: profiles/r3_bneck_probe.txt)
template <typename T, int CMID, int PH, int RX, int RW, int MINB, int ABL = 0>
__global__ __launch_bounds__(512, MINB * 2) void bottleneck_fused_kernel(BnkK p) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int NW = 8;                                  // waves per block
    constexpr int WC = CMID / 32, WP = NW / WC;            // waves along channels (32 each: FM = 2 fragments) x along pixels
    constexpr int FM = 2, NF = 2 * FM;                     // weight fragments per wave and k-step: 2 sub-steps of 32 channels x FM
    constexpr int C4 = 4 * CMID;
    constexpr int NCH = CMID / 64;                         // 64-channel chunks (128-byte rows) of the intermediates
    constexpr int K1S = C4 / 64;                           // k-steps of phase 1
    constexpr int S2 = 9 * NCH, S3 = 4 * NCH, S = K1S + S2 + S3;      // k-steps of phases 2, 3; of the block
    constexpr int HCOLS = 18, HP = (PH + 2) * HCOLS;       // halo pixels
    constexpr int FN1 = ((HP + 15) / 16 + WP - 1) / WP;    // pixel fragments per wave, phase 1
    constexpr int HPR = WP * FN1 * 16;                     // halo rows incl. padding to whole fragments
    constexpr int FN = PH / WP;                            // pixel fragments (patch rows) per wave, phases 2 and 3
    constexpr int NIX = HPR / 8;                           // DMA wave-instructions per pixel-row stage (8 rows x 128 B)
    constexpr int XH = (NIX + NW - 1) / NW;                // ... per wave
    constexpr int XSTAGE = XH * NW * 1024;
    constexpr int PLANE1 = HPR * 128, MID1 = NCH * PLANE1;
    constexpr int PLANE2 = PH * 16 * 128, MID2 = NCH * PLANE2;
    constexpr int WSTEP = WC * NF * 1024;                  // bytes of the weight stream per k-step
    static_assert(WC * WP == NW && PH % WP == 0 && K1S >= RX && RX >= 2 && RW >= 2 && RW <= 8 && (K1S & (K1S - 1)) == 0, "tiling");
    static_assert(MID2 <= RX * XSTAGE, "phase-2 result reuses the pixel-row ring");
    constexpr int COEF = 12 * CMID * 4;                    // (scale, shift) of the three BatchNorms: 2 x (CMID + CMID + 4 CMID) floats
    static_assert(MID1 + RX * XSTAGE + COEF <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[MID1 + RX * XSTAGE + COEF];
    char* const mid1 = smem;
    char* const xring = smem + MID1;
    char* const mid2 = xring;
    // the folded BatchNorm coefficients go to LDS once: a global load in an epilogue would make the compiler wait for ALL
    // outstanding vector-memory operations (it cannot see the ring) and drain the weight stream at every phase boundary
    float* const coef = reinterpret_cast<float*>(smem + MID1 + RX * XSTAGE);      // [sc1 | sh1 | sc2 | sh2 | sc3 (4C) | sh3 (4C)]
    for (int i = threadIdx.x; i < CMID; i += 512) {
        coef[i] = p.sc1[i]; coef[CMID + i] = p.sh1[i]; coef[2 * CMID + i] = p.sc2[i]; coef[3 * CMID + i] = p.sh2[i];
    }
    for (int i = threadIdx.x; i < C4; i += 512) { coef[4 * CMID + i] = p.sc3[i]; coef[8 * CMID + i] = p.sh3[i]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // before the first instruction of the stream (read after phase 1's barriers)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    auto stamp = [&](int i) {
        if ((ABL & 8) && p.dbg && tid == 0) p.dbg[blockIdx.x * 16 + i] = __builtin_readcyclecounter();
    };
    stamp(0);
    // XCD-aware order (block b runs on XCD b % 8): consecutive patches -- vertical neighbours share halo rows -- on one XCD
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, wq = nwg >> 3, wr_ = nwg & 7;
    const int patch = p.xcd_order ? (xcd < wr_ ? xcd * (wq + 1) : wr_ * (wq + 1) + (xcd - wr_) * wq) + (orig >> 3) : orig;
    const int tiles_x = p.W / 16, tiles_y = p.H / PH;
    const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
    const int y0 = by * PH, x0 = bx * 16;

    const u32x4_t rx = make_srd(p.x, p.x_bytes);
    const u32x4_t rw = make_srd(p.wpk, p.w_bytes);
    const unsigned pix_bytes = (unsigned)p.x_ld * 2u;

    // ---- phase-1 pixel rows: DMA source offsets (k-step advance through the scalar offset)
    const int lr = lane >> 3, slot = lane & 7;
    unsigned hoff[XH];
#pragma unroll
    for (int i = 0; i < XH; ++i) {
        const int hp = (i * NW + wid) * 8 + lr;
        const int hy = hp / HCOLS, hx = hp - hy * HCOLS;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const bool ok = hp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        hoff[i] = ok ? (unsigned)((n * p.H + iy) * p.W + ix) * pix_bytes + (unsigned)(slot ^ ((hp >> 1) & 7)) * 16u : OOB_OFF;
    }
    const unsigned wvoff = (ABL & 2) ? OOB_OFF : (unsigned)(wc * NF * 1024 + lane * 16);

    // Every CU needs the whole weight stream of a phase, in no particular order (only the fp32 summation order depends on it).
    // Read in the SAME order by the 32 CUs of an XCD -- all started together, all at the same rate -- every line is asked for 32
    // times within a few hundred cycles and the L2 channel that holds it serialises them while the other channels idle.  So each
    // block starts the k-steps of phase 1 (its input-channel chunks) and of phase 2 (its (chunk, tap) pairs) at its own offset:
    // at any time the CUs of an XCD pull different parts of the stream, and the first touches (HBM) are spread as well.
    const int qx = (orig >> 3) & 31;                        // position among the blocks of this XCD that are resident together
    const int rot1 = __builtin_amdgcn_readfirstlane((qx * K1S) >> 5);
    const int rot2 = __builtin_amdgcn_readfirstlane((qx * S2) >> 5);
    auto elem1 = [&](int t) { return (t + rot1) & (K1S - 1); };                                   // chunk of phase-1 k-step t
    auto elem2 = [&](int i) { const int e = i + rot2; return e >= S2 ? e - S2 : e; };             // (chunk, tap) of phase-2 k-step i
    auto wsoff = [&](int ts) {                              // byte offset of weight stage ts (ts is a compile-time constant at every call)
        return ts < K1S ? elem1(ts) * WSTEP : ts < K1S + S2 ? (K1S + elem2(ts - K1S)) * WSTEP : ts * WSTEP;
    };
    u32x4_t wring[RW][NF];
    // stage t of the stream: its pixel rows (phase 1 only; past K1S the instructions still go out, with an out-of-range offset:
    // every k-step of phase 1 issues the same number of memory instructions, so the counted waits are compile-time constants)
    auto x_piece = [&](int i, int t, int xs) {
        const bool live = t < K1S;
        dma16(rx, xring + xs * XSTAGE + (i * NW + wid) * 1024, live ? hoff[i] : OOB_OFF, live ? elem1(t) * 128 : 0);
    };
    // ---- prologue: weight stages 0 .. RW-2, pixel-row stages 0 .. RX-2
    static_for<RW - 1>([&](auto tt) {
        constexpr int t = decltype(tt)::value;
        wload<0>(wring[t][0], rw, wvoff, wsoff(t));
        wload<1024>(wring[t][1], rw, wvoff, wsoff(t));
        wload<2048>(wring[t][2], rw, wvoff, wsoff(t));
        wload<3072>(wring[t][3], rw, wvoff, wsoff(t));
    });
#pragma unroll
    for (int t = 0; t < RX - 1; ++t) {
#pragma unroll
        for (int i = 0; i < XH; ++i) x_piece(i, t, t);
    }

    stamp(1);
    // =============================================================== phase 1: conv1 over the halo
    {
        f32x4 acc[FM][FN1];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN1; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        int boff[FN1];
#pragma unroll
        for (int b = 0; b < FN1; ++b) {
            const int r = (wp * FN1 + b) * 16 + pl;
            boff[b] = r * 128 + ((g ^ ((r >> 1) & 7)) * 16);
        }
        constexpr int NM = 2 * FM * FN1;                   // MFMAs per k-step and wave
        constexpr int LPW = XH + NF;                       // memory instructions per k-step and wave
        static_for<K1S>([&](auto gg) {
            constexpr int gs = decltype(gg)::value, j = gs % RW;     // this k-step and the ring slot of its weights
            constexpr int ts = gs + RW - 1, js = ts % RW;            // the weight stage it fetches, into the slot stage gs-1 just left
            constexpr int xs = gs + RX - 1, jx = xs % RX;            // the pixel-row stage it fetches, likewise
            wait_frags<pending_at<S, K1S, S2, NCH, NF, XH, FN, RX, RW>(0, gs)>(wring[j]);
            raw_barrier();                                 // everyone's rows of stage gs are in LDS, everyone is done with stage gs-1
            const char* xb = xring + (gs % RX) * XSTAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 bf[FN1];
#pragma unroll
                for (int b = 0; b < FN1; ++b) bf[b] = (ABL & 4) ? make_uint4(1u, 2u, 3u, 4u) : *reinterpret_cast<const uint4*>(xb + (boff[b] ^ (kk * 64)));
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN1; ++b) {
                        const int m = (kk * FM + a) * FN1 + b;
#pragma unroll
                        for (int i = 0; i < LPW; ++i)
                            if ((i * NM) / LPW == m) {
                                if (i < XH) x_piece(i < XH ? i : 0, xs, jx);
                                else if (i == XH) wload<0>(wring[js][0], rw, wvoff, wsoff(ts));
                                else if (i == XH + 1) wload<1024>(wring[js][1], rw, wvoff, wsoff(ts));
                                else if (i == XH + 2) wload<2048>(wring[js][2], rw, wvoff, wsoff(ts));
                                else wload<3072>(wring[js][3], rw, wvoff, wsoff(ts));
                            }
                        if (!(ABL & 1)) Mma<T>::run(as_u4(wring[j][kk * FM + a]), bf[b], acc[a][b]);
                        else asm volatile("" ::"v"(wring[j][kk * FM + a]), "v"(bf[b].x), "v"(bf[b].w));
                    }
            }
        });
        stamp(2);
        // epilogue: ReLU(bn1(.)) -> LDS, zero outside the image (conv2's zero padding is of THIS tensor)
        const int cb = wc * 32 + g * 8;                    // lane holds channels cb .. cb+7 of its pixels
        float sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = coef[cb + q]; sh[q] = coef[CMID + cb + q]; }
#pragma unroll
        for (int b = 0; b < FN1; ++b) {
            const int r = (wp * FN1 + b) * 16 + pl;
            const int hy = r / HCOLS, hx = r - hy * HCOLS;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = r < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            float v[8];
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[a * 4 + q] = ok ? fmaxf(fmaf(acc[a][b][q], sc[a * 4 + q], sh[a * 4 + q]), 0.f) : 0.f;
            // (round 6) the swizzle key of the conv1 halo is its COLUMN, hx & 7: phase 2 reads 16-pixel windows that start at any column, and the
            // pixel-pair key put two lanes of a ds_read_b128 group on one slot for two of the three column shifts (conv3x3_halo_dma_kernel, igemm.hip)
            *reinterpret_cast<uint4*>(mid1 + (cb >> 6) * PLANE1 + r * 128 + ((((cb & 63) >> 3) ^ (hx & 7)) * 16)) = pack8<T>(v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();                                         // the halo of conv1 outputs is complete; the pixel-row ring is dead
    stamp(3);

    // =============================================================== phase 2: conv2 (3x3) on the halo
    {
        f32x4 acc[FM][FN];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        int hp0[FN];                                       // halo pixel of the centre tap
#pragma unroll
        for (int b = 0; b < FN; ++b) hp0[b] = (wp * FN + b + 1) * HCOLS + pl + 1;
        auto step2 = [&](auto ii) {
            constexpr int i = decltype(ii)::value;          // k-step of this phase: chunk i / 9, tap i % 9
            constexpr int gs = K1S + i, j = gs % RW, ts = gs + RW - 1, js = ts % RW;
            const int e = elem2(i);                        // this block's i-th (chunk, tap) pair: uniform, in scalar registers
            const int c = (e * 29) >> 8, t = e - 9 * c;    // e / 9, e % 9 for e < 36
            const int dy = (t * 11) >> 5, dx = t - 3 * dy; // t / 3, t % 3
            const int delta = (dy - 1) * HCOLS + dx - 1;
            wait_frags<pending_at<S, K1S, S2, NCH, NF, XH, FN, RX, RW>(0, gs)>(wring[j]);
            const char* hb = mid1 + c * PLANE1;
            int boff[FN];
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int hp = hp0[b] + delta;
                boff[b] = hp * 128 + ((g ^ ((pl + dx) & 7)) * 16);      // column of the halo pixel: pl + 1 + (dx - 1)
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 bf[FN];
#pragma unroll
                for (int b = 0; b < FN; ++b) bf[b] = (ABL & 4) ? make_uint4(1u, 2u, 3u, 4u) : *reinterpret_cast<const uint4*>(hb + (boff[b] ^ (kk * 64)));
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) {
                        constexpr int NM = 2 * FM * FN;
                        const int m = (kk * FM + a) * FN + b;
                        if (ts < S) {
                            if ((0 * NM) / NF == m) wload<0>(wring[js][0], rw, wvoff, wsoff(ts));
                            if ((1 * NM) / NF == m) wload<1024>(wring[js][1], rw, wvoff, wsoff(ts));
                            if ((2 * NM) / NF == m) wload<2048>(wring[js][2], rw, wvoff, wsoff(ts));
                            if ((3 * NM) / NF == m) wload<3072>(wring[js][3], rw, wvoff, wsoff(ts));
                        }
                        if (!(ABL & 1)) Mma<T>::run(as_u4(wring[j][kk * FM + a]), bf[b], acc[a][b]);
                        else asm volatile("" ::"v"(wring[j][kk * FM + a]), "v"(bf[b].x), "v"(bf[b].w));
                    }
            }
        };
        static_for<S2>(step2);
        stamp(4);
        // epilogue: ReLU(bn2(.)) -> LDS over the dead pixel-row ring
        const int cb = wc * 32 + g * 8;
        float sc[8], sh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = coef[2 * CMID + cb + q]; sh[q] = coef[3 * CMID + cb + q]; }
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int px = (wp * FN + b) * 16 + pl;
            float v[8];
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[a * 4 + q] = fmaxf(fmaf(acc[a][b][q], sc[a * 4 + q], sh[a * 4 + q]), 0.f);
            *reinterpret_cast<uint4*>(mid2 + (cb >> 6) * PLANE2 + px * 128 + ((((cb & 63) >> 3) ^ ((px >> 1) & 7)) * 16)) = pack8<T>(v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();

    stamp(5);
    // =============================================================== phase 3: conv3 (1x1) + residual + ReLU -> HBM
    {
        int boff[FN];
        long opix[FN];
        unsigned roff[FN];                                  // byte offset of the pixel's channels 32 wc + 8 g .. in the block input
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int px = (wp * FN + b) * 16 + pl;
            boff[b] = px * 128 + ((g ^ ((px >> 1) & 7)) * 16);
            opix[b] = (long)(n * p.H + y0 + wp * FN + b) * p.W + x0 + pl;
            roff[b] = (unsigned)opix[b] * pix_bytes + (unsigned)(wc * 32 + g * 8) * 2u;
        }
        T* out = reinterpret_cast<T*>(p.out);
        auto pass3 = [&](auto pp) {
            constexpr int ps = decltype(pp)::value;         // output channels ps*CMID .. +CMID
            const int cb = ps * CMID + wc * 32 + g * 8;
            f32x4 acc[FM][FN];
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4_t resv[FN];                               // the residual rows (the block input): requested before the k-steps of the pass,
#pragma unroll                                              // as instructions of the counted stream
            for (int b = 0; b < FN; ++b) wload<ps * CMID * 2>(resv[b], rx, roff[b], 0);
            auto step3 = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                constexpr int gs = K1S + S2 + ps * NCH + c, j = gs % RW, ts = gs + RW - 1, js = ts % RW;
                wait_frags<pending_at<S, K1S, S2, NCH, NF, XH, FN, RX, RW>(0, gs)>(wring[j]);
                const char* hb = mid2 + c * PLANE2;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    uint4 bf[FN];
#pragma unroll
                    for (int b = 0; b < FN; ++b) bf[b] = (ABL & 4) ? make_uint4(1u, 2u, 3u, 4u) : *reinterpret_cast<const uint4*>(hb + (boff[b] ^ (kk * 64)));
#pragma unroll
                    for (int a = 0; a < FM; ++a)
#pragma unroll
                        for (int b = 0; b < FN; ++b) {
                            constexpr int NM = 2 * FM * FN;
                            const int m = (kk * FM + a) * FN + b;
                            if (ts < S) {
                                if ((0 * NM) / NF == m) wload<0>(wring[js][0], rw, wvoff, wsoff(ts));
                                if ((1 * NM) / NF == m) wload<1024>(wring[js][1], rw, wvoff, wsoff(ts));
                                if ((2 * NM) / NF == m) wload<2048>(wring[js][2], rw, wvoff, wsoff(ts));
                                if ((3 * NM) / NF == m) wload<3072>(wring[js][3], rw, wvoff, wsoff(ts));
                            }
                            if (!(ABL & 1)) Mma<T>::run(as_u4(wring[j][kk * FM + a]), bf[b], acc[a][b]);
                        else asm volatile("" ::"v"(wring[j][kk * FM + a]), "v"(bf[b].x), "v"(bf[b].w));
                        }
                }
            };
            static_for<NCH>(step3);
            wait_regs<pending_at<S, K1S, S2, NCH, NF, XH, FN, RX, RW>(1, ps)>(resv);
            if (ps == 0) stamp(10);
            float sc[8], sh[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { sc[q] = coef[4 * CMID + cb + q]; sh[q] = coef[8 * CMID + cb + q]; }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                float rv[8], v[8];
                Vec16<T>::unpack(as_u4(resv[b]), rv);
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[a * 4 + q] = fmaxf(fmaf(acc[a][b][q], sc[a * 4 + q], sh[a * 4 + q]) + rv[a * 4 + q], 0.f);
                *reinterpret_cast<uint4*>(out + opix[b] * p.out_ld + cb) = pack8<T>(v);
            }
        };
        pass3(std::integral_constant<int, 0>{});
        stamp(6);
        pass3(std::integral_constant<int, 1>{});
        stamp(7);
        pass3(std::integral_constant<int, 2>{});
        stamp(8);
        pass3(std::integral_constant<int, 3>{});
        stamp(9);
    }
}

// fragment-major weight stream of one block (see the kernel): [k-step][wave column wc][sub-step][fragment a][lane] x 16 B
//   fragment row i of fragment a of wave column wc = output channel 32*wc + 8*(i>>2) + 4*a + (i&3) (a lane then ends up with
//   8 consecutive channels), lane (g = lane>>4, i = lane&15) holds its k elements 32*sub + 8*g .. +7 of the k-step's 64
template <typename T>
__global__ void bottleneck_pack_kernel(const T* __restrict__ w1, const T* __restrict__ w2, const T* __restrict__ w3, uint4* __restrict__ out,
                                       int cmid, long units) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= units) return;
    const int wcn = cmid / 32, nch = cmid / 64, c4 = 4 * cmid, k1s = c4 / 64;
    const int lane = (int)(idx & 63), f = (int)((idx >> 6) & 3), wc = (int)((idx >> 8) % wcn);
    const int s = (int)(idx / (256L * wcn));
    const int g = lane >> 4, i = lane & 15, sub = f >> 1, a = f & 1;
    const int lc = 32 * wc + 8 * (i >> 2) + 4 * a + (i & 3);
    const int kk = 32 * sub + 8 * g;
    const T* src;
    if (s < k1s) src = w1 + (long)lc * c4 + s * 64 + kk;                                         // [CMID][4*CMID]
    else if (s < k1s + 9 * nch) {
        const int q = s - k1s, c = q / 9, t = q - 9 * c;
        src = w2 + ((long)lc * 9 + t) * cmid + c * 64 + kk;                                      // [CMID][3][3][CMID]
    } else {
        const int q = s - k1s - 9 * nch, ps = q / nch, c = q - ps * nch;
        src = w3 + (long)(ps * cmid + lc) * cmid + c * 64 + kk;                                  // [4*CMID][CMID]
    }
    out[idx] = *reinterpret_cast<const uint4*>(src);
}

long pack_units(int cmid) {       // 16-byte units of the stream
    const long steps = 4L * cmid / 64 + 13L * (cmid / 64);
    return steps * (cmid / 32) * 256;
}

bool xcd_on() { static int v = -1; if (v < 0) { const char* e = getenv("MSC_XCD_ORDER"); v = (e && e[0] == '0') ? 0 : 1; } return v == 1; }

// patch rows for a block (d->cfg selects explicitly; the caller may time the valid ones).  CMID = 64 (HBM-bound, 140 KB of weights): 8 or
// 4, two resident blocks per CU either way (one block's loads overlap the other's MFMAs) -- measured on ResNet101's layer1
// (32 x 64 x 64): 52 us with 8 rows, 65 us with 4 (twice the blocks, twice the weight traffic); CMID = 128: 8; CMID = 256: 4 when
// that still gives every CU a block, else 2
int pick_ph(const msc_bneck_desc* d) {
    if (d->Cmid == 64) return d->H % 8 == 0 ? 8 : 4;
    if (d->Cmid == 128) return 8;
    return (long)d->N * (d->H / 4) * (d->W / 16) >= 256 && d->H % 4 == 0 ? 4 : 2;
}

template <typename T>
int launch_bneck(const msc_bneck_desc* d, const BnkK& k, int ph, hipStream_t st) {
    const int blocks = d->N * (d->H / ph) * (d->W / 16);
    if (d->Cmid == 64 && ph == 4) hipLaunchKernelGGL((bottleneck_fused_kernel<T, 64, 4, 3, 4, 2>), dim3(blocks), dim3(512), 0, st, k);
    else if (d->Cmid == 64) hipLaunchKernelGGL((bottleneck_fused_kernel<T, 64, 8, 2, 4, 2>), dim3(blocks), dim3(512), 0, st, k);
    else if (d->Cmid == 128) hipLaunchKernelGGL((bottleneck_fused_kernel<T, 128, 8, 4, 5, 1>), dim3(blocks), dim3(512), 0, st, k);
    else if (ph == 4) hipLaunchKernelGGL((bottleneck_fused_kernel<T, 256, 4, 4, 6, 1>), dim3(blocks), dim3(512), 0, st, k);
    else {
        static const int abl = [] { const char* e = getenv("MSC_BNECK_ABL"); return e ? atoi(e) : 0; }();      // probe only
        if constexpr (std::is_same<T, bf16_t>::value) {
            if (abl == 8) { hipLaunchKernelGGL((bottleneck_fused_kernel<T, 256, 2, 4, 8, 1, 8>), dim3(blocks), dim3(512), 0, st, k); return msc_check_launch("bottleneck_fused"); }
            if (abl == 7) { hipLaunchKernelGGL((bottleneck_fused_kernel<T, 256, 2, 4, 8, 1, 7>), dim3(blocks), dim3(512), 0, st, k); return msc_check_launch("bottleneck_fused"); }
        }
        hipLaunchKernelGGL((bottleneck_fused_kernel<T, 256, 2, 4, 8, 1>), dim3(blocks), dim3(512), 0, st, k);
    }
    return msc_check_launch("bottleneck_fused");
}

}  // namespace

static unsigned long long* g_bneck_dbg = nullptr;
// probe hook (tools/bneck_probe.py; not part of the C ABI of include/msc.h): device buffer of 16 stamps per block for MSC_BNECK_ABL=8
extern "C" void msc_bottleneck_debug_buffer(void* buf) { g_bneck_dbg = (unsigned long long*)buf; }

extern "C" int msc_bottleneck_ok(const msc_bneck_desc* d) {
    if (!d) return 0;
    if (d->dtype != MSC_BF16 && d->dtype != MSC_F16) return 0;
    if (d->Cmid != 64 && d->Cmid != 128 && d->Cmid != 256) return 0;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->W % 16) return 0;
    const int ph = d->cfg > 0 ? d->cfg : pick_ph(d);
    if (d->Cmid == 64 ? (ph != 4 && ph != 8) : d->Cmid == 128 ? ph != 8 : (ph != 2 && ph != 4)) return 0;
    if (d->H % ph) return 0;
    if (d->x_ld < 4 * d->Cmid || d->out_ld < 4 * d->Cmid || (d->x_ld * 2) % 16 || (d->out_ld * 2) % 16) return 0;
    if ((((long)d->N * d->H * d->W - 1) * d->x_ld + 4L * d->Cmid) * 2 >= 0x7fffffffL) return 0;       // 31-bit buffer offsets
    return 1;
}

extern "C" int64_t msc_bottleneck_pack_bytes(int Cmid) {
    return (Cmid == 64 || Cmid == 128 || Cmid == 256) ? pack_units(Cmid) * 16 : -1;
}

extern "C" int msc_bottleneck_pack(const void* w1, const void* w2, const void* w3, void* wpk, int Cmid, int dtype, void* stream) {
    if (!w1 || !w2 || !w3 || !wpk) return msc_fail(MSC_ERR_ARG, "msc_bottleneck_pack: null pointer");
    if ((dtype != MSC_BF16 && dtype != MSC_F16) || (Cmid != 64 && Cmid != 128 && Cmid != 256))
        return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bottleneck_pack: 16-bit weights, Cmid in {64, 128, 256} (dtype %d, Cmid %d)", dtype, Cmid);
    if (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3 | (uintptr_t)wpk) & 15) return msc_fail(MSC_ERR_ARG, "msc_bottleneck_pack: pointers must be 16-byte aligned");
    const long units = pack_units(Cmid);
    hipLaunchKernelGGL(bottleneck_pack_kernel<uint16_t>, dim3(ceil_div(units, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w1,
                       (const uint16_t*)w2, (const uint16_t*)w3, (uint4*)wpk, Cmid, units);
    return msc_check_launch("msc_bottleneck_pack");
}

extern "C" int msc_bottleneck_fused(const msc_bneck_desc* d, void* stream) {
    if (!d || !d->x || !d->out || !d->wpk || !d->scale1 || !d->shift1 || !d->scale2 || !d->shift2 || !d->scale3 || !d->shift3)
        return msc_fail(MSC_ERR_ARG, "msc_bottleneck_fused: null pointer");
    if (!msc_bottleneck_ok(d))
        return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bottleneck_fused: not a shape the fused kernel takes (dtype %d, Cmid %d, %dx%dx%d, ld %ld/%ld, cfg %d)",
                        d->dtype, d->Cmid, d->N, d->H, d->W, (long)d->x_ld, (long)d->out_ld, d->cfg);
    if (((uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->wpk) & 15) return msc_fail(MSC_ERR_ARG, "msc_bottleneck_fused: pointers must be 16-byte aligned");
    BnkK k;
    k.dbg = g_bneck_dbg;
    k.x = (const char*)d->x; k.out = (char*)d->out; k.wpk = (const char*)d->wpk;
    k.sc1 = d->scale1; k.sh1 = d->shift1; k.sc2 = d->scale2; k.sh2 = d->shift2; k.sc3 = d->scale3; k.sh3 = d->shift3;
    k.x_ld = d->x_ld; k.out_ld = d->out_ld; k.N = d->N; k.H = d->H; k.W = d->W;
    k.x_bytes = (unsigned)((((long)d->N * d->H * d->W - 1) * d->x_ld + 4L * d->Cmid) * 2);
    k.w_bytes = (unsigned)(pack_units(d->Cmid) * 16);
    k.xcd_order = xcd_on() ? 1 : 0;
    const int ph = d->cfg > 0 ? d->cfg : pick_ph(d);
    if (d->dtype == MSC_F16) return launch_bneck<f16_t>(d, k, ph, (hipStream_t)stream);
    return launch_bneck<bf16_t>(d, k, ph, (hipStream_t)stream);
}
