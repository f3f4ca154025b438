// Streaming kernel for the 1x1 / stride 1 convolutions of the ResNet bottleneck blocks on the large feature maps (gfx950): layer1 and
// layer2 of the encoder (src/unet_models.py:365-371 via torchvision's Bottleneck), forward and data gradient.  Configuration 57 of
// msc_conv_igemm (igemm.hip dispatches here); same descriptor, same epilogue (conv_common.h) as the implicit-GEMM kernels.
#include <stdlib.h>

#include <algorithm>

#include "conv_common.h"
#include "msc_internal.h"

namespace msc_conv {
namespace {

// ------------------------------------------------------------------------------------------------ 1x1, streaming
// The 1x1 / stride 1 layers of layer1 and layer2 (64 <-> 256 channels on 131 072 pixels, 128 <-> 512 on 32 768; forward and data
// gradient) are pure streams: 84 MB in + out for 4.3 GFLOP.  As tiles of the implicit-GEMM kernel they are one or two k-steps per
// block -- a prologue (address set-up, first fill from HBM), one MFMA burst and an epilogue, nothing of which overlaps within the
// block: 2.4 TB/s before the epilogue was rewritten (its 21 VALU instructions per output element were the bound),
// 3.2 TB/s after.  Here a block is persistent: the wave's slice of the weight matrix stays in registers as MFMA fragments for the
// whole launch, the pixel tiles ([TP][Cin], the whole reduction in one stage) stream through two LDS buffers by DMA (tile t+1 in
// flight while tile t is multiplied and stored), the BatchNorm sums are carried in registers across the tiles and folded once per
// block: 3.8 TB/s on the 64 -> 256 forward layers (22 us against 26.6), where the per-layer timing picks it; the implicit-GEMM tiles
// stay ahead on the layers with longer reductions (256 -> 64: 17.6 us against 24.7).
//   waves: WC along the channels (FM fragments of 16 each) x WP along the pixels of the tile
//   LDS image: a row = a pixel's Cin*2 bytes; 16-byte chunks permuted on the source side (128-byte rows as KB = 128 above, longer rows
//   within each 256-byte window as KB = 256) so the fragment reads of 16 consecutive pixels hit distinct banks
//   M % TP == 0 (no partial tiles: the tile advance is the DMA's scalar offset, which the range check does not see)
template <int RB> __device__ __forceinline__ int swz_stream(int chunk, int key) {
    return RB == 128 ? chunk ^ key : ((chunk & ~15) | ((chunk & 15) ^ key));
}
template <typename T, int KS, int FM, int WC, int WP, int TP, int MINB>
__global__ __launch_bounds__(WP * WC * 64, MINB * WP * WC / 4) void conv1x1_stream_kernel(ConvK p) {
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int ES = 2, NW = WP * WC;
    constexpr int RB = KS * 64;                              // bytes of a pixel row (the whole reduction)
    constexpr int WTC = FM * 16, TC = WC * WTC, WTP = TP / WP, FN = WTP / 16, NV = FM * 4;
    constexpr int NI = TP * RB / 1024, XI = NI / NW;         // DMA wave-instructions per tile / per wave
    constexpr int BUF = TP * RB;
    static_assert(RB == 128 || RB % 256 == 0, "row length");
    static_assert(NI % NW == 0 && TP % (WP * 16) == 0 && RB <= 1024, "tile / wave count");
    static_assert(2 * BUF <= 160 * 1024 && 2 * BUF >= NW * WTC * 2 * 4, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    const int ctile = (int)blockIdx.x % p.ntc, slot0 = (int)blockIdx.x / p.ntc, nslot = (int)gridDim.x / p.ntc;
    const int c0 = ctile * TC;
    const int ntm = p.M / TP;
    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const unsigned pix_bytes = (unsigned)p.in_ld * ES;

    unsigned xoff[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int byte = (i * NW + wid) * 1024 + lane * 16;
        const int row = byte / RB, slot = (byte % RB) / 16;
        xoff[i] = (unsigned)row * pix_bytes + (unsigned)swz_stream<RB>(slot, RB == 128 ? (row >> 1) & 7 : row & 15) * 16u;
    }
    auto issue = [&](int t, int buf) {
        const int soff = (int)((unsigned)t * (unsigned)TP * pix_bytes);
#pragma unroll
        for (int i = 0; i < XI; ++i) dma16(rx, smem + buf * BUF + (i * NW + wid) * 1024, xoff[i], soff);
    };
    int t = slot0;
    if (t < ntm) issue(t, 0);

    // the wave's weight fragments: row i = 4g + r of fragment a is channel cb + a*4 + r of lane group g (conv_epilogue's layout)
    uint4 wf[FM][KS];
    {
        const T* w = reinterpret_cast<const T*>(p.wt);
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            const int ch = c0 + wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) wf[a][kk] = *reinterpret_cast<const uint4*>(w + (long)ch * p.Cin + kk * 32 + g * 8);
        }
    }
    const int key = RB == 128 ? (pl >> 1) & 7 : pl;
    int boff[FN];
#pragma unroll
    for (int b = 0; b < FN; ++b) boff[b] = (wp * WTP + b * 16 + pl) * RB;

    float s1[NV], s2[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    int buf = 0;
    for (; t < ntm; t += nslot, buf ^= 1) {
        wait_vmcnt<0>();                         // this tile has landed (and the stores of the tile before it are out)
        raw_barrier();                           // ... for every wave; everyone is done reading the other buffer
        if (t + nslot < ntm) issue(t + nslot, buf ^ 1);
        f32x4 acc[FM][FN];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* sb = smem + buf * BUF;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int so = swz_stream<RB>(kk * 4 + g, key) * 16;
            uint4 bf[FN];
#pragma unroll
            for (int b = 0; b < FN; ++b) bf[b] = *reinterpret_cast<const uint4*>(sb + boff[b] + so);
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) Mma<T>::run(wf[a][kk], bf[b], acc[a][b]);
        }
        conv_epilogue_tile<T, FM, FN, WTP, WP, 0>(p, acc, t * TP, wp, c0 + wc * WTC + g * NV, pl, 0, 0, s1, s2);
    }
    if (p.stats) conv_epilogue_stats<T, FM, WP, WC>(p, s1, s2, wp, wc, pl, c0, reinterpret_cast<float*>(smem));
}

static int num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}

template <typename T, int KS, int FM, int WC, int WP, int TP, int MINB>
int launch_stream(const ConvK& k0, hipStream_t st) {
    ConvK k = k0;
    constexpr int TC = WC * FM * 16, LDS = 2 * TP * KS * 64;
    k.ntc = k.Cout / TC;
    static const int occ_env = [] { const char* e = getenv("MSC_STREAM_OCC"); return e ? atoi(e) : 0; }();
    const int occ = occ_env > 0 ? occ_env : std::max(1, std::min(2, 160 * 1024 / LDS));
    const int ntm = k.M / TP;
    const int slots = std::max(1, std::min(ntm, num_cus() * occ / k.ntc));
    hipLaunchKernelGGL((conv1x1_stream_kernel<T, KS, FM, WC, WP, TP, MINB>), dim3(slots * k.ntc), dim3(WP * WC * 64), 0, st, k);
    return msc_check_launch("conv1x1_stream");
}

// variants: (input channels, channel tile, pixel tile); index into the dispatch switch, -1: none
struct StreamVar { int cin, tc, tp; };
const StreamVar STREAM_VARS[7] = {{64, 256, 64}, {64, 64, 64}, {128, 256, 32}, {128, 128, 64}, {256, 128, 32}, {256, 64, 64}, {512, 128, 64}};
int stream_variant(int Cin, int Cout) {
    for (int i = 0; i < 7; ++i)
        if (STREAM_VARS[i].cin == Cin && Cout % STREAM_VARS[i].tc == 0) return i;
    return -1;
}

template <typename T>
int dispatch(const ConvK& k, hipStream_t st) {
    switch (stream_variant(k.Cin, k.Cout)) {
        case 0: return launch_stream<T, 2, 2, 8, 1, 64, 2>(k, st);
        case 1: return launch_stream<T, 2, 2, 2, 4, 64, 2>(k, st);
        case 2: return launch_stream<T, 4, 2, 8, 1, 32, 2>(k, st);
        case 3: return launch_stream<T, 4, 2, 4, 2, 64, 2>(k, st);
        case 4: return launch_stream<T, 8, 2, 4, 2, 32, 2>(k, st);
        case 5: return launch_stream<T, 8, 2, 2, 4, 64, 2>(k, st);
        default: return launch_stream<T, 16, 2, 4, 2, 64, 1>(k, st);
    }
}

}  // namespace

bool conv1x1_cfg_ok(const ConvK& k, int es) {
    if (es != 2 || k.mode != 0 || k.KH != 1 || k.KW != 1 || k.stride != 1 || k.pad != 0 || k.span_bytes || k.ksplit > 1 || k.fin_w) return false;
    const int v = stream_variant(k.Cin, k.Cout);
    return v >= 0 && k.M % STREAM_VARS[v].tp == 0 && (long)k.M * k.in_ld * es < 0x7fffffffL;
}

int conv1x1_launch(const ConvK& k, int dtype, hipStream_t st) {
    return dtype == MSC_F16 ? dispatch<f16_t>(k, st) : dispatch<bf16_t>(k, st);
}

}  // namespace msc_conv
