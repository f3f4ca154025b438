// Halo-tile kernels of the two 32-channel full-resolution layers of the decoder (gfx950): dec0's 3x3 convolution 32 -> 32 (+ its data
// gradient, + the fused final 1x1 and softmax in inference) and dec1's ConvTranspose2d(k4, s2, p1) 128 -> 32
// (src/unet_models.py:136-141, 400-403).  Configurations 27 / 28 of msc_conv_igemm (igemm.hip validates and dispatches here).
#include <stdlib.h>

#include <type_traits>

#include "conv_common.h"
#include "msc_internal.h"

namespace msc_conv {
namespace {

// ------------------------------------------------------------------------------------------------ halo tile
// 3x3 / stride 1 / pad 1 convolutions of the 32-channel full-resolution layers (dec0 and its data gradient): as an
// implicit GEMM every filter tap re-reads the pixel's 64-byte channel row from L2, and 64-byte row segments are the
// slow LDS-DMA case -- the DMA kernel sits at ~12 TB/s of fill with the MFMA pipes idle.  Here a block owns a 16x16
// pixel patch: the 18x18 halo of input rows goes to LDS once (20 KB), the nine taps read shifted windows of it, the
// 18 KB of weights stay in registers (every block reads the same ones from L2).  bf16, Cin = Cout = 32.
// Persistent over the patches (round 3): the weight fragments, the coefficients and the per-thread halo addressing are set up once per
// block, the filter orientation is a template parameter (the 36 shifted-window reads are immediates) and the epilogue has one
// straight-line body per use (statistics / fused final 1x1 / plain) behind uniform branches.  Before that the kernel issued 1142 VALU
// instructions per wave and patch for 72 MFMAs -- 4x the matrix time in address arithmetic and predicated options.
__device__ __forceinline__ int c32_key(int hx) { return ((0xFC30 >> hx) & 1) << 1; }      // halo column 0..17 -> chunk permutation of the 64-byte pixel

template <typename T, bool FLIP>
__global__ __launch_bounds__(256, 2) void conv3x3_c32_halo_kernel(ConvK p, int npatch) {      // 2 blocks per CU: 256 VGPRs
    constexpr int HCH = (18 * 18 * 4 + 255) / 256;   // halo chunks per thread (the last one of most threads is past the end)
    constexpr int HBUF = HCH * 256;                  // chunks per halo buffer, padded to whole DMA instructions
    __shared__ uint4 halo2[2 * HBUF];                // two halo buffers of [18][18] pixels x 4 chunks of 8 channels
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 16;
    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    // weights [Cout][3][3][Cin]: lane (g, pl) of fragment a holds input channels 8g..8g+7 of ONE output channel; row pl of
    // fragment a is output channel 8*(pl>>2) + 4a + (pl&3), so that the D rows a lane ends up with (4g..4g+3 of both
    // fragments) are the 8 consecutive channels 8g..8g+7 -> one 16-byte store per pixel
    const T* wt = reinterpret_cast<const T*>(p.wt);
    uint4 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            wf[t][a] = *reinterpret_cast<const uint4*>(wt + ((long)(8 * (pl >> 2) + 4 * a + (pl & 3)) * 9 + t) * 32 + g * 8);
    // the thread's halo chunks: position in the halo and byte offset from the patch's first pixel
    int hyx[HCH], hrel[HCH];                         // (hy << 16 | hx + 1); the tensor is below 2 GiB (conv_fill)
#pragma unroll
    for (int i = 0; i < HCH; ++i) {
        const int c = tid + 256 * i, pix = c >> 2;
        const int hy = pix / 18 - 1, hx = pix - (pix / 18) * 18 - 1;
        // (round 6) the pixel's four 16-byte chunks are XOR-permuted by c32_key(halo column): a pixel is 64 bytes, so four pixels share a 256-byte bank
        // window and the 16 lanes of a ds_read_b128 group (8 of channel group g, 8 of g ^ 1, 16 consecutive pixels) met two by two on every read of the
        // linear layout -- SQ_LDS_BANK_CONFLICT 45-48 % of the LDS cycles.  The key (0 or 2 per column, found by exhaustive search over the three column
        // shifts of the taps) makes every read conflict-free.
        hrel[i] = ((hy * p.Wi + hx) * (int)p.in_ld + ((c & 3) ^ c32_key(hx + 1)) * 8) * 2;
        hyx[i] = c >= 18 * 18 * 4 ? -(1 << 28) : hy * 65536 + hx + 1;       // past the end: never inside the image
    }
    T* out = reinterpret_cast<T*>(p.out);
    const T* res = reinterpret_cast<const T*>(p.res);
    const int c0 = 8 * g;
    const bool rlb = p.stats && p.stats_kind == 2;
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr, fin = p.fin_w != nullptr;
    // per-channel coefficients (scale, shift, the two rows of the final 1x1) live in LDS and are read per patch: in registers they
    // would be 32 more live values next to the 72 of the weights
    __shared__ float coef[4][32];
    if (tid < 128) {
        const int k = tid >> 5, ch = tid & 31;
        float v = k == 0 ? 1.f : 0.f;
        if (k == 0 && has_sc) v = p.scale[ch];
        if (k == 1 && has_sh) v = p.shift[ch];
        if (k >= 2 && fin) v = p.fin_w[(k - 2) * 32 + ch];
        coef[k][ch] = v;
    }
    __syncthreads();
    float bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = 0.f;
    // eval: the final 1x1 convolution 32 -> 2 + softmax on the values this epilogue stores (rounded to T first, as msc_final_fwd
    // would read them back): the lane's 8 channels against its 16 weights, then the four channel groups of a pixel (lanes pl,
    // pl+16, pl+32, pl+48) are folded by two cross-lane adds
    float fb0 = 0.f, fb1 = 0.f;
    if (fin && p.fin_b) { fb0 = p.fin_b[0]; fb1 = p.fin_b[1]; }
    int lbase[3];                                                 // halo chunk of (patch row 4*wid, pixel pl + dx), this lane's input channels, dx = -1, 0, 1
#pragma unroll
    for (int d = 0; d < 3; ++d) lbase[d] = ((wid * 4 + 1) * 18 + pl + d) * 4 + (g ^ c32_key(pl + d));
    const long hw = (long)p.Ho * p.Wo;

    // The halo goes HBM -> LDS by DMA (out-of-image chunks as out-of-range offsets: zeros), the NEXT patch's into the other buffer as
    // soon as this patch's has landed: the HBM latency of a patch hides behind the work of the one before it, and one barrier per patch
    // covers both "my halo is complete" and "everyone is done with the buffer the next DMA overwrites".
    auto request = [&](int patch, int buf) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int y0 = by * 16, x0 = bx * 16;
        const int org = (((n * p.Hi + y0) * p.Wi + x0) * (int)p.in_ld) * 2;
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
            const int hy = hyx[i] >> 16, hx = (hyx[i] & 0xffff) - 1;
            const bool ok = (unsigned)(y0 + hy) < (unsigned)p.Hi && (unsigned)(x0 + hx) < (unsigned)p.Wi;
            dma16(rx, reinterpret_cast<char*>(halo2 + buf * HBUF + 256 * i + 64 * wid), ok ? (unsigned)(org + hrel[i]) : OOB_OFF, 0);
        }
    };
    if ((int)blockIdx.x < npatch) request(blockIdx.x, 0);
    int buf = 0;
    for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x, buf ^= 1) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int y0 = by * 16, x0 = bx * 16;
        wait_vmcnt<0>();
        raw_barrier();
        if (patch + (int)gridDim.x < npatch) request(patch + gridDim.x, buf ^ 1);
        const uint4* halo = halo2 + buf * HBUF;
        // stats_kind 2: the activation rows the epilogue masks by are requested before the tap loop (nothing after it could hide them)
        uint4 ypre[4];
        if (rlb) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                ypre[b] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.sy) + ((long)(n * p.Ho + y0 + wid * 4 + b) * p.Wo + x0 + pl) * p.sy_ld + 8 * g);
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the fragments of tap t+1 are read while tap t is multiplied; the scheduling barrier keeps the compiler from hoisting all 36 reads
        // (144 registers) to the top
        uint4 bf[2][4];
        auto rd = [&](auto tt) {
            constexpr int t = decltype(tt)::value;
            constexpr int kh = t / 3, kw = t - kh * 3;
            constexpr int dy = FLIP ? 1 - kh : kh - 1, dx = FLIP ? 1 - kw : kw - 1;
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[t & 1][b] = halo[lbase[dx + 1] + (b + dy) * 18 * 4];
        };
        auto tap = [&](auto tt) {
            constexpr int t = decltype(tt)::value;
            if constexpr (t < 8) rd(std::integral_constant<int, (t < 8 ? t + 1 : 8)>{});
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a) Mma<T>::run(wf[t][a], bf[t & 1][b], acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        };
        rd(std::integral_constant<int, 0>{});
        tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
        tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
        tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
        // D: column = pixel x0+pl; rows 4g..4g+3 of fragments 0 and 1 = output channels 8g..8g+3 and 8g+4..8g+7
        const long opix0 = (long)(n * p.Ho + y0 + wid * 4) * p.Wo + x0 + pl;
        if (rlb) {                      // stats_kind 2: ReLU backward of the layer whose activation is sy, and its bias-gradient sums
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[8], yv[8];
                Vec16<T>::unpack(ypre[b], yv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = yv[j] > 0.f ? acc[j >> 2][b][j & 3] : 0.f;
                    bs[j] += v[j];
                }
                Vec16<T>::store(out + (opix0 + (long)b * p.Wo) * p.out_ld + c0, v);
            }
        } else {
            float fa0 = 0.f, fa1 = 0.f;
            int cofs = c0;
            asm volatile("" : "+v"(cofs));                   // opaque per patch: the reads below stay in the loop
            float sc[8], sh[8], fw0[8], fw1[8];
            if (has_sc || has_sh) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sc[j] = coef[0][cofs + j]; sh[j] = coef[1][cofs + j]; }
            }
            if (fin) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { fw0[j] = coef[2][cofs + j]; fw1[j] = coef[3][cofs + j]; }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long opix = opix0 + (long)b * p.Wo;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc[j >> 2][b][j & 3];
                if (has_sc) {
                    asm volatile("" ::: "memory");          // keeps the options as branches (conv_common.h)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
                } else if (has_sh) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = v[j] + sh[j];
                }
                if (res) {
                    float rv[8];
                    Vec16<T>::load(res + opix * p.res_ld + c0, rv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += rv[j];
                }
                if (p.relu) {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (!p.fin_skip) Vec16<T>::store(out + opix * p.out_ld + c0, v);
                if (fin) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const T tv = ElemIO<T>::from(v[j]);
                        const float r = ElemIO<T>::load(&tv);
                        a0 = fmaf(r, fw0[j], a0);
                        a1 = fmaf(r, fw1[j], a1);
                    }
                    a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64);
                    a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64);
                    // every lane group now holds the pixel's two logits: group b keeps row b, so the softmax below runs once for the four rows
                    if (g == b) { fa0 = a0; fa1 = a1; }
                }
            }
            if (fin) {
                const float a0 = fa0 + fb0, a1 = fa1 + fb1;
                const long o0 = (long)n * 2 * hw + (long)(y0 + wid * 4 + g) * p.Wo + x0 + pl;
                if (p.fin_logits) { p.fin_logits[o0] = a0; p.fin_logits[o0 + hw] = a1; }
                if (p.fin_probs) {      // numpy softmax of src/utils.py:231-273: subtract max, exp, divide by the sum
                    const float m = fmaxf(a0, a1);
                    const float e0 = expf(a0 - m), e1 = expf(a1 - m);
                    const float sden = e0 + e1;
                    p.fin_probs[o0] = e0 / sden; p.fin_probs[o0 + hw] = e1 / sden;
                }
            }
        }
    }
    if (rlb) {
        // fold the 16 pixel lanes of a channel group, then the four waves through LDS (the halo is no longer read), and add the
        // block's 32 sums (of all its patches) to this XCD's slot ([MSC_BN_SLOTS][32][2] doubles, first of each pair) with one coalesced atomic
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[j] = row16_sum(bs[j]);
        float* red = reinterpret_cast<float*>(halo2);
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wid * 32 + c0 + j] = bs[j];
        }
        __syncthreads();
        if (tid < 32) {
            const float a = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
            atomicAdd(p.stats + ((long)msc_xcc_id() * 32 + tid) * 2, (double)a);
        }
    }
}

// ConvTranspose2d(k4, s2, p1) 128 -> 32 channels (dec1's up-sampling to full resolution): per output-parity phase the DMA
// kernel re-reads four taps of 256-byte input rows for a 32-channel output tile.  Here a block owns 8x16 input pixels
// (16x32 outputs): their 10x18 halo goes to LDS once (46 KB), each phase's four taps of weights (32 KB) follow, both with
// the 16-byte chunks of a row XOR-swizzled by the pixel index so that 16 lanes reading 16 different pixels hit 16 bank groups.
template <typename T>
__global__ __launch_bounds__(512) void deconv4_c128_c32_halo_kernel(ConvK p) {      // one block of 8 waves per CU (two halo buffers): up to 256 VGPRs
    constexpr int HR = 10, HC = 18;                   // halo rows / columns
    constexpr int NCH = HR * HC * 16;                 // 16-byte chunks of a halo: [pixel][16 chunks of 8 channels], chunk ^= pixel & 15
    constexpr int HCH = (NCH + 511) / 512;            // ... per thread (DMA wave-instructions per wave)
    constexpr int HBUF = HCH * 512;                   // chunks per buffer, padded to whole instructions
    __shared__ uint4 halo2[2 * HBUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int tiles_x = p.Wi / 16, tiles_y = p.Hi / 8;
    const int npatch = p.N * tiles_x * tiles_y;
    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const T* wt = reinterpret_cast<const T*>(p.wt);          // [Cout][4][4][Cin]
    T* out = reinterpret_cast<T*>(p.out);
    const T* res = reinterpret_cast<const T*>(p.res);
    // Round 5: Cout = 32 G (dec2's 128 -> 128 up-sampling: G = 4).  A block keeps ONE group of 32 output channels -- block b works on group
    // b % G, so the weights of its phases still enter the registers once -- and walks the patches b / G, b / G + gridDim.x / G, ...; the four
    // groups of a patch read the same halo (from the L2: four blocks, not four passes of one block over 128 KB of weight fragments).
    const int G = p.Cout / 32, cg = (int)blockIdx.x % G, bidx = (int)blockIdx.x / G, nblk = (int)gridDim.x / G;
    const int c0 = 32 * cg + 8 * g;
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = 1.f; sh[j] = 0.f; }
    if (has_sc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = p.scale[c0 + j];
    }
    if (has_sh) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sh[j] = p.shift[c0 + j];
    }
    // Two waves = one output-parity phase (rows 0-3 and 4-7 of the 8 x 16 patch: two waves per SIMD, so that the epilogue and the
    // fragment reads of one overlap the MFMAs of the other).  The 4 taps x 128 input channels x 32 output channels of
    // its phase are 32 weight fragments = 128 VGPRs, fetched ONCE straight from L2 and held; only the pixel fragments come from
    // LDS (one read per two MFMAs).  With the weights in LDS too (the first version: one phase at a time for all waves) every
    // MFMA cost one fragment read and the LDS port, not the matrix pipe, set the pace.
    // Round 3: the block is PERSISTENT -- it walks patches blockIdx.x, + gridDim.x, ... with the weights of its phases kept in
    // registers (with one patch per block the 128 KB of weight fragments were re-fetched 4096 times per launch: 512 MB of L2 -> CU
    // traffic, as much as input and output together) -- and the halo of the next patch arrives by DMA in the second buffer while this
    // one is multiplied (before: a synchronous load between two barriers, with 25 VALU instructions of index arithmetic per chunk).
    // row pl of weight fragment a is output channel 8*(pl>>2) + 4a + (pl&3): a lane ends with channels 8g..8g+7
    const int ph = wid & 3, py = ph >> 1, px = ph & 1, b0 = (wid >> 2) * 4;
    const int kh0 = (py + 1) & 1, kw0 = (px + 1) & 1;                 // taps kh0, kh0+2 / kw0, kw0+2
    uint4 aw[4][4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kh = kh0 + 2 * (t >> 1), kw = kw0 + 2 * (t & 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int co = 32 * cg + 8 * (pl >> 2) + 4 * a + (pl & 3);
                aw[t][kk][a] = *reinterpret_cast<const uint4*>(wt + ((long)(co * 4 + kh) * 4 + kw) * 128 + (kk * 4 + g) * 8);
            }
    }
    // the thread's halo chunks: LDS slot c = 256 i + tid is chunk (c & 15) ^ ((pix & 7) << 1) of halo pixel pix = c >> 4.  (Round 6: the key was
    // pix & 15.  A ds_read_b128 lane group holds 8 lanes of channel group g and 8 of g ^ 1, reading 16 consecutive pixels; with the full pixel index as
    // key a lane of one half and a lane of the other whose pixels differ in bit 0 only fall on one slot -- SQ_LDS_BANK_CONFLICT 33 % of the LDS cycles.
    // Keyed on bits 1-3 alone, the slot's bit 0 is the channel group's: the two halves cannot meet, and 8 consecutive pixels differ in the key.)
    int hyx[HCH], hrel[HCH];                         // (hy << 16 | hx + 1), byte offset from the patch's first pixel (the tensor is below 2 GiB)
#pragma unroll
    for (int i = 0; i < HCH; ++i) {
        const int c = tid + 512 * i, pix = c >> 4;
        const int hy = pix / HC - 1, hx = pix - (pix / HC) * HC - 1;
        hrel[i] = ((hy * p.Wi + hx) * (int)p.in_ld + ((c & 15) ^ ((pix & 7) << 1)) * 8) * 2;
        hyx[i] = c >= NCH ? -(1 << 28) : hy * 65536 + hx + 1;          // past the end: never inside the image
    }
    auto request = [&](int patch, int buf) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int qy0 = by * 8, qx0 = bx * 16;
        const int org = (((n * p.Hi + qy0) * p.Wi + qx0) * (int)p.in_ld) * 2;
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
            const int hy = hyx[i] >> 16, hx = (hyx[i] & 0xffff) - 1;
            const bool ok = (unsigned)(qy0 + hy) < (unsigned)p.Hi && (unsigned)(qx0 + hx) < (unsigned)p.Wi;
            dma16(rx, reinterpret_cast<char*>(halo2 + buf * HBUF + 512 * i + 64 * wid), ok ? (unsigned)(org + hrel[i]) : OOB_OFF, 0);
        }
    };
    if (bidx < npatch) request(bidx, 0);
    int buf = 0;
    for (int patch = bidx; patch < npatch; patch += nblk, buf ^= 1) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int qy0 = by * 8, qx0 = bx * 16;
        wait_vmcnt<0>();                                              // this patch's halo has landed ...
        raw_barrier();                                                // ... for every wave, and everyone is done with the other buffer
        if (patch + nblk < npatch) request(patch + nblk, buf ^ 1);
        const uint4* halo = halo2 + buf * HBUF;
#pragma unroll 1
        for (int b = b0; b < b0 + 4; ++b) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kh = kh0 + 2 * (t >> 1), kw = kw0 + 2 * (t & 1);
                const int dy = (py + 1 - kh) / 2, dx = (px + 1 - kw) / 2;  // input offset of this tap: -1, 0 or +1
                const int pix = (b + 1 + dy) * HC + (pl + 1 + dx);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const uint4 bf = halo[pix * 16 + ((kk * 4 + g) ^ ((pix & 7) << 1))];
                    Mma<T>::run(aw[t][kk][0], bf, acc[0]);
                    Mma<T>::run(aw[t][kk][1], bf, acc[1]);
                }
            }
            const int oy = 2 * (qy0 + b) + py, ox = 2 * (qx0 + pl) + px;
            const long opix = (long)(n * p.Ho + oy) * p.Wo + ox;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[j >> 2][j & 3];
            if (has_sc) {
                asm volatile("" ::: "memory");                        // the options stay branches (conv_common.h)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
            } else if (has_sh) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += sh[j];
            }
            if (res) {
                float rv[8];
                Vec16<T>::load(res + opix * p.res_ld + c0, rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
            if (p.relu) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            Vec16<T>::store(out + opix * p.out_ld + c0, v);
        }
    }
}

// The stem: Conv2d(3, 64, k7, s2, p3) on the prepared input xp [N][H+6][W+8][4] (msc_stem_prepare: the image at (3,3), a zero fourth
// channel), which msc_conv_igemm sees as KH = 7, KW = 1, Cin = 32 (8 pixels x 4 channels of a row), stride 2, in_ld = 4, and the
// weights packed [64][7][32] (msc_stem_pack) -- src/unet_models.py:360 via torchvision's resnet conv1.  As an implicit GEMM that is
// seven 64-byte k-steps per tile (the slow LDS-DMA case) for 67 MB of output: 58-65 us.  Here a block owns 8 x 16 output pixels: the
// 21 x 38 input pixels under them (6.4 KB) go to LDS once per patch by DMA, double-buffered over the patches of a persistent block, and
// the B fragment of (kernel row kh, output pixel x, k-chunk g) is the 16 bytes at input row 2y + kh, pixel pair x + g -- 16-byte aligned
// whatever x, because the stride of 2 pixels is 16 bytes.  Eight waves: four pairs of output rows x two halves of the 64 channels,
// each with its 2 x 7 weight fragments in registers.
template <typename T>
__global__ __launch_bounds__(512) void stem7_halo_kernel(ConvK p, int npatch) {
    constexpr int HR = 21, HCK = 19;                  // halo rows; 16-byte chunks (pixel pairs) per row
    constexpr int NCH = HR * HCK;                     // 399 chunks
    constexpr int HBUF = 512;                         // chunks per buffer: one DMA instruction per wave
    __shared__ uint4 halo2[3 * HBUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int wp = wid >> 1, wc = wid & 1;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 8;
    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const T* wt = reinterpret_cast<const T*>(p.wt);          // [64][7][32]
    // row pl of weight fragment a is output channel 32 wc + 8*(pl>>2) + 4a + (pl&3): a lane ends with channels 32 wc + 8g .. + 7
    uint4 wf[7][2];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            wf[kh][a] = *reinterpret_cast<const uint4*>(wt + ((long)(32 * wc + 8 * (pl >> 2) + 4 * a + (pl & 3)) * 7 + kh) * 32 + g * 8);
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {      // used ahead of the loop: the compiler's wait for them stays out of it
        u32x4_t w0 = __builtin_bit_cast(u32x4_t, wf[kh][0]), w1 = __builtin_bit_cast(u32x4_t, wf[kh][1]);
        asm volatile("" : "+v"(w0), "+v"(w1));
        wf[kh][0] = __builtin_bit_cast(uint4, w0); wf[kh][1] = __builtin_bit_cast(uint4, w1);
    }
    // the thread's halo chunk: LDS slot c = tid is pixel pair c % 19 of halo row c / 19
    const int hr = tid / HCK, hj = tid - hr * HCK;
    const int hrel = (hr * p.Wi * 4 + hj * 8) * 2;            // bytes from the patch's first input pixel (in_ld = 4)
    auto request = [&](int patch, int buf) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int org = (((n * p.Hi + 16 * by) * p.Wi + 32 * bx) * 4) * 2;
        dma16(rx, reinterpret_cast<char*>(halo2 + buf * HBUF + 64 * wid), tid < NCH && patch < npatch ? (unsigned)(org + hrel) : OOB_OFF, 0);
    };
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // Round 6 (as down4_c32_halo_kernel below): three halo buffers, two patches in flight, every vector-memory instruction of the loop inline asm with a
    // counted wait, the epilogue written out here (BatchNorm statistics of the raw accumulators and / or scale, shift, ReLU; what msc_conv_cfg_ok admits
    // for configuration 58).  With the shared epilogue in the loop and one patch ahead, the wait at the top was vmcnt(0): for the halo AND for the stores
    // of the patch before.  Program order of a wave: prologue = two requests; patch i = the request of patch i+2 (past the end: out of range, same
    // count), then 2 stores -- "at most 1 outstanding" at the top of patch i proves its halo, older than the request of patch i+1, has landed.
    const int cb = 32 * wc + 8 * g;
    const bool has_sc = p.scale != nullptr, has_sh = p.shift != nullptr, k3 = p.stats != nullptr;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = has_sc ? p.scale[cb + j] : 1.f; sh[j] = has_sh ? p.shift[cb + j] : 0.f; }
    {
        float t = 0.f;                      // the coefficients are used ahead of the loop (their wait stays out of it)
#pragma unroll
        for (int j = 0; j < 8; ++j) t += sc[j] + sh[j];
        asm volatile("" ::"v"(t));
    }
    const u32x4_t ro = make_srd(p.out, (unsigned)((((long)p.M - 1) * p.out_ld + p.Cout) * 2));
    const int lbase = (4 * wp * HCK + pl + g);                // chunk of (kernel row 0, first of the wave's two output rows, pixel pl)
    const int step = (int)gridDim.x;
    request((int)blockIdx.x, 0);
    request((int)blockIdx.x + step, 1);
    int buf = 0;
    for (int patch = blockIdx.x; patch < npatch; patch += step, buf = buf == 2 ? 0 : buf + 1) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        wait_vmcnt<1>();
        raw_barrier();
        request(patch + 2 * step, buf == 0 ? 2 : buf - 1);
        const uint4* halo = halo2 + buf * HBUF;
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const uint4 bf = halo[lbase + (2 * b + kh) * HCK];
#pragma unroll
                for (int a = 0; a < 2; ++a) Mma<T>::run(wf[kh][a], bf, acc[a][b]);
            }
        const int m0 = (n * p.Ho + 8 * by) * p.Wo + 16 * bx;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float v[8];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[a * 4 + r] = acc[a][b][r];
            if (k3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
            }
            if (has_sc) {
                asm volatile("" ::: "memory");                        // the options stay branches (conv_common.h)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
            } else if (has_sh) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += sh[j];
            }
            if (p.relu) {
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            }
            bstore16(ro, (unsigned)((m0 + (wp * 2 + b) * p.Wo + pl) * (int)p.out_ld + cb) * 2u, Vec16<T>::pack(v));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.stats) {
        __syncthreads();
        conv_epilogue_stats<T, 2, 4, 2>(p, s1, s2, wp, wc, pl, 0, reinterpret_cast<float*>(halo2));
    }
}

// Conv2d(k4, s2, p1) 32 -> 128 channels on a full-resolution 32-channel map: the DATA GRADIENT of dec1's ConvTranspose2d 128 -> 32
// (src/unet_models.py:136-141; dX[ci](y, x) = sum_{co, kh, kw} dY[co](2y - 1 + kh, 2x - 1 + kw) W[ci][co][kh][kw]).  As an implicit GEMM it is
// sixteen 64-byte k-steps per tile -- the slow LDS-DMA case -- and ran 155 us for 268 MB of traffic (ResNet101, batch 32, 256x256: the launch
// furthest above its HBM floor in the train step, round 6).  Here a block owns 8 x 16 output pixels: the 18 x 34 input pixels under them go to
// LDS once per patch by DMA, two patches ahead of the one a persistent block multiplies, as TWO column-parity planes -- a tap (kh, kw) reads
// columns 2 x + kw, all of one parity, so in a plane the 16 pixels of a fragment are neighbours (64 bytes apart) and the 16-byte chunk of a
// pixel is XOR-swizzled by (column >> 1) & 2, which makes the sixteen lanes of every ds_read_b128 service group hit sixteen different bank
// quads for both column offsets (kw >> 1 = 0, 1; found by exhaustive search over the instruction's lane groups; SQ_LDS_BANK_CONFLICT = 0).
// Eight waves: two halves of the patch rows x four groups of 32 output channels, each wave with its 16 taps x 2 weight fragments in registers.
// EVERY vector-memory instruction of the loop is inline asm with counted waits (as in bottleneck.hip): with the shared epilogue of
// conv_common.h in the loop the compiler's own s_waitcnt vmcnt(0) -- for the mask loads, and conservatively at the head of the MFMA section --
// drained the halo requests it cannot see right after they were issued: 74 % of the wave cycles parked, 152 us (SQ_WAIT_ANY, first version).
// Epilogues: none, or stats_kind 2 (ReLU backward by [sy > 0] + bias-gradient sums); msc_conv_cfg_ok admits nothing else for configuration 59.
__device__ __forceinline__ void bload16(u32x4_t& r, u32x4_t srd, unsigned voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(srd) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_loaded4(u32x4_t (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [cnt] "n"(N) : "memory");
}

template <typename T>
__global__ __launch_bounds__(512) void down4_c32_halo_kernel(ConvK p, int npatch) {
    constexpr int HR = 18, PCOLS = 18, RU = 2 * PCOLS * 4;     // halo rows; columns per parity plane (17 used); 16-byte units per halo row
    constexpr int NU = HR * RU;                               // 2592 units per patch
    constexpr int XH = 6, HBUF = XH * 8 * 64;                 // DMA instructions per wave and patch; units per buffer (>= NU)
    static_assert(HBUF >= NU, "halo buffer");
    __shared__ uint4 halo2[3 * HBUF];      // three buffers: the patch being multiplied and the next two in flight
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int wp = wid >> 2, wc = wid & 3;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 8;
    const bool k2 = p.stats != nullptr;
    const u32x4_t rx = make_srd(p.in, p.in_bytes);
    const u32x4_t ro = make_srd(p.out, (unsigned)((((long)p.M - 1) * p.out_ld + p.Cout) * 2));
    const u32x4_t rs = make_srd(k2 ? p.sy : p.out, k2 ? (unsigned)((((long)p.M - 1) * p.sy_ld + p.Cout) * 2) : 16u);
    const T* wt = reinterpret_cast<const T*>(p.wt);          // [128][4][4][32]
    // row pl of weight fragment a is output channel 32 wc + 8*(pl>>2) + 4a + (pl&3): a lane ends with channels 32 wc + 8g .. + 7
    uint4 wf[16][2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            wf[t][a] = *reinterpret_cast<const uint4*>(wt + ((long)(32 * wc + 8 * (pl >> 2) + 4 * a + (pl & 3)) * 16 + t) * 32 + g * 8);
    // The fragments are USED here, ahead of the loop: left to their first use inside it, the compiler's wait for them (s_waitcnt vmcnt(31) ... vmcnt(0),
    // one per fragment) is part of the loop body and drains the halo requests and the stores of every patch, not just the first
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        u32x4_t w0 = __builtin_bit_cast(u32x4_t, wf[t][0]), w1 = __builtin_bit_cast(u32x4_t, wf[t][1]);
        asm volatile("" : "+v"(w0), "+v"(w1));
        wf[t][0] = __builtin_bit_cast(uint4, w0); wf[t][1] = __builtin_bit_cast(uint4, w1);
    }
    // the thread's halo units: LDS unit u = (i*8 + wid)*64 + lane is [halo row][column parity][plane column][chunk slot]; the slot holds channel
    // chunk slot ^ key(plane column).  The decode is redone per request (a dozen integer operations per unit) instead of living in 18 registers across
    // the loop.  A patch past the end is requested all the same, with every lane out of range: each iteration issues the same number of memory
    // instructions, so the counted waits are constants
    const int pix_bytes = (int)p.in_ld * 2;
    auto request = [&](int patch, int buf) {
        const bool live = patch < npatch;
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        const int iy0 = 16 * by - 1, ix0 = 32 * bx - 1;
        const int org = ((n * p.Hi + iy0) * p.Wi + ix0) * pix_bytes;      // of the halo's first pixel (outside the image on the top / left edge)
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));                                    // keeps the decode below inside the loop
#pragma unroll
        for (int i = 0; i < XH; ++i) {
            const int u = (i * 8 + wid) * 64 + lane_;
            const int r = u / RU, rem = u - r * RU;
            const int par = rem / (PCOLS * 4), rem2 = rem - par * (PCOLS * 4);
            const int col = rem2 >> 2, ch = (rem2 & 3) ^ ((col >> 1) & 2);
            const int cx = 2 * col + par;
            const bool ok = live && u < NU && col < 17 && (unsigned)(iy0 + r) < (unsigned)p.Hi && (unsigned)(ix0 + cx) < (unsigned)p.Wi;
            dma16(rx, reinterpret_cast<char*>(halo2 + buf * HBUF + (i * 8 + wid) * 64), ok ? (unsigned)(org + (r * p.Wi + cx) * pix_bytes + ch * 16) : OOB_OFF, 0);
        }
    };
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // unit of (halo row 0, parity 0, the lane's pixel at column offset s = kw >> 1, its k-chunk g)
    const int l0 = pl * 4 + (g ^ ((pl >> 1) & 2)), l1 = (pl + 1) * 4 + (g ^ (((pl + 1) >> 1) & 2));
    const int cb = 32 * wc + 8 * g;
    // Memory instructions of a wave in program order: prologue = halo requests of its first two patches (XH each); patch i = 4 mask loads (the
    // wave's four pixel rows; out of range when there is no mask), XH halo requests of patch i+2, then 4 stores.  Loads complete in order among
    // loads, so "at most XH outstanding" proves everything older than the youngest XH loads has landed -- stores in flight only make the wait stricter:
    //   top of patch i:       its halo is older than the requests of patch i+1           -> vmcnt(XH)
    //   first epilogue of i:  the mask loads are older than the requests of patch i+2    -> vmcnt(XH)
    const int step = (int)gridDim.x;
    request(blockIdx.x, 0);
    request(blockIdx.x + step, 1);
    int buf = 0;
    for (int patch = blockIdx.x; patch < npatch; patch += step, buf = buf == 2 ? 0 : buf + 1) {
        const int bx = patch % tiles_x, by = (patch / tiles_x) % tiles_y, n = patch / (tiles_x * tiles_y);
        wait_vmcnt<XH>();
        raw_barrier();
        const int m0 = (n * p.Ho + 8 * by) * p.Wo + 16 * bx;
        u32x4_t sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bload16(sv[q], rs, k2 ? (unsigned)((m0 + (wp * 4 + q) * p.Wo + pl) * (int)p.sy_ld + cb) * 2u : OOB_OFF);
        request(patch + 2 * step, buf == 0 ? 2 : buf - 1);
        const uint4* halo = halo2 + buf * HBUF;
        // the wave's four patch rows in two passes of two: 16 accumulator registers beside the 128 of the weights
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            f32x4 acc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int lb = (kw >> 1) ? l1 : l0;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const uint4 bf = halo[lb + (2 * (wp * 4 + bh * 2 + b) + kh) * RU + (kw & 1) * (PCOLS * 4)];
#pragma unroll
                        for (int a = 0; a < 2; ++a) Mma<T>::run(wf[kh * 4 + kw][a], bf, acc[a][b]);
                    }
                }
            if (bh == 0) wait_loaded4<XH>(sv);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[8];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[a * 4 + r] = acc[a][b][r];
                if (k2) {                            // dh = acc * [sy > 0]; the bias gradient is the sum of what is stored
                    float yv[8];
                    const u32x4_t t = sv[bh * 2 + b];
                    Vec16<T>::unpack(make_uint4(t.x, t.y, t.z, t.w), yv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v[j] = yv[j] > 0.f ? v[j] : 0.f;
                        s1[j] += v[j];
                    }
                }
                bstore16(ro, (unsigned)((m0 + (wp * 4 + bh * 2 + b) * p.Wo + pl) * (int)p.out_ld + cb) * 2u, Vec16<T>::pack(v));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.stats) {
        __syncthreads();
        conv_epilogue_stats<T, 2, 2, 4>(p, s1, s2, wp, wc, pl, 0, reinterpret_cast<float*>(halo2));
    }
}

template <typename T>
int conv_launch(const ConvK& k, hipStream_t st) {
    const int patches = k.N * (k.Ho / 16) * (k.Wo / 16);
    static const int persist = [] { const char* e = getenv("MSC_C32_BLOCKS"); return e ? atoi(e) : 512; }();       // 0: one block per patch (measured 100 us; 512: 68, 1024: 71, 2048: 78)
    const dim3 grid(persist > 0 && patches > persist ? persist : patches);
    if (k.flip) hipLaunchKernelGGL((conv3x3_c32_halo_kernel<T, true>), grid, dim3(256), 0, st, k, patches);
    else hipLaunchKernelGGL((conv3x3_c32_halo_kernel<T, false>), grid, dim3(256), 0, st, k, patches);
    return msc_check_launch("conv3x3_c32_halo");
}

template <typename T>
int deconv_launch(const ConvK& k, hipStream_t st) {
    const int G = k.Cout / 32;                                  // groups of 32 output channels: block b -> group b % G
    const int patches = k.N * (k.Hi / 8) * (k.Wi / 16);
    static const int persist = [] { const char* e = getenv("MSC_DECONV_BLOCKS"); return e ? atoi(e) : 256; }();      // one resident block per CU (98 KB of LDS)
    int blocks = persist > 0 && patches * G > persist ? persist / G * G : patches * G;
    if (blocks < G) blocks = G;
    hipLaunchKernelGGL(deconv4_c128_c32_halo_kernel<T>, dim3(blocks), dim3(512), 0, st, k);
    return msc_check_launch("deconv4_c128_c32_halo");
}

template <typename T>
int stem_launch(const ConvK& k, hipStream_t st) {
    const int patches = k.N * (k.Ho / 8) * (k.Wo / 16);
    static const int persist = [] { const char* e = getenv("MSC_STEM_BLOCKS"); return e ? atoi(e) : 512; }();      // measured (round 6, three buffers + counted waits): 256: 27.8 us, 512: 25.7, 1024: 30.5; before: 0 (one block per patch) 52, 256: 36, 512: 39
    hipLaunchKernelGGL(stem7_halo_kernel<T>, dim3(persist > 0 && patches > persist ? persist : patches), dim3(512), 0, st, k, patches);
    return msc_check_launch("stem7_halo");
}

template <typename T>
int down_launch(const ConvK& k, hipStream_t st) {
    const int patches = k.N * (k.Ho / 8) * (k.Wo / 16);
    static const int persist = [] { const char* e = getenv("MSC_DOWN4_BLOCKS"); return e ? atoi(e) : 256; }();      // one resident block per CU (144 KB of LDS)
    hipLaunchKernelGGL(down4_c32_halo_kernel<T>, dim3(persist > 0 && patches > persist ? persist : patches), dim3(512), 0, st, k, patches);
    return msc_check_launch("down4_c32_halo");
}

}  // namespace

int halo32_stem_launch(const ConvK& k, int dtype, hipStream_t st) { return dtype == MSC_F16 ? stem_launch<f16_t>(k, st) : stem_launch<bf16_t>(k, st); }
int halo32_conv_launch(const ConvK& k, int dtype, hipStream_t st) { return dtype == MSC_F16 ? conv_launch<f16_t>(k, st) : conv_launch<bf16_t>(k, st); }
int halo32_deconv_launch(const ConvK& k, int dtype, hipStream_t st) { return dtype == MSC_F16 ? deconv_launch<f16_t>(k, st) : deconv_launch<bf16_t>(k, st); }
int halo32_down_launch(const ConvK& k, int dtype, hipStream_t st) { return dtype == MSC_F16 ? down_launch<f16_t>(k, st) : down_launch<bf16_t>(k, st); }

}  // namespace msc_conv
