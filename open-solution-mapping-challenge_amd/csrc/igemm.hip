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
#include "common.h"
#include "msc_internal.h"

namespace {

struct ConvK {
    const char* in; const char* wt; char* out; const char* res;
    const float* scale; const float* shift; float* stats;
    long in_ld, out_ld, res_ld;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, flip, relu;
    int M, Hq, Wq;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

constexpr int ROWB = 80;  // LDS row pitch: 64 B of K + 16 B pad (16-B slots 5r+g mod 16 spread rows)

template <typename T, int TP, int TC, int WP, int WC, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
    constexpr int ES = sizeof(T);
    constexpr int KE = 64 / ES;   // K elements per step
    constexpr int CE = 16 / ES;   // elements per 16-byte chunk
    constexpr int WTP = TP / WP, WTC = TC / WC;
    constexpr int FM = WTC / 16, FN = WTP / 16;
    constexpr int NV = FM * 4;
    constexpr int XI = TP / 64, WI = (TC + 63) / 64;
    static_assert(WP * WC == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) char smem[2 * (TP + TC) * ROWB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * TP;
    const int c0 = blockIdx.y * TC;
    const int ph = MODE ? (int)blockIdx.z : 0;
    const int py = ph >> 1, px = ph & 1;

    int kh0 = 0, kw0 = 0, nkh = p.KH, nkw = p.KW;
    if (MODE) {
        kh0 = (py + p.pad) & 1; kw0 = (px + p.pad) & 1;
        nkh = p.KH > kh0 ? (p.KH - kh0 + 1) / 2 : 0;
        nkw = p.KW > kw0 ? (p.KW - kw0 + 1) / 2 : 0;
    }
    const int cps = p.Cin / KE;            // k-steps per tap
    const int nsteps = nkh * nkw * cps;

    // ---- per-thread load coordinates
    const int lrow = tid >> 2, kc = tid & 3;
    int xn[XI], xby[XI], xbx[XI];
    bool xv[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + lrow + i * 64;
        xv[i] = m < p.M;
        const int mm = xv[i] ? m : 0;
        const int n = mm / (p.Hq * p.Wq);
        const int rem = mm - n * (p.Hq * p.Wq);
        const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
        xn[i] = n * p.Hi;
        xby[i] = MODE ? qy : qy * p.stride;
        xbx[i] = MODE ? qx : qx * p.stride;
    }
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wt = reinterpret_cast<const T*>(p.wt);

    uint4 xr[XI], wr[WI];
    auto gload = [&](int s) {
        const int tap = s / cps;
        const int cch = s - tap * cps;
        const int khi = tap / nkw, kwi = tap - khi * nkw;
        const int kh = MODE ? kh0 + 2 * khi : khi;
        const int kw = MODE ? kw0 + 2 * kwi : kwi;
        int dy, dx;
        if (MODE) { dy = (py + p.pad - kh) / 2; dx = (px + p.pad - kw) / 2; }
        else if (p.flip) { dy = p.pad - kh; dx = p.pad - kw; }
        else { dy = kh - p.pad; dx = kw - p.pad; }
        const int coff = cch * KE + kc * CE;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int iy = xby[i] + dy, ix = xbx[i] + dx;
            const bool ok = xv[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(in + ((long)(xn[i] + iy) * p.Wi + ix) * p.in_ld + coff);
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = lrow + i * 64;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((TC >= 64 || row < TC) && c0 + row < p.Cout)
                v = *reinterpret_cast<const uint4*>(wt + ((long)((c0 + row) * p.KH + kh) * p.KW + kw) * p.Cin + coff);
            wr[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        char* sx = smem + buf * (TP + TC) * ROWB;
        char* sw = sx + TP * ROWB;
#pragma unroll
        for (int i = 0; i < XI; ++i) *reinterpret_cast<uint4*>(sx + (lrow + i * 64) * ROWB + kc * 16) = xr[i];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int row = lrow + i * 64;
            if (TC >= 64 || row < TC) *reinterpret_cast<uint4*>(sw + row * ROWB + kc * 16) = wr[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) gload(s + 1);
            const char* sx = smem + (s & 1) * (TP + TC) * ROWB;
            const char* sw = sx + TP * ROWB;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                // fragment row i=pl of fragment a carries local channel (i>>2)*NV + a*4 + (i&3)
                const int row = wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3);
                af[a] = *reinterpret_cast<const uint4*>(sw + row * ROWB + g * 16);
            }
#pragma unroll
            for (int b = 0; b < FN; ++b)
                bf[b] = *reinterpret_cast<const uint4*>(sx + (wp * WTP + b * 16 + pl) * ROWB + g * 16);
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) Mma<T>::run(af[a], bf[b], acc[a][b]);
            if (s + 1 < nsteps) lstore((s + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds channels cb .. cb+NV-1 of pixel rows (b*16+pl), b < FN
    const int cb = c0 + wc * WTC + g * NV;
    float sc[NV], sh[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        sc[j] = p.scale ? p.scale[cb + j] : 1.f;
        sh[j] = p.shift ? p.shift[cb + j] : 0.f;
    }
    float s1[NV], s2[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    T* out = reinterpret_cast<T*>(p.out);
    const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int m = m0 + wp * WTP + b * 16 + pl;
        float v[NV];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a * 4 + r] = acc[a][b][r];
        if (p.stats) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
        if (m < p.M) {
            long opix = m;
            if (MODE) {
                const int n = m / (p.Hq * p.Wq);
                const int rem = m - n * (p.Hq * p.Wq);
                const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
                opix = ((long)n * p.Ho + 2 * qy + py) * p.Wo + 2 * qx + px;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = v[j] * sc[j] + sh[j];
            if (res) {
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    float rv[CE];
                    Vec16<T>::load(res + opix * p.res_ld + cb + j, rv);
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[j + e] += rv[e];
                }
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = fmaxf(v[j], 0.f);
            }
#pragma unroll
            for (int j = 0; j < NV; j += CE) Vec16<T>::store(out + opix * p.out_ld + cb + j, v + j);
        }
    }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        }
        if (pl == 0) {
            // layout [Cout][slices][2]: msc_bn_finalize gives each channel one wavefront over its slices
            const long nsl = (long)gridDim.x * WP, sl = (long)blockIdx.x * WP + wp;
#pragma unroll
            for (int j = 0; j < NV; ++j) *reinterpret_cast<float2*>(p.stats + ((cb + j) * nsl + sl) * 2) = make_float2(s1[j], s2[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient:  dW[a][kh][kw][b] += sum_m P[m][a] * Q[pix(m)*stride - pad + (kh,kw)][b]
//   conv  wgrad: P = dY (a = cout), Q = X  (b = cin)
//   convT wgrad: P = X  (a = cin, coarse grid), Q = dOut (b = cout, fine grid), stride 2
// GEMM K = pixels, which is the strided dimension of NHWC: both operands are staged
// pixel-major in LDS and the k-contiguous fragments are gathered element-wise from LDS.
struct WgK {
    const char* p; const char* q; float* dw;
    long p_ld, q_ld;
    int N, Hp, Wp, A, Hq, Wq, B, KH, KW, stride, pad;
    int M, mchunk, tiles_b;
};

template <typename T, int TA, int TB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgK p) {
    constexpr int ES = sizeof(T);
    constexpr int KP = 64 / ES;      // pixels per k-step
    constexpr int CE = 16 / ES;
    constexpr int WTA = TA / 2, WTB = TB / 2;
    constexpr int FM = WTA / 16, FN = WTB / 16;
    constexpr int CPA = TA / CE, CPB = TB / CE;            // 16-B chunks per pixel row
    constexpr int PI = (KP * CPA + 255) / 256, QI = (KP * CPB + 255) / 256;
    constexpr int LDA = TA * ES + 16, LDB = TB * ES + 16;  // LDS row pitch (bytes)
    constexpr int BUF = KP * (LDA + LDB);
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wa = wid >> 1, wb = wid & 1;
    const int g = lane >> 4, pl = lane & 15;
    const int ta = blockIdx.x / p.tiles_b, tb = blockIdx.x - ta * p.tiles_b;
    const int a0 = ta * TA, b0 = tb * TB;
    const int tap = blockIdx.y;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int mbeg = blockIdx.z * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int nsteps = mend > mbeg ? (mend - mbeg + KP - 1) / KP : 0;

    const T* P = reinterpret_cast<const T*>(p.p);
    const T* Q = reinterpret_cast<const T*>(p.q);
    uint4 pr[PI], qr[QI];
    auto gload = [&](int s) {
        const int mb = mbeg + s * KP;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPA, cc = c - row * CPA;
            const int m = mb + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < KP && m < mend) v = *reinterpret_cast<const uint4*>(P + (long)m * p.p_ld + a0 + cc * CE);
            pr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPB, cc = c - row * CPB;
            const int m = mb + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < KP && m < mend) {
                const int n = m / (p.Hp * p.Wp);
                const int rem = m - n * (p.Hp * p.Wp);
                const int y = rem / p.Wp, x = rem - y * p.Wp;
                const int iy = y * p.stride - p.pad + kh, ix = x * p.stride - p.pad + kw;
                if ((unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq)
                    v = *reinterpret_cast<const uint4*>(Q + ((long)(n * p.Hq + iy) * p.Wq + ix) * p.q_ld + b0 + cc * CE);
            }
            qr[i] = v;
        }
    };
    auto lstore = [&](int buf) {
        char* sp = smem + buf * BUF;
        char* sq = sp + KP * LDA;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPA, cc = c - row * CPA;
            if (row < KP) *reinterpret_cast<uint4*>(sp + row * LDA + cc * 16) = pr[i];
        }
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPB, cc = c - row * CPB;
            if (row < KP) *reinterpret_cast<uint4*>(sq + row * LDB + cc * 16) = qr[i];
        }
    };
    // k-contiguous fragment of channel `ch` (local) for lane group g, gathered from a pixel-major tile
    auto frag = [&](const char* base, int ld, int ch) -> uint4 {
        uint4 r;
        if (ES == 2) {
            const uint16_t* s = reinterpret_cast<const uint16_t*>(base) + ch;
            const int ldh = ld / 2;
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = s[(8 * g + 2 * j) * ldh];
                const uint32_t hi = s[(8 * g + 2 * j + 1) * ldh];
                w[j] = lo | (hi << 16);
            }
            r = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            const uint32_t* s = reinterpret_cast<const uint32_t*>(base) + ch;
            const int ldw = ld / 4;
            r = make_uint4(s[(4 * g) * ldw], s[(4 * g + 1) * ldw], s[(4 * g + 2) * ldw], s[(4 * g + 3) * ldw]);
        }
        return r;
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) gload(s + 1);
            const char* sp = smem + (s & 1) * BUF;
            const char* sq = sp + KP * LDA;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) af[a] = frag(sp, LDA, wa * WTA + a * 16 + pl);
#pragma unroll
            for (int b = 0; b < FN; ++b) bf[b] = frag(sq, LDB, wb * WTB + b * 16 + pl);
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) Mma<T>::run(af[a], bf[b], acc[a][b]);
            if (s + 1 < nsteps) lstore((s + 1) & 1);
            __syncthreads();
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
                    if (ia < p.A && ib < p.B) atomicAdd(p.dw + (ia * taps + tap) * p.B + ib, acc[a][b][r]);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 of the same kernel: operands go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per
// wave-instruction) instead of through VGPRs + ds_write_b128, which removes the LDS-write pass that
// bounded v1 (ds_write_b128 sustains ~79 B/clk/CU: 16 KiB per k-step = ~200 cycles against ~257
// cycles of MFMA).  An LDS-DMA destination is wave-uniform base + lane*16, so the tile image is
// linear [row][64 B]; bank conflicts of the ds_read_b128 fragment reads are removed by permuting
// the 16-byte k-chunks of every row on the SOURCE side (lane loads chunk slot^S[q]) and applying
// the same involution on the read: q = (row>>2)&3 for pixel rows, (row/NV)&3 for weight rows,
// S = {0,2,3,1}, which makes the four lane groups of ds_read_b128 hit 16 distinct 16-byte slots.
// Out-of-image taps load from a 64-byte zero page instead of being zero-filled in registers.
template <typename T, int TP, int TC, int WP, int WC, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_dma_kernel(ConvK p, const char* __restrict__ zero_page) {
    constexpr int ES = sizeof(T);
    constexpr int KE = 64 / ES;
    constexpr int CE = 16 / ES;
    constexpr int WTP = TP / WP, WTC = TC / WC;
    constexpr int FM = WTC / 16, FN = WTP / 16;
    constexpr int NV = FM * 4;
    constexpr int XI = TP / 64, WI = (TC + 63) / 64;
    constexpr int BUF = (TP + TC) * 64;
    static_assert(WP * WC == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / WC, wc = wid % WC;
    const int g = lane >> 4, pl = lane & 15;
    const int m0 = blockIdx.x * TP;
    const int c0 = blockIdx.y * TC;
    const int ph = MODE ? (int)blockIdx.z : 0;
    const int py = ph >> 1, px = ph & 1;
    auto swz = [](int q) { return (0x1320 >> (4 * (q & 3))) & 3; };   // S = {0,2,3,1}

    int kh0 = 0, kw0 = 0, nkh = p.KH, nkw = p.KW;
    if (MODE) {
        kh0 = (py + p.pad) & 1; kw0 = (px + p.pad) & 1;
        nkh = p.KH > kh0 ? (p.KH - kh0 + 1) / 2 : 0;
        nkw = p.KW > kw0 ? (p.KW - kw0 + 1) / 2 : 0;
    }
    const int cps = p.Cin / KE;
    const int nsteps = nkh * nkw * cps;

    const int lrow = tid >> 2, slot = tid & 3;
    const int kcx = slot ^ swz(lrow >> 2);           // source k-chunk this lane fetches for pixel rows
    const int kcw = slot ^ swz(lrow / NV);           // ... and for weight rows
    int xn[XI], xby[XI], xbx[XI];
    bool xv[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + lrow + i * 64;
        xv[i] = m < p.M;
        const int mm = xv[i] ? m : 0;
        const int n = mm / (p.Hq * p.Wq);
        const int rem = mm - n * (p.Hq * p.Wq);
        const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
        xn[i] = n * p.Hi;
        xby[i] = MODE ? qy : qy * p.stride;
        xbx[i] = MODE ? qx : qx * p.stride;
    }
    const T* in = reinterpret_cast<const T*>(p.in);
    const T* wt = reinterpret_cast<const T*>(p.wt);
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    auto issue = [&](int s, int buf) {
        const int tap = s / cps;
        const int cch = s - tap * cps;
        const int khi = tap / nkw, kwi = tap - khi * nkw;
        const int kh = MODE ? kh0 + 2 * khi : khi;
        const int kw = MODE ? kw0 + 2 * kwi : kwi;
        int dy, dx;
        if (MODE) { dy = (py + p.pad - kh) / 2; dx = (px + p.pad - kw) / 2; }
        else if (p.flip) { dy = p.pad - kh; dx = p.pad - kw; }
        else { dy = kh - p.pad; dx = kw - p.pad; }
        char* sx = smem + buf * BUF;
        char* sw = sx + TP * 64;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int iy = xby[i] + dy, ix = xbx[i] + dx;
            const bool ok = xv[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            const char* src = ok ? reinterpret_cast<const char*>(in + ((long)(xn[i] + iy) * p.Wi + ix) * p.in_ld + cch * KE + kcx * CE)
                                 : zero_page;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sx + (i * 4 + wid) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            if (TC >= 64 || wid < TC / 16) {       // wave-uniform: a 32-row weight tile is filled by waves 0 and 1
                const int row = lrow + i * 64;
                const bool ok = c0 + row < p.Cout;
                const char* src = ok ? reinterpret_cast<const char*>(wt + ((long)((c0 + row) * p.KH + kh) * p.KW + kw) * p.Cin + cch * KE + kcw * CE)
                                     : zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sw + (i * 4 + wid) * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int rslot = (g ^ swz(pl >> 2)) * 16;       // both fragment kinds: (row>>2)&3 resp. (row/NV)&3 equals pl>>2
    if (nsteps > 0) {
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
            const char* sx = smem + (s & 1) * BUF;
            const char* sw = sx + TP * 64;
            uint4 af[FM], bf[FN];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                const int row = wc * WTC + (pl >> 2) * NV + a * 4 + (pl & 3);
                af[a] = *reinterpret_cast<const uint4*>(sw + row * 64 + rslot);
            }
#pragma unroll
            for (int b = 0; b < FN; ++b)
                bf[b] = *reinterpret_cast<const uint4*>(sx + (wp * WTP + b * 16 + pl) * 64 + rslot);
#pragma unroll
            for (int a = 0; a < FM; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) Mma<T>::run(af[a], bf[b], acc[a][b]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile has landed in the other buffer
            __syncthreads();
        }
    }

    const int cb = c0 + wc * WTC + g * NV;
    float sc[NV], sh[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        sc[j] = p.scale ? p.scale[cb + j] : 1.f;
        sh[j] = p.shift ? p.shift[cb + j] : 0.f;
    }
    float s1[NV], s2[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    T* out = reinterpret_cast<T*>(p.out);
    const T* res = reinterpret_cast<const T*>(p.res);
#pragma unroll
    for (int b = 0; b < FN; ++b) {
        const int m = m0 + wp * WTP + b * 16 + pl;
        float v[NV];
#pragma unroll
        for (int a = 0; a < FM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a * 4 + r] = acc[a][b][r];
        if (p.stats) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        }
        if (m < p.M) {
            long opix = m;
            if (MODE) {
                const int n = m / (p.Hq * p.Wq);
                const int rem = m - n * (p.Hq * p.Wq);
                const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
                opix = ((long)n * p.Ho + 2 * qy + py) * p.Wo + 2 * qx + px;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = v[j] * sc[j] + sh[j];
            if (res) {
#pragma unroll
                for (int j = 0; j < NV; j += CE) {
                    float rv[CE];
                    Vec16<T>::load(res + opix * p.res_ld + cb + j, rv);
#pragma unroll
                    for (int e = 0; e < CE; ++e) v[j + e] += rv[e];
                }
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j] = fmaxf(v[j], 0.f);
            }
#pragma unroll
            for (int j = 0; j < NV; j += CE) Vec16<T>::store(out + opix * p.out_ld + cb + j, v + j);
        }
    }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[j] += __shfl_xor(s1[j], o, 64);
                s2[j] += __shfl_xor(s2[j], o, 64);
            }
        }
        if (pl == 0) {
            const long nsl = (long)gridDim.x * WP, sl = (long)blockIdx.x * WP + wp;
#pragma unroll
            for (int j = 0; j < NV; ++j) *reinterpret_cast<float2*>(p.stats + ((cb + j) * nsl + sl) * 2) = make_float2(s1[j], s2[j]);
        }
    }
}

// 64-byte zero page per device for the out-of-image taps of the DMA kernels (allocated on first use,
// never freed: the only piece of library-owned device memory)
const char* zero_page_for_current_device() {
    static const char* pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void* ptr = nullptr;
        if (hipMalloc(&ptr, 256) != hipSuccess) return nullptr;
        if (hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = (const char*)ptr;
    }
    return pages[dev];
}

bool use_v1_conv() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MSC_CONV_V1"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

template <typename T, int TP, int TC, int WP, int WC>
int launch_conv(const ConvK& k, int mode, hipStream_t st) {
    dim3 grid(ceil_div(k.M, TP), ceil_div(k.Cout, TC), mode ? 4 : 1);
    if (use_v1_conv()) {
        if (mode) hipLaunchKernelGGL((conv_igemm_kernel<T, TP, TC, WP, WC, 1>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((conv_igemm_kernel<T, TP, TC, WP, WC, 0>), grid, dim3(256), 0, st, k);
        return msc_check_launch("conv_igemm");
    }
    const char* zp = zero_page_for_current_device();
    if (!zp) return msc_fail(MSC_ERR_HIP, "conv_igemm: cannot allocate the zero page");
    if (mode) hipLaunchKernelGGL((conv_igemm_dma_kernel<T, TP, TC, WP, WC, 1>), grid, dim3(256), 0, st, k, zp);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<T, TP, TC, WP, WC, 0>), grid, dim3(256), 0, st, k, zp);
    return msc_check_launch("conv_igemm_dma");
}

// tile choice: TC follows Cout (128 / 64 / 32); the pixel tile shrinks when the launch would not
// fill the 256 CUs.
void pick_tile(int M, int Cout, int* tp, int* tc) {
    if (Cout % 128 == 0) {
        *tc = 128;
        *tp = ((long)ceil_div(M, 128) * (Cout / 128) >= 512) ? 128 : 64;
        if (*tp == 64) *tc = 64;
    } else if (Cout % 64 == 0) {
        *tc = 64;
        *tp = ((long)ceil_div(M, 128) * (Cout / 64) >= 512) ? 128 : 64;
    } else {
        *tc = 32;
        *tp = 256;
    }
}

template <typename T>
int conv_dispatch(const ConvK& k, int mode, hipStream_t st) {
    int tp, tc;
    pick_tile(k.M, k.Cout, &tp, &tc);
    if (tc == 128) return launch_conv<T, 128, 128, 2, 2>(k, mode, st);
    if (tc == 64 && tp == 128) return launch_conv<T, 128, 64, 4, 1>(k, mode, st);
    if (tc == 64) return launch_conv<T, 64, 64, 2, 2>(k, mode, st);
    return launch_conv<T, 256, 32, 4, 1>(k, mode, st);
}

}  // namespace

static int conv_fill(const msc_conv_desc* d, ConvK* k) {
    if (!d || !d->in || !d->wt || !d->out) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: null pointer");
    const int es = d->dtype == MSC_BF16 ? 2 : 4;
    if (d->dtype != MSC_BF16 && d->dtype != MSC_F32) return msc_fail(MSC_ERR_ARG, "msc_conv_igemm: dtype %d", d->dtype);
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
    k->in_ld = d->in_ld; k->out_ld = d->out_ld; k->res_ld = d->res_ld;
    k->N = d->N; k->Hi = d->Hi; k->Wi = d->Wi; k->Cin = d->Cin; k->Ho = d->Ho; k->Wo = d->Wo; k->Cout = d->Cout;
    k->KH = d->KH; k->KW = d->KW; k->stride = d->stride; k->pad = d->pad; k->flip = d->flip; k->relu = d->relu;
    if (d->mode == 1) {
        if (d->stride != 2 || (d->Ho & 1) || (d->Wo & 1)) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: transposed mode needs stride 2 and even output size");
        if (d->stats) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_igemm: stats not available in transposed mode");
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
    return MSC_OK;
}

extern "C" int msc_conv_stats_slices(const msc_conv_desc* d) {
    ConvK k;
    if (conv_fill(d, &k) != MSC_OK) return -1;
    int tp, tc;
    pick_tile(k.M, k.Cout, &tp, &tc);
    const int wpx = (tc == 128 || (tc == 64 && tp == 64)) ? 2 : 4;
    return ceil_div(k.M, tp) * wpx;
}

extern "C" int msc_conv_igemm(const msc_conv_desc* d, void* stream) {
    ConvK k;
    int rc = conv_fill(d, &k);
    if (rc != MSC_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == MSC_BF16) return conv_dispatch<bf16_t>(k, d->mode, st);
    return conv_dispatch<float>(k, d->mode, st);
}

extern "C" int msc_conv_wgrad(const msc_wgrad_desc* d, void* stream) {
    if (!d || !d->p || !d->q || !d->dw) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: null pointer");
    if (d->dtype != MSC_BF16 && d->dtype != MSC_F32) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: dtype %d", d->dtype);
    const int es = d->dtype == MSC_BF16 ? 2 : 4;
    if (d->A % 32 || d->B % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_conv_wgrad: channel counts must be multiples of 32 (A=%d B=%d)", d->A, d->B);
    const bool q_ok = (d->q_ld * es) % 16 == 0 ||
                      (d->KW == 1 && d->pad == 0 && (d->stride * d->q_ld * es) % 16 == 0 && ((int64_t)d->Wq * d->q_ld * es) % 16 == 0);
    if ((d->p_ld * es) % 16 || !q_ok || (((uintptr_t)d->p | (uintptr_t)d->q) & 15))
        return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: operands must keep 16-byte alignment");
    WgK k;
    k.p = (const char*)d->p; k.q = (const char*)d->q; k.dw = d->dw; k.p_ld = d->p_ld; k.q_ld = d->q_ld;
    k.N = d->N; k.Hp = d->Hp; k.Wp = d->Wp; k.A = d->A; k.Hq = d->Hq; k.Wq = d->Wq; k.B = d->B;
    k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad;
    const long m = (long)d->N * d->Hp * d->Wp;
    if (m <= 0 || m > 0x7fffffffL) return msc_fail(MSC_ERR_ARG, "msc_conv_wgrad: bad pixel count %ld", m);
    k.M = (int)m;
    const int kp = 64 / es;
    const bool big = (d->A % 128 == 0) && (d->B % 128 == 0);
    const int ta = big ? 128 : (d->A % 64 == 0 ? 64 : 32), tbs = big ? 128 : (d->B % 64 == 0 ? 64 : 32);
    const int tiles = (d->A / ta) * (d->B / tbs) * d->KH * d->KW;
    // split the pixel (K) dimension until ~4 blocks per CU are in flight, at least 8 k-steps each
    int splits = ceil_div(1024, tiles);
    const int max_splits = ceil_div(k.M, 8 * kp);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int mchunk = ceil_div(k.M, splits);
    mchunk = ceil_div(mchunk, kp) * kp;
    splits = ceil_div(k.M, mchunk);
    k.mchunk = mchunk; k.tiles_b = d->B / tbs;
    dim3 grid((d->A / ta) * (d->B / tbs), d->KH * d->KW, splits);
    hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH(T, TA, TB) hipLaunchKernelGGL((conv_wgrad_kernel<T, TA, TB>), grid, dim3(256), 0, st, k)
    if (d->dtype == MSC_BF16) {
        if (big) WG_LAUNCH(bf16_t, 128, 128);
        else if (ta == 64 && tbs == 64) WG_LAUNCH(bf16_t, 64, 64);
        else if (ta == 64) WG_LAUNCH(bf16_t, 64, 32);
        else if (tbs == 64) WG_LAUNCH(bf16_t, 32, 64);
        else WG_LAUNCH(bf16_t, 32, 32);
    } else {
        if (big) WG_LAUNCH(float, 128, 128);
        else if (ta == 64 && tbs == 64) WG_LAUNCH(float, 64, 64);
        else if (ta == 64) WG_LAUNCH(float, 64, 32);
        else if (tbs == 64) WG_LAUNCH(float, 32, 64);
        else WG_LAUNCH(float, 32, 32);
    }
#undef WG_LAUNCH
    return msc_check_launch("conv_wgrad");
}
