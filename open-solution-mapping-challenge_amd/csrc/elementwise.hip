// HBM-bound kernels of the U-Net hot path on gfx950: weight packing, stem input preparation,
// MaxPool2d(2,2), BatchNorm2d training statistics / apply / backward, decoder ReLU backward, bias
// gradient, the final 1x1 conv fused with softmax, and Adam.  All activations NHWC with a channel
// stride; every global access is a 16-byte vector per lane (4 f32 / 8 bf16), lanes run along the
// channel (contiguous) dimension first.
#include <stdlib.h>

#include "common.h"
#include "msc_internal.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_grid(long long work) {
    long long b = (work + EW_THREADS - 1) / EW_THREADS;
    if (b > 256 * 16) b = 256 * 16;  // grid-stride beyond ~16 blocks per CU
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------ packing
template <typename T>
__global__ void pack_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        ElemIO<T>::store(dst + i, src[i]);
}

// [A][T][B] -> [B][T][A], tiled through LDS so both sides are coalesced
template <typename T>
__global__ void pack_transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, int A, int Tn, int B) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int a0 = blockIdx.y * 32, b0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, b = b0 + tx;
        tile[r][tx] = (a < A && b < B) ? src[((long)a * Tn + t) * B + b] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, a = a0 + tx;
        if (a < A && b < B) ElemIO<T>::store(dst + ((long)b * Tn + t) * A + a, tile[tx][r]);
    }
}

// conv1 weight [co][3][7][7] -> [co][7][8][4]
template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ w, T* __restrict__ dst, int cout) {
    const int n = cout * 7 * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ci = i & 3, kw = (i >> 2) & 7, kh = (i >> 5) % 7, co = i / (7 * 32);
        float v = 0.f;
        if (ci < 3 && kw < 7) v = w[((co * 3 + ci) * 7 + kh) * 7 + kw];
        ElemIO<T>::store(dst + i, v);
    }
}
__global__ void stem_unpack_grad_kernel(const float* __restrict__ dp, float* __restrict__ dw, int cout) {
    const int n = cout * 3 * 49;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int kw = i % 7, kh = (i / 7) % 7, ci = (i / 49) % 3, co = i / 147;
        dw[i] += dp[((co * 7 + kh) * 8 + kw) * 4 + ci];
    }
}

// x f32 NCHW [N,3,H,W] -> xp [N][H+6][W+8][4], image at (3,3), zero elsewhere
template <typename T>
__global__ void stem_prepare_kernel(const float* __restrict__ x, T* __restrict__ xp, int N, int H, int W) {
    const int Hp = H + 6, Wp = W + 8;
    const long total = (long)N * Hp * Wp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % Wp);
        const int py = (int)((i / Wp) % Hp);
        const int n = (int)(i / ((long)Wp * Hp));
        const int y = py - 3, xx = px - 3;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const long base = ((long)n * 3 * H + y) * W + xx;
            v[0] = x[base]; v[1] = x[base + (long)H * W]; v[2] = x[base + 2L * H * W];
        }
        T* d = xp + i * 4;
        if (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            d[0] = ElemIO<T>::from(v[0]); d[1] = ElemIO<T>::from(v[1]); d[2] = ElemIO<T>::from(v[2]); d[3] = ElemIO<T>::from(v[3]);
        }
    }
}

// ------------------------------------------------------------------ maxpool 2x2 / 2
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ in, long in_ld, T* __restrict__ out, long out_ld,
                                    int N, int Ho, int Wo, int C) {
    constexpr int CE = Vec16<T>::N;
    const int cv = C / CE;
    const long total = (long)N * Ho * Wo * cv;
    const int Wi = Wo * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * CE;
        const long pix = i / cv;
        const int ox = (int)(pix % Wo);
        const long row = pix / Wo;  // n*Ho + oy
        const long ibase = (row * 2 * Wi + 2 * ox);
        float a[CE], b[CE], m[CE];
        Vec16<T>::load(in + ibase * in_ld + c, m);
        Vec16<T>::load(in + (ibase + 1) * in_ld + c, a);
#pragma unroll
        for (int e = 0; e < CE; ++e) m[e] = a[e] > m[e] ? a[e] : m[e];
        Vec16<T>::load(in + (ibase + Wi) * in_ld + c, a);
        Vec16<T>::load(in + (ibase + Wi + 1) * in_ld + c, b);
#pragma unroll
        for (int e = 0; e < CE; ++e) { m[e] = a[e] > m[e] ? a[e] : m[e]; m[e] = b[e] > m[e] ? b[e] : m[e]; }
        Vec16<T>::store(out + pix * out_ld + c, m);
    }
}

template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ dout, long dout_ld, const T* __restrict__ in, long in_ld,
                                    T* __restrict__ din, long din_ld, int N, int Ho, int Wo, int C, int accumulate) {
    constexpr int CE = Vec16<T>::N;
    const int cv = C / CE;
    const long total = (long)N * Ho * Wo * cv;
    const int Wi = Wo * 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * CE;
        const long pix = i / cv;
        const int ox = (int)(pix % Wo);
        const long row = pix / Wo;
        const long ibase = (row * 2 * Wi + 2 * ox);
        const long off[4] = {ibase, ibase + 1, ibase + Wi, ibase + Wi + 1};
        float v[4][CE], g[CE];
#pragma unroll
        for (int k = 0; k < 4; ++k) Vec16<T>::load(in + off[k] * in_ld + c, v[k]);
        Vec16<T>::load(dout + pix * dout_ld + c, g);
        float o[4][CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            int best = 0;
            float bm = v[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k][e] > bm) { bm = v[k][e]; best = k; }  // first maximum wins
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][e] = (k == best) ? g[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (accumulate) {
                float old[CE];
                Vec16<T>::load(din + off[k] * din_ld + c, old);
#pragma unroll
                for (int e = 0; e < CE; ++e) o[k][e] += old[e];
            }
            Vec16<T>::store(din + off[k] * din_ld + c, o[k]);
        }
    }
}

// ------------------------------------------------------------------ BatchNorm (statistics kernels: reduce.hip)
// eval mode: fold running statistics into the conv epilogue, scale = gamma/sqrt(rv+eps), shift = beta - rm*scale
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = (gamma ? gamma[c] : 1.f) / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
    }
}

// BatchNorm2d training-mode apply / backward-apply with the statistics FINALISED IN THE PROLOGUE: the producer (conv epilogue
// or column reduce) left one partial sum pair per XCD and channel ([MSC_BN_SLOTS][C][2], common.h); a block owns CT channels
// x a pixel range, its first CT threads turn the 8 slots of their channel into the per-channel coefficients (double
// arithmetic, as the former finalize kernels did) and park them in LDS -- 208 dependent 6-8 us finalize launches per
// ResNet101 train step are gone.  The blocks of the first pixel range also publish what later kernels need (scale / shift /
// mean / invstd, running statistics; dgamma / dbeta).
// Block -> (pixel block, channel tile) of the channel-tiled BatchNorm kernels (1-D grid of npb * nct blocks).  XCD-aware (round 4): workgroup
// b runs on XCD b % 8; every XCD gets a CONTIGUOUS run of pixel blocks, channel tile fastest -- the same pixel ranges the convolution
// kernels give it (conv_igemm_dma_kernel's tile order).  The write-back at a kernel boundary leaves a producer's lines clean in ITS XCD's
// L2 (profiles/r4_run8_store_policy_ab.txt: dropping them costs 3 %); with pixel blocks dealt round-robin over the XCDs, as before, seven
// eighths of what a BatchNorm pass reads had been written through another XCD's L2 and came from the fabric.
__device__ __forceinline__ void bn_block(int nct, int xcd_order, int& pb, int& ct) {
    int wgid = (int)blockIdx.x;
    if (xcd_order) {
        const int nwg = (int)gridDim.x, xcd = wgid & 7, wq = nwg >> 3, wr = nwg & 7;
        const int cnt = wq + (xcd < wr ? 1 : 0), j = wgid >> 3;
        // xcd_order 2: the XCD's run is walked from its END.  The convolutions write a range front to back; of a tensor larger than the L2
        // (layer1 / layer2: 4-8 MB per XCD against 4 MB) only the tail is still resident when the BatchNorm pass starts -- which then
        // leaves ITS tail, the front of the range, for the next convolution to start on.  Measured: +0.05 ms per step (two A/B pairs,
        // 11.03-11.05 -> 11.08-11.10 ms); kept as MSC_BN_XCD=2, the default walks forward.
        wgid = (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (xcd_order == 2 ? cnt - 1 - j : j);
    } else {                      // the former 2-D grid: pixel block fastest
        const int npb = (int)gridDim.x / nct;
        pb = wgid % npb; ct = wgid / npb;
        return;
    }
    pb = wgid / nct;
    ct = wgid - pb * nct;
}

template <typename T, int CT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, long y_ld, const T* __restrict__ res, long res_ld,
                                                       T* __restrict__ out, long out_ld, BnFwdFin f, int relu, long pixels, int C, long ppb,
                                                       int xcd_order, uint8_t* __restrict__ mask, long mask_ld) {
    constexpr int CE = Vec16<T>::N, VC = CT / CE, R = 256 / VC;
    __shared__ float s_sc[CT], s_sh[CT];
    int pb, ct;
    bn_block(C / CT, xcd_order, pb, ct);
    const int tid = threadIdx.x, c0 = ct * CT;
    if (tid < CT) {
        const int c = c0 + tid;
        float sc, sh;
        if (f.slots) {
            bn_fwd_coeffs(f, C, c, pb == 0, sc, sh);
        } else {                        // coefficients given (no statistics to finalise)
            sc = f.scale[c]; sh = f.shift[c];
        }
        s_sc[tid] = sc; s_sh[tid] = sh;
    }
    __syncthreads();
    const int col = tid % VC, r = tid / VC;
    const int c = c0 + col * CE;
    float sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = s_sc[col * CE + e]; sh[e] = s_sh[col * CE + e]; }
    const long p1 = min(pixels, ((long)pb + 1) * ppb);
    for (long pix = (long)pb * ppb + r; pix < p1; pix += R) {
        float v[CE], rr[CE];
        Vec16<T>::load(y + pix * y_ld + c, v);
#pragma unroll
        for (int e = 0; e < CE; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        if (res) {
            Vec16<T>::load(res + pix * res_ld + c, rr);
#pragma unroll
            for (int e = 0; e < CE; ++e) v[e] += rr[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < CE; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const uint4 pk = Vec16<T>::pack(v);
        store16(out + pix * out_ld + c, pk);
        if (mask) {                  // ReLU mask of this lane's CE channels as one byte (bit e = channel c + e): what the backward reads instead of `out`
            // taken from the STORED (rounded) value, so that it is what [out > 0] gives (MSC_RELU_BITS=0 and the reference mask on the stored
            // activation): an fp16 value below 2^-25 rounds to 0 and must not count as active (round-4 advisory)
            float w[CE];
            Vec16<T>::unpack(pk, w);
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < CE; ++e) m |= (w[e] > 0.f ? 1u : 0u) << e;
            mask[pix * mask_ld + c / CE] = (uint8_t)m;
        }
    }
}

struct BnBwdFin {
    const double* slots; double count; const float* gamma; const float* save_mean; const float* save_invstd;
    float* dgamma; float* dbeta;
};

// RSUM (round 4): the launch also reduces (sum dh, sum dh*ry) into `rslots` -- the BatchNorm-backward sums of the layer that produced the
// RESIDUAL (the downsample branch of a stage's first block: its output gradient IS the dh this kernel writes to `dres`, its pre-BN tensor
// is ry), which a separate msc_bn_bwd_reduce launch would read back
template <typename T, int CT, bool RSUM>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, long dout_ld, const T* __restrict__ out, long out_ld,
                                                           const T* __restrict__ y, long y_ld, int relu, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, BnBwdFin f, T* __restrict__ dy, long dy_ld,
                                                           T* __restrict__ dres, long dres_ld, int dres_acc, long pixels, int C, long ppb,
                                                           int xcd_order, const T* __restrict__ ry, long ry_ld, double* __restrict__ rslots) {
    constexpr int CE = Vec16<T>::N, VC = CT / CE, R = 256 / VC;
    __shared__ float s_a[CT], s_b[CT], s_k[CT], s_sc[CT], s_sh[CT];
    __shared__ float rred[RSUM ? 256 * 2 * CE : 1];
    float t1[CE], t2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
    int pb, ct;
    bn_block(C / CT, xcd_order, pb, ct);
    const int tid = threadIdx.x, c0 = ct * CT;
    if (tid < CT) {
        const int c = c0 + tid;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int x = 0; x < MSC_BN_SLOTS; ++x) {
            const double2 v = *reinterpret_cast<const double2*>(f.slots + ((long)x * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
        const double mu = f.save_mean[c], is = f.save_invstd[c], g = f.gamma ? f.gamma[c] : 1.0;
        const double dbe = s1;                        // sum dh
        const double dga = is * (s2 - mu * s1);       // sum dh * xhat
        // dy = g*is*(dh - dbe/M - xhat*dga/M),  xhat = (y-mu)*is   ->   dy = a*dh + b*y + k0
        const double a = g * is;
        const double b = -g * is * is * dga / f.count;
        const double k0 = -g * is * dbe / f.count - b * mu;
        s_a[tid] = (float)a; s_b[tid] = (float)b; s_k[tid] = (float)k0;
        s_sc[tid] = relu == 2 ? scale[c] : 0.f;
        s_sh[tid] = relu == 2 ? shift[c] : 0.f;
        if (pb == 0) {
            if (f.dgamma) f.dgamma[c] += (float)dga;
            if (f.dbeta) f.dbeta[c] += (float)dbe;
        }
    }
    __syncthreads();
    const int col = tid % VC, r = tid / VC;
    const int c = c0 + col * CE;
    float ca[CE], cb[CE], ck[CE], sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        ca[e] = s_a[col * CE + e]; cb[e] = s_b[col * CE + e]; ck[e] = s_k[col * CE + e];
        sc[e] = s_sc[col * CE + e]; sh[e] = s_sh[col * CE + e];
    }
    const long p1 = min(pixels, ((long)pb + 1) * ppb);
    for (long pix = (long)pb * ppb + r; pix < p1; pix += R) {
        float d[CE], o[CE], yy[CE], rr[CE];
        Vec16<T>::load(dout + pix * dout_ld + c, d);
        Vec16<T>::load(y + pix * y_ld + c, yy);
        if (relu == 1) {
            Vec16<T>::load(out + pix * out_ld + c, o);
#pragma unroll
            for (int e = 0; e < CE; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        } else if (relu == 3) {      // `out` is the byte mask msc_bn_apply wrote (one byte per CE channels, out_ld bytes per pixel)
            const unsigned m = reinterpret_cast<const uint8_t*>(out)[pix * out_ld + c / CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) d[e] = ((m >> e) & 1u) ? d[e] : 0.f;
        } else if (relu == 2) {
#pragma unroll
            for (int e = 0; e < CE; ++e) d[e] = fmaf(yy[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
        }
        if (dres) {
            if (dres_acc) {
                Vec16<T>::load(dres + pix * dres_ld + c, rr);
#pragma unroll
                for (int e = 0; e < CE; ++e) rr[e] += d[e];
                Vec16<T>::store(dres + pix * dres_ld + c, rr);
            } else {
                Vec16<T>::store(dres + pix * dres_ld + c, d);
            }
        }
        if (RSUM) {
            float rv[CE];
            Vec16<T>::load(ry + pix * ry_ld + c, rv);
#pragma unroll
            for (int e = 0; e < CE; ++e) { t1[e] += d[e]; t2[e] = fmaf(d[e], rv[e], t2[e]); }
        }
#pragma unroll
        for (int e = 0; e < CE; ++e) yy[e] = ca[e] * d[e] + cb[e] * yy[e] + ck[e];
        Vec16<T>::store(dy + pix * dy_ld + c, yy);
    }
    if (RSUM) {
        float* mine = rred + tid * 2 * CE;
#pragma unroll
        for (int e = 0; e < CE; ++e) { mine[e] = t1[e]; mine[CE + e] = t2[e]; }
        __syncthreads();
        if (r == 0) {
            for (int k = 1; k < R; ++k) {
                const float* o = rred + (k * VC + col) * 2 * CE;
#pragma unroll
                for (int e = 0; e < CE; ++e) { t1[e] += o[e]; t2[e] += o[CE + e]; }
            }
        }
        __syncthreads();
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < CE; ++e) *reinterpret_cast<float2*>(rred + (col * CE + e) * 2) = make_float2(t1[e], t2[e]);
        }
        __syncthreads();
        double* slot = rslots + ((long)msc_xcc_id() * C + c0) * 2;
        for (int i = tid; i < CT * 2; i += 256) atomicAdd(slot + i, (double)rred[i]);
    }
}

// ---- round 4: the stem's BatchNorm + ReLU + MaxPool2d(2,2) (src/unet_models.py:360-363) without the full-resolution activation.
// relu(bn(y)) of the 7x7 conv is read by nothing but the pool (the stem's weight gradient reads the image, the BatchNorm backward takes
// its ReLU mask from y): forward = one pass y -> pooled (67 MB read + 17 MB written instead of 268 + 17); backward = the pool's gradient
// routing, the ReLU mask and the BatchNorm backward from (pooled gradient, y) in two passes (sums, apply) instead of three kernels over
// five full-resolution tensors.  The window's maximum is taken over the fp32 activations (first maximum wins, as torch; the stored
// pooled value is the rounded maximum = the maximum of the rounded values), and the backward recomputes the same comparison.
template <typename T, int CT>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const T* __restrict__ y, long y_ld, T* __restrict__ out, long out_ld, BnFwdFin f,
                                                            int N, int Ho, int Wo, int C, long ppb, int xcd_order) {
    constexpr int CE = Vec16<T>::N, VC = CT / CE, R = 256 / VC;
    __shared__ float s_sc[CT], s_sh[CT];
    int pb, ct;
    bn_block(C / CT, xcd_order, pb, ct);
    const int tid = threadIdx.x, c0 = ct * CT;
    if (tid < CT) {
        const int c = c0 + tid;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int x = 0; x < MSC_BN_SLOTS; ++x) {
            const double2 v = *reinterpret_cast<const double2*>(f.slots + ((long)x * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
        const double mean = s1 / f.count;
        double var = s2 / f.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
        const float sc = g * invstd, sh = b - (float)mean * sc;
        if (pb == 0) {
            f.scale[c] = sc;
            f.shift[c] = sh;
            if (f.save_mean) f.save_mean[c] = (float)mean;
            if (f.save_invstd) f.save_invstd[c] = invstd;
            if (f.running_mean) f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
            if (f.running_var) {
                const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
                f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
            }
        }
        s_sc[tid] = sc; s_sh[tid] = sh;
    }
    __syncthreads();
    const int col = tid % VC, r = tid / VC;
    const int c = c0 + col * CE;
    float sc[CE], sh[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { sc[e] = s_sc[col * CE + e]; sh[e] = s_sh[col * CE + e]; }
    const long total = (long)N * Ho * Wo, p1 = min(total, ((long)pb + 1) * ppb);
    const int Wi = 2 * Wo;
    for (long q = (long)pb * ppb + r; q < p1; q += R) {
        const long n = q / ((long)Ho * Wo), rem = q - n * ((long)Ho * Wo);
        const int py = (int)(rem / Wo), px = (int)(rem - (long)py * Wo);
        const long base = ((n * 2 * Ho + 2 * py) * Wi + 2 * px);
        float best[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) best[e] = 0.f;                 // relu: the maximum of the four max(v, 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[CE];
            Vec16<T>::load(y + (base + (k >> 1) * Wi + (k & 1)) * y_ld + c, v);
#pragma unroll
            for (int e = 0; e < CE; ++e) best[e] = fmaxf(best[e], fmaf(v[e], sc[e], sh[e]));
        }
        Vec16<T>::store(out + q * out_ld + c, best);
    }
}

// PASS 0: (sum dh, sum dh*y) into the per-XCD slots; PASS 1: dy = a*dh + b*y + k written over y.  dh = the pooled gradient at the window's
// first maximum of relu(scale*y + shift), if that maximum is positive; zero elsewhere.
template <typename T, int CT, int PASS>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const T* __restrict__ dpool, long dpool_ld, T* __restrict__ y, long y_ld,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, double* __restrict__ slots,
                                                          BnBwdFin f, int N, int Ho, int Wo, int C, long ppb, int xcd_order) {
    constexpr int CE = Vec16<T>::N, VC = CT / CE, R = 256 / VC;
    __shared__ float s_a[CT], s_b[CT], s_k[CT], s_sc[CT], s_sh[CT];
    __shared__ float red[PASS == 0 ? 256 * 2 * CE : 1];
    int pb, ct;
    bn_block(C / CT, xcd_order, pb, ct);
    const int tid = threadIdx.x, c0 = ct * CT;
    if (tid < CT) {
        const int c = c0 + tid;
        s_sc[tid] = scale[c]; s_sh[tid] = shift[c];
        if (PASS == 1) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int x = 0; x < MSC_BN_SLOTS; ++x) {
                const double2 v = *reinterpret_cast<const double2*>(f.slots + ((long)x * C + c) * 2);
                s1 += v.x; s2 += v.y;
            }
            const double mu = f.save_mean[c], is = f.save_invstd[c], g = f.gamma ? f.gamma[c] : 1.0;
            const double dbe = s1, dga = is * (s2 - mu * s1);
            const double a = g * is, b = -g * is * is * dga / f.count, k0 = -g * is * dbe / f.count - b * mu;
            s_a[tid] = (float)a; s_b[tid] = (float)b; s_k[tid] = (float)k0;
            if (pb == 0) {
                if (f.dgamma) f.dgamma[c] += (float)dga;
                if (f.dbeta) f.dbeta[c] += (float)dbe;
            }
        }
    }
    __syncthreads();
    const int col = tid % VC, r = tid / VC;
    const int c = c0 + col * CE;
    float sc[CE], sh[CE], ca[CE], cb[CE], ck[CE], s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        sc[e] = s_sc[col * CE + e]; sh[e] = s_sh[col * CE + e];
        ca[e] = PASS ? s_a[col * CE + e] : 0.f; cb[e] = PASS ? s_b[col * CE + e] : 0.f; ck[e] = PASS ? s_k[col * CE + e] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
    const long total = (long)N * Ho * Wo, p1 = min(total, ((long)pb + 1) * ppb);
    const int Wi = 2 * Wo;
    for (long q = (long)pb * ppb + r; q < p1; q += R) {
        const long n = q / ((long)Ho * Wo), rem = q - n * ((long)Ho * Wo);
        const int py = (int)(rem / Wo), px = (int)(rem - (long)py * Wo);
        const long base = ((n * 2 * Ho + 2 * py) * Wi + 2 * px);
        float g[CE], yy[4][CE];
        Vec16<T>::load(dpool + q * dpool_ld + c, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) Vec16<T>::load(y + (base + (k >> 1) * Wi + (k & 1)) * y_ld + c, yy[k]);
        int arg[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            float best = 0.f;
            arg[e] = -1;                                            // no positive activation in the window: the gradient stops at the ReLU
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = fmaf(yy[k][e], sc[e], sh[e]);
                if (v > best) { best = v; arg[e] = k; }             // strict: the first maximum wins
            }
        }
        if (PASS == 0) {
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float ys = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) ys = arg[e] == k ? yy[k][e] : ys;
                const float dh = arg[e] >= 0 ? g[e] : 0.f;
                s1[e] += dh;
                s2[e] = fmaf(dh, ys, s2[e]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float o[CE];
#pragma unroll
                for (int e = 0; e < CE; ++e) o[e] = ca[e] * (arg[e] == k ? g[e] : 0.f) + cb[e] * yy[k][e] + ck[e];
                Vec16<T>::store(y + (base + (k >> 1) * Wi + (k & 1)) * y_ld + c, o);
            }
        }
    }
    if (PASS == 0) {
        // fold the block's partial sums through LDS, one coalesced double atomic per (channel, sum) into this XCD's slot (as colreduce_kernel)
        float* mine = red + tid * 2 * CE;
#pragma unroll
        for (int e = 0; e < CE; ++e) { mine[e] = s1[e]; mine[CE + e] = s2[e]; }
        __syncthreads();
        if (r == 0) {
            for (int k = 1; k < R; ++k) {
                const float* o = red + (k * VC + col) * 2 * CE;
#pragma unroll
                for (int e = 0; e < CE; ++e) { s1[e] += o[e]; s2[e] += o[CE + e]; }
            }
        }
        __syncthreads();
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < CE; ++e) *reinterpret_cast<float2*>(red + (col * CE + e) * 2) = make_float2(s1[e], s2[e]);
        }
        __syncthreads();
        double* slot = slots + ((long)msc_xcc_id() * C + c0) * 2;
        for (int i = tid; i < CT * 2; i += 256) atomicAdd(slot + i, (double)red[i]);
    }
}

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ y, long y_ld,
                                T* __restrict__ dx, long dx_ld, int accumulate, long pixels, int C) {
    constexpr int CE = Vec16<T>::N;
    const int cv = C / CE;
    const long total = pixels * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * CE;
        const long pix = i / cv;
        float d[CE], o[CE], r[CE];
        Vec16<T>::load(dy + pix * dy_ld + c, d);
        Vec16<T>::load(y + pix * y_ld + c, o);
#pragma unroll
        for (int e = 0; e < CE; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        if (accumulate) {
            Vec16<T>::load(dx + pix * dx_ld + c, r);
#pragma unroll
            for (int e = 0; e < CE; ++e) d[e] += r[e];
        }
        Vec16<T>::store(dx + pix * dx_ld + c, d);
    }
}

// ------------------------------------------------------------------ final 1x1 conv (C -> 2) + softmax
// one thread per pixel: reads C contiguous values (C*sizeof(T) bytes), writes 2 logits/probs into NCHW planes
template <typename T>
__global__ void final_fwd_kernel(const T* __restrict__ in, long in_ld, const float* __restrict__ w,
                                 const float* __restrict__ b, float* __restrict__ logits, float* __restrict__ probs,
                                 int N, long HW, int C) {
    constexpr int CE = Vec16<T>::N;
    extern __shared__ float sw[];  // [2][C]
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    const long total = (long)N * HW;
    const float b0 = b ? b[0] : 0.f, b1 = b ? b[1] : 0.f;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        float a0 = 0.f, a1 = 0.f;
        for (int c = 0; c < C; c += CE) {
            float v[CE];
            Vec16<T>::load(in + p * in_ld + c, v);
#pragma unroll
            for (int e = 0; e < CE; ++e) { a0 = fmaf(v[e], sw[c + e], a0); a1 = fmaf(v[e], sw[C + c + e], a1); }
        }
        a0 += b0; a1 += b1;
        const long n = p / HW, hw = p - n * HW;
        const long o0 = (n * 2) * HW + hw;
        if (logits) { logits[o0] = a0; logits[o0 + HW] = a1; }
        if (probs) {
            // numpy softmax of src/utils.py:231-273: subtract max, exp, divide by the sum
            const float m = fmaxf(a0, a1);
            const float e0 = expf(a0 - m), e1 = expf(a1 - m);
            const float s = e0 + e1;
            probs[o0] = e0 / s; probs[o0 + HW] = e1 / s;
        }
    }
}

// VC = C / CE lanes per pixel, each owning one 16-byte chunk of the pixel's channels: a wave-instruction covers 64/VC whole
// pixel rows (coalesced), din needs no cross-lane traffic, and the per-channel sums (dw, the producer's bias gradient) stay in
// registers until the end of the block -- one lane per pixel with a wave reduction per element and iteration kept the VALU
// busier than the memory pipe (0.19 ms for 285 MB).  UNR pixels per lane are in flight together.
// ORD (ordered_ws of msc_final_bwd): the four waves' sums are added in wave order and the block's 3 C + 2 sums go to column blockIdx.x of
// `ws` ([3 C + 2][MSC_FINAL_BWD_WS_ROWS]) instead of the gradients; final_bwd_finish_kernel adds each row up in a fixed tree.
template <typename T, int VC, bool ORD>
__global__ __launch_bounds__(256) void final_bwd_kernel(const float* __restrict__ dlogits, const T* __restrict__ in, long in_ld,
                                                        const float* __restrict__ w, T* __restrict__ din, long din_ld,
                                                        float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dbin,
                                                        float* __restrict__ ws, int N, long HW) {
    constexpr int CE = Vec16<T>::N, C = VC * CE, PPB = 256 / VC, UNR = 4, NS = 3 * C + 2;
    __shared__ float acc[(ORD ? 4 : 1) * NS];        // dw[2][C], dbias_in[C], db[2] (ORD: per wave)
    const int tid = threadIdx.x, chunk = tid % VC, sub = tid / VC;
    for (int i = tid; i < (ORD ? 4 : 1) * NS; i += 256) acc[i] = 0.f;
    float w0[CE], w1[CE], a0[CE], a1[CE], ab[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { w0[e] = w[chunk * CE + e]; w1[e] = w[C + chunk * CE + e]; a0[e] = 0.f; a1[e] = 0.f; ab[e] = 0.f; }
    float g0s = 0.f, g1s = 0.f;
    const long total = (long)N * HW;
    const long stride = (long)gridDim.x * PPB * UNR;
    for (long base = (long)blockIdx.x * PPB * UNR + sub; base < total; base += stride) {
        float g0[UNR], g1[UNR], v[UNR][CE];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long p = base + (long)u * PPB;
            ok[u] = p < total;
            g0[u] = 0.f; g1[u] = 0.f;
#pragma unroll
            for (int e = 0; e < CE; ++e) v[u][e] = 0.f;
            if (ok[u]) {
                const long n = p / HW, hw = p - n * HW;
                g0[u] = dlogits[(n * 2) * HW + hw];
                g1[u] = dlogits[(n * 2 + 1) * HW + hw];
                Vec16<T>::load(in + p * in_ld + chunk * CE, v[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float d[CE];
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                // dec0 ends in a ReLU (src/unet_models.py:401): gradient flows only where its output > 0
                d[e] = v[u][e] > 0.f ? g0[u] * w0[e] + g1[u] * w1[e] : 0.f;
                a0[e] += g0[u] * v[u][e];
                a1[e] += g1[u] * v[u][e];
            }
            if (ok[u]) {
                // the bias gradient counts what din HOLDS (the rounded value), as a pass over din would
                const uint4 packed = Vec16<T>::pack(d);
                *reinterpret_cast<uint4*>(din + (base + (long)u * PPB) * din_ld + chunk * CE) = packed;
                Vec16<T>::unpack(packed, d);
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) ab[e] += ok[u] ? d[e] : 0.f;
            if (chunk == 0) { g0s += g0[u]; g1s += g1[u]; }
        }
    }
    // fold the lanes that own the same chunk (lane % VC), then the four waves through LDS
#pragma unroll
    for (int e = 0; e < CE; ++e) {
#pragma unroll
        for (int o = VC; o < 64; o <<= 1) {
            a0[e] += __shfl_xor(a0[e], o, 64);
            a1[e] += __shfl_xor(a1[e], o, 64);
            ab[e] += __shfl_xor(ab[e], o, 64);
        }
    }
#pragma unroll
    for (int o = VC; o < 64; o <<= 1) { g0s += __shfl_xor(g0s, o, 64); g1s += __shfl_xor(g1s, o, 64); }
    __syncthreads();
    if constexpr (ORD) {
        if ((tid & 63) < VC) {      // lane `chunk` of every wave holds that wave's sums of its channels
            float* wa = acc + (tid >> 6) * NS;
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                wa[chunk * CE + e] = a0[e];
                wa[C + chunk * CE + e] = a1[e];
                wa[2 * C + chunk * CE + e] = ab[e];
            }
            if (chunk == 0) { wa[3 * C] = g0s; wa[3 * C + 1] = g1s; }
        }
        __syncthreads();
        for (int i = tid; i < NS; i += 256) ws[(long)i * MSC_FINAL_BWD_WS_ROWS + blockIdx.x] = ((acc[i] + acc[NS + i]) + acc[2 * NS + i]) + acc[3 * NS + i];
        return;
    }
    if ((tid & 63) < VC) {
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            atomicAdd(&acc[chunk * CE + e], a0[e]);
            atomicAdd(&acc[C + chunk * CE + e], a1[e]);
            atomicAdd(&acc[2 * C + chunk * CE + e], ab[e]);
        }
        if (chunk == 0) { atomicAdd(&acc[3 * C], g0s); atomicAdd(&acc[3 * C + 1], g1s); }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) atomicAdd(dw + i, acc[i]);
    if (dbin) for (int i = tid; i < C; i += 256) atomicAdd(dbin + i, acc[2 * C + i]);
    if (tid < 2 && db) atomicAdd(db + tid, acc[3 * C + tid]);
}

// block i: element i of the gradients += the sum of the per-block values of row i -- thread t adds columns t, t + 256, ... in that order,
// the 256 partial sums combine in a fixed shuffle tree and wave order
__global__ __launch_bounds__(256) void final_bwd_finish_kernel(const float* __restrict__ ws, int cols, int C, float* __restrict__ dw,
                                                               float* __restrict__ db, float* __restrict__ dbin) {
    __shared__ float w4[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* r = ws + (long)i * MSC_FINAL_BWD_WS_ROWS;
    float s = 0.f;
    for (int k = tid; k < cols; k += 256) s += r[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((tid & 63) == 0) w4[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const float t = (w4[0] + w4[1]) + (w4[2] + w4[3]);
        if (i < 2 * C) dw[i] += t;
        else if (i < 3 * C) { if (dbin) dbin[i - 2 * C] += t; }
        else if (db) db[i - 3 * C] += t;
    }
}

// ------------------------------------------------------------------ Adam (+L2), torch.optim.Adam semantics
// `state` (device f32[MSC_OPT_STATE], optional; include/msc.h) = step count, learning rate and the dynamic loss scale: lets a
// captured hipGraph replay the step with the bias corrections / lr / scale of the CURRENT iteration.  msc_adam_tick advances it
// inside the graph: a step whose gradients msc_grad_check found non-finite is SKIPPED (no parameter, moment or step-count
// change) and halves the scale; `growth` clean steps in a row double it (torch.cuda.amp.GradScaler's rule).
__global__ void adam_tick_kernel(float* state) {
    // the gradients of THIS step carry the scale the loss kernel read: the Adam kernels divide by that one, whatever happens to the scale below
    state[MSC_OPT_UNSCALE] = state[MSC_OPT_SCALE] > 0.f ? 1.f / state[MSC_OPT_SCALE] : 1.f;
    if (state[MSC_OPT_OVERFLOW] != 0.f) {
        state[MSC_OPT_OVERFLOW] = 0.f;
        state[MSC_OPT_SKIP] = 1.f;
        state[MSC_OPT_SKIPPED] += 1.f;
        state[MSC_OPT_GOOD] = 0.f;
        if (state[MSC_OPT_SCALE] > 1.f) state[MSC_OPT_SCALE] *= 0.5f;
    } else {
        state[MSC_OPT_SKIP] = 0.f;
        state[MSC_OPT_STEP] += 1.f;
        if (state[MSC_OPT_GROWTH] > 0.f && (state[MSC_OPT_GOOD] += 1.f) >= state[MSC_OPT_GROWTH]) {
            state[MSC_OPT_GOOD] = 0.f;
            if (state[MSC_OPT_SCALE] > 0.f && state[MSC_OPT_SCALE] < 16777216.f) state[MSC_OPT_SCALE] *= 2.f;
        }
    }
}

// any non-finite gradient element raises the overflow flag (every writer stores the same value: no atomics needed)
__global__ void grad_check_kernel(const float* __restrict__ g, long n, float* __restrict__ state) {
    const long n4 = n / 4;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        // x - x is 0 for finite x and NaN for +-inf / NaN: one test per vector
        const float t = (G.x - G.x) + (G.y - G.y) + (G.z - G.z) + (G.w - G.w);
        bad |= !(t == 0.f);
    }
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) bad |= !((g[i] - g[i]) == 0.f);
    if (bad) state[MSC_OPT_OVERFLOW] = 1.f;
}

struct AdamC { float lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale; };

// false: this step is skipped (overflow)
__device__ __forceinline__ bool adam_coeffs(AdamC& c, const float* __restrict__ state) {
    if (state) {
        if (state[MSC_OPT_SKIP] != 0.f) return false;
        const float step = state[MSC_OPT_STEP];
        c.lr = state[MSC_OPT_LR];
        c.bc1 = 1.f - powf(c.b1, step);
        c.bc2_sqrt = sqrtf(1.f - powf(c.b2, step));
        if (state[MSC_OPT_UNSCALE] > 0.f) c.gscale *= state[MSC_OPT_UNSCALE];
    }
    return true;
}

__device__ __forceinline__ void adam_update(const AdamC& c, float& p, float g, float& m, float& v) {
    const float gr = g * c.gscale + c.wd * p;
    m = c.b1 * m + (1.f - c.b1) * gr;
    v = c.b2 * v + (1.f - c.b2) * gr * gr;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p -= (c.lr / c.bc1) * (m / denom);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, AdamC c, const float* __restrict__ state) {
    if (!adam_coeffs(c, state)) return;
    const long n4 = n / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w};
        float mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_update(c, pp[e], gg[e], mm[e], vv[e]);
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    // tail
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_update(c, pp, g[i], mm, vv);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

// Adam over a TABLE of tensors that also writes the 16-bit compute copies the convolutions read (msc_adam_pack): the separate
// msc_pack_multi pass re-read 602 MB of fp32 masters per step that this kernel has in registers.  Block b works on
// items[block_item[b]], piece block_local[b]:
//   item without `trans`: 2048 consecutive elements (8 per thread), optional `direct` copy in the tensor's own layout;
//   item with `trans`   : one 64 (a) x 32 (b) tile of tap t of the [A][T][B] master (local index as msc_pack_multi: (t*ceil(A/64) +
//                         a_tile)*ceil(B/32) + b_tile), B % 4 == 0: 128-byte row segments of p / g / m / v, the `direct` copy from
//                         registers, the [B][T][A] copy through an LDS transpose (128-byte segments along a).
template <typename T>
__global__ __launch_bounds__(256) void adam_pack_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const msc_adam_item* __restrict__ items,
                                                        const int32_t* __restrict__ block_item, const int32_t* __restrict__ block_local,
                                                        AdamC c, const float* __restrict__ state) {
    __shared__ float tile[64][33];
    if (!adam_coeffs(c, state)) return;
    const msc_adam_item it = items[block_item[blockIdx.x]];
    const int lb = block_local[blockIdx.x];
    T* direct = reinterpret_cast<T*>(it.direct);
    if (!it.trans) {
        const long i0 = (long)lb * 2048 + threadIdx.x * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long i = i0 + h * 4;
            if (i + 4 <= it.n) {
                const long e0 = it.off + i;
                float4 P = *reinterpret_cast<float4*>(p + e0);
                const float4 G = *reinterpret_cast<const float4*>(g + e0);
                float4 M = *reinterpret_cast<float4*>(m + e0);
                float4 V = *reinterpret_cast<float4*>(v + e0);
                adam_update(c, P.x, G.x, M.x, V.x); adam_update(c, P.y, G.y, M.y, V.y);
                adam_update(c, P.z, G.z, M.z, V.z); adam_update(c, P.w, G.w, M.w, V.w);
                *reinterpret_cast<float4*>(p + e0) = P;
                *reinterpret_cast<float4*>(m + e0) = M;
                *reinterpret_cast<float4*>(v + e0) = V;
                if (direct) {
                    ElemIO<T>::store(direct + i, P.x); ElemIO<T>::store(direct + i + 1, P.y);
                    ElemIO<T>::store(direct + i + 2, P.z); ElemIO<T>::store(direct + i + 3, P.w);
                }
            } else {
                for (long k = i; k < it.n && k < i + 4; ++k) {
                    const long e = it.off + k;
                    float pp = p[e], mm = m[e], vv = v[e];
                    adam_update(c, pp, g[e], mm, vv);
                    p[e] = pp; m[e] = mm; v[e] = vv;
                    if (direct) ElemIO<T>::store(direct + k, pp);
                }
            }
        }
        return;
    }
    const int tiles_b = (it.B + 31) / 32, tiles_a = (it.A + 63) / 64;
    const int t = lb / (tiles_a * tiles_b);
    const int rem = lb - t * (tiles_a * tiles_b);
    const int a0 = (rem / tiles_b) * 64, b0 = (rem % tiles_b) * 32;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = ty + 32 * h;
        const int a = a0 + r, b = b0 + tx * 4;
        float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a < it.A && b < it.B) {
            const long k = ((long)a * it.T + t) * it.B + b, e0 = it.off + k;
            P = *reinterpret_cast<float4*>(p + e0);
            const float4 G = *reinterpret_cast<const float4*>(g + e0);
            float4 M = *reinterpret_cast<float4*>(m + e0);
            float4 V = *reinterpret_cast<float4*>(v + e0);
            adam_update(c, P.x, G.x, M.x, V.x); adam_update(c, P.y, G.y, M.y, V.y);
            adam_update(c, P.z, G.z, M.z, V.z); adam_update(c, P.w, G.w, M.w, V.w);
            *reinterpret_cast<float4*>(p + e0) = P;
            *reinterpret_cast<float4*>(m + e0) = M;
            *reinterpret_cast<float4*>(v + e0) = V;
            if (direct) {
                ElemIO<T>::store(direct + k, P.x); ElemIO<T>::store(direct + k + 1, P.y);
                ElemIO<T>::store(direct + k + 2, P.z); ElemIO<T>::store(direct + k + 3, P.w);
            }
        }
        tile[r][tx * 4] = P.x; tile[r][tx * 4 + 1] = P.y; tile[r][tx * 4 + 2] = P.z; tile[r][tx * 4 + 3] = P.w;
    }
    __syncthreads();
    T* trans = reinterpret_cast<T*>(it.trans);
    const int sx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    for (int r = sy; r < 32; r += 4) {
        const int b = b0 + r, a = a0 + sx;
        if (a < it.A && b < it.B) ElemIO<T>::store(trans + ((long)b * it.T + t) * it.A + a, tile[sx][r]);
    }
}

}  // namespace

#define DT_CHECK(name, dtype) \
    if (!msc_dtype_ok(dtype)) return msc_fail(MSC_ERR_ARG, name ": dtype %d", (int)(dtype))
#define VEC_CHECK(name, dtype, C) \
    if ((C) % msc_dtype_vec(dtype)) return msc_fail(MSC_ERR_UNSUPPORTED, name ": C=%d must be a multiple of the 16-byte vector", (int)(C))

extern "C" int msc_pack_cast(const float* src, void* dst, int dtype, int64_t n, void* stream) {
    DT_CHECK("msc_pack_cast", dtype);
    if (!src || !dst || n < 0) return msc_fail(MSC_ERR_ARG, "msc_pack_cast: bad argument");
    if (n == 0) return MSC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(pack_cast_kernel<f16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, src, (f16_t*)dst, (long)n);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(pack_cast_kernel<bf16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, src, (bf16_t*)dst, (long)n);
    else hipLaunchKernelGGL(pack_cast_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, src, (float*)dst, (long)n);
    return msc_check_launch("msc_pack_cast");
}

// ---- 16-bit wire format of the data-parallel gradient exchange (distributed.py): every rank casts its fp32 gradient
// range to the compute dtype (msc_pack_cast), an all-to-all hands each rank the `world` partial shards of its own
// slice, which are summed in fp32 here and rounded ONCE to the wire dtype; after the all-gather the result is widened
// back into the fp32 gradient buffer.  Halves the bytes of the fp32 ring all-reduce on the per-link-bound xGMI ring.
namespace {
template <typename T>
__global__ void grad_reduce_kernel(const T* __restrict__ recv, T* __restrict__ out, int world, long shard) {
    constexpr int CE = Vec16<T>::N;
    const long nv = shard / CE;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        float acc[CE];
#pragma unroll
        for (int e = 0; e < CE; ++e) acc[e] = 0.f;
        for (int w = 0; w < world; ++w) {
            float v[CE];
            Vec16<T>::load(recv + (long)w * shard + i * CE, v);
#pragma unroll
            for (int e = 0; e < CE; ++e) acc[e] += v[e];
        }
        Vec16<T>::store(out + i * CE, acc);
    }
}
template <typename T>
__global__ void grad_unpack_kernel(const T* __restrict__ in, float* __restrict__ g, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] = ElemIO<T>::load(in + i);
}
}  // namespace

extern "C" int msc_grad_reduce(const void* recv, void* out, int dtype, int world, int64_t shard, void* stream) {
    DT_CHECK("msc_grad_reduce", dtype);
    if (!recv || !out || world < 1 || shard < 0 || shard % msc_dtype_vec(dtype) || (((uintptr_t)recv | (uintptr_t)out) & 15))
        return msc_fail(MSC_ERR_ARG, "msc_grad_reduce: bad argument (shard must be a multiple of the 16-byte vector)");
    if (shard == 0) return MSC_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(shard / msc_dtype_vec(dtype));
    if (dtype == MSC_F16) hipLaunchKernelGGL(grad_reduce_kernel<f16_t>, dim3(grid), dim3(EW_THREADS), 0, st, (const f16_t*)recv, (f16_t*)out, world, (long)shard);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(grad_reduce_kernel<bf16_t>, dim3(grid), dim3(EW_THREADS), 0, st, (const bf16_t*)recv, (bf16_t*)out, world, (long)shard);
    else hipLaunchKernelGGL(grad_reduce_kernel<float>, dim3(grid), dim3(EW_THREADS), 0, st, (const float*)recv, (float*)out, world, (long)shard);
    return msc_check_launch("msc_grad_reduce");
}

extern "C" int msc_grad_unpack(const void* in, float* g, int dtype, int64_t n, void* stream) {
    DT_CHECK("msc_grad_unpack", dtype);
    if (!in || !g || n < 0) return msc_fail(MSC_ERR_ARG, "msc_grad_unpack: bad argument");
    if (n == 0) return MSC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(grad_unpack_kernel<f16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, (const f16_t*)in, g, (long)n);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(grad_unpack_kernel<bf16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, (const bf16_t*)in, g, (long)n);
    else hipLaunchKernelGGL(grad_unpack_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, (const float*)in, g, (long)n);
    return msc_check_launch("msc_grad_unpack");
}

extern "C" int msc_pack_transpose(const float* src, void* dst, int dtype, int A, int T, int B, void* stream) {
    DT_CHECK("msc_pack_transpose", dtype);
    if (!src || !dst || A <= 0 || T <= 0 || B <= 0 || T > 65535) return msc_fail(MSC_ERR_ARG, "msc_pack_transpose: bad argument");
    dim3 grid(ceil_div(B, 32), ceil_div(A, 32), T);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(pack_transpose_kernel<f16_t>, grid, dim3(256), 0, st, src, (f16_t*)dst, A, T, B);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(pack_transpose_kernel<bf16_t>, grid, dim3(256), 0, st, src, (bf16_t*)dst, A, T, B);
    else hipLaunchKernelGGL(pack_transpose_kernel<float>, grid, dim3(256), 0, st, src, (float*)dst, A, T, B);
    return msc_check_launch("msc_pack_transpose");
}

// All compute copies of the weights in ONE launch (the per-tensor launches cost ~230 x 4.7 us per step):
// block b works on items[block_item[b]], piece block_local[b] (2048 elements of a cast, or one 64x32 (a x b) tile of
// one tap of a transpose: a 64 x 32 tile, see below).
namespace {
template <typename T>
__global__ void pack_multi_kernel(const msc_pack_item* __restrict__ items, const int32_t* __restrict__ block_item,
                                  const int32_t* __restrict__ block_local) {
    __shared__ float tile[64][33];
    const msc_pack_item it = items[block_item[blockIdx.x]];
    const int lb = block_local[blockIdx.x];
    T* dst = reinterpret_cast<T*>(it.dst);
    if (it.kind == 0) {
        // 2048 elements per block, 8 consecutive ones per thread: two 16-byte loads, one 16-byte (bf16) or two (f32) stores
        const long i = (long)lb * 2048 + threadIdx.x * 8;
        if (i + 8 <= it.n && ((((uintptr_t)(it.src + i)) | ((uintptr_t)(dst + i))) & 15) == 0) {
            float v[8];
            const float4 lo = *reinterpret_cast<const float4*>(it.src + i), hi = *reinterpret_cast<const float4*>(it.src + i + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            constexpr int CE = Vec16<T>::N;
#pragma unroll
            for (int j = 0; j < 8; j += CE) Vec16<T>::store(dst + i + j, v + j);
        } else {
            for (int j = 0; j < 8; ++j)
                if (i + j < it.n) ElemIO<T>::store(dst + i + j, it.src[i + j]);
        }
        return;
    }
    // tile = 64 (a) x 32 (b): 128-byte row segments on the fp32 source side AND on the bf16 destination side
    const int tiles_b = (it.B + 31) / 32, tiles_a = (it.A + 63) / 64;
    const int t = lb / (tiles_a * tiles_b);
    const int rem = lb - t * (tiles_a * tiles_b);
    const int a0 = (rem / tiles_b) * 64, b0 = (rem % tiles_b) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 64; r += 8) {
        const int a = a0 + r, b = b0 + tx;
        tile[r][tx] = (a < it.A && b < it.B) ? it.src[((long)a * it.T + t) * it.B + b] : 0.f;
    }
    __syncthreads();
    const int sx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    for (int r = sy; r < 32; r += 4) {
        const int b = b0 + r, a = a0 + sx;
        if (a < it.A && b < it.B) ElemIO<T>::store(dst + ((long)b * it.T + t) * it.A + a, tile[sx][r]);
    }
}
}  // namespace

extern "C" int msc_pack_multi(const msc_pack_item* items, const int32_t* block_item, const int32_t* block_local, int nblocks,
                              int dtype, void* stream) {
    DT_CHECK("msc_pack_multi", dtype);
    if (!items || !block_item || !block_local || nblocks < 0) return msc_fail(MSC_ERR_ARG, "msc_pack_multi: bad argument");
    if (nblocks == 0) return MSC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(pack_multi_kernel<f16_t>, dim3(nblocks), dim3(256), 0, st, items, block_item, block_local);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, st, items, block_item, block_local);
    else hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(nblocks), dim3(256), 0, st, items, block_item, block_local);
    return msc_check_launch("msc_pack_multi");
}

extern "C" int msc_stem_pack(const float* w, void* dst, int dtype, int cout, void* stream) {
    DT_CHECK("msc_stem_pack", dtype);
    if (!w || !dst || cout <= 0) return msc_fail(MSC_ERR_ARG, "msc_stem_pack: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int n = cout * 7 * 32;
    if (dtype == MSC_F16) hipLaunchKernelGGL(stem_pack_kernel<f16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, w, (f16_t*)dst, cout);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, w, (bf16_t*)dst, cout);
    else hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(ew_grid(n)), dim3(EW_THREADS), 0, st, w, (float*)dst, cout);
    return msc_check_launch("msc_stem_pack");
}

extern "C" int msc_stem_unpack_grad(const float* dpacked, float* dw, int cout, void* stream) {
    if (!dpacked || !dw || cout <= 0) return msc_fail(MSC_ERR_ARG, "msc_stem_unpack_grad: bad argument");
    hipLaunchKernelGGL(stem_unpack_grad_kernel, dim3(ew_grid(cout * 147)), dim3(EW_THREADS), 0, (hipStream_t)stream, dpacked, dw, cout);
    return msc_check_launch("msc_stem_unpack_grad");
}

extern "C" int msc_stem_prepare(const float* x, void* xp, int dtype, int N, int H, int W, void* stream) {
    DT_CHECK("msc_stem_prepare", dtype);
    if (!x || !xp || N <= 0 || H <= 0 || W <= 0) return msc_fail(MSC_ERR_ARG, "msc_stem_prepare: bad argument");
    const long total = (long)N * (H + 6) * (W + 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(stem_prepare_kernel<f16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, x, (f16_t*)xp, N, H, W);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(stem_prepare_kernel<bf16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, x, (bf16_t*)xp, N, H, W);
    else hipLaunchKernelGGL(stem_prepare_kernel<float>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, x, (float*)xp, N, H, W);
    return msc_check_launch("msc_stem_prepare");
}

extern "C" int msc_maxpool2_fwd(const void* in, int64_t in_ld, void* out, int64_t out_ld, int dtype, int N, int Ho, int Wo, int C, void* stream) {
    DT_CHECK("msc_maxpool2_fwd", dtype);
    VEC_CHECK("msc_maxpool2_fwd", dtype, C);
    if (!in || !out) return msc_fail(MSC_ERR_ARG, "msc_maxpool2_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * Ho * Wo * (C / msc_dtype_vec(dtype));
    if (dtype == MSC_F16) hipLaunchKernelGGL(maxpool2_fwd_kernel<f16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const f16_t*)in, (long)in_ld, (f16_t*)out, (long)out_ld, N, Ho, Wo, C);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const bf16_t*)in, (long)in_ld, (bf16_t*)out, (long)out_ld, N, Ho, Wo, C);
    else hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const float*)in, (long)in_ld, (float*)out, (long)out_ld, N, Ho, Wo, C);
    return msc_check_launch("msc_maxpool2_fwd");
}

extern "C" int msc_maxpool2_bwd(const void* dout, int64_t dout_ld, const void* in, int64_t in_ld, void* din, int64_t din_ld,
                                int dtype, int N, int Ho, int Wo, int C, int accumulate, void* stream) {
    DT_CHECK("msc_maxpool2_bwd", dtype);
    VEC_CHECK("msc_maxpool2_bwd", dtype, C);
    if (!dout || !in || !din) return msc_fail(MSC_ERR_ARG, "msc_maxpool2_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * Ho * Wo * (C / msc_dtype_vec(dtype));
    if (dtype == MSC_F16) hipLaunchKernelGGL(maxpool2_bwd_kernel<f16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const f16_t*)dout, (long)dout_ld, (const f16_t*)in, (long)in_ld, (f16_t*)din, (long)din_ld, N, Ho, Wo, C, accumulate);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const bf16_t*)dout, (long)dout_ld, (const bf16_t*)in, (long)in_ld, (bf16_t*)din, (long)din_ld, N, Ho, Wo, C, accumulate);
    else hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const float*)dout, (long)dout_ld, (const float*)in, (long)in_ld, (float*)din, (long)din_ld, N, Ho, Wo, C, accumulate);
    return msc_check_launch("msc_maxpool2_bwd");
}

extern "C" int msc_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                            float eps, float* scale, float* shift, int C, void* stream) {
    if (!running_mean || !running_var || !scale || !shift || C <= 0) return msc_fail(MSC_ERR_ARG, "msc_bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale, shift, C);
    return msc_check_launch("msc_bn_fold");
}

namespace {
static int bn_xcd_order() { static const int v = [] { const char* e = getenv("MSC_BN_XCD"); return e ? atoi(e) : 1; }(); return v; }
// grid of the channel-tiled BatchNorm kernels: CT channels x `ppb` pixels per block, about 1024 blocks in all (four per CU).  With the
// round-robin block order of rounds 1-3 the count did not matter (512 ... 4096: +-0); with the XCD-aware order fewer, longer blocks are
// ahead -- 256: 11.60, 512: 10.93, 768: 10.76, 1024: 10.62-10.68, 2048: 10.70-10.77, 4096 / 8192: 10.81 ms per ResNet101 train step
// (profiles/r4_run25_bn_blocks_ab.txt)
template <int R>
long bn_ppb(long pixels, int ctiles) {
    static int target = -1;      // MSC_BN_BLOCKS: blocks per launch aimed at (A/B measurements)
    if (target < 0) { const char* e = getenv("MSC_BN_BLOCKS"); target = e ? atoi(e) : 1024; if (target < 64) target = 64; }
    long ppb = ceil_div(pixels * ctiles, target);
    ppb = (ppb + R - 1) / R * R;
    return ppb < R ? R : ppb;
}
template <typename T, int CT>
void launch_bn_apply(const void* y, long y_ld, const void* res, long res_ld, void* out, long out_ld, const BnFwdFin& f, int relu, long pixels, int C,
                     uint8_t* mask, long mask_ld, hipStream_t st) {
    constexpr int R = 256 / (CT / Vec16<T>::N);
    const long ppb = bn_ppb<R>(pixels, C / CT);
    hipLaunchKernelGGL((bn_apply_kernel<T, CT>), dim3(ceil_div(pixels, ppb) * (C / CT)), dim3(256), 0, st, (const T*)y, y_ld, (const T*)res, res_ld, (T*)out,
                       out_ld, f, relu, pixels, C, ppb, bn_xcd_order(), mask, mask_ld);
}
template <typename T, int CT>
void launch_bn_bwd_apply(const void* dout, long dout_ld, const void* out, long out_ld, const void* y, long y_ld, int relu, const float* scale,
                         const float* shift, const BnBwdFin& f, void* dy, long dy_ld, void* dres, long dres_ld, int dres_acc, long pixels, int C,
                         const void* ry, long ry_ld, double* rslots, hipStream_t st) {
    constexpr int R = 256 / (CT / Vec16<T>::N);
    const long ppb = bn_ppb<R>(pixels, C / CT);
    const dim3 grid(ceil_div(pixels, ppb) * (C / CT));
    if (ry) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, CT, true>), grid, dim3(256), 0, st, (const T*)dout, dout_ld, (const T*)out, out_ld, (const T*)y, y_ld, relu,
                               scale, shift, f, (T*)dy, dy_ld, (T*)dres, dres_ld, dres_acc, pixels, C, ppb, bn_xcd_order(), (const T*)ry, ry_ld, rslots);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, CT, false>), grid, dim3(256), 0, st, (const T*)dout, dout_ld, (const T*)out, out_ld, (const T*)y, y_ld, relu,
                            scale, shift, f, (T*)dy, dy_ld, (T*)dres, dres_ld, dres_acc, pixels, C, ppb, bn_xcd_order(), (const T*)nullptr, 0L, (double*)nullptr);
}
}  // namespace

#define MSC_BN_DISPATCH(FN, ...) \
    do { \
        if (C % 64 == 0) { \
            if (dtype == MSC_F16) FN<f16_t, 64>(__VA_ARGS__); else if (dtype == MSC_BF16) FN<bf16_t, 64>(__VA_ARGS__); else FN<float, 64>(__VA_ARGS__); \
        } else { \
            if (dtype == MSC_F16) FN<f16_t, 32>(__VA_ARGS__); else if (dtype == MSC_BF16) FN<bf16_t, 32>(__VA_ARGS__); else FN<float, 32>(__VA_ARGS__); \
        } \
    } while (0)

extern "C" int msc_bn_apply(const void* y, int64_t y_ld, const void* res, int64_t res_ld, void* out, int64_t out_ld,
                            const double* slots, int64_t count, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                            uint8_t* relu_mask, int64_t relu_mask_ld, int relu, int dtype, int64_t pixels, int C, void* stream) {
    DT_CHECK("msc_bn_apply", dtype);
    if (relu_mask && (!relu || relu_mask_ld * msc_dtype_vec(dtype) < C)) return msc_fail(MSC_ERR_ARG, "msc_bn_apply: relu_mask needs relu and C / %d bytes per pixel", msc_dtype_vec(dtype));
    if (C % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_apply: C=%d must be a multiple of 32", C);
    if (!y || !out || !scale || !shift || pixels <= 0 || (slots && count <= 0)) return msc_fail(MSC_ERR_ARG, "msc_bn_apply: bad argument");
    const BnFwdFin f = {slots, (double)count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
    MSC_BN_DISPATCH(launch_bn_apply, y, (long)y_ld, res, (long)res_ld, out, (long)out_ld, f, relu, (long)pixels, C, relu_mask, (long)relu_mask_ld, (hipStream_t)stream);
    return msc_check_launch("msc_bn_apply");
}

extern "C" int msc_bn_bwd_apply(const void* dout, int64_t dout_ld, const void* out, int64_t out_ld, const void* y, int64_t y_ld,
                                int relu, const float* scale, const float* shift, const double* slots, int64_t count, const float* gamma,
                                const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, void* dy, int64_t dy_ld,
                                void* dres, int64_t dres_ld, int dres_acc, const void* res_y, int64_t res_y_ld, double* res_slots, int dtype,
                                int64_t pixels, int C, void* stream) {
    DT_CHECK("msc_bn_bwd_apply", dtype);
    if (res_y && (!res_slots || !dres)) return msc_fail(MSC_ERR_ARG, "msc_bn_bwd_apply: res_y needs res_slots and dres");
    if (C % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_bwd_apply: C=%d must be a multiple of 32", C);
    if (!dout || !y || !slots || !save_mean || !save_invstd || !dy || count <= 0 || pixels <= 0 || relu < 0 || relu > 3 || ((relu == 1 || relu == 3) && !out) ||
        (relu == 2 && (!scale || !shift)))
        return msc_fail(MSC_ERR_ARG, "msc_bn_bwd_apply: bad argument");
    const BnBwdFin f = {slots, (double)count, gamma, save_mean, save_invstd, dgamma, dbeta};
    MSC_BN_DISPATCH(launch_bn_bwd_apply, dout, (long)dout_ld, out, (long)out_ld, y, (long)y_ld, relu, scale, shift, f, dy, (long)dy_ld, dres, (long)dres_ld,
                    dres_acc, (long)pixels, C, res_y, (long)res_y_ld, res_slots, (hipStream_t)stream);
    return msc_check_launch("msc_bn_bwd_apply");
}

namespace {
template <typename T, int CT>
void launch_bn_apply_pool(const void* y, long y_ld, void* out, long out_ld, const BnFwdFin& f, int N, int Ho, int Wo, int C, hipStream_t st) {
    constexpr int R = 256 / (CT / Vec16<T>::N);
    const long pixels = (long)N * Ho * Wo, ppb = bn_ppb<R>(pixels, C / CT);
    hipLaunchKernelGGL((bn_apply_pool_kernel<T, CT>), dim3(ceil_div(pixels, ppb) * (C / CT)), dim3(256), 0, st, (const T*)y, y_ld, (T*)out, out_ld, f, N, Ho,
                       Wo, C, ppb, bn_xcd_order());
}
template <typename T, int CT>
void launch_bn_pool_bwd(int pass, const void* dpool, long dpool_ld, void* y, long y_ld, const float* scale, const float* shift, double* slots,
                        const BnBwdFin& f, int N, int Ho, int Wo, int C, hipStream_t st) {
    constexpr int R = 256 / (CT / Vec16<T>::N);
    const long pixels = (long)N * Ho * Wo, ppb = bn_ppb<R>(pixels, C / CT);
    const dim3 grid(ceil_div(pixels, ppb) * (C / CT));
    if (pass == 0) hipLaunchKernelGGL((bn_pool_bwd_kernel<T, CT, 0>), grid, dim3(256), 0, st, (const T*)dpool, dpool_ld, (T*)y, y_ld, scale, shift, slots, f, N, Ho, Wo, C, ppb, bn_xcd_order());
    else hipLaunchKernelGGL((bn_pool_bwd_kernel<T, CT, 1>), grid, dim3(256), 0, st, (const T*)dpool, dpool_ld, (T*)y, y_ld, scale, shift, slots, f, N, Ho, Wo, C, ppb, bn_xcd_order());
}
}  // namespace

extern "C" int msc_bn_apply_pool(const void* y, int64_t y_ld, void* out, int64_t out_ld, const double* slots, int64_t count, const float* gamma,
                                 const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                 float* save_mean, float* save_invstd, int dtype, int N, int Ho, int Wo, int C, void* stream) {
    DT_CHECK("msc_bn_apply_pool", dtype);
    if (C % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_apply_pool: C=%d must be a multiple of 32", C);
    if (!y || !out || !slots || !scale || !shift || count <= 0 || N <= 0 || Ho <= 0 || Wo <= 0) return msc_fail(MSC_ERR_ARG, "msc_bn_apply_pool: bad argument");
    const BnFwdFin f = {slots, (double)count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
    MSC_BN_DISPATCH(launch_bn_apply_pool, y, (long)y_ld, out, (long)out_ld, f, N, Ho, Wo, C, (hipStream_t)stream);
    return msc_check_launch("msc_bn_apply_pool");
}

extern "C" int msc_bn_pool_bwd_reduce(const void* dpool, int64_t dpool_ld, const void* y, int64_t y_ld, const float* scale, const float* shift,
                                      double* slots, int dtype, int N, int Ho, int Wo, int C, void* stream) {
    DT_CHECK("msc_bn_pool_bwd_reduce", dtype);
    if (C % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_pool_bwd_reduce: C=%d must be a multiple of 32", C);
    if (!dpool || !y || !scale || !shift || !slots || N <= 0 || Ho <= 0 || Wo <= 0) return msc_fail(MSC_ERR_ARG, "msc_bn_pool_bwd_reduce: bad argument");
    const BnBwdFin f = {slots, 1.0, nullptr, nullptr, nullptr, nullptr, nullptr};
    MSC_BN_DISPATCH(launch_bn_pool_bwd, 0, dpool, (long)dpool_ld, const_cast<void*>(y), (long)y_ld, scale, shift, slots, f, N, Ho, Wo, C, (hipStream_t)stream);
    return msc_check_launch("msc_bn_pool_bwd_reduce");
}

extern "C" int msc_bn_pool_bwd_apply(const void* dpool, int64_t dpool_ld, void* y, int64_t y_ld, const float* scale, const float* shift,
                                     const double* slots, int64_t count, const float* gamma, const float* save_mean, const float* save_invstd,
                                     float* dgamma, float* dbeta, int dtype, int N, int Ho, int Wo, int C, void* stream) {
    DT_CHECK("msc_bn_pool_bwd_apply", dtype);
    if (C % 32) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_pool_bwd_apply: C=%d must be a multiple of 32", C);
    if (!dpool || !y || !scale || !shift || !slots || !save_mean || !save_invstd || count <= 0 || N <= 0 || Ho <= 0 || Wo <= 0)
        return msc_fail(MSC_ERR_ARG, "msc_bn_pool_bwd_apply: bad argument");
    const BnBwdFin f = {slots, (double)count, gamma, save_mean, save_invstd, dgamma, dbeta};
    MSC_BN_DISPATCH(launch_bn_pool_bwd, 1, dpool, (long)dpool_ld, y, (long)y_ld, scale, shift, const_cast<double*>(slots), f, N, Ho, Wo, C, (hipStream_t)stream);
    return msc_check_launch("msc_bn_pool_bwd_apply");
}

// Round 5: stream-ordered fills and copies are KERNELS, not hipMemsetAsync / hipMemcpyAsync.  Captured into a hipGraph those calls become memset /
// memcpy NODES, and a replayed training step that held them could run with garbage gradients when the program had been built while other work was
// still pending on the device (same launch lists eagerly -- fine; the same graph with these two entry points as kernels -- fine,
// with or without the device synchronise that otherwise hides it; probes/replay_order_probe.py, tests/test_gpu_replay_hazard.py).  A captured step now holds kernel nodes only.
// MSC_MEMOPS_KERNEL=0 brings the runtime calls back (A/B).
__global__ __launch_bounds__(256) void fill_zero_kernel(char* __restrict__ p, long bytes) {
    const long head = min((long)((16 - ((uintptr_t)p & 15)) & 15), bytes);          // bytes in front of the first 16-byte boundary
    const long n16 = (bytes - head) / 16, tail0 = head + n16 * 16;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) q[i] = z;
    if (blockIdx.x == 0) {
        if ((long)threadIdx.x < head) p[threadIdx.x] = 0;
        if (tail0 + (long)threadIdx.x < bytes && threadIdx.x < 16) p[tail0 + threadIdx.x] = 0;
    }
}
__global__ __launch_bounds__(256) void copy_kernel(char* __restrict__ d, const char* __restrict__ s, long bytes) {
    if ((((uintptr_t)d ^ (uintptr_t)s) & 15) == 0) {                                   // same phase within 16 bytes: vector body
        const long head = min((long)((16 - ((uintptr_t)d & 15)) & 15), bytes);
        const long n16 = (bytes - head) / 16, tail0 = head + n16 * 16;
        uint4* q = reinterpret_cast<uint4*>(d + head);
        const uint4* r = reinterpret_cast<const uint4*>(s + head);
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) q[i] = r[i];
        if (blockIdx.x == 0) {
            if ((long)threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
            if (tail0 + (long)threadIdx.x < bytes && threadIdx.x < 16) d[tail0 + threadIdx.x] = s[tail0 + threadIdx.x];
        }
    } else {
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < bytes; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
    }
}
static bool memops_as_kernels() {
    static const bool off = [] { const char* e = getenv("MSC_MEMOPS_KERNEL"); return e && e[0] == '0'; }();
    return !off;
}
static int memop_blocks(long bytes) {
    long blocks = (bytes / 16 + 255) / 256;
    return (int)(blocks > 4096 ? 4096 : blocks < 1 ? 1 : blocks);
}

extern "C" int msc_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    if (!dst || !src || bytes < 0) return msc_fail(MSC_ERR_ARG, "msc_copy: bad argument");
    if (bytes && memops_as_kernels()) {
        hipLaunchKernelGGL(copy_kernel, dim3(memop_blocks(bytes)), dim3(256), 0, (hipStream_t)stream, (char*)dst, (const char*)src, (long)bytes);
        return msc_check_launch("msc_copy");
    }
    if (bytes && hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return msc_fail(MSC_ERR_HIP, "msc_copy: copy failed");
    return MSC_OK;
}

extern "C" int msc_memset_zero(void* ptr, int64_t bytes, void* stream) {
    if (!ptr || bytes < 0) return msc_fail(MSC_ERR_ARG, "msc_memset_zero: bad argument");
    if (bytes && memops_as_kernels()) {
        hipLaunchKernelGGL(fill_zero_kernel, dim3(memop_blocks(bytes)), dim3(256), 0, (hipStream_t)stream, (char*)ptr, (long)bytes);
        return msc_check_launch("msc_memset_zero");
    }
    if (bytes && hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) return msc_fail(MSC_ERR_HIP, "msc_memset_zero: memset failed");
    return MSC_OK;
}

extern "C" int msc_relu_bwd(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, void* dx, int64_t dx_ld,
                            int accumulate, int dtype, int64_t pixels, int C, void* stream) {
    DT_CHECK("msc_relu_bwd", dtype);
    VEC_CHECK("msc_relu_bwd", dtype, C);
    if (!dy || !y || !dx) return msc_fail(MSC_ERR_ARG, "msc_relu_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long total = pixels * (C / msc_dtype_vec(dtype));
    if (dtype == MSC_F16) hipLaunchKernelGGL(relu_bwd_kernel<f16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const f16_t*)dy, (long)dy_ld, (const f16_t*)y, (long)y_ld, (f16_t*)dx, (long)dx_ld, accumulate, (long)pixels, C);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const bf16_t*)dy, (long)dy_ld, (const bf16_t*)y, (long)y_ld, (bf16_t*)dx, (long)dx_ld, accumulate, (long)pixels, C);
    else hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(ew_grid(total)), dim3(EW_THREADS), 0, st, (const float*)dy, (long)dy_ld, (const float*)y, (long)y_ld, (float*)dx, (long)dx_ld, accumulate, (long)pixels, C);
    return msc_check_launch("msc_relu_bwd");
}

extern "C" int msc_final_fwd(const void* in, int64_t in_ld, const float* w, const float* b, float* logits, float* probs,
                             int dtype, int N, int H, int W, int C, void* stream) {
    DT_CHECK("msc_final_fwd", dtype);
    VEC_CHECK("msc_final_fwd", dtype, C);
    if (!in || !w || (!logits && !probs)) return msc_fail(MSC_ERR_ARG, "msc_final_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    const size_t shm = 2 * C * sizeof(float);
    if (dtype == MSC_F16) hipLaunchKernelGGL(final_fwd_kernel<f16_t>, dim3(ew_grid((long)N * hw)), dim3(EW_THREADS), shm, st, (const f16_t*)in, (long)in_ld, w, b, logits, probs, N, hw, C);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(final_fwd_kernel<bf16_t>, dim3(ew_grid((long)N * hw)), dim3(EW_THREADS), shm, st, (const bf16_t*)in, (long)in_ld, w, b, logits, probs, N, hw, C);
    else hipLaunchKernelGGL(final_fwd_kernel<float>, dim3(ew_grid((long)N * hw)), dim3(EW_THREADS), shm, st, (const float*)in, (long)in_ld, w, b, logits, probs, N, hw, C);
    return msc_check_launch("msc_final_fwd");
}

template <typename T>
static int final_bwd_launch(const float* dlogits, const void* in, long in_ld, const float* w, void* din, long din_ld, float* dw,
                            float* db, float* dbin, float* ws, int N, long hw, int C, hipStream_t st) {
    constexpr int CE = Vec16<T>::N;
    const int vc = C / CE;
    long blocks = ceil_div((long)N * hw, (256 / vc) * 4);
    if (blocks > MSC_FINAL_BWD_WS_ROWS) blocks = MSC_FINAL_BWD_WS_ROWS;
    // the unordered form ends every block in 3 C + 2 float atomics on the same few lines (serialised by the L2): fewer, longer blocks.  Measured on the train step
    // (round 6, tools/gpu_tail_atomics_ab.sh): 1024 blocks 99.7 us, 512: 86.6-86.9, 256: 112-117, 128: 207.  MSC_FINAL_BWD_BLOCKS sets the cap
    static const long cap = [] { const char* e = getenv("MSC_FINAL_BWD_BLOCKS"); return e ? atol(e) : 512L; }();
    if (!ws && blocks > cap) blocks = cap;
#define MSC_FB(VC) \
    if (ws) hipLaunchKernelGGL((final_bwd_kernel<T, VC, true>), dim3((int)blocks), dim3(256), 0, st, dlogits, (const T*)in, in_ld, w, (T*)din, din_ld, dw, db, dbin, ws, N, hw); \
    else hipLaunchKernelGGL((final_bwd_kernel<T, VC, false>), dim3((int)blocks), dim3(256), 0, st, dlogits, (const T*)in, in_ld, w, (T*)din, din_ld, dw, db, dbin, ws, N, hw)
    switch (vc) {
        case 1: MSC_FB(1); break;
        case 2: MSC_FB(2); break;
        case 4: MSC_FB(4); break;
        case 8: MSC_FB(8); break;
        case 16: MSC_FB(16); break;
        default: return msc_fail(MSC_ERR_UNSUPPORTED, "msc_final_bwd: C=%d (supported: C*sizeof(dtype)/16 a power of two up to 16)", C);
    }
#undef MSC_FB
    if (ws) hipLaunchKernelGGL(final_bwd_finish_kernel, dim3(3 * C + 2), dim3(256), 0, st, ws, (int)blocks, C, dw, db, dbin);
    return msc_check_launch("msc_final_bwd");
}

extern "C" int msc_final_bwd(const float* dlogits, const void* in, int64_t in_ld, const float* w, void* din, int64_t din_ld,
                             float* dw, float* db, float* dbias_in, float* ordered_ws, int dtype, int N, int H, int W, int C, void* stream) {
    DT_CHECK("msc_final_bwd", dtype);
    VEC_CHECK("msc_final_bwd", dtype, C);
    if (!dlogits || !in || !w || !din || !dw) return msc_fail(MSC_ERR_ARG, "msc_final_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    if (dtype == MSC_F16) return final_bwd_launch<f16_t>(dlogits, in, in_ld, w, din, din_ld, dw, db, dbias_in, ordered_ws, N, hw, C, st);
    if (dtype == MSC_BF16) return final_bwd_launch<bf16_t>(dlogits, in, in_ld, w, din, din_ld, dw, db, dbias_in, ordered_ws, N, hw, C, st);
    return final_bwd_launch<float>(dlogits, in, in_ld, w, din, din_ld, dw, db, dbias_in, ordered_ws, N, hw, C, st);
}

extern "C" int msc_adam_tick(float* state, void* stream) {
    if (!state) return msc_fail(MSC_ERR_ARG, "msc_adam_tick: null state");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    return msc_check_launch("msc_adam_tick");
}

static AdamC adam_host_coeffs(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale) {
    const float bc1 = 1.f - powf(beta1, (float)(step < 1 ? 1 : step));
    const float bc2 = 1.f - powf(beta2, (float)(step < 1 ? 1 : step));
    return AdamC{lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale};
}

extern "C" int msc_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, const float* state, void* stream) {
    if (!p || !g || !m || !v || n < 0 || (!state && step < 1)) return msc_fail(MSC_ERR_ARG, "msc_adam_step: bad argument");
    if (n == 0) return MSC_OK;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return msc_fail(MSC_ERR_ARG, "msc_adam_step: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                       adam_host_coeffs(lr, beta1, beta2, eps, weight_decay, step, grad_scale), state);
    return msc_check_launch("msc_adam_step");
}

extern "C" int msc_adam_pack(float* p, const float* g, float* m, float* v, const msc_adam_item* items, const int32_t* block_item,
                             const int32_t* block_local, int nblocks, int dtype, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, float grad_scale, const float* state, void* stream) {
    DT_CHECK("msc_adam_pack", dtype);
    if (!p || !g || !m || !v || !items || !block_item || !block_local || nblocks < 0 || (!state && step < 1))
        return msc_fail(MSC_ERR_ARG, "msc_adam_pack: bad argument");
    if (nblocks == 0) return MSC_OK;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return msc_fail(MSC_ERR_ARG, "msc_adam_pack: buffers must be 16-byte aligned");
    const AdamC c = adam_host_coeffs(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) hipLaunchKernelGGL(adam_pack_kernel<f16_t>, dim3(nblocks), dim3(256), 0, st, p, g, m, v, items, block_item, block_local, c, state);
    else if (dtype == MSC_BF16) hipLaunchKernelGGL(adam_pack_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, st, p, g, m, v, items, block_item, block_local, c, state);
    else hipLaunchKernelGGL(adam_pack_kernel<float>, dim3(nblocks), dim3(256), 0, st, p, g, m, v, items, block_item, block_local, c, state);
    return msc_check_launch("msc_adam_pack");
}

extern "C" int msc_grad_check(const float* g, int64_t n, float* state, void* stream) {
    if (!g || !state || n < 0 || ((uintptr_t)g & 15)) return msc_fail(MSC_ERR_ARG, "msc_grad_check: bad argument");
    if (n == 0) return MSC_OK;
    hipLaunchKernelGGL(grad_check_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, g, (long)n, state);
    return msc_check_launch("msc_grad_check");
}
