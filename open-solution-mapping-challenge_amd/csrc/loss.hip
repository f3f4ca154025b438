// Fused loss kernels (HBM-bound streaming over NCHW f32 logits / targets) for gfx950.
//
// Reference semantics (file:line under /root/reference):
//   plain CE                     src/steps/pytorch/validation.py:25-28
//   distance x size weighted CE  src/models.py:310-381   (get_weights :339-370)
//   soft Dice on softmax (or, cfg.dice_sigmoid, on sigmoid: src/models.py:437-442), class 1 only, sums over the whole batch   src/models.py:421-454, validation.py:8-16
//   mix                          src/models.py:384-418 (weights from neptune.yaml:42-43,55-57)
// Two phases so that the four global sums can be all-reduced over ranks in between (the reference
// computes the loss on the DataParallel-gathered full batch, src/steps/pytorch/models.py:92,104).
#include <stdlib.h>

#include "common.h"
#include "msc_internal.h"

namespace {

struct PixelTerms { float ce, w, p0, p1, t1, q1; int tcls; };      // q1: the Dice activation of class 1 (softmax p1 or sigmoid(l1))

__device__ __forceinline__ PixelTerms pixel_terms(const float* logits, const float* target, int tc, const msc_loss_cfg& cfg,
                                                  long n, long hw, long HW) {
    PixelTerms r;
    const float l0 = logits[(n * 2) * HW + hw], l1 = logits[(n * 2 + 1) * HW + hw];
    const float t = target[(n * tc) * HW + hw];
    r.tcls = (int)t;  // .long(): truncation
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float s = e0 + e1;
    const float lse = m + logf(s);
    r.p0 = e0 / s; r.p1 = e1 / s;
    r.q1 = cfg.dice_sigmoid ? 1.f / (1.f + expf(-l1)) : r.p1;
    r.ce = lse - (r.tcls == 1 ? l1 : l0);
    r.t1 = r.tcls == 1 ? 1.f : 0.f;
    float w = 1.f;
    if (cfg.weighted) {
        const float d = target[(n * tc + 1) * HW + hw];
        const float sz = target[(n * tc + 2) * HW + hw];
        const float dw = d == 0.f ? 1.f : 1.f + cfg.w0 * expf(-(d * d) / (cfg.sigma * cfg.sigma));
        const float s1 = sz == 0.f ? 1.f : sz;
        const float sw = s1 == 1.f ? 1.f : cfg.size_c / s1;
        w = dw * sw;
    }
    r.w = w;
    return r;
}

__global__ void loss_sums_kernel(const float* __restrict__ logits, const float* __restrict__ target, int tc,
                                 msc_loss_cfg cfg, double* sums, int N, long HW) {
    __shared__ double red[4][4];
    const long total = (long)N * HW;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long n = p / HW, hw = p - n * HW;
        const PixelTerms t = pixel_terms(logits, target, tc, cfg, n, hw, HW);
        a0 += (double)(t.w * t.ce);
        a1 += (double)(t.q1 * t.t1);
        a2 += (double)t.q1;
        a3 += (double)t.t1;
    }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2); a3 = wave_sum_d(a3);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid][0] = a0; red[wid][1] = a1; red[wid][2] = a2; red[wid][3] = a3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double s = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k][threadIdx.x];
        atomicAdd(sums + threadIdx.x, s);
    }
}

__global__ void loss_grad_kernel(const float* __restrict__ logits, const float* __restrict__ target, int tc,
                                 msc_loss_cfg cfg, const double* __restrict__ sums, double total_pixels, float gscale,
                                 const float* __restrict__ scale_state, float* loss, float* __restrict__ dlogits, int N, long HW) {
    if (scale_state && scale_state[MSC_OPT_SCALE] > 0.f) gscale *= scale_state[MSC_OPT_SCALE];      // dynamic loss scale (fp16 training)
    const double A = 2.0 * sums[1] + cfg.smooth;
    const double B = sums[2] + sums[3] + cfg.smooth + cfg.eps;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss)
        loss[0] = (float)(cfg.ce_weight * sums[0] / total_pixels + cfg.dice_weight * (1.0 - A / B));
    const float ce_k = (float)(cfg.ce_weight / total_pixels) * gscale;
    const float dk_a = (float)(cfg.dice_weight * A / (B * B)) * gscale;   // d(1-A/B)/dp1 = (A - 2 t B)/B^2
    const float dk_b = (float)(cfg.dice_weight * 2.0 / B) * gscale;
    const long total = (long)N * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const long n = p / HW, hw = p - n * HW;
        const PixelTerms t = pixel_terms(logits, target, tc, cfg, n, hw, HW);
        const float g1_ce = ce_k * t.w * (t.p1 - t.t1);          // d ce / d l1 ; d/d l0 is its negative (2 classes)
        // d dice / d l1: softmax dq1/dl1 = p1 p0 = -dq1/dl0; sigmoid dq1/dl1 = q1 (1 - q1) and l0 does not enter
        const float dq = cfg.dice_sigmoid ? t.q1 * (1.f - t.q1) : t.p1 * t.p0;
        const float ddice = (dk_a - dk_b * t.t1) * dq;
        dlogits[(n * 2) * HW + hw] = -g1_ce - (cfg.dice_sigmoid ? 0.f : ddice);
        dlogits[(n * 2 + 1) * HW + hw] = g1_ce + ddice;
    }
}

}  // namespace

static int loss_check(const char* name, const float* logits, const float* target, int tc, const msc_loss_cfg* cfg, int N, int H, int W) {
    if (!logits || !target || !cfg) return msc_fail(MSC_ERR_ARG, "%s: null pointer", name);
    if (N <= 0 || H <= 0 || W <= 0) return msc_fail(MSC_ERR_ARG, "%s: empty batch", name);
    if (cfg->weighted ? tc != 3 : tc < 1) return msc_fail(MSC_ERR_ARG, "%s: target has %d channels", name, tc);
    return MSC_OK;
}

extern "C" int msc_loss_sums(const float* logits, const float* target, int tc, const msc_loss_cfg* cfg, double* sums,
                             int N, int H, int W, void* stream) {
    int rc = loss_check("msc_loss_sums", logits, target, tc, cfg, N, H, W);
    if (rc) return rc;
    if (!sums) return msc_fail(MSC_ERR_ARG, "msc_loss_sums: null sums");
    hipStream_t st = (hipStream_t)stream;
    if (msc_memset_zero(sums, 4 * sizeof(double), stream) != MSC_OK) return MSC_ERR_HIP;       // (a kernel under MSC_MEMOPS_KERNEL=1)
    const long total = (long)N * H * W;
    long blocks = (total + 255) / 256;
    // every block ends in four fp64 atomics on the SAME four addresses, which the L2 serialises (~12 ns each).  Measured on the train step (round 6,
    // tools/gpu_tail_atomics_ab.sh): 2048 blocks 34.2 us, 1024: 25.9, 512: 28.6, 256: 42.9, 128: 78.5 (too few waves for the loads).  MSC_LOSS_BLOCKS sets the cap
    static const long cap = [] { const char* e = getenv("MSC_LOSS_BLOCKS"); return e ? atol(e) : 1024L; }();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(loss_sums_kernel, dim3((int)blocks), dim3(256), 0, st, logits, target, tc, *cfg, sums, N, (long)H * W);
    return msc_check_launch("msc_loss_sums");
}

extern "C" int msc_loss_grad(const float* logits, const float* target, int tc, const msc_loss_cfg* cfg, const double* sums,
                             double total_pixels, float grad_scale, const float* scale_state, float* loss, float* dlogits, int N, int H, int W,
                             void* stream) {
    int rc = loss_check("msc_loss_grad", logits, target, tc, cfg, N, H, W);
    if (rc) return rc;
    if (!sums || !dlogits || total_pixels <= 0) return msc_fail(MSC_ERR_ARG, "msc_loss_grad: bad argument");
    const long total = (long)N * H * W;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(loss_grad_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, logits, target, tc, *cfg, sums,
                       total_pixels, grad_scale, scale_state, loss, dlogits, N, (long)H * W);
    return msc_check_launch("msc_loss_grad");
}
