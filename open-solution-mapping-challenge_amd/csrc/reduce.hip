// Per-channel reductions over the pixel dimension of NHWC tensors on gfx950 (HBM-bound):
// BatchNorm2d training statistics finalisation, BatchNorm2d backward sums, conv-bias gradients.
//
// Pattern: column reduce.  A block owns 128 pixels x up to 64 16-byte channel vectors; lanes run along
// the channel dimension (coalesced 16-B loads), row lanes stride over pixels, LDS folds the row lanes.
// BatchNorm-backward sums (K = 2) are then added to the per-XCD slot of the layer ([MSC_BN_SLOTS][C][2], common.h), which
// msc_bn_bwd_apply sums in its prologue; bias gradients (K = 1) write one partial per (channel, pixel chunk) to a [C][S]
// workspace that a second kernel (one wavefront per channel) adds up.
#include "common.h"
#include "msc_internal.h"

namespace {

constexpr int RED_PIX = 128;       // pixels per block: halved (down to the row-lane count) until ~1024 blocks exist,
                                   // doubled (up to 2048) while more than 4096 would: the finalize pass reads one partial per block

// K = 2: (sum dh, sum dh*y) with dh = dout*[out>0] (BatchNorm backward);  K = 1: sum d (bias gradient)
template <typename T, int K>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ dout, long dout_ld, const T* __restrict__ out, long out_ld,
                                                        const T* __restrict__ y, long y_ld, int relu, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, void* __restrict__ partials_,
                                                        long pixels, int C, int cols, int S, int ppb, T* __restrict__ dx, long dx_ld, int chunks, int xcd_order) {
    constexpr int CE = Vec16<T>::N;
    __shared__ float red[256 * K * CE];
    float* partials = reinterpret_cast<float*>(partials_);          // K = 1: [C][S] f32 workspace; K = 2: double slots, below
    const int tid = threadIdx.x;
    const int col = tid % cols, r = tid / cols, R = 256 / cols;
    // 1-D grid of S * chunks blocks; XCD-aware (round 4, as the BatchNorm apply kernels): workgroup b runs on XCD b % 8, every XCD gets a
    // contiguous run of pixel slices (channel chunk fastest) -- the pixel range the data-gradient conv that wrote `dout` gave it
    int wgid = (int)blockIdx.x, ps, cy;
    if (xcd_order) {
        const int nwg = (int)gridDim.x, xcd = wgid & 7, wq = nwg >> 3, wr = nwg & 7;
        wgid = (xcd < wr ? xcd * (wq + 1) : wr * (wq + 1) + (xcd - wr) * wq) + (wgid >> 3);
        ps = wgid / chunks; cy = wgid - ps * chunks;
    } else {
        ps = wgid % S; cy = wgid / S;
    }
    const int vc = cy * cols + col;
    const long p0 = (long)ps * ppb;
    const long p1 = min(pixels, p0 + ppb);
    float s1[CE], s2[CE];
#pragma unroll
    for (int e = 0; e < CE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const bool live = vc * CE < C;
    if (live) {
        float sc[CE], sh[CE];
        if (K == 2 && relu == 2) {
#pragma unroll
            for (int e = 0; e < CE; ++e) { sc[e] = scale[vc * CE + e]; sh[e] = shift[vc * CE + e]; }
        }
        for (long p = p0 + r; p < p1; p += R) {
            float d[CE];
            Vec16<T>::load(dout + p * dout_ld + vc * CE, d);
            if (relu == 1) {
                float o[CE];
                Vec16<T>::load(out + p * out_ld + vc * CE, o);
#pragma unroll
                for (int e = 0; e < CE; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
            }
            if (K == 1 && dx) Vec16<T>::store(dx + p * dx_ld + vc * CE, d);      // the masked gradient itself (ReLU backward)
            if (K == 2) {
                float yy[CE];
                Vec16<T>::load(y + p * y_ld + vc * CE, yy);
                if (relu == 2) {            // the forward's pre-activation, recomputed instead of reading `out`
#pragma unroll
                    for (int e = 0; e < CE; ++e) d[e] = fmaf(yy[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < CE; ++e) s2[e] += d[e] * yy[e];
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) s1[e] += d[e];
        }
    }
    float* mine = red + tid * K * CE;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        mine[e] = s1[e];
        if (K == 2) mine[CE + e] = s2[e];
    }
    __syncthreads();
    if (r == 0 && live) {
        for (int k = 1; k < R; ++k) {
            const float* o = red + (k * cols + col) * K * CE;
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                s1[e] += o[e];
                if (K == 2) s2[e] += o[CE + e];
            }
        }
        if (K == 2) {              // BatchNorm-backward sums: back to LDS as [channel][2] for a coalesced atomic pass below
#pragma unroll
            for (int e = 0; e < CE; ++e) *reinterpret_cast<float2*>(red + (col * CE + e) * 2) = make_float2(s1[e], s2[e]);
        } else {
#pragma unroll
            for (int e = 0; e < CE; ++e) partials[(long)(vc * CE + e) * S + ps] = s1[e];
        }
    }
    if (K == 2) {
        // one slot per XCD ([MSC_BN_SLOTS][C][2], common.h); consecutive lanes -> consecutive floats: an atomic costs the L2 per
        // touched line, so one instruction covers whole lines
        __syncthreads();
        const int cbase = cy * cols * CE;
        const int nval = min(cols * CE, C - cbase) * 2;
        double* slot = reinterpret_cast<double*>(partials_) + ((long)msc_xcc_id() * C + cbase) * 2;      // double: see conv_epilogue (igemm.hip)
        for (int f = tid; f < nval; f += 256) atomicAdd(slot + f, (double)red[f]);
    }
}

// one wavefront per channel over partials [C][S][K]
template <int K>
__device__ __forceinline__ void channel_sums(const float* __restrict__ partials, int S, int c, int lane, double* a, double* b) {
    double s1 = 0.0, s2 = 0.0;
    const float* p = partials + (long)c * S * K;
    for (int s = lane; s < S; s += 64) {
        if (K == 2) {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * s);
            s1 += v.x; s2 += v.y;
        } else {
            s1 += p[s];
        }
    }
    *a = wave_sum_d(s1);
    *b = K == 2 ? wave_sum_d(s2) : 0.0;
}

__global__ __launch_bounds__(256) void bias_finalize_kernel(const float* __restrict__ partials, int S, int C, float* db) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1, s2;
    channel_sums<1>(partials, S, c, lane, &s1, &s2);
    if (lane == 0) db[c] += (float)s1;
}

// bias gradient from the per-XCD slots a data-gradient conv with stats_kind 2 filled ([MSC_BN_SLOTS][Cs][2] doubles, first of
// each pair): db[c] += sum over the slots, c < C <= Cs
__global__ void bias_slots_finalize_kernel(const double* __restrict__ slots, int Cs, float* __restrict__ db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
#pragma unroll
    for (int x = 0; x < MSC_BN_SLOTS; ++x) s += slots[((long)x * Cs + c) * 2];
    db[c] += (float)s;
}

// the same for several layers in one launch (one block per layer): the decoder's ReLU layers all at once
struct BiasSlotsBatch { msc_bias_slots_item it[MSC_BIAS_SLOTS_MAX]; };
__global__ __launch_bounds__(256) void bias_slots_finalize_multi_kernel(BiasSlotsBatch b) {
    const msc_bias_slots_item it = b.it[blockIdx.x];
    for (int c = threadIdx.x; c < it.C; c += 256) {
        double s = 0.0;
#pragma unroll
        for (int x = 0; x < MSC_BN_SLOTS; ++x) s += it.slots[((long)x * it.Cs + c) * 2];
        it.db[c] += (float)s;
    }
}

// launch geometry shared by the reduce launches and the workspace-size queries
struct RedGeom { int cols, chunks, ppb, S; };
bool red_geom(long pixels, int C, int ce, RedGeom* g) {
    const int cv = C / ce;
    int cols = 64;
    while (cols > cv) cols >>= 1;                 // 4..64, power of two
    if (cols < 1 || cv % cols || C % ce) return false;
    const int rows = 256 / cols;
    int ppb = RED_PIX;
    while (ppb > rows && ppb > 16 && (long)ceil_div(pixels, ppb) * (cv / cols) < 1024) ppb >>= 1;
    while (ppb < 2048 && (long)ceil_div(pixels, ppb) * (cv / cols) > 4096) ppb <<= 1;
    g->cols = cols; g->chunks = cv / cols; g->ppb = ppb; g->S = ceil_div(pixels, ppb);
    return true;
}

template <typename T, int K>
int launch_colreduce(const void* dout, long dout_ld, const void* out, long out_ld, const void* y, long y_ld, int relu,
                     const float* scale, const float* shift, void* partials, long pixels, int C, hipStream_t st,
                     void* dx = nullptr, long dx_ld = 0) {
    RedGeom g;
    if (!red_geom(pixels, C, Vec16<T>::N, &g)) return msc_fail(MSC_ERR_UNSUPPORTED, "column reduce: C=%d not supported", C);
    static const int xo = [] { const char* e = getenv("MSC_BN_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
    hipLaunchKernelGGL((colreduce_kernel<T, K>), dim3(g.S * g.chunks), dim3(256), 0, st, (const T*)dout, dout_ld, (const T*)out, out_ld, (const T*)y, y_ld,
                       relu, scale, shift, partials, pixels, C, g.cols, g.S, g.ppb, (T*)dx, dx_ld, g.chunks, xo);
    return msc_check_launch("colreduce");
}

}  // namespace

#define DT_CHECK(name, dtype) \
    if (!msc_dtype_ok(dtype)) return msc_fail(MSC_ERR_ARG, name ": dtype %d", (int)(dtype))

extern "C" int msc_bn_bwd_reduce(const void* dout, int64_t dout_ld, const void* out, int64_t out_ld, const void* y, int64_t y_ld,
                                 int relu, const float* scale, const float* shift, double* slots, int dtype, int64_t pixels, int C,
                                 void* stream) {
    DT_CHECK("msc_bn_bwd_reduce", dtype);
    void* partials = slots;
    if (!dout || !y || !partials || relu < 0 || relu > 2 || (relu == 1 && !out) || (relu == 2 && (!scale || !shift)) || pixels <= 0)
        return msc_fail(MSC_ERR_ARG, "msc_bn_bwd_reduce: bad argument");
    if (C % (4 * msc_dtype_vec(dtype))) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bn_bwd_reduce: C=%d must be a multiple of %d", C, 4 * msc_dtype_vec(dtype));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MSC_F16) return launch_colreduce<f16_t, 2>(dout, dout_ld, out, out_ld, y, y_ld, relu, scale, shift, partials, pixels, C, st);
    else if (dtype == MSC_BF16) return launch_colreduce<bf16_t, 2>(dout, dout_ld, out, out_ld, y, y_ld, relu, scale, shift, partials, pixels, C, st);
    return launch_colreduce<float, 2>(dout, dout_ld, out, out_ld, y, y_ld, relu, scale, shift, partials, pixels, C, st);
}

extern "C" int64_t msc_bias_grad_workspace_bytes(int64_t pixels, int C, int dtype) {
    RedGeom g;
    if (pixels <= 0 || C <= 0 || !red_geom(pixels, C, msc_dtype_vec(dtype), &g)) return 0;
    return (int64_t)g.S * C * (int64_t)sizeof(float);
}

extern "C" int msc_bias_grad(const void* dy, int64_t dy_ld, float* db, void* workspace, int dtype, int64_t pixels, int C, void* stream) {
    DT_CHECK("msc_bias_grad", dtype);
    if (!dy || !db || !workspace || pixels <= 0) return msc_fail(MSC_ERR_ARG, "msc_bias_grad: bad argument");
    if (C % (4 * msc_dtype_vec(dtype))) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_bias_grad: C=%d must be a multiple of %d", C, 4 * msc_dtype_vec(dtype));
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MSC_F16  ? launch_colreduce<f16_t, 1>(dy, dy_ld, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, (float*)workspace, pixels, C, st)
           : dtype == MSC_BF16 ? launch_colreduce<bf16_t, 1>(dy, dy_ld, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, (float*)workspace, pixels, C, st)
                               : launch_colreduce<float, 1>(dy, dy_ld, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, (float*)workspace, pixels, C, st);
    if (rc) return rc;
    RedGeom g;
    red_geom(pixels, C, msc_dtype_vec(dtype), &g);
    hipLaunchKernelGGL(bias_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, g.S, C, db);
    return msc_check_launch("msc_bias_grad");
}

extern "C" int msc_bias_slots_finalize(const double* slots, int Cs, float* db, int C, void* stream) {
    if (!slots || !db || C <= 0 || Cs < C) return msc_fail(MSC_ERR_ARG, "msc_bias_slots_finalize: bad argument");
    hipLaunchKernelGGL(bias_slots_finalize_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, slots, Cs, db, C);
    return msc_check_launch("msc_bias_slots_finalize");
}

extern "C" int msc_bias_slots_finalize_multi(const msc_bias_slots_item* items, int n, void* stream) {
    if (!items || n < 1 || n > MSC_BIAS_SLOTS_MAX) return msc_fail(MSC_ERR_ARG, "msc_bias_slots_finalize_multi: 1..%d items", MSC_BIAS_SLOTS_MAX);
    BiasSlotsBatch b;
    for (int i = 0; i < n; ++i) {
        if (!items[i].slots || !items[i].db || items[i].C <= 0 || items[i].Cs < items[i].C)
            return msc_fail(MSC_ERR_ARG, "msc_bias_slots_finalize_multi: bad item %d", i);
        b.it[i] = items[i];
    }
    hipLaunchKernelGGL(bias_slots_finalize_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, b);
    return msc_check_launch("msc_bias_slots_finalize_multi");
}

extern "C" int msc_relu_bias_grad(const void* dy, int64_t dy_ld, const void* y, int64_t y_ld, void* dx, int64_t dx_ld, float* db,
                                  void* workspace, int dtype, int64_t pixels, int C, void* stream) {
    DT_CHECK("msc_relu_bias_grad", dtype);
    if (!dy || !y || !dx || !db || !workspace || pixels <= 0) return msc_fail(MSC_ERR_ARG, "msc_relu_bias_grad: bad argument");
    if (C % (4 * msc_dtype_vec(dtype))) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_relu_bias_grad: C=%d must be a multiple of %d", C, 4 * msc_dtype_vec(dtype));
    hipStream_t st = (hipStream_t)stream;
    int rc = dtype == MSC_F16  ? launch_colreduce<f16_t, 1>(dy, dy_ld, y, y_ld, nullptr, 0, 1, nullptr, nullptr, (float*)workspace, pixels, C, st, dx, dx_ld)
           : dtype == MSC_BF16 ? launch_colreduce<bf16_t, 1>(dy, dy_ld, y, y_ld, nullptr, 0, 1, nullptr, nullptr, (float*)workspace, pixels, C, st, dx, dx_ld)
                               : launch_colreduce<float, 1>(dy, dy_ld, y, y_ld, nullptr, 0, 1, nullptr, nullptr, (float*)workspace, pixels, C, st, dx, dx_ld);
    if (rc) return rc;
    RedGeom g;
    red_geom(pixels, C, msc_dtype_vec(dtype), &g);
    hipLaunchKernelGGL(bias_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, g.S, C, db);
    return msc_check_launch("msc_relu_bias_grad");
}
