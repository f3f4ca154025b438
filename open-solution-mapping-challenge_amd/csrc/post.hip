// Mask post-processing on gfx950: the per-image Python / scipy / scikit-image / pydensecrf loops of
// src/postprocessing.py:48-258 and src/utils.py:328-339 as batched HIP kernels.  All of it is
// HBM/latency-bound integer & byte work over [B,H,W] planes: one launch covers the whole batch
// (grid.y = image), lanes run along x so every access is coalesced.
//
// Integer results (threshold layers, labels, erosion/dilation, dropped objects) are bit-exact
// against oracle/post_ref.py; float results (resize, score, CRF) within the tolerances in tests/.
#include <stdlib.h>

#include "common.h"
#include "msc_internal.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int n) {  // scipy.ndimage 'reflect' (d c b a | a b c d | d c b a)
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - i - 1;
    return min(max(i, 0), n - 1);
}

// ------------------------------------------------------------------ resize / crop / threshold / argmax
// The reference thresholds the FLOAT64 map skimage's resize returns (src/postprocessing.py:60,83).  The interpolant is therefore
// computed here with exactly numpy's roundings: one rounding per operation in the order of the expression (no fused multiply-add:
// hipcc contracts a*b+c by default), then skimage's clip to the input range joined with cval 0 (warp(..., clip=True)); thresholds
// are compared in double against it, and only the stored float32 map (what the scoring reads) is rounded.
__device__ __forceinline__ double bilinear_f64(const float* __restrict__ p, int h, int w, int oy, int ox, double fy, double fx, double lo, double hi) {
#pragma clang fp contract(off)
    const double ys = fy * (oy + 0.5) - 0.5, xs = fx * (ox + 0.5) - 0.5;
    // scipy 'constant' mode: no interpolation beyond the edges -> cval
    if (!(ys >= 0.0 && ys <= (double)(h - 1) && xs >= 0.0 && xs <= (double)(w - 1))) return fmin(fmax(0.0, lo), hi);
    int y0 = (int)floor(ys), x0 = (int)floor(xs);
    y0 = min(max(y0, 0), h - 1); x0 = min(max(x0, 0), w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    // scipy's NI_GeometricTransform: weights 1-a and 1-(1-a) per axis, per tap (coefficient * row weight) * column weight, taps in raster order
    const double wy0 = 1.0 - (ys - y0), wx0 = 1.0 - (xs - x0);
    const double wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
    double v = ((double)p[(long)y0 * w + x0] * wy0) * wx0;
    v = v + ((double)p[(long)y0 * w + x1] * wy0) * wx1;
    v = v + ((double)p[(long)y1 * w + x0] * wy1) * wx0;
    v = v + ((double)p[(long)y1 * w + x1] * wy1) * wx1;
    return fmin(fmax(v, lo), hi);
}

// per image: (min(image.min(), 0), max(image.max(), 0)) -- the range skimage clips the resized map to; one block per image
__global__ void image_minmax_kernel(const float* __restrict__ in, float* __restrict__ minmax, long per_image) {
    __shared__ float smn[16], smx[16];
    const float* p = in + blockIdx.x * per_image;
    float mn = 0.f, mx = 0.f;
    for (long i = threadIdx.x; i < per_image; i += blockDim.x) { const float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    for (int o = 32; o; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
        minmax[2 * blockIdx.x] = mn; minmax[2 * blockIdx.x + 1] = mx;
    }
}

template <typename OUT>
__global__ void resize_bilinear_kernel(const float* __restrict__ in, OUT* __restrict__ out, const float* __restrict__ minmax, int C, int planes, int h, int w, int H, int W) {
    const long total = (long)planes * H * W;
    const double fy = (double)h / H, fx = (double)w / W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W), oy = (int)((i / W) % H);
        const long pl = i / ((long)W * H);
        const long b = pl / C;
        out[i] = (OUT)bilinear_f64(in + pl * (long)h * w, h, w, oy, ox, fy, fx, (double)minmax[2 * b], (double)minmax[2 * b + 1]);
    }
}

// resize + categorize_multilayer_image in one pass: the float32 map for the scoring, the layers from the double interpolant
__global__ void resize_threshold_kernel(const float* __restrict__ in, float* __restrict__ out, uint8_t* __restrict__ layers, const float* __restrict__ minmax,
                                        int B, int C, int h, int w, int H, int W, const int32_t* __restrict__ layer_class, const double* __restrict__ layer_thr, int L) {
    const long HW = (long)H * W, total = (long)B * HW;
    const double fy = (double)h / H, fx = (double)w / W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW, b = i / HW;
        const int ox = (int)(hw % W), oy = (int)(hw / W);
        const double lo = (double)minmax[2 * b], hi = (double)minmax[2 * b + 1];
        for (int c = 0; c < C; ++c) {
            const double v = bilinear_f64(in + (b * C + c) * (long)h * w, h, w, oy, ox, fy, fx, lo, hi);
            out[(b * C + c) * HW + hw] = (float)v;
            for (int l = 0; l < L; ++l)
                if (layer_class[l] == c) layers[(b * L + l) * HW + hw] = v > layer_thr[l] ? 1 : 0;
        }
    }
}

__global__ void crop_center_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int hc, int wc, int hs, int ws) {
    const long total = (long)planes * hc * wc;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % wc), y = (int)((i / wc) % hc);
        const long pl = i / ((long)wc * hc);
        out[i] = in[(pl * h + y + hs) * w + x + ws];
    }
}

template <typename IN>
__global__ void threshold_layers_kernel(const IN* __restrict__ probs, uint8_t* __restrict__ layers, int B, int C, long HW,
                                        const int32_t* __restrict__ layer_class, const double* __restrict__ layer_thr, int L) {
    const long total = (long)B * L * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW;
        const int l = (int)((i / HW) % L);
        const long b = i / (HW * L);
        layers[i] = (double)probs[(b * C + layer_class[l]) * HW + hw] > layer_thr[l] ? 1 : 0;      // numpy: array > float64 scalar compares in double
    }
}

__global__ void argmax_channels_kernel(const float* __restrict__ probs, int32_t* __restrict__ out, int B, int C, long HW) {
    const long total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW, b = i / HW;
        int best = 0;
        float bv = probs[(b * C) * HW + hw];
        for (int c = 1; c < C; ++c) {
            const float v = probs[(b * C + c) * HW + hw];
            if (v > bv) { bv = v; best = c; }
        }
        out[i] = best;
    }
}

// ------------------------------------------------------------------ k x k rectangle min / max filters
template <typename T, bool IS_MAX>
__global__ void rect_filter_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int lo, int hi) {
    const long total = (long)B * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const long b = i / ((long)W * H);
        const T* p = in + b * (long)H * W;
        T acc = p[(long)reflect_idx(y + lo, H) * W + reflect_idx(x + lo, W)];
        for (int dy = lo; dy <= hi; ++dy) {
            const int yy = reflect_idx(y + dy, H);
            for (int dx = lo; dx <= hi; ++dx) {
                const T v = p[(long)yy * W + reflect_idx(x + dx, W)];
                acc = IS_MAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
            }
        }
        out[i] = acc;
    }
}

// round 6: the same filter, separable and tiled.  min / max over a k x k window is the min / max over the rows of the row-wise min / max, with scipy's
// 'reflect' indices on either axis (a reflected index only repeats values the window holds anyway) -- identical results.  A block owns a 32 x 64 tile:
// its halo goes to LDS once (coalesced rows), the row pass and the column pass read LDS.  The 2-D loop above re-reads k*k global values per pixel through
// byte-wide gathers: 434 us for the 5 x 5 erosion of 256 planes of 300 x 300 (the watershed markers), 106 us for the 2 x 2 label dilation.
constexpr int RF_TH = 32, RF_TW = 64, RF_MAXK = 17;      // window extents hi - lo + 1 up to 17
template <typename T, bool IS_MAX>
__global__ __launch_bounds__(256) void rect_filter_tiled_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int lo, int hi) {
    __shared__ T a[(RF_TH + RF_MAXK - 1) * (RF_TW + RF_MAXK - 1)];
    __shared__ T r[(RF_TH + RF_MAXK - 1) * RF_TW];
    const int ext = hi - lo, hh = RF_TH + ext, ww = RF_TW + ext;
    const int x0 = blockIdx.x * RF_TW, y0 = blockIdx.y * RF_TH;
    const T* p = in + (long)blockIdx.z * H * W;
    for (int i = threadIdx.x; i < hh * ww; i += 256) {
        const int yy = i / ww, xx = i - yy * ww;
        // rows / columns past H - 1 + hi feed no pixel of the image (partial tiles): any in-range value will do
        const int y = reflect_idx(min(y0 + lo + yy, H - 1 + hi), H), x = reflect_idx(min(x0 + lo + xx, W - 1 + hi), W);
        a[i] = p[(long)y * W + x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hh * RF_TW; i += 256) {
        const int yy = i / RF_TW, x = i - yy * RF_TW;
        T v = a[yy * ww + x];
        for (int d = 1; d <= ext; ++d) {
            const T u = a[yy * ww + x + d];
            v = IS_MAX ? (u > v ? u : v) : (u < v ? u : v);
        }
        r[i] = v;
    }
    __syncthreads();
    T* o = out + (long)blockIdx.z * H * W;
    for (int i = threadIdx.x; i < RF_TH * RF_TW; i += 256) {
        const int y = i / RF_TW, x = i - y * RF_TW;
        if (y0 + y >= H || x0 + x >= W) continue;
        T v = r[i];
        for (int d = 1; d <= ext; ++d) {
            const T u = r[(y + d) * RF_TW + x];
            v = IS_MAX ? (u > v ? u : v) : (u < v ? u : v);
        }
        o[(long)(y0 + y) * W + x0 + x] = v;
    }
}
template <typename T, bool IS_MAX>
static void launch_rect_filter(const T* in, T* out, int B, int H, int W, int lo, int hi, hipStream_t st) {
    static const bool tiled_off = [] { const char* e = getenv("MSC_RECT_TILED"); return e && e[0] == '0'; }();      // A/B: the 2-D loop of rounds 1-5
    if (hi - lo + 1 <= RF_MAXK && !tiled_off)
        hipLaunchKernelGGL((rect_filter_tiled_kernel<T, IS_MAX>), dim3((W + RF_TW - 1) / RF_TW, (H + RF_TH - 1) / RF_TH, B), dim3(256), 0, st, in, out, H, W, lo, hi);
    else {
        long blocks = ((long)B * H * W + 255) / 256;
        hipLaunchKernelGGL((rect_filter_kernel<T, IS_MAX>), dim3((unsigned)(blocks > 4096 ? 4096 : blocks < 1 ? 1 : blocks)), dim3(256), 0, st, in, out, B, H, W, lo, hi);
    }
}

// ------------------------------------------------------------------ connected components, 4-connectivity
// Union-find over pixel indices with atomicMin links: the root of a component is its smallest pixel
// index = its first pixel in raster order, so numbering roots by an exclusive prefix count in raster
// order reproduces scipy.ndimage.label's numbering exactly.
__device__ __forceinline__ int uf_load(const int* L, int a) {
    return __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(const int* L, int a) {
    int p = uf_load(L, a);
    while (p != a) { a = p; p = uf_load(L, a); }
    return a;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
    bool done = false;
    while (!done) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a < b) {
            const int old = atomicMin(L + b, a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(L + a, b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    }
}

// One wavefront per image row: every foreground pixel starts as a child of the first pixel of its horizontal run
// (ballot of 64 pixels -> run starts -> highest start at or below the lane; the run open at a chunk boundary is
// carried in a scalar), so the horizontal unions are done before any atomic is issued.
__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ L, int H, int W, long rows) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int y = (int)(row % H);
    const uint8_t* m = mask + row * W;
    int* l = L + row * W;
    int carry = -1;                                   // start of the run that reaches the previous chunk's last pixel
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool fg = x < W && m[x] != 0;
        const unsigned long long f = __ballot(fg);
        const unsigned long long starts = f & ~((f << 1) | (carry >= 0 ? 1ull : 0ull));
        const unsigned long long below = starts & (~0ull >> (63 - lane));
        const int sx = below ? x0 + 63 - __clzll((long long)below) : carry;
        if (x < W) l[x] = fg ? y * W + sx : -1;
        const int last = __shfl(fg ? sx : -1, 63, 64);
        carry = (f >> 63) ? last : -1;
    }
}
// vertical unions, once per pair of touching runs: at the first pixel where the two runs overlap
__global__ void ccl_merge_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ L, int H, int W) {
    const long HW = (long)H * W;
    const long b = blockIdx.y;
    const uint8_t* m = mask + b * HW;
    int* l = L + b * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        if (p < W || !m[p] || !m[p - W]) continue;
        const int x = (int)(p % W);
        if (x > 0 && m[p - 1] && m[p - W - 1]) continue;
        uf_union(l, (int)p, (int)(p - W));
    }
}
__global__ void ccl_compress_kernel(int32_t* __restrict__ L, long HW) {
    const long b = blockIdx.y;
    int* l = L + b * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int v = uf_load(l, (int)p);
        if (v >= 0 && v != (int)p) {
            const int r = uf_find(l, v);
            __hip_atomic_store(l + p, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// round 6: the unions of a strip of rows in LDS.  ccl_merge_kernel chases parent pointers through global memory (a dependent L2 round trip per hop, atomicMin
// links contended by every run that touches a component): 137 us per 64 tiles, and ccl_compress_kernel's chains were as deep as the image (93 us).  Here a
// block owns CCL_STRIP_PIX / W rows: run labelling (as ccl_init_kernel), the vertical unions and the path compression all happen on an int array in LDS,
// the strip leaves as finished roots; the rows where two strips meet are united in global memory (ccl_seam_kernel: one row per seam) and the final
// compression finds every root within a hop or two.  Label values are pixel indices of the PLANE throughout (LDS index + base), so the root of a
// component stays its first pixel in raster order.
constexpr int CCL_STRIP_PIX = 16384;       // 64 KB of LDS: two 1024-thread blocks per CU
__device__ __forceinline__ int ufl_load(const int* lab, int base, int a) {
    return __hip_atomic_load(lab + (a - base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int ufl_find(const int* lab, int base, int a) {
    int p = ufl_load(lab, base, a);
    while (p != a) { a = p; p = ufl_load(lab, base, a); }
    return a;
}
__device__ __forceinline__ void ufl_union(int* lab, int base, int a, int b) {
    bool done = false;
    while (!done) {
        a = ufl_find(lab, base, a);
        b = ufl_find(lab, base, b);
        if (a < b) {
            const int old = atomicMin(lab + (b - base), a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            const int old = atomicMin(lab + (a - base), b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    }
}
__global__ __launch_bounds__(1024) void ccl_strip_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ L, int H, int W, int SR) {
    __shared__ int lab[CCL_STRIP_PIX];
    const long HW = (long)H * W;
    const long b = blockIdx.y;
    const int y0 = blockIdx.x * SR;
    const int rows = min(SR, H - y0);
    const int base = y0 * W, total = rows * W;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // runs of every row: a foreground pixel starts as a child of the first pixel of its horizontal run (ccl_init_kernel)
    for (int r = wid; r < rows; r += 16) {
        const uint8_t* m = mask + b * HW + base + (long)r * W;
        int* l = lab + r * W;
        int carry = -1;
        for (int x0 = 0; x0 < W; x0 += 64) {
            const int x = x0 + lane;
            const bool fg = x < W && m[x] != 0;
            const unsigned long long f = __ballot(fg);
            const unsigned long long starts = f & ~((f << 1) | (carry >= 0 ? 1ull : 0ull));
            const unsigned long long below = starts & (~0ull >> (63 - lane));
            const int sx = below ? x0 + 63 - __clzll((long long)below) : carry;
            if (x < W) l[x] = fg ? base + r * W + sx : -1;
            const int last = __shfl(fg ? sx : -1, 63, 64);
            carry = (f >> 63) ? last : -1;
        }
    }
    __syncthreads();
    // vertical unions inside the strip, once per pair of touching runs (ccl_merge_kernel's rule; foreground = label >= 0)
    for (int i = W + threadIdx.x; i < total; i += 1024) {
        if (lab[i] < 0 || lab[i - W] < 0) continue;
        const int x = i % W;
        if (x > 0 && lab[i - 1] >= 0 && lab[i - W - 1] >= 0) continue;
        ufl_union(lab, base, base + i, base + i - W);
    }
    __syncthreads();
    int* out = L + b * HW + base;
    for (int i = threadIdx.x; i < total; i += 1024) {
        const int v = lab[i];
        out[i] = v >= 0 ? ufl_find(lab, base, v) : -1;
    }
}
// the first row of every strip but the first against the row above it, in global memory (labels are strip roots by now)
__global__ void ccl_seam_kernel(int32_t* __restrict__ L, int H, int W, int SR) {
    const long HW = (long)H * W;
    int* l = L + blockIdx.y * HW;
    const int y = (blockIdx.x + 1) * SR;
    if (y >= H) return;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int p = y * W + x;
        if (uf_load(l, p) < 0 || uf_load(l, p - W) < 0) continue;
        if (x > 0 && uf_load(l, p - 1) >= 0 && uf_load(l, p - W - 1) >= 0) continue;
        uf_union(l, p, p - W);
    }
}
// one block per image: rank[p] = number of roots before p (raster order), for roots only.  The image is walked in chunks of
// 1024 consecutive pixels (coalesced: lane = pixel), the roots of a chunk are ranked by wave ballots + a scan of the 16 wave
// totals; one thread per contiguous 88-pixel segment (64 cache lines per wave-instruction) took 99 us per 128 images.
__global__ __launch_bounds__(1024) void ccl_rank_kernel(const int32_t* __restrict__ L, int32_t* __restrict__ rank,
                                                        int32_t* __restrict__ counts, long HW) {
    __shared__ int wtot[16];
    __shared__ int base_s;
    const long b = blockIdx.x;
    const int* l = L + b * HW;
    int* r = rank + b * HW;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (long p0 = 0; p0 < HW; p0 += 1024) {
        const long p = p0 + threadIdx.x;
        const bool root = p < HW && l[p] == (int)p;
        const unsigned long long m = __ballot(root);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wid] = __popcll(m);
        __syncthreads();
        int off = base_s;
#pragma unroll
        for (int w = 0; w < 16; ++w) off += w < wid ? wtot[w] : 0;
        if (root) r[p] = off + before;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) t += wtot[w];
            base_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && counts) counts[b] = base_s;
}
__global__ void ccl_relabel_kernel(int32_t* __restrict__ L, const int32_t* __restrict__ rank, long HW) {
    const long b = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int root = L[b * HW + p];
        L[b * HW + p] = root >= 0 ? rank[b * HW + root] + 1 : 0;
    }
}

// ------------------------------------------------------------------ add_dropped_objects
__global__ void dropped_mark_kernel(const uint8_t* __restrict__ processed, const int32_t* __restrict__ lab, int32_t* __restrict__ alive, long HW) {
    const long b = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int l = lab[b * HW + p];
        // the reference tests np.any(np.where(overlap)), i.e. the INDEX arrays: an overlap consisting of pixel (0,0) alone has
        // only zero indices and counts as "no surviving pixel" (src/utils.py:337)
        if (l > 0 && processed[b * HW + p] && p != 0) alive[b * HW + l - 1] = 1;
    }
}
__global__ void dropped_apply_kernel(const uint8_t* __restrict__ processed, const int32_t* __restrict__ lab,
                                     const int32_t* __restrict__ alive, uint8_t* __restrict__ out, long HW, int bool_sum) {
    const long b = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int l = lab[b * HW + p];
        const int add = (l > 0 && !alive[b * HW + l - 1]) ? 1 : 0;
        const int v = processed[b * HW + p] + add;            // `reconstructed += (labeled == i)`: a sum for integer masks ...
        out[b * HW + p] = (uint8_t)(bool_sum ? (v != 0) : v);   // ... a logical or for the bool masks of categorize_*
    }
}

// ------------------------------------------------------------------ watershed (EXTENSION: WATERSHED.md; no reference function)
// Synchronous immersion flood of the 8-bit relief of 1 - P inside the mask, from labelled markers; one workgroup per
// image-layer.  The definition (WATERSHED.md) is a sequence of synchronous rounds, level by level; run literally it is
// 256 levels x (rounds until nothing changes) block-wide barriers -- hundreds of L2 round trips per layer (round 2: 85 us).
// The same result without walking the levels: the round in which a pixel is labelled is its ARRIVAL TIME
//     T(p) = (level, round), markers at (-1, 0);   T(p) = min over 4-neighbours q of  next(T(q), h(p)),
//     next((l, r), h) = (l, r + 1) if h <= l  (p is eligible already: the round after q)   else (h, 1)  (first round of p's own level)
// -- a shortest-path problem with a strictly increasing step, so plain relaxation in ANY order reaches its unique solution --
// and the label it takes in that round is the smallest one among the neighbours labelled BEFORE it:
//     label(p) = min { label(q) : T(q) < T(p) },
// a second monotone fixed point over the (acyclic) order T defines.  Both are swept over the work list of unlabelled mask pixels
// until nothing changes: the number of sweeps is the flood DISTANCE in pixels (a few for the eroded-mask markers), not the number
// of grey levels.  Bit-identical to the round-by-round oracle (tests/test_gpu_post.py; the arrival-time form is also checked
// against it on the CPU in tests/test_oracle_watershed.py).  T is packed as ((level + 1) << 20) | round: 0 for markers.
constexpr int WS_INF = 0x7fffffff;
constexpr int WS_TILE = 64, WS_HALO = WS_TILE + 2;

__device__ __forceinline__ int ws_step(int tq, int h1) {      // arrival time of a pixel of relief h1 - 1 reached from a neighbour that arrived at tq
    return h1 > (tq >> 20) ? ((h1 << 20) | 1) : tq + 1;
}

// relief, arrival times of the markers (0) / everything else (infinity), the work list of unlabelled mask pixels: one workgroup per layer
__global__ __launch_bounds__(1024) void watershed_init_kernel(const float* __restrict__ prob, const uint8_t* __restrict__ mask, int32_t* labels,
                                                              uint8_t* hq, int32_t* tarr, int32_t* list, int32_t* nlist, int H, int W) {
    __shared__ int n_list;
    const long HW = (long)H * W;
    const long base = (long)blockIdx.x * HW;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    for (long p = threadIdx.x; p < HW; p += blockDim.x) {
        float v = floorf((1.0f - prob[base + p]) * 255.0f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        hq[base + p] = (uint8_t)v;
        int t = WS_INF;
        if (!mask[base + p]) labels[base + p] = 0;
        else if (labels[base + p] != 0) t = 0;
        else list[base + atomicAdd(&n_list, 1)] = (int32_t)p;
        tarr[base + p] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) nlist[blockIdx.x] = n_list;
}

// Block-Jacobi pass of either fixed point (both are order-independent): a block takes a 64x64 tile with a one-pixel ring of its
// neighbours' current values into LDS, relaxes it to ITS fixed point there (a sweep is ~100 cycles instead of an L2 round trip),
// and writes the tile back.  Values only ever decrease towards the solution, so stale ring values are harmless; what a pass cannot
// do is carry information across more than one tile border -- a few passes cover the flood distances of eroded-mask markers, and
// the finish kernels below complete whatever is left, so the result does not depend on the number of passes.
template <bool LABELS>
__global__ __launch_bounds__(256) void watershed_tile_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ hq, int32_t* tarr,
                                                             int32_t* labels, int H, int W) {
    __shared__ int sT[WS_HALO * WS_HALO];
    __shared__ int sL[LABELS ? WS_HALO * WS_HALO : 1];
    __shared__ uint8_t sH[WS_HALO * WS_HALO];
    __shared__ uint16_t sQ[WS_TILE * WS_TILE];       // the tile's own work list: the unlabelled mask pixels (typically a fifth of the tile)
    __shared__ int nQ;
    const long HW = (long)H * W;
    const long base = (long)blockIdx.z * HW;
    const int x0 = blockIdx.x * WS_TILE, y0 = blockIdx.y * WS_TILE;
    if (threadIdx.x == 0) nQ = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < WS_HALO * WS_HALO; i += blockDim.x) {
        const int hy = i / WS_HALO, hx = i - hy * WS_HALO;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        int t = WS_INF, l = 0, h = 0, wk = 0;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const long p = base + (long)y * W + x;
            t = tarr[p];
            h = hq[p];
            if (LABELS) l = labels[p];
            const bool interior = hy >= 1 && hy <= WS_TILE && hx >= 1 && hx <= WS_TILE;
            wk = interior && mask[p] && t != 0 && (!LABELS || t != WS_INF);      // markers (t == 0) never change
        }
        sT[i] = t; sH[i] = (uint8_t)h;
        if (LABELS) sL[i] = l;
        if (wk) sQ[atomicAdd(&nQ, 1)] = (uint16_t)i;
    }
    __syncthreads();
    const int n = nQ;
    if (n == 0) return;                              // no unlabelled mask pixel in this tile
    for (;;) {
        int changed = 0;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const int i = sQ[k];
            if (!LABELS) {
                const int h1 = (int)sH[i] + 1, cur = sT[i];
                int best = cur;
                const int nb[4] = {sT[i - WS_HALO], sT[i + WS_HALO], sT[i - 1], sT[i + 1]};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (nb[q] != WS_INF) { const int c = ws_step(nb[q], h1); best = c < best ? c : best; }
                if (best < cur) { sT[i] = best; changed = 1; }
            } else {
                const int t = sT[i], cur = sL[i];
                int best = cur > 0 ? cur : WS_INF;
                const int off[4] = {-WS_HALO, WS_HALO, -1, 1};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (sT[i + off[q]] < t) { const int lq = sL[i + off[q]]; if (lq > 0 && lq < best) best = lq; }
                if (best != WS_INF && best != cur) { sL[i] = best; changed = 1; }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int i = sQ[k];
        const int ty = i / WS_HALO - 1, tx = i % WS_HALO - 1;
        const long p = base + (long)(y0 + ty) * W + x0 + tx;
        if (LABELS) labels[p] = sL[i];
        else tarr[p] = sT[i];
    }
}

// the same two relaxations over a layer's work list by ONE workgroup, swept until nothing changes: after the tiled passes this is
// normally a single confirming sweep; on masks whose flood paths wind through many tiles it does the remaining work
template <bool LABELS>
__global__ __launch_bounds__(1024) void watershed_finish_kernel(const uint8_t* __restrict__ hq, int32_t* tarr, int32_t* labels,
                                                                const int32_t* __restrict__ list, const int32_t* __restrict__ nlist, int H, int W) {
    const long HW = (long)H * W;
    const long base = (long)blockIdx.x * HW;
    int32_t* L = labels + base;
    const uint8_t* Hq = hq + base;
    int32_t* T = tarr + base;
    const int32_t* Q = list + base;
    const int n = nlist[blockIdx.x];
    for (;;) {
        int changed = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const long p = Q[i];
            const int y = (int)(p / W), x = (int)(p - (long)y * W);
            if (!LABELS) {
                const int h1 = (int)Hq[p] + 1;
                const int cur = T[p];
                int best = cur;
                auto relax = [&](long q) {
                    const int tq = T[q];
                    if (tq != WS_INF) { const int c = ws_step(tq, h1); best = c < best ? c : best; }
                };
                if (y > 0) relax(p - W);
                if (y + 1 < H) relax(p + W);
                if (x > 0) relax(p - 1);
                if (x + 1 < W) relax(p + 1);
                if (best < cur) { T[p] = best; changed = 1; }
            } else {
                const int t = T[p];
                if (t == WS_INF) continue;             // not reachable from any marker (cannot happen for masks from erode_image)
                const int cur = L[p];
                int best = cur > 0 ? cur : WS_INF;
                auto take = [&](long q) {
                    if (T[q] < t) { const int lq = L[q]; if (lq > 0 && lq < best) best = lq; }
                };
                if (y > 0) take(p - W);
                if (y + 1 < H) take(p + W);
                if (x > 0) take(p - 1);
                if (x + 1 < W) take(p + 1);
                if (best != WS_INF && best != cur) { L[p] = best; changed = 1; }
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
}

// ------------------------------------------------------------------ build_score
__global__ void score_accum_kernel(const int32_t* __restrict__ labels, const float* __restrict__ probs, double* __restrict__ sums,
                                   int32_t* __restrict__ areas, long HW, int max_labels) {
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const long stride = (long)gridDim.x * blockDim.x;
    const long iters = (HW + stride - 1) / stride;
    for (long it = 0; it < iters; ++it) {
        const long p = it * stride + blockIdx.x * (long)blockDim.x + threadIdx.x;
        int lab = 0;
        float pr = 0.f;
        if (p < HW) { lab = labels[b * HW + p]; pr = probs[b * HW + p]; }
        bool active = lab > 0 && lab <= max_labels;
        // wave-level aggregation: one atomic per distinct label per wave (a row segment has 1-3)
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int l0 = __shfl(lab, leader, 64);
            const bool mine = active && lab == l0;
            const double s = wave_sum_d(mine ? (double)pr : 0.0);
            const unsigned long long mm = __ballot(mine);
            if (lane == leader) {
                atomicAdd(sums + b * max_labels + l0 - 1, s);
                atomicAdd(areas + b * max_labels + l0 - 1, (int)__popcll(mm));
            }
            active = active && !mine;
            todo &= ~mm;
        }
    }
}
// round 6: the same sums with the label accumulators of a 4096-pixel chunk in LDS (ds_add_f64 / ds_add_u32) and ONE pass of global atomics per chunk and
// touched label.  The first form issued its atomics per wave (64 pixels): ~1400 waves per plane adding into a few dozen addresses serialise in the
// L2 -- 138 us per 64 tiles for 92 MB of reads.  Labels above SCORE_LDS_MAX go the old way (msc_build_score falls back to score_accum_kernel).
constexpr int SCORE_LDS_MAX = 2048, SCORE_CHUNK = 4096;
__global__ __launch_bounds__(256) void score_accum_lds_kernel(const int32_t* __restrict__ labels, const float* __restrict__ probs, double* __restrict__ sums,
                                                              int32_t* __restrict__ areas, long HW, int max_labels) {
    __shared__ double s_sum[SCORE_LDS_MAX];
    __shared__ int s_area[SCORE_LDS_MAX];
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < max_labels; i += 256) { s_sum[i] = 0.0; s_area[i] = 0; }
    __syncthreads();
    const long p0 = (long)blockIdx.x * SCORE_CHUNK;
    int lab[SCORE_CHUNK / 256];
    float pr[SCORE_CHUNK / 256];
#pragma unroll
    for (int k = 0; k < SCORE_CHUNK / 256; ++k) {          // all loads of the chunk in flight together
        const long p = p0 + k * 256 + threadIdx.x;
        lab[k] = p < HW ? labels[b * HW + p] : 0;
        pr[k] = p < HW ? probs[b * HW + p] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SCORE_CHUNK / 256; ++k) {
        bool active = lab[k] > 0 && lab[k] <= max_labels;
        unsigned long long todo = __ballot(active);
        while (todo) {                                      // one LDS add per distinct label of the wave's 64 pixels (a row segment has 1-3)
            const int leader = __ffsll((long long)todo) - 1;
            const int l0 = __shfl(lab[k], leader, 64);
            const bool mine = active && lab[k] == l0;
            const double s = wave_sum_d(mine ? (double)pr[k] : 0.0);
            const unsigned long long mm = __ballot(mine);
            if (lane == leader) {
                atomicAdd(&s_sum[l0 - 1], s);
                atomicAdd(&s_area[l0 - 1], (int)__popcll(mm));
            }
            active = active && !mine;
            todo &= ~mm;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < max_labels; i += 256) {
        const int a = s_area[i];
        if (a) {
            atomicAdd(sums + b * max_labels + i, s_sum[i]);
            atomicAdd(areas + b * max_labels + i, a);
        }
    }
}
__global__ void score_final_kernel(const double* __restrict__ sums, const int32_t* __restrict__ areas, double* __restrict__ score, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int a = areas[i];
        score[i] = a > 0 ? sums[i] / (double)a * sqrt((double)a) : 0.0;
    }
}

// ------------------------------------------------------------------ dense CRF, exact windowed mean field
struct CrfP { int H, W, rg, rb; float inv2g, inv2b, inv2rgb, compat_g, compat_b; };

__device__ __forceinline__ float rgb_d2(const uint8_t* a, const uint8_t* b) {
    const float d0 = (float)a[0] - (float)b[0], d1 = (float)a[1] - (float)b[1], d2 = (float)a[2] - (float)b[2];
    return d0 * d0 + d1 * d1 + d2 * d2;
}

__global__ void crf_norm_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ ng, float* __restrict__ nb, CrfP c) {
    const long HW = (long)c.H * c.W, b = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % c.W), y = (int)(p / c.W);
        const uint8_t* me = rgb + (b * HW + p) * 3;
        float sg = 0.f, sb = 0.f;
        for (int dy = -c.rg; dy <= c.rg; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)c.H) continue;
            for (int dx = -c.rg; dx <= c.rg; ++dx)
                if ((unsigned)(x + dx) < (unsigned)c.W) sg += expf(-(float)(dy * dy + dx * dx) * c.inv2g);
        }
        for (int dy = -c.rb; dy <= c.rb; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)c.H) continue;
            for (int dx = -c.rb; dx <= c.rb; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)c.W) continue;
                sb += expf(-(float)(dy * dy + dx * dx) * c.inv2b - rgb_d2(me, rgb + (b * HW + (long)yy * c.W + xx) * 3) * c.inv2rgb);
            }
        }
        ng[b * HW + p] = 1.f / sqrtf(sg + 1e-20f);
        nb[b * HW + p] = 1.f / sqrtf(sb + 1e-20f);
    }
}

__device__ __forceinline__ void softmax2(float a0, float a1, float* q0, float* q1) {
    const float m = fmaxf(a0, a1);
    const float e0 = expf(a0 - m), e1 = expf(a1 - m);
    const float s = e0 + e1;
    *q0 = e0 / s; *q1 = e1 / s;
}

__global__ void crf_init_kernel(const float* __restrict__ probs, float* __restrict__ q, long HW) {
    const long b = blockIdx.y;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const float u0 = -logf(fminf(fmaxf(probs[(b * 2) * HW + p], 1e-5f), 1.f));
        const float u1 = -logf(fminf(fmaxf(probs[(b * 2 + 1) * HW + p], 1e-5f), 1.f));
        softmax2(-u0, -u1, &q[(b * 2) * HW + p], &q[(b * 2 + 1) * HW + p]);
    }
}

__global__ void crf_iter_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ rgb, const float* __restrict__ ng,
                                const float* __restrict__ nb, const float* __restrict__ qin, float* __restrict__ qout, CrfP c) {
    const long HW = (long)c.H * c.W, b = blockIdx.y;
    const float* q0 = qin + (b * 2) * HW;
    const float* q1 = q0 + HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % c.W), y = (int)(p / c.W);
        const uint8_t* me = rgb + (b * HW + p) * 3;
        float g0 = 0.f, g1 = 0.f, b0 = 0.f, b1 = 0.f;
        for (int dy = -c.rg; dy <= c.rg; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)c.H) continue;
            for (int dx = -c.rg; dx <= c.rg; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)c.W) continue;
                const long j = (long)yy * c.W + xx;
                const float k = expf(-(float)(dy * dy + dx * dx) * c.inv2g) * ng[b * HW + j];
                g0 += k * q0[j]; g1 += k * q1[j];
            }
        }
        for (int dy = -c.rb; dy <= c.rb; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)c.H) continue;
            for (int dx = -c.rb; dx <= c.rb; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)c.W) continue;
                const long j = (long)yy * c.W + xx;
                const float k = expf(-(float)(dy * dy + dx * dx) * c.inv2b - rgb_d2(me, rgb + (b * HW + j) * 3) * c.inv2rgb) * nb[b * HW + j];
                b0 += k * q0[j]; b1 += k * q1[j];
            }
        }
        const float u0 = -logf(fminf(fmaxf(probs[(b * 2) * HW + p], 1e-5f), 1.f));
        const float u1 = -logf(fminf(fmaxf(probs[(b * 2 + 1) * HW + p], 1e-5f), 1.f));
        const float n_g = ng[b * HW + p], n_b = nb[b * HW + p];
        const float t0 = -u0 + c.compat_g * g0 * n_g + c.compat_b * b0 * n_b;
        const float t1 = -u1 + c.compat_g * g1 * n_g + c.compat_b * b1 * n_b;
        softmax2(t0, t1, &qout[(b * 2) * HW + p], &qout[(b * 2 + 1) * HW + p]);
    }
}

// ---- tiled form (window radius <= CRF_RMAX): a block owns a 32x32 tile; the (32+2r)^2 halo of what every tap needs -- the
// neighbour's Q already multiplied by its two normalisers, and its colour -- goes to LDS once per iteration and the 121 taps
// read it from there.  The naive kernels above load 5 global values per tap (605 per pixel and kernel): 672 us per iteration for
// 32 images of 256x256; tiled, the iteration is bound by the per-tap arithmetic (3 colour differences, a dot product, one exp2,
// four FMAs): a thread owns FOUR vertically adjacent pixels, so a halo value it loads serves up to four (pixel, tap) pairs -- with
// one pixel per thread the two ds_read_b128 per tap, not the VALU, set the pace (222 us per iteration for 32 images; the LDS pipe
// moves 128 B/clk for the whole CU).  Spatial weights are tables of (2r+1)^2 floats in LDS (broadcast
// reads), folded into the exponent for the bilateral kernel: k = exp2(c_sp[t] - |dI|^2 * inv2rgb * log2(e)).
constexpr int CRF_T = 32, CRF_PV = 4, CRF_RMAX = 8, CRF_HMAX = CRF_T + 2 * CRF_RMAX;      // 32x32 tile, 256 threads x 4 vertical pixels

struct CrfHalo {
    float4 a[CRF_HMAX * CRF_HMAX];      // (q0*ng, q1*ng, q0*nb, q1*nb) of the halo pixel (zeros outside the image)
    float4 c[CRF_HMAX * CRF_HMAX];      // (r, g, b, inside)
    float gsp[(2 * CRF_RMAX + 1) * (2 * CRF_RMAX + 1)];      // Gaussian kernel: exp(-d^2 * inv2g)
    float bsp[(2 * CRF_RMAX + 1) * (2 * CRF_RMAX + 1)];      // bilateral kernel, spatial part as an exponent of 2: -d^2 * inv2b * log2(e)
};

__device__ __forceinline__ void crf_tables(CrfHalo& s, const CrfP& c, int r) {
    const int w = 2 * r + 1;
    for (int i = threadIdx.x; i < w * w; i += blockDim.x) {
        const int dy = i / w - r, dx = i % w - r;
        const float d2 = (float)(dy * dy + dx * dx);
        s.gsp[i] = (abs(dy) <= c.rg && abs(dx) <= c.rg) ? expf(-d2 * c.inv2g) : 0.f;
        s.bsp[i] = (abs(dy) <= c.rb && abs(dx) <= c.rb) ? -d2 * c.inv2b * 1.44269504f : -1e30f;
    }
}

// normalisers n = 1 / sqrt(sum_j k(i, j) + 1e-20) of both kernels (NORMALIZE_SYMMETRIC), window clipped to the image
__global__ __launch_bounds__(256) void crf_norm_tiled_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ ng, float* __restrict__ nb, CrfP c, int r) {
    __shared__ CrfHalo s;
    const long HW = (long)c.H * c.W, b = blockIdx.z;
    const int x0 = blockIdx.x * CRF_T, y0 = blockIdx.y * CRF_T, hw = CRF_T + 2 * r;
    crf_tables(s, c, r);
    for (int i = threadIdx.x; i < hw * hw; i += blockDim.x) {
        const int yy = y0 - r + i / hw, xx = x0 - r + i % hw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
            const uint8_t* px = rgb + (b * HW + (long)yy * c.W + xx) * 3;
            v = make_float4((float)px[0], (float)px[1], (float)px[2], 1.f);
        }
        s.c[i] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;      // this thread's pixels: (tx, ty .. ty+3) of the tile
    const int x = x0 + tx;
    if (x >= c.W || y0 + ty >= c.H) return;
    float4 me[CRF_PV];
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) me[k] = s.c[(ty + k + r) * hw + tx + r];
    const float krgb = c.inv2rgb * 1.44269504f;
    float sg[CRF_PV] = {0.f, 0.f, 0.f, 0.f}, sb[CRF_PV] = {0.f, 0.f, 0.f, 0.f};
    const int w = 2 * r + 1;
    for (int hy = 0; hy < CRF_PV + 2 * r; ++hy)            // halo rows this thread's four windows cover
        for (int dx = -r; dx <= r; ++dx) {
            const float4 o = s.c[(ty + hy) * hw + tx + r + dx];
#pragma unroll
            for (int k = 0; k < CRF_PV; ++k) {
                const int dy = hy - k - r;                  // offset of this halo row from pixel k
                if (dy < -r || dy > r) continue;
                const int t = (dy + r) * w + dx + r;
                const float d0 = o.x - me[k].x, d1 = o.y - me[k].y, d2 = o.z - me[k].z;
                sg[k] += s.gsp[t] * o.w;
                sb[k] += __builtin_amdgcn_exp2f(s.bsp[t] - (d0 * d0 + d1 * d1 + d2 * d2) * krgb) * o.w;      // v_exp_f32: the argument is <= 0
            }
        }
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const int y = y0 + ty + k;
        if (y >= c.H) break;
        ng[b * HW + (long)y * c.W + x] = 1.f / sqrtf(sg[k] + 1e-20f);
        nb[b * HW + (long)y * c.W + x] = 1.f / sqrtf(sb[k] + 1e-20f);
    }
}

// one mean-field iteration: Q <- softmax(-U + compat_g * K_g Q + compat_b * K_b Q)
__global__ __launch_bounds__(256) void crf_iter_tiled_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ rgb, const float* __restrict__ ng,
                                                             const float* __restrict__ nb, const float* __restrict__ qin, float* __restrict__ qout, CrfP c, int r) {
    __shared__ CrfHalo s;
    const long HW = (long)c.H * c.W, b = blockIdx.z;
    const int x0 = blockIdx.x * CRF_T, y0 = blockIdx.y * CRF_T, hw = CRF_T + 2 * r;
    const float* q0 = qin + (b * 2) * HW;
    const float* q1 = q0 + HW;
    crf_tables(s, c, r);
    for (int i = threadIdx.x; i < hw * hw; i += blockDim.x) {
        const int yy = y0 - r + i / hw, xx = x0 - r + i % hw;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vc = va;
        if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
            const long j = (long)yy * c.W + xx;
            const float a0 = q0[j], a1 = q1[j], n_g = ng[b * HW + j], n_b = nb[b * HW + j];
            const uint8_t* px = rgb + (b * HW + j) * 3;
            va = make_float4(a0 * n_g, a1 * n_g, a0 * n_b, a1 * n_b);
            vc = make_float4((float)px[0], (float)px[1], (float)px[2], 1.f);
        }
        s.a[i] = va;
        s.c[i] = vc;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;
    const int x = x0 + tx;
    if (x >= c.W || y0 + ty >= c.H) return;
    float4 me[CRF_PV];
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) me[k] = s.c[(ty + k + r) * hw + tx + r];
    const float krgb = c.inv2rgb * 1.44269504f;
    float g0[CRF_PV] = {0.f, 0.f, 0.f, 0.f}, g1[CRF_PV] = {0.f, 0.f, 0.f, 0.f}, b0[CRF_PV] = {0.f, 0.f, 0.f, 0.f}, b1[CRF_PV] = {0.f, 0.f, 0.f, 0.f};
    const int w = 2 * r + 1;
    for (int hy = 0; hy < CRF_PV + 2 * r; ++hy)
        for (int dx = -r; dx <= r; ++dx) {
            const int h = (ty + hy) * hw + tx + r + dx;
            const float4 o = s.c[h];
            const float4 qv = s.a[h];                       // out-of-image neighbours carry Q = 0
#pragma unroll
            for (int k = 0; k < CRF_PV; ++k) {
                const int dy = hy - k - r;
                if (dy < -r || dy > r) continue;
                const int t = (dy + r) * w + dx + r;
                const float d0 = o.x - me[k].x, d1 = o.y - me[k].y, d2 = o.z - me[k].z;
                const float kg = s.gsp[t];
                const float kb = __builtin_amdgcn_exp2f(s.bsp[t] - (d0 * d0 + d1 * d1 + d2 * d2) * krgb);
                g0[k] += kg * qv.x; g1[k] += kg * qv.y;
                b0[k] += kb * qv.z; b1[k] += kb * qv.w;
            }
        }
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const int y = y0 + ty + k;
        if (y >= c.H) break;
        const long p = (long)y * c.W + x;
        const float u0 = -logf(fminf(fmaxf(probs[(b * 2) * HW + p], 1e-5f), 1.f));
        const float u1 = -logf(fminf(fmaxf(probs[(b * 2 + 1) * HW + p], 1e-5f), 1.f));
        const float n_g = ng[b * HW + p], n_b = nb[b * HW + p];
        const float t0 = -u0 + c.compat_g * g0[k] * n_g + c.compat_b * b0[k] * n_b;
        const float t1 = -u1 + c.compat_g * g1[k] * n_g + c.compat_b * b1[k] * n_b;
        softmax2(t0, t1, &qout[(b * 2) * HW + p], &qout[(b * 2 + 1) * HW + p]);
    }
}

// ---- round 4: the same tiled iteration for window radii <= 5 (the reference's sxy = 1 for both kernels) with the inner loops laid out
// for the machine.  The first tiled form keeps 32 bytes per halo pixel and the two spatial tables in LDS (74 KB per block: two blocks =
// two waves per SIMD) and reads the tables with two broadcast ds_read_b32 per (tap, pixel) inside a loop the compiler cannot unroll
// (runtime radius): every k-iteration is a dependent LDS read -> exp -> FMA chain that nothing else covers -- 857 us per iteration
// for 128 images against ~260 us of VALU issue.  Here: radius as template parameter (the 11 columns of a halo row are straight-line
// code: all their LDS reads are in flight together); colours packed to one dword per halo pixel (20 bytes per pixel, 35 KB per
// block: four blocks per CU); the spatial weights are SEPARABLE -- exp(-(dx^2+dy^2)c) = gx[dx]*gy[dy], and for the bilateral kernel the
// exponent is bx[dx] + by[dy] -- so the column factors sit in registers and the row factors cost one v_exp_f32 per (halo row, pixel).
template <int R>
struct CrfHaloR {
    static constexpr int HW_ = CRF_T + 2 * R;
    float4 a[HW_ * HW_];        // (q0*ng, q1*ng, q0*nb, q1*nb); zeros outside the image
    uint32_t c[HW_ * HW_];      // r | g << 8 | b << 16 | inside << 24
};

// NORM: the normalisers (sum of the kernel weights over the window clipped to the image) instead of one iteration
template <int R> struct CrfCols { float gx[2 * R + 1], bx[2 * R + 1]; };      // column factors, computed on the host: kernel arguments = SGPR operands

// ---- round 5: the same kernel on PACKED fp32 math.  The vector peak the roofline quotes (157 TFLOP/s) is the rate of v_pk_fma_f32 -- two
// lanes' worth of fp32 FMA per issue slot; scalar-fp32 code tops out at half of it, and the round-4 form (13 VALU + 1 v_exp_f32 per tap and
// pixel, all unpacked) sat at 0.20 of that peak = 0.40 of what unpacked code can reach.  Here a lane's four vertical pixels are two PAIRS
// (k, k+1): colour differences, squared distance, exponent argument, the Gaussian factor and the four accumulations are v_pk_* over the
// pair (13 packed ops + 2 v_exp_f32 per tap and pair: ~10.5 issue slots per tap and pixel instead of 17); a pair's window is 12 halo rows
// (the row that only one of the two pixels reaches runs with a zero / -1e30 factor for the other).  Same arithmetic per element as the
// unpacked form (IEEE fma either way), so the results agree with it to the last bit wherever the compiler contracts the same products.
typedef float crf_f2 __attribute__((ext_vector_type(2)));

template <int R, bool NORM, bool PK = false>
__global__ __launch_bounds__(256, 4) void crf_r_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ rgb, float* __restrict__ ng,
                                                    float* __restrict__ nb, const float* __restrict__ qin, float* __restrict__ qout, CrfP c,
                                                    CrfCols<R> cols) {
    __shared__ CrfHaloR<R> s;
    constexpr int hw = CRF_T + 2 * R;
    const long HW = (long)c.H * c.W, b = blockIdx.z;
    const int x0 = blockIdx.x * CRF_T, y0 = blockIdx.y * CRF_T;
    const float* q0 = NORM ? nullptr : qin + (b * 2) * HW;
    const float* q1 = NORM ? nullptr : q0 + HW;
    for (int i = threadIdx.x; i < hw * hw; i += blockDim.x) {
        const int yy = y0 - R + i / hw, xx = x0 - R + i % hw;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t vc = 0;
        if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
            const long j = (long)yy * c.W + xx;
            const uint8_t* px = rgb + (b * HW + j) * 3;
            vc = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | (1u << 24);
            if (!NORM) {
                const float a0 = q0[j], a1 = q1[j], n_g = ng[b * HW + j], n_b = nb[b * HW + j];
                va = make_float4(a0 * n_g, a1 * n_g, a0 * n_b, a1 * n_b);
            }
        }
        if (!NORM) s.a[i] = va;
        s.c[i] = vc;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;
    const int x = x0 + tx;
    if (x >= c.W || y0 + ty >= c.H) return;
    // column factors of the two kernels come as kernel arguments (scalar registers: the dx loop below is unrolled, so cols.gx[i] is a
    // fixed SGPR); a kernel whose own radius is smaller has zeros / -1e30 there
    const float kg2 = c.inv2g * 1.44269504f, kb2 = c.inv2b * 1.44269504f, krgb = c.inv2rgb * 1.44269504f;
    float g0[CRF_PV], g1[CRF_PV], b0[CRF_PV], b1[CRF_PV];
    if constexpr (PK) {
        static_assert(CRF_PV % 2 == 0, "pairs of vertical pixels");
        constexpr int NP = CRF_PV / 2;
        crf_f2 mr[NP], mg[NP], mb[NP], G0[NP], G1[NP], B0[NP], B1[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const uint32_t m0 = s.c[(ty + 2 * j + R) * hw + tx + R], m1 = s.c[(ty + 2 * j + 1 + R) * hw + tx + R];
            mr[j] = crf_f2{(float)(m0 & 255u), (float)(m1 & 255u)};
            mg[j] = crf_f2{(float)((m0 >> 8) & 255u), (float)((m1 >> 8) & 255u)};
            mb[j] = crf_f2{(float)((m0 >> 16) & 255u), (float)((m1 >> 16) & 255u)};
            G0[j] = G1[j] = B0[j] = B1[j] = crf_f2{0.f, 0.f};
        }
        for (int hy = 0; hy < CRF_PV + 2 * R; ++hy) {
            crf_f2 gy[NP], by[NP];
            bool on[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int dy0 = hy - 2 * j - R, dy1 = dy0 - 1, a0 = dy0 < 0 ? -dy0 : dy0, a1 = dy1 < 0 ? -dy1 : dy1;
                on[j] = a0 <= R || a1 <= R;                                   // wave-uniform: the pair's 12-row window
                gy[j] = crf_f2{a0 <= c.rg ? __builtin_amdgcn_exp2f(-(float)(dy0 * dy0) * kg2) : 0.f,
                               a1 <= c.rg ? __builtin_amdgcn_exp2f(-(float)(dy1 * dy1) * kg2) : 0.f};
                by[j] = crf_f2{a0 <= c.rb ? -(float)(dy0 * dy0) * kb2 : -1e30f, a1 <= c.rb ? -(float)(dy1 * dy1) * kb2 : -1e30f};
            }
            const int h0 = (ty + hy) * hw + tx;
            uint32_t oc[2 * R + 1];
            float4 qv[2 * R + 1];
#pragma unroll
            for (int i = 0; i <= 2 * R; ++i) {
                oc[i] = s.c[h0 + i];
                if (!NORM) qv[i] = s.a[h0 + i];
            }
#pragma unroll
            for (int i = 0; i <= 2 * R; ++i) {
                const float orr = (float)(oc[i] & 255u), og = (float)((oc[i] >> 8) & 255u), ob = (float)((oc[i] >> 16) & 255u);
                const float ins = (float)(oc[i] >> 24);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    if (!on[j]) continue;
                    const crf_f2 d0 = orr - mr[j], d1 = og - mg[j], d2 = ob - mb[j];
                    const crf_f2 dist = d0 * d0 + d1 * d1 + d2 * d2;
                    const crf_f2 arg = (cols.bx[i] + by[j]) - dist * krgb;       // <= 0
                    const crf_f2 kb = crf_f2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                    const crf_f2 kg = cols.gx[i] * gy[j];
                    if (NORM) {
                        G0[j] += kg * ins;
                        B0[j] += kb * ins;
                    } else {
                        G0[j] += kg * qv[i].x; G1[j] += kg * qv[i].y;
                        B0[j] += kb * qv[i].z; B1[j] += kb * qv[i].w;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            g0[2 * j] = G0[j].x; g0[2 * j + 1] = G0[j].y; g1[2 * j] = G1[j].x; g1[2 * j + 1] = G1[j].y;
            b0[2 * j] = B0[j].x; b0[2 * j + 1] = B0[j].y; b1[2 * j] = B1[j].x; b1[2 * j + 1] = B1[j].y;
        }
    } else {
    float mr[CRF_PV], mg[CRF_PV], mb[CRF_PV];
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const uint32_t m = s.c[(ty + k + R) * hw + tx + R];
        mr[k] = (float)(m & 255u); mg[k] = (float)((m >> 8) & 255u); mb[k] = (float)((m >> 16) & 255u);
    }
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) { g0[k] = 0.f; g1[k] = 0.f; b0[k] = 0.f; b1[k] = 0.f; }
    for (int hy = 0; hy < CRF_PV + 2 * R; ++hy) {
        // row factors for the (up to four) pixels this halo row is in the window of; dy is the same for every lane.  Rows outside a pixel's
        // window are skipped with a wave-uniform branch -- measured against the branch-free form (factors 0 / -1e30 for those rows, the
        // four pixels' chains interleaved by the scheduler): that one needs 182-254 registers (two or three waves per SIMD) or spills at
        // 128, and ran 1.18-1.25 ms per 32 images against 0.90 ms for this one at four waves per SIMD (profiles/r4_run7_crf_variants.txt)
        float gy[CRF_PV], by[CRF_PV];
        bool on[CRF_PV];
#pragma unroll
        for (int k = 0; k < CRF_PV; ++k) {
            const int dy = hy - k - R, ad = dy < 0 ? -dy : dy;
            on[k] = ad <= R;
            gy[k] = ad <= c.rg ? __builtin_amdgcn_exp2f(-(float)(dy * dy) * kg2) : 0.f;
            by[k] = ad <= c.rb ? -(float)(dy * dy) * kb2 : -1e30f;
        }
        const int h0 = (ty + hy) * hw + tx;
        uint32_t oc[2 * R + 1];
        float4 qv[2 * R + 1];
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) {
            oc[i] = s.c[h0 + i];
            if (!NORM) qv[i] = s.a[h0 + i];
        }
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) {
            const float orr = (float)(oc[i] & 255u), og = (float)((oc[i] >> 8) & 255u), ob = (float)((oc[i] >> 16) & 255u);
            const float ins = (float)(oc[i] >> 24);
#pragma unroll
            for (int k = 0; k < CRF_PV; ++k) {
                if (!on[k]) continue;                                        // wave-uniform
                const float d0 = orr - mr[k], d1 = og - mg[k], d2 = ob - mb[k];
                const float kg = cols.gx[i] * gy[k];
                const float kb = __builtin_amdgcn_exp2f((cols.bx[i] + by[k]) - (d0 * d0 + d1 * d1 + d2 * d2) * krgb);      // argument <= 0
                if (NORM) {
                    g0[k] += kg * ins;
                    b0[k] += kb * ins;
                } else {
                    g0[k] += kg * qv[i].x; g1[k] += kg * qv[i].y;            // out-of-image neighbours carry Q = 0
                    b0[k] += kb * qv[i].z; b1[k] += kb * qv[i].w;
                }
            }
        }
    }
    }
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const int y = y0 + ty + k;
        if (y >= c.H) break;
        const long p = (long)y * c.W + x;
        if (NORM) {
            ng[b * HW + p] = 1.f / sqrtf(g0[k] + 1e-20f);
            nb[b * HW + p] = 1.f / sqrtf(b0[k] + 1e-20f);
        } else {
            const float u0 = -logf(fminf(fmaxf(probs[(b * 2) * HW + p], 1e-5f), 1.f));
            const float u1 = -logf(fminf(fmaxf(probs[(b * 2 + 1) * HW + p], 1e-5f), 1.f));
            const float n_g = ng[b * HW + p], n_b = nb[b * HW + p];
            const float t0 = -u0 + c.compat_g * g0[k] * n_g + c.compat_b * b0[k] * n_b;
            const float t1 = -u1 + c.compat_g * g1[k] * n_g + c.compat_b * b1[k] * n_b;
            softmax2(t0, t1, &qout[(b * 2) * HW + p], &qout[(b * 2 + 1) * HW + p]);
        }
    }
}

// ---- round 6: the bilateral kernel on EXACT integer colour distances from a dot product, the Gaussian kernel as a separable blur in its own launch.
// (a) The colours are 8-bit: |o - m|^2 = |o|^2 + |m|^2 - 2 o.m, and every term is an integer below 2^24, i.e. exact in fp32 whatever the order -- the
//     three subtractions and three multiply-adds per tap and pair become one add and three multiply-adds against the centre pixel's -2 m (per halo pixel
//     |o|^2 is computed once when the tile is loaded; 1e30 outside the image, which zeroes the weight), and the exponent argument is the same
//     fma(dist, -k, bx + by) as before: the weights are those of crf_r_kernel bit for bit.
// (b) exp(-(dx^2+dy^2)/2s^2) does not depend on the image: the Gaussian message is a separable 11 + 11 tap blur of Q * n_g (crf_gauss_kernel, ~1/10 of the
//     multiply-adds of the 121-tap form), its normaliser the product of two 1-D window sums.  The hot loop keeps 8 packed operations + 2 v_exp_f32 per tap
//     and pixel pair (13 + 2 before), 16 bytes of LDS per halo pixel (20 before).
template <int R>
struct CrfHaloX {
    static constexpr int HW_ = CRF_T + 2 * R;
    float4 v[HW_ * HW_];        // (bits: r | g << 8 | b << 16, |rgb|^2 (1e30 outside the image), q0 * nb, q1 * nb)
};

// Gaussian message of one iteration: gmsg[b][p] = (sum_w kg * q0 * ng, sum_w kg * q1 * ng), window clipped to the image (zeros outside)
template <int R>
__global__ __launch_bounds__(256) void crf_gauss_kernel(const float* __restrict__ qin, const float* __restrict__ ng, float2* __restrict__ gmsg, CrfP c, CrfCols<R> cols) {
    constexpr int hw = CRF_T + 2 * R;
    __shared__ float2 qn[hw * hw];
    __shared__ float2 hb[hw * CRF_T];
    const long HW = (long)c.H * c.W, b = blockIdx.z;
    const int x0 = blockIdx.x * CRF_T, y0 = blockIdx.y * CRF_T;
    const float* q0 = qin + (b * 2) * HW;
    const float* q1 = q0 + HW;
    for (int i = threadIdx.x; i < hw * hw; i += 256) {
        const int yy = y0 - R + i / hw, xx = x0 - R + i % hw;
        float2 v = make_float2(0.f, 0.f);
        if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
            const long j = (long)yy * c.W + xx;
            const float n = ng[b * HW + j];
            v = make_float2(q0[j] * n, q1[j] * n);
        }
        qn[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hw * CRF_T; i += 256) {
        const int row = i / CRF_T, x = i % CRF_T;
        float h0 = 0.f, h1 = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) {
            const float2 v = qn[row * hw + x + t];
            h0 = fmaf(cols.gx[t], v.x, h0); h1 = fmaf(cols.gx[t], v.y, h1);
        }
        hb[i] = make_float2(h0, h1);
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;
    const int x = x0 + tx;
    if (x >= c.W) return;
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const int y = y0 + ty + k;
        if (y >= c.H) break;
        float g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) {
            const float2 v = hb[(ty + k + t) * CRF_T + tx];
            g0 = fmaf(cols.gx[t], v.x, g0); g1 = fmaf(cols.gx[t], v.y, g1);      // the spatial kernel is symmetric: one table for both axes
        }
        gmsg[b * HW + (long)y * c.W + x] = make_float2(g0, g1);
    }
}

// FUSEG: the separable Gaussian message of the tile is computed by the same block first, in the LDS the bilateral halo then overwrites (Q * n_g of the
// halo: 14 KB, row-blurred: 10.5 KB, against 28 KB) -- no crf_gauss_kernel launch, no message buffer written and read back (64 of 400 us per iteration)
template <int R, bool NORM, bool FUSEG = false>
__global__ __launch_bounds__(256, 4) void crf_x_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ rgb, float* __restrict__ ng,
                                                       float* __restrict__ nb, const float* __restrict__ qin, const float2* __restrict__ gmsg,
                                                       float* __restrict__ qout, CrfP c, CrfCols<R> cols) {
    __shared__ CrfHaloX<R> s;
    constexpr int hw = CRF_T + 2 * R;
    const long HW = (long)c.H * c.W, b = blockIdx.z;
    const int x0 = blockIdx.x * CRF_T, y0 = blockIdx.y * CRF_T;
    const float* q0 = NORM ? nullptr : qin + (b * 2) * HW;
    const float* q1 = NORM ? nullptr : q0 + HW;
    float2 gfused[CRF_PV];
    if constexpr (FUSEG && !NORM) {
        static_assert(sizeof(float2) * (hw * hw + hw * CRF_T) <= sizeof(CrfHaloX<R>), "the Gaussian scratch fits under the bilateral halo");
        float2* qn = reinterpret_cast<float2*>(&s);
        float2* hb = qn + hw * hw;
        for (int i = threadIdx.x; i < hw * hw; i += 256) {
            const int yy = y0 - R + i / hw, xx = x0 - R + i % hw;
            float2 v = make_float2(0.f, 0.f);
            if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
                const long j = (long)yy * c.W + xx;
                const float n = ng[b * HW + j];
                v = make_float2(q0[j] * n, q1[j] * n);
            }
            qn[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < hw * CRF_T; i += 256) {
            const int row = i / CRF_T, x = i % CRF_T;
            float h0 = 0.f, h1 = 0.f;
#pragma unroll
            for (int t = 0; t <= 2 * R; ++t) {
                const float2 v = qn[row * hw + x + t];
                h0 = fmaf(cols.gx[t], v.x, h0); h1 = fmaf(cols.gx[t], v.y, h1);
            }
            hb[i] = make_float2(h0, h1);
        }
        __syncthreads();
        {
            const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;
#pragma unroll
            for (int k = 0; k < CRF_PV; ++k) {
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (int t = 0; t <= 2 * R; ++t) {
                    const float2 v = hb[(ty + k + t) * CRF_T + tx];
                    g0 = fmaf(cols.gx[t], v.x, g0); g1 = fmaf(cols.gx[t], v.y, g1);
                }
                gfused[k] = make_float2(g0, g1);
            }
        }
        __syncthreads();               // the bilateral halo goes over the scratch
    }
    for (int i = threadIdx.x; i < hw * hw; i += blockDim.x) {
        const int yy = y0 - R + i / hw, xx = x0 - R + i % hw;
        float4 v = make_float4(0.f, 1e30f, 0.f, 0.f);
        if ((unsigned)yy < (unsigned)c.H && (unsigned)xx < (unsigned)c.W) {
            const long j = (long)yy * c.W + xx;
            const uint8_t* px = rgb + (b * HW + j) * 3;
            const uint32_t r = px[0], g = px[1], bl = px[2];
            v.x = __uint_as_float(r | (g << 8) | (bl << 16));
            v.y = (float)(r * r + g * g + bl * bl);
            if (!NORM) {
                const float n_b = nb[b * HW + j];
                v.z = q0[j] * n_b; v.w = q1[j] * n_b;
            }
        }
        s.v[i] = v;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = (threadIdx.x >> 5) * CRF_PV;
    const int x = x0 + tx;
    if (x >= c.W || y0 + ty >= c.H) return;
    const float kb2 = c.inv2b * 1.44269504f, krgb = c.inv2rgb * 1.44269504f;
    static_assert(CRF_PV % 2 == 0, "pairs of vertical pixels");
    constexpr int NP = CRF_PV / 2;
    crf_f2 m2r[NP], m2g[NP], m2b[NP], km[NP], B0[NP], B1[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const float4 a0 = s.v[(ty + 2 * j + R) * hw + tx + R], a1 = s.v[(ty + 2 * j + 1 + R) * hw + tx + R];
        const uint32_t c0 = __float_as_uint(a0.x), c1 = __float_as_uint(a1.x);
        m2r[j] = crf_f2{-2.f * (float)(c0 & 255u), -2.f * (float)(c1 & 255u)};
        m2g[j] = crf_f2{-2.f * (float)((c0 >> 8) & 255u), -2.f * (float)((c1 >> 8) & 255u)};
        m2b[j] = crf_f2{-2.f * (float)((c0 >> 16) & 255u), -2.f * (float)((c1 >> 16) & 255u)};
        km[j] = crf_f2{a0.y, a1.y};              // the centre pixels are inside the image or not stored
        B0[j] = B1[j] = crf_f2{0.f, 0.f};
    }
    // one halo row against the pairs J0 .. J1-1 (compile-time: no branch inside, so the chains of the taps and of the two pairs interleave; a pair's
    // window is the 12 halo rows 2j .. 2j + 2R + 1 -- the first two rows belong to pair 0 alone, the last two to the last pair)
    auto row = [&](int hy, auto j0_tag, auto j1_tag) __attribute__((always_inline)) {
        constexpr int J0 = decltype(j0_tag)::value, J1 = decltype(j1_tag)::value;
        crf_f2 by[NP];
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const int dy0 = hy - 2 * j - R, dy1 = dy0 - 1, a0 = dy0 < 0 ? -dy0 : dy0, a1 = dy1 < 0 ? -dy1 : dy1;
            by[j] = crf_f2{a0 <= c.rb ? -(float)(dy0 * dy0) * kb2 : -1e30f, a1 <= c.rb ? -(float)(dy1 * dy1) * kb2 : -1e30f};
        }
        const int h0 = (ty + hy) * hw + tx;
        float4 hv[2 * R + 1];
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) hv[i] = s.v[h0 + i];
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) {
            const uint32_t oc = __float_as_uint(hv[i].x);
            const float orr = (float)(oc & 255u), og = (float)((oc >> 8) & 255u), ob = (float)((oc >> 16) & 255u);
#pragma unroll
            for (int j = J0; j < J1; ++j) {
                crf_f2 dist = km[j] + hv[i].y;                               // integers below 2^24 all the way: exact
                dist = orr * m2r[j] + dist;
                dist = og * m2g[j] + dist;
                dist = ob * m2b[j] + dist;
                const crf_f2 arg = (cols.bx[i] + by[j]) - dist * krgb;       // <= 0; -1e30 outside the window, -3e26 outside the image
                const crf_f2 kb = crf_f2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                if (NORM) {
                    B0[j] += kb;
                } else {
                    B0[j] += kb * hv[i].z; B1[j] += kb * hv[i].w;
                }
            }
        }
    };
    static_assert(NP == 2, "two pairs of vertical pixels per lane");
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    row(0, I0{}, I1{}); row(1, I0{}, I1{});
    for (int hy = 2; hy < 2 * R + 2; ++hy) row(hy, I0{}, I2{});
    row(2 * R + 2, I1{}, I2{}); row(2 * R + 3, I1{}, I2{});
    float sx = 0.f;
    if (NORM) {
#pragma unroll
        for (int i = 0; i <= 2 * R; ++i) sx += (unsigned)(x + i - R) < (unsigned)c.W ? cols.gx[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < CRF_PV; ++k) {
        const int y = y0 + ty + k;
        if (y >= c.H) break;
        const long p = (long)y * c.W + x;
        const float b0 = (k & 1) ? B0[k / 2].y : B0[k / 2].x, b1 = (k & 1) ? B1[k / 2].y : B1[k / 2].x;
        if (NORM) {
            float sy = 0.f;
#pragma unroll
            for (int i = 0; i <= 2 * R; ++i) sy += (unsigned)(y + i - R) < (unsigned)c.H ? cols.gx[i] : 0.f;
            ng[b * HW + p] = 1.f / sqrtf(sx * sy + 1e-20f);
            nb[b * HW + p] = 1.f / sqrtf(b0 + 1e-20f);
        } else {
            const float u0 = -logf(fminf(fmaxf(probs[(b * 2) * HW + p], 1e-5f), 1.f));
            const float u1 = -logf(fminf(fmaxf(probs[(b * 2 + 1) * HW + p], 1e-5f), 1.f));
            const float n_g = ng[b * HW + p], n_b = nb[b * HW + p];
            const float2 gm = (FUSEG && !NORM) ? gfused[k] : gmsg[b * HW + p];
            const float t0 = -u0 + c.compat_g * gm.x * n_g + c.compat_b * b0 * n_b;
            const float t1 = -u1 + c.compat_g * gm.y * n_g + c.compat_b * b1 * n_b;
            softmax2(t0, t1, &qout[(b * 2) * HW + p], &qout[(b * 2 + 1) * HW + p]);
        }
    }
}

// ------------------------------------------------------------------ test-time augmentation (src/loaders.py:401-517)
// spec bits: 0 = ud flip, 1 = lr flip (the reference's elif chain: ud wins), 2-3 = rotation / 90 (counter-clockwise,
// as skimage.rotate / np.rot90).  transformed = rot90^k(flip(image)).
__device__ __forceinline__ void tta_src_of(int i, int j, int spec, int H, int W, int* y, int* x) {
    const int k = (spec >> 2) & 3;
    int yy, xx;                                  // position before the rotation
    if (k == 0) { yy = i; xx = j; }
    else if (k == 1) { yy = j; xx = W - 1 - i; }
    else if (k == 2) { yy = H - 1 - i; xx = W - 1 - j; }
    else { yy = H - 1 - j; xx = i; }
    if (spec & 1) yy = H - 1 - yy;
    else if (spec & 2) xx = W - 1 - xx;
    *y = yy; *x = xx;
}
__device__ __forceinline__ void tta_dst_of(int y, int x, int spec, int H, int W, int* i, int* j) {
    if (spec & 1) y = H - 1 - y;
    else if (spec & 2) x = W - 1 - x;
    const int k = (spec >> 2) & 3;
    if (k == 0) { *i = y; *j = x; }
    else if (k == 1) { *i = W - 1 - x; *j = y; }
    else if (k == 2) { *i = H - 1 - y; *j = W - 1 - x; }
    else { *i = x; *j = H - 1 - y; }
}
__global__ void tta_transform_kernel(const float* __restrict__ x, float* __restrict__ out, long planes, int H, int W,
                                     const int32_t* __restrict__ specs, int V) {
    const long HW = (long)H * W, total = (long)V * planes * HW;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int j = (int)(t % W), i = (int)((t / W) % H);
        const long pl = (t / HW) % planes;
        const int v = (int)(t / (HW * planes));
        int y, xx;
        tta_src_of(i, j, specs[v], H, W, &y, &xx);
        out[t] = x[pl * HW + (long)y * W + xx];
    }
}
__global__ void tta_aggregate_kernel(const float* __restrict__ preds, float* __restrict__ out, long planes, int H, int W,
                                     const int32_t* __restrict__ specs, int V, int method) {
    const long HW = (long)H * W, total = planes * HW;
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), y = (int)((t / W) % H);
        const long pl = t / HW;
        double acc = method == 2 ? -1e300 : (method == 3 ? 1e300 : 0.0);
        for (int v = 0; v < V; ++v) {
            int i, j;
            tta_dst_of(y, x, specs[v], H, W, &i, &j);
            const float p = preds[((long)v * planes + pl) * HW + (long)i * W + j];
            if (method == 0) acc += (double)p;                       // mean
            else if (method == 1) acc += (double)logf(p);            // gmean = exp(mean(log p))
            else if (method == 2) acc = fmax(acc, (double)p);        // max
            else acc = fmin(acc, (double)p);                         // min
        }
        float r;
        if (method == 0) r = (float)(acc / V);
        else if (method == 1) r = expf((float)(acc / V));
        else r = (float)acc;
        out[t] = r;
    }
}

inline dim3 plane_grid(long HW, int B) {
    long bx = (HW + 255) / 256;
    if (bx > 1024) bx = 1024;
    return dim3((int)bx, B);
}
inline int flat_grid(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

#define POST_DIMS(name) \
    if (B <= 0 || H <= 0 || W <= 0 || (long)H * W > 0x3fffffffL) return msc_fail(MSC_ERR_ARG, name ": bad dims B=%d H=%d W=%d", B, H, W)

extern "C" int msc_resize_bilinear(const float* in, void* out, int out_f64, float* minmax_ws, int B, int C, int h, int w, int H, int W, void* stream) {
    if (!in || !out || !minmax_ws || B <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return msc_fail(MSC_ERR_ARG, "msc_resize_bilinear: bad argument");
    hipLaunchKernelGGL(image_minmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, in, minmax_ws, (long)C * h * w);
    if (out_f64)
        hipLaunchKernelGGL(resize_bilinear_kernel<double>, dim3(flat_grid((long)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, in, (double*)out, minmax_ws, C, B * C, h, w, H, W);
    else
        hipLaunchKernelGGL(resize_bilinear_kernel<float>, dim3(flat_grid((long)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, in, (float*)out, minmax_ws, C, B * C, h, w, H, W);
    return msc_check_launch("msc_resize_bilinear");
}

extern "C" int msc_resize_threshold(const float* in, float* out, uint8_t* layers, float* minmax_ws, int B, int C, int h, int w, int H, int W,
                                    const int32_t* layer_class, const double* layer_thr, int L, void* stream) {
    if (!in || !out || !layers || !minmax_ws || !layer_class || !layer_thr || B <= 0 || C <= 0 || L <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
        return msc_fail(MSC_ERR_ARG, "msc_resize_threshold: bad argument");
    hipLaunchKernelGGL(image_minmax_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, in, minmax_ws, (long)C * h * w);
    hipLaunchKernelGGL(resize_threshold_kernel, dim3(flat_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, layers, minmax_ws, B, C, h, w, H, W, layer_class, layer_thr, L);
    return msc_check_launch("msc_resize_threshold");
}

extern "C" int msc_crop_center(const float* in, float* out, int B, int C, int h, int w, int hc, int wc, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || hc <= 0 || wc <= 0 || hc > h || wc > w) return msc_fail(MSC_ERR_ARG, "msc_crop_center: bad argument");
    const int hs = (int)((h - hc) / 2.), ws = (int)((w - wc) / 2.);
    // the reference slices [hs:-hs]: it is only well defined when the margins are symmetric and > 0
    if (hs <= 0 || ws <= 0 || h - 2 * hs != hc || w - 2 * ws != wc)
        return msc_fail(MSC_ERR_UNSUPPORTED, "msc_crop_center: reference slicing [s:-s] needs symmetric non-zero margins (h=%d hc=%d w=%d wc=%d)", h, hc, w, wc);
    hipLaunchKernelGGL(crop_center_kernel, dim3(flat_grid((long)B * C * hc * wc)), dim3(256), 0, (hipStream_t)stream, in, out, B * C, h, w, hc, wc, hs, ws);
    return msc_check_launch("msc_crop_center");
}

extern "C" int msc_threshold_layers(const void* probs, int probs_f64, uint8_t* layers, int B, int C, int H, int W,
                                    const int32_t* layer_class, const double* layer_thr, int L, void* stream) {
    POST_DIMS("msc_threshold_layers");
    if (!probs || !layers || !layer_class || !layer_thr || L <= 0 || C <= 0) return msc_fail(MSC_ERR_ARG, "msc_threshold_layers: bad argument");
    if (probs_f64)
        hipLaunchKernelGGL(threshold_layers_kernel<double>, dim3(flat_grid((long)B * L * H * W)), dim3(256), 0, (hipStream_t)stream, (const double*)probs, layers, B, C, (long)H * W, layer_class, layer_thr, L);
    else
        hipLaunchKernelGGL(threshold_layers_kernel<float>, dim3(flat_grid((long)B * L * H * W)), dim3(256), 0, (hipStream_t)stream, (const float*)probs, layers, B, C, (long)H * W, layer_class, layer_thr, L);
    return msc_check_launch("msc_threshold_layers");
}

extern "C" int msc_argmax_channels(const float* probs, int32_t* out, int B, int C, int H, int W, void* stream) {
    POST_DIMS("msc_argmax_channels");
    if (!probs || !out || C <= 0) return msc_fail(MSC_ERR_ARG, "msc_argmax_channels: bad argument");
    hipLaunchKernelGGL(argmax_channels_kernel, dim3(flat_grid((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, probs, out, B, C, (long)H * W);
    return msc_check_launch("msc_argmax_channels");
}

static int window(int k, int H, int W, const char* name, int* lo, int* hi) {
    if (k <= 0) return msc_fail(MSC_ERR_ARG, "%s: selem size must be > 0 (the reference returns the input unchanged otherwise)", name);
    *lo = -((k - 1) / 2);
    *hi = *lo + k - 1;
    if (*hi >= H || *hi >= W) return msc_fail(MSC_ERR_UNSUPPORTED, "%s: selem %d larger than the image", name, k);
    return MSC_OK;
}

extern "C" int msc_erode_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int k, void* stream) {
    POST_DIMS("msc_erode_u8");
    if (!in || !out || in == out) return msc_fail(MSC_ERR_ARG, "msc_erode_u8: bad pointers");
    int lo, hi, rc = window(k, H, W, "msc_erode_u8", &lo, &hi);
    if (rc) return rc;
    launch_rect_filter<uint8_t, false>(in, out, B, H, W, lo, hi, (hipStream_t)stream);
    return msc_check_launch("msc_erode_u8");
}

extern "C" int msc_dilate_i32(const int32_t* in, int32_t* out, int B, int H, int W, int k, void* stream) {
    POST_DIMS("msc_dilate_i32");
    if (!in || !out || in == out) return msc_fail(MSC_ERR_ARG, "msc_dilate_i32: bad pointers");
    int lo, hi, rc = window(k, H, W, "msc_dilate_i32", &lo, &hi);
    if (rc) return rc;
    launch_rect_filter<int32_t, true>(in, out, B, H, W, lo, hi, (hipStream_t)stream);
    return msc_check_launch("msc_dilate_i32");
}

extern "C" int msc_rect_filter_u8(const uint8_t* in, uint8_t* out, int B, int H, int W, int lo, int hi, int is_max, void* stream) {
    POST_DIMS("msc_rect_filter_u8");
    if (!in || !out || in == out || lo > 0 || hi < 0 || hi >= H || hi >= W || -lo >= H || -lo >= W)
        return msc_fail(MSC_ERR_ARG, "msc_rect_filter_u8: bad argument");
    if (is_max) launch_rect_filter<uint8_t, true>(in, out, B, H, W, lo, hi, (hipStream_t)stream);
    else launch_rect_filter<uint8_t, false>(in, out, B, H, W, lo, hi, (hipStream_t)stream);
    return msc_check_launch("msc_rect_filter_u8");
}

extern "C" int64_t msc_label_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * H * W * 4;
}

extern "C" int msc_label4(const uint8_t* mask, int32_t* labels, int32_t* counts, void* workspace, int B, int H, int W, void* stream) {
    POST_DIMS("msc_label4");
    if (!mask || !labels || !workspace) return msc_fail(MSC_ERR_ARG, "msc_label4: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const dim3 g = plane_grid(HW, B);
    int32_t* rank = (int32_t*)workspace;
    static const bool strip_off = [] { const char* e = getenv("MSC_CCL_STRIP"); return e && e[0] == '0'; }();      // A/B: the global-memory unions of rounds 1-5
    const int SR = W <= CCL_STRIP_PIX ? CCL_STRIP_PIX / W : 0;      // rows per strip
    if (SR >= 2 && HW < (1L << 31) && !strip_off) {
        const int strips = ceil_div(H, SR);
        hipLaunchKernelGGL(ccl_strip_kernel, dim3(strips, B), dim3(1024), 0, st, mask, labels, H, W, SR);
        if (strips > 1) hipLaunchKernelGGL(ccl_seam_kernel, dim3(strips - 1, B), dim3(256), 0, st, labels, H, W, SR);
    } else {
        hipLaunchKernelGGL(ccl_init_kernel, dim3(ceil_div((long)B * H, 4)), dim3(256), 0, st, mask, labels, H, W, (long)B * H);
        hipLaunchKernelGGL(ccl_merge_kernel, g, dim3(256), 0, st, mask, labels, H, W);
    }
    hipLaunchKernelGGL(ccl_compress_kernel, g, dim3(256), 0, st, labels, HW);
    hipLaunchKernelGGL(ccl_rank_kernel, dim3(B), dim3(1024), 0, st, labels, rank, counts, HW);
    hipLaunchKernelGGL(ccl_relabel_kernel, g, dim3(256), 0, st, labels, rank, HW);
    return msc_check_launch("msc_label4");
}

extern "C" int64_t msc_watershed_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    return (int64_t)B * H * W * 9 + (int64_t)B * 4 + 16;          // int32 arrival times + int32 work list + 8-bit relief + work-list lengths
}

extern "C" int msc_watershed(const float* prob, const uint8_t* mask, int32_t* labels, void* workspace, int B, int H, int W, void* stream) {
    POST_DIMS("msc_watershed");
    if (!prob || !mask || !labels || !workspace) return msc_fail(MSC_ERR_ARG, "msc_watershed: null pointer");
    if (((uintptr_t)workspace) & 3) return msc_fail(MSC_ERR_ARG, "msc_watershed: workspace must be 4-byte aligned");
    const long n = (long)B * H * W;
    int32_t* tarr = (int32_t*)workspace;
    int32_t* list = tarr + n;
    int32_t* nlist = list + n;
    uint8_t* hq = (uint8_t*)(nlist + B + ((4 - (B & 3)) & 3));
    hipStream_t st = (hipStream_t)stream;
    static const int passes = [] { const char* e = getenv("MSC_WS_PASSES"); return e ? atoi(e) : 3; }();      // 0: the single-workgroup kernels alone (A/B)
    const dim3 gt(ceil_div(W, WS_TILE), ceil_div(H, WS_TILE), B);
    hipLaunchKernelGGL(watershed_init_kernel, dim3(B), dim3(1024), 0, st, prob, mask, labels, hq, tarr, list, nlist, H, W);
    for (int p = 0; p < passes; ++p) hipLaunchKernelGGL(watershed_tile_kernel<false>, gt, dim3(256), 0, st, mask, hq, tarr, labels, H, W);
    hipLaunchKernelGGL(watershed_finish_kernel<false>, dim3(B), dim3(1024), 0, st, hq, tarr, labels, list, nlist, H, W);
    for (int p = 0; p < passes; ++p) hipLaunchKernelGGL(watershed_tile_kernel<true>, gt, dim3(256), 0, st, mask, hq, tarr, labels, H, W);
    hipLaunchKernelGGL(watershed_finish_kernel<true>, dim3(B), dim3(1024), 0, st, hq, tarr, labels, list, nlist, H, W);
    return msc_check_launch("msc_watershed");
}

extern "C" int msc_add_dropped(const uint8_t* processed, const int32_t* labels_orig, uint8_t* out, void* workspace,
                               int B, int H, int W, int bool_sum, void* stream) {
    POST_DIMS("msc_add_dropped");
    if (!processed || !labels_orig || !out || !workspace) return msc_fail(MSC_ERR_ARG, "msc_add_dropped: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    if (msc_memset_zero(workspace, (int64_t)B * HW * 4, st) != MSC_OK) return MSC_ERR_HIP;      // a kernel, not a memset node (elementwise.hip, round 5)
    const dim3 g = plane_grid(HW, B);
    hipLaunchKernelGGL(dropped_mark_kernel, g, dim3(256), 0, st, processed, labels_orig, (int32_t*)workspace, HW);
    hipLaunchKernelGGL(dropped_apply_kernel, g, dim3(256), 0, st, processed, labels_orig, (const int32_t*)workspace, out, HW, bool_sum);
    return msc_check_launch("msc_add_dropped");
}

extern "C" int msc_build_score(const int32_t* labels, const float* probs, double* sums, int32_t* areas, double* score,
                               int B, int H, int W, int max_labels, void* stream) {
    POST_DIMS("msc_build_score");
    if (!labels || !probs || !sums || !areas || !score || max_labels <= 0) return msc_fail(MSC_ERR_ARG, "msc_build_score: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const long n = (long)B * max_labels;
    if (msc_memset_zero(sums, (int64_t)(n * sizeof(double)), st) != MSC_OK || msc_memset_zero(areas, (int64_t)(n * sizeof(int32_t)), st) != MSC_OK)
        return msc_fail(MSC_ERR_HIP, "msc_build_score: memset failed");
    static const bool lds_off = [] { const char* e = getenv("MSC_SCORE_LDS"); return e && e[0] == '0'; }();      // A/B: the per-wave global atomics of rounds 1-5
    if (max_labels <= SCORE_LDS_MAX && !lds_off)
        hipLaunchKernelGGL(score_accum_lds_kernel, dim3((unsigned)ceil_div(HW, SCORE_CHUNK), B), dim3(256), 0, st, labels, probs, sums, areas, HW, max_labels);
    else
        hipLaunchKernelGGL(score_accum_kernel, plane_grid(HW, B), dim3(256), 0, st, labels, probs, sums, areas, HW, max_labels);
    hipLaunchKernelGGL(score_final_kernel, dim3(flat_grid(n)), dim3(256), 0, st, sums, areas, score, n);
    return msc_check_launch("msc_build_score");
}

extern "C" int64_t msc_crf_workspace_bytes(int B, int H, int W, int radius) {
    (void)radius;
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)B * H * W * 4 * (2 + 2 + 2 + 2);  // two norms + two ping-pong Q buffers of 2 channels + the Gaussian message of an iteration (round 6)
}

extern "C" int msc_dense_crf(const float* probs, const uint8_t* rgb, float* out, void* workspace, int B, int H, int W,
                             float sxy_g, float compat_g, float sxy_b, float srgb, float compat_b, int iterations, void* stream) {
    POST_DIMS("msc_dense_crf");
    if (!probs || !rgb || !out || !workspace || sxy_g <= 0 || sxy_b <= 0 || srgb <= 0 || iterations < 0)
        return msc_fail(MSC_ERR_ARG, "msc_dense_crf: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    CrfP c;
    c.H = H; c.W = W;
    c.rg = (int)ceilf(5.f * sxy_g); c.rb = (int)ceilf(5.f * sxy_b);
    c.inv2g = 0.5f / (sxy_g * sxy_g); c.inv2b = 0.5f / (sxy_b * sxy_b); c.inv2rgb = 0.5f / (srgb * srgb);
    c.compat_g = compat_g; c.compat_b = compat_b;
    float* ng = (float*)workspace;
    float* nb = ng + (long)B * HW;
    float* qa = nb + (long)B * HW;
    float* qb = qa + 2L * B * HW;
    float2* gmsg = (float2*)(qb + 2L * B * HW);
    const dim3 g = plane_grid(HW, B);
    const int r = c.rg > c.rb ? c.rg : c.rb;
    static const bool naive = [] { const char* e = getenv("MSC_CRF_NAIVE"); return e && e[0] == '1'; }();      // A/B: the global-memory kernels
    const bool tiled = r <= CRF_RMAX && !naive;
    const dim3 gt(ceil_div(W, CRF_T), ceil_div(H, CRF_T), B);
    static const bool r5_off = [] { const char* e = getenv("MSC_CRF_R5"); return e && e[0] == '0'; }();                // A/B: the first tiled form
    const bool fast = tiled && r <= 5 && !r5_off;
    CrfCols<5> cols;
    for (int i = 0; i <= 10; ++i) {
        const int dx = i - 5, ad = dx < 0 ? -dx : dx;
        cols.gx[i] = ad <= c.rg ? exp2f(-(float)(dx * dx) * c.inv2g * 1.44269504f) : 0.f;
        cols.bx[i] = ad <= c.rb ? -(float)(dx * dx) * c.inv2b * 1.44269504f : -1e30f;
    }
    static const bool pk_off = [] { const char* e = getenv("MSC_CRF_PK"); return e && e[0] == '0'; }();                // A/B: the unpacked round-4 inner loop
    static const bool x_off = [] { const char* e = getenv("MSC_CRF_X"); return e && e[0] == '0'; }();                  // A/B: round 5's kernel (Gaussian + bilateral in one 121-tap loop)
    const bool xform = fast && !pk_off && !x_off;
    if (xform) hipLaunchKernelGGL((crf_x_kernel<5, true>), gt, dim3(256), 0, st, (const float*)nullptr, rgb, ng, nb, (const float*)nullptr, (const float2*)nullptr, (float*)nullptr, c, cols);
    else if (fast && !pk_off) hipLaunchKernelGGL((crf_r_kernel<5, true, true>), gt, dim3(256), 0, st, (const float*)nullptr, rgb, ng, nb, (const float*)nullptr, (float*)nullptr, c, cols);
    else if (fast) hipLaunchKernelGGL((crf_r_kernel<5, true>), gt, dim3(256), 0, st, (const float*)nullptr, rgb, ng, nb, (const float*)nullptr, (float*)nullptr, c, cols);
    else if (tiled) hipLaunchKernelGGL(crf_norm_tiled_kernel, gt, dim3(256), 0, st, rgb, ng, nb, c, r);
    else hipLaunchKernelGGL(crf_norm_kernel, g, dim3(256), 0, st, rgb, ng, nb, c);
    hipLaunchKernelGGL(crf_init_kernel, g, dim3(256), 0, st, probs, iterations == 0 ? out : qa, HW);
    float* cur = qa;
    for (int it = 0; it < iterations; ++it) {
        float* dst = (it == iterations - 1) ? out : (cur == qa ? qb : qa);
        static const bool fuseg_off = [] { const char* e = getenv("MSC_CRF_FUSEG"); return e && e[0] == '0'; }();            // A/B: the Gaussian message as its own launch
        if (xform && !fuseg_off) {
            hipLaunchKernelGGL((crf_x_kernel<5, false, true>), gt, dim3(256), 0, st, probs, rgb, ng, nb, (const float*)cur, (const float2*)nullptr, dst, c, cols);
        } else if (xform) {
            hipLaunchKernelGGL((crf_gauss_kernel<5>), gt, dim3(256), 0, st, (const float*)cur, (const float*)ng, gmsg, c, cols);
            hipLaunchKernelGGL((crf_x_kernel<5, false>), gt, dim3(256), 0, st, probs, rgb, ng, nb, (const float*)cur, (const float2*)gmsg, dst, c, cols);
        } else if (fast && !pk_off) hipLaunchKernelGGL((crf_r_kernel<5, false, true>), gt, dim3(256), 0, st, probs, rgb, ng, nb, (const float*)cur, dst, c, cols);
        else if (fast) hipLaunchKernelGGL((crf_r_kernel<5, false>), gt, dim3(256), 0, st, probs, rgb, ng, nb, (const float*)cur, dst, c, cols);
        else if (tiled) hipLaunchKernelGGL(crf_iter_tiled_kernel, gt, dim3(256), 0, st, probs, rgb, ng, nb, cur, dst, c, r);
        else hipLaunchKernelGGL(crf_iter_kernel, g, dim3(256), 0, st, probs, rgb, ng, nb, cur, dst, c);
        cur = dst;
    }
    return msc_check_launch("msc_dense_crf");
}

static int tta_check(const char* name, const void* a, const void* b, const int32_t* specs, int N, int C, int H, int W, int V, const int32_t* host_specs) {
    if (!a || !b || !specs || N <= 0 || C <= 0 || H <= 0 || W <= 0 || V <= 0) return msc_fail(MSC_ERR_ARG, "%s: bad argument", name);
    (void)host_specs;
    return MSC_OK;
}

extern "C" int msc_tta_transform(const float* x, float* out, const int32_t* specs, int N, int C, int H, int W, int V, int any_quarter_turn, void* stream) {
    int rc = tta_check("msc_tta_transform", x, out, specs, N, C, H, W, V, nullptr);
    if (rc) return rc;
    if (any_quarter_turn && H != W) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_tta_transform: 90/270 degree rotations need square images");
    hipLaunchKernelGGL(tta_transform_kernel, dim3(flat_grid((long)V * N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, out, (long)N * C, H, W, specs, V);
    return msc_check_launch("msc_tta_transform");
}

extern "C" int msc_tta_aggregate(const float* preds, float* out, const int32_t* specs, int N, int C, int H, int W, int V, int method, int any_quarter_turn, void* stream) {
    int rc = tta_check("msc_tta_aggregate", preds, out, specs, N, C, H, W, V, nullptr);
    if (rc) return rc;
    if (method < 0 || method > 3) return msc_fail(MSC_ERR_ARG, "msc_tta_aggregate: method %d (0 mean, 1 gmean, 2 max, 3 min)", method);
    if (any_quarter_turn && H != W) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_tta_aggregate: 90/270 degree rotations need square images");
    hipLaunchKernelGGL(tta_aggregate_kernel, dim3(flat_grid((long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, preds, out, (long)N * C, H, W, specs, V, method);
    return msc_check_launch("msc_tta_aggregate");
}
