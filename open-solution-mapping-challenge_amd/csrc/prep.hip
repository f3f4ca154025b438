// Target preparation on gfx950: the body of overlay_mask_one_image (src/preparation.py:44-84), given the decoded
// instance masks of one image (plain overlay; msc_prep_morph adds the eroded / eroded+dilated variants, :61-77):
//   * instances without a pixel in the interior [2:-2, 2:-2] are skipped             (is_on_border, :197-198, :111)
//   * mask_overlayed = category number of the last category covering the pixel       (:65-68, :118)
//   * per instance the exact Euclidean distance to its nearest pixel (scipy distance_transform_edt(1 - mask),
//     :146-151), per pixel the two smallest over instances: distances = d1 + d2 as float16, second nearest as
//     float64 (clean_distances, :154-163; one instance -> d2 = d1; none -> 0)
//   * get_size_matrix (:181-187): component area per pixel, background 1
//   * optional border class from the second-nearest distance (:73-76)
// The reference stacks one full-image EDT per building (np.dstack) and sorts the stack; here a column sweep gives
// every instance its vertical distance g, and each pixel takes min over x' of (x-x')^2 + g^2 inside the instance's
// column range -- integers throughout, so sqrt() of the result is bit-identical to scipy's.
#include "common.h"
#include "msc_internal.h"

namespace {

constexpr unsigned short G_INF = 0xffff;

// area and "has an interior pixel" per instance
__global__ __launch_bounds__(256) void inst_stats_kernel(const uint8_t* __restrict__ masks, int H, int W, int border, int* __restrict__ area,
                                                         int* __restrict__ interior) {
    const int i = blockIdx.y;
    const uint8_t* m = masks + (long)i * H * W;
    int a = 0, in = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < H * W; p += gridDim.x * blockDim.x) {
        if (m[p]) {
            ++a;
            const int y = p / W, x = p - y * W;
            if (y >= border && y < H - border && x >= border && x < W - border) in = 1;
        }
    }
    a = (int)wave_sum((float)a);      // exact: a block strides over < 2^24 pixels
    in = __any(in);
    if ((threadIdx.x & 63) == 0) {
        if (a && area) atomicAdd(area + i, a);
        if (in && interior) atomicOr(interior + i, 1);
    }
}

// keep[i]: interior pixel present, and not one of the image-covering instances that update_distances() discards
// because the distances accumulated so far sum to zero (src/preparation.py:147)
__global__ void keep_kernel(const int* __restrict__ area, const int* __restrict__ interior, int* __restrict__ keep, int* __restrict__ xlo,
                            int* __restrict__ xhi, int n, int hw, int W) {
    if (blockIdx.x || threadIdx.x) return;
    bool seen_partial = false;
    for (int i = 0; i < n; ++i) {
        int k = interior[i] ? 1 : 0;
        if (k && area[i] == hw && !seen_partial) k = 2;          // overlayed, but contributes no distance layer
        if (k == 1) seen_partial = true;
        keep[i] = k;
        xlo[i] = W; xhi[i] = -1;
    }
}

// one thread per (instance, column): vertical distance to the nearest instance pixel of that column
__global__ void col_dist_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ keep, unsigned short* __restrict__ g,
                                int* __restrict__ xlo, int* __restrict__ xhi, int n, int H, int W) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * W) return;
    const int i = t / W, x = t - i * W;
    if (!keep[i]) return;
    const uint8_t* m = masks + (long)i * H * W + x;
    unsigned short* gi = g + (long)i * H * W + x;
    unsigned d = G_INF;
    bool any = false;
    for (int y = 0; y < H; ++y) {
        if (m[(long)y * W]) { d = 0; any = true; }
        else if (d != G_INF) ++d;
        gi[(long)y * W] = (unsigned short)d;
    }
    if (!any) return;
    d = G_INF;
    for (int y = H - 1; y >= 0; --y) {
        if (m[(long)y * W]) d = 0;
        else if (d != G_INF) ++d;
        if (d < gi[(long)y * W]) gi[(long)y * W] = (unsigned short)d;
    }
    atomicMin(xlo + i, x);
    atomicMax(xhi + i, x);
}

__device__ __forceinline__ unsigned short f64_to_f16(double v) {       // round to nearest even, v >= 0 and finite
    if (v == 0.0) return 0;
    int e;
    const double f = frexp(v, &e);            // v = f * 2^e, f in [0.5, 1)
    int eb = e - 1 + 15;                      // biased exponent of 1.xxx * 2^(e-1)
    if (eb >= 31) return 0x7c00;
    if (eb <= 0) {                            // subnormal half: units of 2^-24
        const double q = rint(ldexp(v, 24));  // rint = nearest even
        return (unsigned short)q;
    }
    const double q = rint(ldexp(f, 11));      // 11 significant bits: [1024, 2048]
    unsigned mant = (unsigned)q;
    if (mant == 2048u) { mant = 1024u; ++eb; if (eb >= 31) return 0x7c00; }
    return (unsigned short)((eb << 10) | (mant - 1024u));
}

// one thread per pixel: squared distance to every kept instance, two smallest; category overlay
__global__ __launch_bounds__(256) void two_nearest_kernel(const unsigned short* __restrict__ g, const int* __restrict__ keep,
                                                          const int* __restrict__ xlo, const int* __restrict__ xhi,
                                                          const int* __restrict__ category, uint8_t* __restrict__ overlay,
                                                          unsigned short* __restrict__ dist16, double* __restrict__ second, int n, int H, int W) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    long b1 = -1, b2 = -1;        // smallest and second smallest squared distance (-1: none yet)
    int cat = 0, layers = 0;
    for (int i = 0; i < n; ++i) {
        const int k = keep[i];
        if (!k) continue;
        const unsigned short* row = g + ((long)i * H + y) * W;
        if (row[x] == 0) { const int c = category ? category[i] : 1; cat = c > cat ? c : cat; }
        if (k != 1) continue;
        long best = -1;
        for (int xx = xlo[i]; xx <= xhi[i]; ++xx) {
            const unsigned gg = row[xx];
            if (gg == G_INF) continue;
            const long dx = x - xx;
            const long d = dx * dx + (long)gg * gg;
            if (best < 0 || d < best) best = d;
        }
        // an instance eroded to nothing: scipy's distance_transform_edt of an array without background measures to a
        // virtual pixel at (y, x) = (-1, 0), and the reference stacks that layer like any other (src/preparation.py:150)
        if (best < 0) best = (long)(y + 1) * (y + 1) + (long)x * x;
        ++layers;
        if (b1 < 0 || best < b1) { b2 = b1; b1 = best; }
        else if (b2 < 0 || best < b2) b2 = best;
    }
    double d1 = 0.0, d2 = 0.0;
    if (layers >= 1) { d1 = sqrt((double)b1); d2 = layers >= 2 ? sqrt((double)b2) : d1; }
    overlay[p] = (uint8_t)cat;
    dist16[p] = f64_to_f16(d1 + d2);
    if (second) second[p] = d2;
}

// get_simple_eroded_mask / get_simple_eroded_dilated_mask (src/preparation.py:166-178): instances larger than
// small^2 pixels take their eroded mask, the others stay as they are (other == NULL) or take the dilated one
__global__ void select_kernel(const uint8_t* __restrict__ orig, const uint8_t* __restrict__ eroded, const uint8_t* __restrict__ other,
                              const int* __restrict__ area, int thr, uint8_t* __restrict__ out, long hw, long total) {
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int i = (int)(p / hw);
        out[p] = area[i] > thr ? eroded[p] : (other ? other[p] : (uint8_t)(orig[p] != 0));
    }
}

__global__ void paint_kernel(uint8_t* __restrict__ overlay, const uint8_t* __restrict__ mask, int value, long hw) {
    const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (p < hw && mask[p]) overlay[p] = (uint8_t)value;
}

__global__ void border_kernel(uint8_t* __restrict__ overlay, const double* __restrict__ second, double width, int id, int hw) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    // (second < width) & (~mask): bitwise NOT of a uint8 keeps bit 0 only where the mask value is even
    if (second[p] < width && !(overlay[p] & 1)) overlay[p] = (uint8_t)id;
}

__global__ void max_u8_kernel(const uint8_t* __restrict__ v, int hw, int* __restrict__ out) {
    int m = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += gridDim.x * blockDim.x) m = v[p] > m ? v[p] : m;
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__global__ void area_count_kernel(const int32_t* __restrict__ labels, int* __restrict__ areas, long total, int hw, int max_labels) {
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int l = labels[p];
        if (l > 0 && l <= max_labels) atomicAdd(areas + (p / hw) * (long)(max_labels + 1) + l, 1);
    }
}

__global__ void area_gather_kernel(const int32_t* __restrict__ labels, const int* __restrict__ areas, int32_t* __restrict__ sizes, long total,
                                   int hw, int max_labels) {
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int l = labels[p];
        sizes[p] = (l > 0 && l <= max_labels) ? areas[(p / hw) * (long)(max_labels + 1) + l] : 1;
    }
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int64_t msc_prep_workspace_bytes(int n, int H, int W) {
    if (n < 0 || H <= 0 || W <= 0 || H >= 65535 || W >= 65535) return -1;
    const size_t nn = n > 0 ? n : 1;
    return (int64_t)(up256(nn * (size_t)H * W * 2) + 5 * up256(nn * 4) + 256);
}

extern "C" int msc_prep_targets(const uint8_t* masks, const uint8_t* border_masks, const int32_t* category_nr, int n, int H, int W,
                                uint8_t* mask_overlayed, uint16_t* distances_f16, double* second_nearest, int32_t* kept, void* workspace,
                                void* stream) {
    if ((n > 0 && !masks) || !mask_overlayed || !distances_f16 || !workspace || n < 0 || H <= 0 || W <= 0 || H >= 65535 || W >= 65535)
        return msc_fail(MSC_ERR_ARG, "msc_prep_targets: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t nn = n > 0 ? n : 1;
    char* w = (char*)workspace;
    unsigned short* g = (unsigned short*)w;      w += up256(nn * (size_t)H * W * 2);
    int* area = (int*)w;                         w += up256(nn * 4);
    int* interior = (int*)w;                     w += up256(nn * 4);
    int* keep = (int*)w;                         w += up256(nn * 4);
    int* xlo = (int*)w;                          w += up256(nn * 4);
    int* xhi = (int*)w;
    const int hw = H * W;
    if (n > 0) {
        if (msc_memset_zero(area, (int64_t)(2 * up256(nn * 4)), st) != MSC_OK) return MSC_ERR_HIP;
        int bx = ceil_div(hw, 256 * 8);
        if (border_masks && border_masks != masks) {     // is_on_border() looks at the annotation, the rest at its eroded / dilated form
            hipLaunchKernelGGL(inst_stats_kernel, dim3(bx < 1 ? 1 : bx, n), dim3(256), 0, st, border_masks, H, W, 2, (int*)nullptr, interior);
            hipLaunchKernelGGL(inst_stats_kernel, dim3(bx < 1 ? 1 : bx, n), dim3(256), 0, st, masks, H, W, 2, area, (int*)nullptr);
        } else {
            hipLaunchKernelGGL(inst_stats_kernel, dim3(bx < 1 ? 1 : bx, n), dim3(256), 0, st, masks, H, W, 2, area, interior);
        }
        hipLaunchKernelGGL(keep_kernel, dim3(1), dim3(1), 0, st, area, interior, keep, xlo, xhi, n, hw, W);
        hipLaunchKernelGGL(col_dist_kernel, dim3(ceil_div((long)n * W, 256)), dim3(256), 0, st, masks, keep, g, xlo, xhi, n, H, W);
    }
    hipLaunchKernelGGL(two_nearest_kernel, dim3(ceil_div(hw, 256)), dim3(256), 0, st, g, keep, xlo, xhi, category_nr, mask_overlayed,
                       distances_f16, second_nearest, n, H, W);
    if (kept && n > 0 && msc_copy(kept, keep, (int64_t)n * 4, st) != MSC_OK)
        return msc_fail(MSC_ERR_HIP, "msc_prep_targets: copy");
    return msc_check_launch("msc_prep_targets");
}

extern "C" int64_t msc_prep_morph_workspace_bytes(int n, int H, int W) {
    if (n <= 0 || H <= 0 || W <= 0) return -1;
    return (int64_t)(2 * up256((size_t)n * H * W) + up256((size_t)n * 4));
}

extern "C" int msc_prep_morph(const uint8_t* masks, int n, int H, int W, int erode, int dilate, int small_annotations_size, uint8_t* chosen,
                              void* workspace, void* stream) {
    if (!masks || !chosen || !workspace || n <= 0 || H <= 0 || W <= 0 || erode <= 0 || dilate < 0 || chosen == masks)
        return msc_fail(MSC_ERR_ARG, "msc_prep_morph: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W, total = (long)n * hw;
    char* w = (char*)workspace;
    uint8_t* eroded = (uint8_t*)w;       w += up256((size_t)total);
    uint8_t* dilated = (uint8_t*)w;      w += up256((size_t)total);
    int* area = (int*)w;
    if (msc_memset_zero(area, (int64_t)n * 4, st) != MSC_OK) return MSC_ERR_HIP;
    int bx = ceil_div(hw, 256 * 8);
    hipLaunchKernelGGL(inst_stats_kernel, dim3(bx < 1 ? 1 : bx, n), dim3(256), 0, st, masks, H, W, 2, area, (int*)nullptr);
    // skimage binary_erosion / binary_dilation with rectangle(k, k) = scipy's: windows -(k/2).. resp. -((k-1)/2)..
    int rc = msc_rect_filter_u8(masks, eroded, n, H, W, -(erode / 2), -(erode / 2) + erode - 1, 0, stream);
    if (rc) return rc;
    if (dilate > 0) {
        rc = msc_rect_filter_u8(masks, dilated, n, H, W, -((dilate - 1) / 2), -((dilate - 1) / 2) + dilate - 1, 1, stream);
        if (rc) return rc;
    }
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(select_kernel, dim3((int)blocks), dim3(256), 0, st, masks, eroded, dilate > 0 ? dilated : (const uint8_t*)nullptr, area,
                       small_annotations_size * small_annotations_size, chosen, hw, total);
    return msc_check_launch("msc_prep_morph");
}

extern "C" int msc_prep_paint(uint8_t* mask_overlayed, const uint8_t* mask, int value, int H, int W, void* stream) {
    if (!mask_overlayed || !mask || H <= 0 || W <= 0 || value < 0 || value > 255) return msc_fail(MSC_ERR_ARG, "msc_prep_paint: bad argument");
    hipLaunchKernelGGL(paint_kernel, dim3(ceil_div((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, mask_overlayed, mask, value, (long)H * W);
    return msc_check_launch("msc_prep_paint");
}

extern "C" int msc_prep_border(uint8_t* mask_overlayed, const double* second_nearest, double border_width, int32_t* scratch, int H, int W,
                               void* stream) {
    if (!mask_overlayed || !second_nearest || !scratch || H <= 0 || W <= 0) return msc_fail(MSC_ERR_ARG, "msc_prep_border: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int hw = H * W;
    int mx = 0;
    if (msc_memset_zero(scratch, 4, st) != MSC_OK) return MSC_ERR_HIP;
    hipLaunchKernelGGL(max_u8_kernel, dim3(ceil_div(hw, 256 * 16) < 1 ? 1 : ceil_div(hw, 256 * 16)), dim3(256), 0, st, mask_overlayed, hw, scratch);
    if (hipMemcpyAsync(&mx, scratch, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return msc_fail(MSC_ERR_HIP, "msc_prep_border: copy");
    hipLaunchKernelGGL(border_kernel, dim3(ceil_div(hw, 256)), dim3(256), 0, st, mask_overlayed, second_nearest, border_width, mx + 1, hw);
    return msc_check_launch("msc_prep_border");
}

extern "C" int msc_size_matrix(const int32_t* labels, int32_t* sizes, int32_t* areas, int B, int H, int W, int max_labels, void* stream) {
    if (!labels || !sizes || !areas || B <= 0 || H <= 0 || W <= 0 || max_labels < 0) return msc_fail(MSC_ERR_ARG, "msc_size_matrix: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)B * H * W;
    if (msc_memset_zero(areas, (int64_t)B * (max_labels + 1) * 4, st) != MSC_OK) return MSC_ERR_HIP;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(area_count_kernel, dim3((int)blocks), dim3(256), 0, st, labels, areas, total, H * W, max_labels);
    hipLaunchKernelGGL(area_gather_kernel, dim3((int)blocks), dim3(256), 0, st, labels, areas, sizes, total, H * W, max_labels);
    return msc_check_launch("msc_size_matrix");
}
