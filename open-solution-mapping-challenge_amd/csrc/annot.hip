// Annotation encoding on gfx950: labelled instance images -> COCO run-length strings + bounding boxes.
// Replaces, for every instance of every layer at once, the reference's per-instance host loop
//   decompose (full-image copy per label, src/utils.py:61-73) -> cocomask.encode -> toBbox (src/utils.py:106-127),
// whose algorithm is pycocotools 2.0.0 (environment.yml:27) common/maskApi.c: rleEncode, rleToString, rleToBbox.
//
// Formulation (HBM-bound scans, no per-instance pass): in column-major pixel order j = x*H + y a maximal run of
// one non-zero label is a "segment"; the 1-runs of instance i are exactly the segments with label i and its 0-runs
// are the gaps between consecutive ones.  So:
//   1. transpose the int32 label images to column-major;
//   2. flag segment starts, inclusive scan -> segment index; scatter (start, end, key = layer:label);
//   3. stable radix sort of the segments by key -> grouped per instance, still in pixel order;
//   4. per segment, from its sorted neighbours: gap, length, trailing 0-run, their delta codes (x[i] -= x[i-2] for
//      i > 2) and character counts (5 bits per char, continuation bit 0x20, +48); scan -> string offsets;
//   5. emit characters, instance table (layer, label, string range, bounding box by atomic min/max).
// Scan and sort are rocPRIM device primitives (rocprim::inclusive_scan / exclusive_scan / radix_sort_pairs); the rest are the kernels below.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "common.h"
#include "msc_internal.h"

namespace {

constexpr int TABLE_W = 8;      // int32 per instance: layer, label, str_begin, str_end, xs, ys, xe, ye

__global__ __launch_bounds__(256) void transpose_cm_kernel(const int* __restrict__ in, int* __restrict__ out, int H, int W) {
    __shared__ int tile[32][33];
    const long img = (long)blockIdx.z * H * W;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int y = y0 + r, x = x0 + threadIdx.x;
        if (y < H && x < W) tile[r][threadIdx.x] = in[img + (long)y * W + x];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int x = x0 + r, y = y0 + threadIdx.x;
        if (y < H && x < W) out[img + (long)x * H + y] = tile[threadIdx.x][r];
    }
}

__global__ void seg_flags_kernel(const int* __restrict__ cm, unsigned* __restrict__ flag, long total, int a) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int j = (int)(g % a);
        const int v = cm[g];
        const int prev = j ? cm[g - 1] : 0;
        flag[g] = (v != 0 && v != prev) ? 1u : 0u;
    }
}

__global__ void seg_scatter_kernel(const int* __restrict__ cm, const unsigned* __restrict__ incl, int* __restrict__ S, int* __restrict__ E,
                                   unsigned long long* __restrict__ key, unsigned* __restrict__ val, long total, int a) {
    for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int v = cm[g];
        if (v == 0) continue;
        const int j = (int)(g % a);
        const unsigned k = incl[g] - 1u;
        const int prev = j ? cm[g - 1] : 0;
        if (v != prev) {
            S[k] = j;
            key[k] = ((unsigned long long)(g / a) << 24) | (unsigned)v;
            val[k] = k;
        }
        if (j == a - 1 || cm[g + 1] != v) E[k] = j;
    }
}

__device__ __forceinline__ int rle_chars(long x) {            // rleToString's loop, counting only
    int n = 0;
    bool more = true;
    while (more) {
        const int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        ++n;
    }
    return n;
}

// per sorted segment: three slots (gap, length, trailing zero run or nothing) -> delta-coded value + char count
__global__ void seg_codes_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, const int* __restrict__ S,
                                 const int* __restrict__ E, int* __restrict__ xval, unsigned* __restrict__ nch, unsigned* __restrict__ first,
                                 int n, int a) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        const bool r1 = i > 0 && keys[i - 1] == key;             // at least one earlier segment of this instance
        const bool r2 = r1 && i > 1 && keys[i - 2] == key;
        const bool last = i == n - 1 || keys[i + 1] != key;
        const unsigned k = vals[i];
        const int s = S[k], e = E[k];
        int gap = s, len = e - s + 1, pgap = 0, plen = 0;
        if (r1) {
            const unsigned k1 = vals[i - 1];
            gap = s - (E[k1] + 1);
            plen = E[k1] - S[k1] + 1;
            if (r2) pgap = S[k1] - (E[vals[i - 2]] + 1);
        }
        const long xg = (long)gap - (r2 ? pgap : 0);             // count index 2r   > 2  <=>  r >= 2
        const long xl = (long)len - (r1 ? plen : 0);             // count index 2r+1 > 2  <=>  r >= 1
        xval[3 * i] = (int)xg;     nch[3 * i] = rle_chars(xg);
        xval[3 * i + 1] = (int)xl; nch[3 * i + 1] = rle_chars(xl);
        if (last && e < a - 1) {
            const long xt = (long)(a - 1 - e) - (r1 ? gap : 0);  // count index 2r+2 > 2  <=>  r >= 1
            xval[3 * i + 2] = (int)xt; nch[3 * i + 2] = rle_chars(xt);
        } else {
            xval[3 * i + 2] = 0; nch[3 * i + 2] = 0;
        }
        first[i] = r1 ? 0u : 1u;
    }
}

__global__ void table_init_kernel(int* __restrict__ table, int n, int H, int W) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int* t = table + (long)i * TABLE_W;
        t[4] = W; t[5] = H; t[6] = 0; t[7] = 0;
    }
}

__device__ __forceinline__ void rle_emit(char* dst, long x) {
    bool more = true;
    while (more) {
        char c = (char)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        *dst++ = (char)(c + 48);
    }
}

// off = exclusive scan of nch, inst = inclusive scan of first
__global__ void seg_emit_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ vals, const int* __restrict__ S,
                                const int* __restrict__ E, const int* __restrict__ xval, const unsigned* __restrict__ nch,
                                const unsigned* __restrict__ off, const unsigned* __restrict__ inst, char* __restrict__ chars,
                                int* __restrict__ table, int n, int H) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long key = keys[i];
        const bool first = i == 0 || keys[i - 1] != key;
        const bool last = i == n - 1 || keys[i + 1] != key;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (nch[3 * i + t]) rle_emit(chars + off[3 * i + t], (long)xval[3 * i + t]);
        int* row = table + (long)(inst[i] - 1u) * TABLE_W;
        if (first) { row[0] = (int)(key >> 24); row[1] = (int)(key & 0xffffffu); row[2] = (int)off[3 * i]; }
        if (last) row[3] = (int)(off[3 * i + 2] + nch[3 * i + 2]);
        // rleToBbox: run start and last pixel; a run that crosses a column boundary spans the full height
        const unsigned k = vals[i];
        const int s = S[k], e = E[k];
        const int xs = s / H, ys = s - xs * H, xe = e / H, ye = e - xe * H;
        atomicMin(row + 4, xs);
        atomicMax(row + 6, xe);
        if (xe > xs) { atomicMin(row + 5, 0); atomicMax(row + 7, H - 1); }
        else { atomicMin(row + 5, ys); atomicMax(row + 7, ye); }
    }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws1 {                       // segment extraction
    size_t cm, incl, S, E, key, val, tmp, tmp_bytes, total;
    Ws1(int layers, int H, int W) {
        const size_t px = (size_t)layers * H * W, cap = px;      // touching instances: every pixel can start a run
        size_t o = 0;
        cm = o;   o += align256(px * 4);
        incl = o; o += align256(px * 4);
        S = o;    o += align256(cap * 4);
        E = o;    o += align256(cap * 4);
        key = o;  o += align256(cap * 8);
        val = o;  o += align256(cap * 4);
        tmp_bytes = 0;
        (void)rocprim::inclusive_scan(nullptr, tmp_bytes, (unsigned*)nullptr, (unsigned*)nullptr, px, rocprim::plus<unsigned>());
        tmp = o;  o += align256(tmp_bytes);
        total = o;
    }
};

struct Ws2 {                       // sort, codes, output
    size_t key, val, xval, nch, off, first, table, chars, tmp, tmp_bytes, total;
    Ws2(int nseg) {
        const size_t n = nseg > 0 ? (size_t)nseg : 1;
        size_t o = 0;
        key = o;   o += align256(n * 8);
        val = o;   o += align256(n * 4);
        xval = o;  o += align256(3 * n * 4);
        nch = o;   o += align256(3 * n * 4);
        off = o;   o += align256(3 * n * 4);
        first = o; o += align256(n * 4);
        table = o; o += align256(n * TABLE_W * 4);
        chars = o; o += align256(21 * n);
        size_t a = 0, b = 0, c = 0;
        (void)rocprim::radix_sort_pairs(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                        (unsigned*)nullptr, n, 0u, 40u);
        (void)rocprim::exclusive_scan(nullptr, b, (unsigned*)nullptr, (unsigned*)nullptr, 0u, 3 * n, rocprim::plus<unsigned>());
        (void)rocprim::inclusive_scan(nullptr, c, (unsigned*)nullptr, (unsigned*)nullptr, n, rocprim::plus<unsigned>());
        tmp_bytes = a > b ? a : b;
        if (c > tmp_bytes) tmp_bytes = c;
        tmp = o;   o += align256(tmp_bytes);
        total = o;
    }
};

bool rle_shape_ok(int layers, int H, int W) {
    return layers > 0 && layers < 65536 && H > 0 && W > 0 && (long)H * W < (1L << 24) && (long)layers * H * W < 0x7fffffffL;
}

int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

#define HIP_OK(expr, what) \
    if ((expr) != hipSuccess) return msc_fail(MSC_ERR_HIP, what ": %s", hipGetErrorString(hipGetLastError()))

extern "C" int64_t msc_rle_segments_workspace(int layers, int H, int W) {
    if (!rle_shape_ok(layers, H, W)) return -1;
    return (int64_t)Ws1(layers, H, W).total;
}

extern "C" int msc_rle_segments(const int32_t* labels, int layers, int H, int W, void* ws, int64_t ws_bytes, int32_t* nseg, void* stream) {
    if (!labels || !ws || !nseg || !rle_shape_ok(layers, H, W)) return msc_fail(MSC_ERR_ARG, "msc_rle_segments: bad argument");
    const Ws1 L(layers, H, W);
    if ((size_t)ws_bytes < L.total) return msc_fail(MSC_ERR_ARG, "msc_rle_segments: workspace %lld < %zu bytes", (long long)ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    const int a = H * W;
    const long total = (long)layers * a;
    int* cm = (int*)(w + L.cm);
    unsigned* incl = (unsigned*)(w + L.incl);
    hipLaunchKernelGGL(transpose_cm_kernel, dim3(ceil_div(W, 32), ceil_div(H, 32), layers), dim3(32, 8), 0, st, labels, cm, H, W);
    hipLaunchKernelGGL(seg_flags_kernel, dim3(grid_for(total)), dim3(256), 0, st, cm, incl, total, a);
    size_t tb = L.tmp_bytes;
    HIP_OK(rocprim::inclusive_scan(w + L.tmp, tb, incl, incl, (size_t)total, rocprim::plus<unsigned>(), st), "msc_rle_segments: scan");
    hipLaunchKernelGGL(seg_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, st, cm, incl, (int*)(w + L.S), (int*)(w + L.E),
                       (unsigned long long*)(w + L.key), (unsigned*)(w + L.val), total, a);
    unsigned n = 0;
    HIP_OK(hipMemcpyAsync(&n, incl + total - 1, sizeof(n), hipMemcpyDeviceToHost, st), "msc_rle_segments: copy");
    HIP_OK(hipStreamSynchronize(st), "msc_rle_segments: sync");
    *nseg = (int32_t)n;
    return msc_check_launch("msc_rle_segments");
}

extern "C" int64_t msc_rle_encode_workspace(int nseg) {
    if (nseg < 0) return -1;
    return (int64_t)Ws2(nseg).total;
}

extern "C" int msc_rle_encode(const void* seg_ws, int layers, int H, int W, int nseg, void* ws, int64_t ws_bytes, int32_t* n_inst,
                              int64_t* n_chars, const int32_t** table, const char** chars, void* stream) {
    if (!seg_ws || !ws || !n_inst || !n_chars || !table || !chars || nseg < 0 || !rle_shape_ok(layers, H, W))
        return msc_fail(MSC_ERR_ARG, "msc_rle_encode: bad argument");
    const Ws1 L1(layers, H, W);
    const Ws2 L(nseg);
    if ((size_t)ws_bytes < L.total) return msc_fail(MSC_ERR_ARG, "msc_rle_encode: workspace %lld < %zu bytes", (long long)ws_bytes, L.total);
    if (21L * nseg >= 0x7fffffffL) return msc_fail(MSC_ERR_UNSUPPORTED, "msc_rle_encode: %d segments exceed 32-bit string offsets; split the batch", nseg);
    char* w = (char*)ws;
    *table = (const int32_t*)(w + L.table);
    *chars = w + L.chars;
    *n_inst = 0; *n_chars = 0;
    if (nseg == 0) return MSC_OK;
    hipStream_t st = (hipStream_t)stream;
    const char* w1 = (const char*)seg_ws;
    const int* S = (const int*)(w1 + L1.S);
    const int* E = (const int*)(w1 + L1.E);
    unsigned long long* key = (unsigned long long*)(w + L.key);
    unsigned* val = (unsigned*)(w + L.val);
    int* xval = (int*)(w + L.xval);
    unsigned* nch = (unsigned*)(w + L.nch);
    unsigned* off = (unsigned*)(w + L.off);
    unsigned* first = (unsigned*)(w + L.first);
    size_t tb = L.tmp_bytes;
    HIP_OK(rocprim::radix_sort_pairs(w + L.tmp, tb, (const unsigned long long*)(w1 + L1.key), key, (const unsigned*)(w1 + L1.val), val,
                                     (size_t)nseg, 0u, 40u, st), "msc_rle_encode: sort");
    hipLaunchKernelGGL(seg_codes_kernel, dim3(grid_for(nseg)), dim3(256), 0, st, key, val, S, E, xval, nch, first, nseg, H * W);
    tb = L.tmp_bytes;
    HIP_OK(rocprim::exclusive_scan(w + L.tmp, tb, nch, off, 0u, (size_t)(3 * nseg), rocprim::plus<unsigned>(), st), "msc_rle_encode: scan");
    tb = L.tmp_bytes;
    HIP_OK(rocprim::inclusive_scan(w + L.tmp, tb, first, first, (size_t)nseg, rocprim::plus<unsigned>(), st), "msc_rle_encode: scan");
    hipLaunchKernelGGL(table_init_kernel, dim3(grid_for(nseg)), dim3(256), 0, st, (int*)(w + L.table), nseg, H, W);
    hipLaunchKernelGGL(seg_emit_kernel, dim3(grid_for(nseg)), dim3(256), 0, st, key, val, S, E, xval, nch, off, first, w + L.chars,
                       (int*)(w + L.table), nseg, H);
    unsigned tail[3] = {0, 0, 0};        // instances, last offset, last count
    HIP_OK(hipMemcpyAsync(&tail[0], first + nseg - 1, 4, hipMemcpyDeviceToHost, st), "msc_rle_encode: copy");
    HIP_OK(hipMemcpyAsync(&tail[1], off + 3 * (size_t)nseg - 1, 4, hipMemcpyDeviceToHost, st), "msc_rle_encode: copy");
    HIP_OK(hipMemcpyAsync(&tail[2], nch + 3 * (size_t)nseg - 1, 4, hipMemcpyDeviceToHost, st), "msc_rle_encode: copy");
    HIP_OK(hipStreamSynchronize(st), "msc_rle_encode: sync");
    *n_inst = (int32_t)tail[0];
    *n_chars = (int64_t)tail[1] + tail[2];
    return msc_check_launch("msc_rle_encode");
}


// ------------------------------------------------------------------ annotations as JSON text (host)
// What create_annotations (src/utils.py:76-115) hands to json.dumps -- [{"image_id", "category_id", "score", "segmentation":
// {"size", "counts"}, "bbox"}, ...] -- written directly from the encoder's table: the reference builds one Python dict per
// instance, which at a few dozen instances per tile costs more than the whole device chain (0.2 ms per image on the GPU box's
// host).  Host memory in, host memory out; no device work.
#include <charconv>

namespace {
struct JsonOut {
    char* p; char* end; int64_t need;
    void raw(const char* s, size_t n) { need += (int64_t)n; if (p && n <= (size_t)(end - p)) { memcpy(p, s, n); p += n; } else p = end; }
    void lit(const char* s) { raw(s, strlen(s)); }
    void i64(long long v) { char b[24]; auto r = std::to_chars(b, b + sizeof(b), v); raw(b, (size_t)(r.ptr - b)); }
    void f64(double v) {                   // shortest text that reads back as the same double (what Python's repr gives json.dumps)
        if (v != v) { lit("NaN"); return; }                              // json.dumps' spellings of the non-finite values (they round-trip
        if (v - v != 0.0) { lit(v > 0 ? "Infinity" : "-Infinity"); return; }   // through json.loads; std::to_chars' "nan" / "inf" do not)
        char b[40];
        auto r = std::to_chars(b, b + sizeof(b), v);
        size_t n = (size_t)(r.ptr - b);
        bool plain = true;
        for (size_t i = 0; i < n; ++i) if (b[i] == '.' || b[i] == 'e' || b[i] == 'n' || b[i] == 'i') plain = false;
        raw(b, n);
        if (plain) lit(".0");
    }
    void str(const char* s, size_t n) {    // COCO count strings use the characters '0' .. 'o' (48 .. 111): '\\' (92) needs escaping, '"' cannot occur
        raw("\"", 1);
        size_t a = 0;
        for (size_t i = 0; i < n; ++i)
            if (s[i] == '\\') { raw(s + a, i - a); raw("\\\\", 2); a = i + 1; }
        raw(s + a, n - a);
        raw("\"", 1);
    }
};
}  // namespace

// table: i32 [n_inst][8] (layer, label, string begin, string end, xs, ys, xe, ye) sorted by (layer, label) as msc_rle_encode returns
// it (copied to the host); per encoded layer k < layers: image_ids[k], category_ids[k], counts[k] = number of score entries,
// scores + score_off[k] = its scores (label id - 1 indexes them).  Instances 1 .. min(largest id present, counts[k]) of every layer
// are written in id order, ids without pixels as empty masks (decompose(), src/utils.py:61-73, zip-truncated like :96-104); a layer
// without any instance contributes one empty mask if it has a score.  Returns the bytes the document needs (written if <= cap).
extern "C" int64_t msc_annotations_json(const int32_t* table, int n_inst, const char* chars, int layers, const int64_t* image_ids,
                                        const int32_t* category_ids, const int32_t* counts, const double* scores, const int64_t* score_off,
                                        int H, int W, char* out, int64_t cap) {
    if ((n_inst > 0 && (!table || !chars)) || layers < 0 || (layers > 0 && (!image_ids || !category_ids || !counts || !scores || !score_off)) || cap < 0)
        return msc_fail(MSC_ERR_ARG, "msc_annotations_json: bad argument");
    JsonOut o{out, out ? out + cap : out, 0};
    // the single count of an all-background mask (maskApi.c rleToString of [H*W])
    char empty[16];
    size_t ne = 0;
    {
        long x = (long)H * W;
        bool more = true;
        while (more) {
            long c = x & 0x1f;
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            empty[ne++] = (char)(c + 48);
        }
    }
    o.lit("[");
    bool first = true;
    int row = 0;
    for (int k = 0; k < layers; ++k) {
        int r0 = row;
        while (row < n_inst && table[(long)row * 8] == k) ++row;
        const int present = row - r0;
        const int max_id = present ? table[(long)(row - 1) * 8 + 1] : 1;      // decompose() of an instance-free layer yields ONE empty mask
        const int m = max_id < counts[k] ? max_id : counts[k];
        int r = r0;
        for (int id = 1; id <= m; ++id) {
            while (r < row && table[(long)r * 8 + 1] < id) ++r;
            const bool have = r < row && table[(long)r * 8 + 1] == id;
            const int32_t* t = table + (long)r * 8;
            if (!first) o.lit(", ");
            first = false;
            o.lit("{\"image_id\": "); o.i64(image_ids[k]);
            o.lit(", \"category_id\": "); o.i64(category_ids[k]);
            o.lit(", \"score\": "); o.f64(scores[score_off[k] + id - 1]);
            o.lit(", \"segmentation\": {\"size\": ["); o.i64(H); o.lit(", "); o.i64(W); o.lit("], \"counts\": ");
            if (have) o.str(chars + t[2], (size_t)(t[3] - t[2])); else o.str(empty, ne);
            o.lit("}, \"bbox\": [");
            if (have) { o.f64((double)t[4]); o.lit(", "); o.f64((double)t[5]); o.lit(", "); o.f64((double)(t[6] - t[4] + 1)); o.lit(", "); o.f64((double)(t[7] - t[5] + 1)); }
            else o.lit("0.0, 0.0, 0.0, 0.0");
            o.lit("]}");
        }
    }
    o.lit("]");
    return o.need;
}
