/* TEST INFRASTRUCTURE -- plain-C restatement of the default mask post-processing chain, used as the
 * single-thread CPU baseline ("port") in bench.py and cross-checked against oracle/post_ref.py in
 * tests/test_oracle.py.  Never linked into the product.
 *
 * Restates, per image (reference file:line under /root/reference):
 *   resize_image                 src/postprocessing.py:48-61  (scipy map_coordinates order 1, mode 'constant': no
 *                                interpolation beyond the edges -> 0)
 *   categorize_multilayer_image  src/postprocessing.py:77-84  (CATEGORY_LAYERS [1,1] -> threshold 0.5 per class)
 *   label_multilayer_image       src/postprocessing.py:127-132, src/utils.py:328-330 (4-connectivity, raster order)
 *   dilate_image                 src/postprocessing.py:159-180 (k x k max filter of the label image, skimage origin)
 *   build_score                  src/postprocessing.py:228-236 (mean prob x sqrt(area) per label)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int find_root(int32_t* parent, int a) {
    while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; }
    return a;
}

/* mask u8[h*w] -> labels i32[h*w]; returns the number of components */
int msc_ref_label4(const uint8_t* mask, int32_t* labels, int h, int w) {
    int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * (size_t)h * w);
    for (int i = 0; i < h * w; ++i) parent[i] = i;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int i = y * w + x;
            if (!mask[i]) continue;
            if (x > 0 && mask[i - 1]) { int a = find_root(parent, i), b = find_root(parent, i - 1); if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; } }
            if (y > 0 && mask[i - w]) { int a = find_root(parent, i), b = find_root(parent, i - w); if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; } }
        }
    int n = 0;
    for (int i = 0; i < h * w; ++i) {            /* roots are first pixels in raster order: number them as met */
        if (!mask[i]) { labels[i] = 0; continue; }
        const int r = find_root(parent, i);
        if (r == i) labels[i] = ++n; else labels[i] = labels[r];
    }
    free(parent);
    return n;
}

static int reflect_idx(int i, int n) { if (i < 0) i = -i - 1; if (i >= n) i = 2 * n - i - 1; return i < 0 ? 0 : (i >= n ? n - 1 : i); }

void msc_ref_dilate_i32(const int32_t* in, int32_t* out, int h, int w, int k) {
    const int lo = -((k - 1) / 2), hi = lo + k - 1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int32_t m = in[reflect_idx(y + lo, h) * w + reflect_idx(x + lo, w)];
            for (int dy = lo; dy <= hi; ++dy)
                for (int dx = lo; dx <= hi; ++dx) {
                    const int32_t v = in[reflect_idx(y + dy, h) * w + reflect_idx(x + dx, w)];
                    if (v > m) m = v;
                }
            out[y * w + x] = m;
        }
}

/* scipy's arithmetic rounding for rounding (see oracle/post_ref.py resize_image): weights 1-a and 1-(1-a), per tap (coefficient * row weight) *
 * column weight, taps in raster order; then skimage's clip to [lo, hi] = the input range joined with cval 0.  Build with -ffp-contract=off. */
void msc_ref_resize(const float* in, double* out, int h, int w, int H, int W, double lo, double hi) {
    const double fy = (double)h / H, fx = (double)w / W;
    for (int oy = 0; oy < H; ++oy)
        for (int ox = 0; ox < W; ++ox) {
            const double ys = fy * (oy + 0.5) - 0.5, xs = fx * (ox + 0.5) - 0.5;
            double r = 0.0;
            if (ys >= 0.0 && ys <= h - 1 && xs >= 0.0 && xs <= w - 1) {
                int y0 = (int)floor(ys), x0 = (int)floor(xs);
                const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
                const double wy0 = 1.0 - (ys - y0), wx0 = 1.0 - (xs - x0);
                const double wy1 = 1.0 - wy0, wx1 = 1.0 - wx0;
                r = ((double)in[y0 * w + x0] * wy0) * wx0;
                r = r + ((double)in[y0 * w + x1] * wy0) * wx1;
                r = r + ((double)in[y1 * w + x0] * wy1) * wx0;
                r = r + ((double)in[y1 * w + x1] * wy1) * wx1;
            }
            out[oy * W + ox] = r < lo ? lo : (r > hi ? hi : r);
        }
}

/* whole chain for one image: probs f32[2][h][w] -> labels i32[2][H][W], counts[2], scores f64[2][max_labels].
 * returns 0, or -1 if a layer has more than max_labels components */
int msc_ref_postprocess(const float* probs, int h, int w, int H, int W, int dilate, int32_t* labels, int32_t* counts,
                        double* scores, int max_labels) {
    const size_t HW = (size_t)H * W;
    double* r = (double*)malloc(sizeof(double) * HW);
    uint8_t* m = (uint8_t*)malloc(HW);
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * HW);
    double* sum = (double*)malloc(sizeof(double) * (size_t)(max_labels + 1));
    int64_t* area = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_labels + 1));
    int rc = 0;
    double lo = 0.0, hi = 0.0;
    for (size_t i = 0; i < (size_t)2 * h * w; ++i) { if (probs[i] < lo) lo = probs[i]; if (probs[i] > hi) hi = probs[i]; }
    for (int c = 0; c < 2 && rc == 0; ++c) {
        msc_ref_resize(probs + (size_t)c * h * w, r, h, w, H, W, lo, hi);
        for (size_t i = 0; i < HW; ++i) m[i] = r[i] > 0.5;
        int32_t* lab = labels + c * HW;
        const int n = msc_ref_label4(m, dilate > 0 ? tmp : lab, H, W);
        if (dilate > 0) msc_ref_dilate_i32(tmp, lab, H, W, dilate);
        counts[c] = n;
        if (n > max_labels) { rc = -1; break; }
        memset(sum, 0, sizeof(double) * (size_t)(n + 1));
        memset(area, 0, sizeof(int64_t) * (size_t)(n + 1));
        for (size_t i = 0; i < HW; ++i) { sum[lab[i]] += r[i]; area[lab[i]]++; }
        for (int l = 1; l <= n; ++l) scores[c * max_labels + l - 1] = area[l] ? sum[l] / (double)area[l] * sqrt((double)area[l]) : 0.0;
    }
    free(r); free(m); free(tmp); free(sum); free(area);
    return rc;
}
