/* TEST INFRASTRUCTURE -- CPU oracle: dense-CRF mean field on the PERMUTOHEDRAL LATTICE.  Never linked into the product.
 *
 * The reference's dense_crf (src/postprocessing.py:183-225) calls pydensecrf (environment.yml:15, an unpinned git HEAD that is not
 * vendored in /root/reference and not installable here), a wrapper of Kraehenbuehl & Koltun's densecrf ("Efficient Inference in
 * Fully Connected CRFs with Gaussian Edge Potentials", NIPS 2011 -- the paper the docstring at src/postprocessing.py:189-192 cites).
 * densecrf evaluates the message passing step  (K Q)_i = sum_j k(f_i, f_j) Q_j  approximately, with the high-dimensional Gaussian
 * filter of Adams, Baek & Davis, "Fast High-Dimensional Filtering Using the Permutohedral Lattice" (Eurographics 2010).  This file
 * restates that published algorithm -- it is written from the two papers, with densecrf's conventions where the papers leave a
 * choice (listed below) -- so that the distance between the lattice filter and the EXACT windowed filter of oracle/crf_ref.py (which
 * the HIP kernel msc_dense_crf implements) can be stated and tested (tests/test_oracle_crf.py).  PARITY WITH pydensecrf ITSELF STAYS
 * UNPINNED: there is no binary of it here to compare with.
 *
 * Lattice (Adams et al. 2010, sections 3-4), d = feature dimension (2: Gaussian kernel, 5: bilateral kernel):
 *   elevate   f in R^d -> the hyperplane H_d = {x in R^(d+1): sum x = 0} with the triangular basis E, scaled so that splat + blur
 *             + slice together have unit variance per feature: scale_i = (d+1) * sqrt(2/3) / sqrt((i+1)(i+2))   (p. 5-6)
 *   simplex   round to the nearest remainder-0 lattice point (multiples of d+1), rank the residuals, walk back onto H_d   (p. 6-7)
 *   splat     barycentric weights to the d+1 vertices of the enclosing simplex (p. 10), vertices kept in a hash table
 *   blur      along each of the d+1 lattice directions with the kernel [1/2, 1, 1/2] (densecrf's un-normalised form; the constant
 *             alpha = 1 / (1 + 2^-d) applied at the slice compensates)
 *   slice     the same barycentric weights
 * Mean field (Kraehenbuehl & Koltun 2011, Algorithm 1, as densecrf runs it): Q = softmax(-U); repeat: Q = softmax(-U + sum_k w_k *
 * n_k * Filter_k(n_k * Q)) with the symmetric normalisation n_k = 1 / sqrt(Filter_k(1) + 1e-20) and a Potts compatibility.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/build/libcrf_lattice_ref.so oracle/crf_lattice_ref.c -lm   (__graft_entry__.build()) */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int d, n, m;          /* feature dimension, points, lattice vertices */
    int* offset;          /* [n][d+1] vertex index of each simplex corner */
    float* bary;          /* [n][d+1] barycentric weight of each corner */
    int* nb1;             /* [d+1][m] neighbour -1 along direction j (or -1) */
    int* nb2;             /* [d+1][m] neighbour +1 */
} lattice;

/* ---- open-addressing hash table of lattice keys (d shorts each; the (d+1)-th coordinate is minus their sum) ---- */
typedef struct { int d, cap, filled; short* keys; int* table; } hasht;

static uint64_t key_hash(const short* k, int d) {
    uint64_t h = 0;
    for (int i = 0; i < d; ++i) { h += (uint64_t)(int64_t)k[i]; h *= 1664525u; }
    return h;
}
static void ht_init(hasht* t, int d, int expect) {
    t->d = d; t->filled = 0;
    t->cap = 1;
    while (t->cap < 2 * expect + 16) t->cap <<= 1;
    t->keys = (short*)malloc((size_t)(expect + 16) * d * sizeof(short));
    t->table = (int*)malloc((size_t)t->cap * sizeof(int));
    for (int i = 0; i < t->cap; ++i) t->table[i] = -1;
}
static int ht_find(hasht* t, const short* k, int create) {
    uint64_t h = key_hash(k, t->d) & (uint64_t)(t->cap - 1);
    for (;;) {
        int e = t->table[h];
        if (e < 0) {
            if (!create) return -1;
            memcpy(t->keys + (size_t)t->filled * t->d, k, (size_t)t->d * sizeof(short));
            t->table[h] = t->filled;
            return t->filled++;
        }
        if (!memcmp(t->keys + (size_t)e * t->d, k, (size_t)t->d * sizeof(short))) return e;
        h = (h + 1) & (uint64_t)(t->cap - 1);
    }
}

/* features: [n][d] floats (already divided by the kernel's standard deviations) */
static lattice* lattice_build(const float* feat, int n, int d) {
    lattice* L = (lattice*)calloc(1, sizeof(lattice));
    L->d = d; L->n = n;
    L->offset = (int*)malloc((size_t)n * (d + 1) * sizeof(int));
    L->bary = (float*)malloc((size_t)n * (d + 1) * sizeof(float));
    hasht ht;
    ht_init(&ht, d, n * (d + 1));
    float* scale = (float*)malloc(d * sizeof(float));
    float* elev = (float*)malloc((d + 1) * sizeof(float));
    float* rem0 = (float*)malloc((d + 1) * sizeof(float));
    float* bc = (float*)malloc((d + 2) * sizeof(float));
    short* rank = (short*)malloc((d + 1) * sizeof(short));
    short* canon = (short*)malloc((size_t)(d + 1) * (d + 1) * sizeof(short));
    short* key = (short*)malloc((d + 1) * sizeof(short));
    /* canonical simplex: vertex r = (r, ..., r, r-(d+1), ..., r-(d+1)) with d+1-r leading entries */
    for (int i = 0; i <= d; ++i) {
        for (int j = 0; j <= d - i; ++j) canon[i * (d + 1) + j] = (short)i;
        for (int j = d - i + 1; j <= d; ++j) canon[i * (d + 1) + j] = (short)(i - (d + 1));
    }
    const float inv_std = sqrtf(2.0f / 3.0f) * (float)(d + 1);
    for (int i = 0; i < d; ++i) scale[i] = 1.0f / sqrtf((float)((i + 2) * (i + 1))) * inv_std;
    const float down = 1.0f / (float)(d + 1), up = (float)(d + 1);
    for (int p = 0; p < n; ++p) {
        const float* f = feat + (size_t)p * d;
        float sm = 0.f;
        for (int j = d; j > 0; --j) {
            const float cf = f[j - 1] * scale[j - 1];
            elev[j] = sm - (float)j * cf;
            sm += cf;
        }
        elev[0] = sm;
        /* nearest remainder-0 point */
        int sum = 0;
        for (int i = 0; i <= d; ++i) {
            const float v = down * elev[i];
            const float u = ceilf(v) * up, l = floorf(v) * up;
            const int rd2 = (u - elev[i] < elev[i] - l) ? (int)u : (int)l;
            rem0[i] = (float)rd2;
            sum += (int)lrintf((float)rd2 * down);
        }
        /* rank of the residuals (0 = largest) */
        for (int i = 0; i <= d; ++i) rank[i] = 0;
        for (int i = 0; i < d; ++i) {
            const float di = elev[i] - rem0[i];
            for (int j = i + 1; j <= d; ++j) {
                if (di < elev[j] - rem0[j]) rank[i]++;
                else rank[j]++;
            }
        }
        /* back onto the hyperplane */
        for (int i = 0; i <= d; ++i) {
            rank[i] = (short)(rank[i] + sum);
            if (rank[i] < 0) { rank[i] = (short)(rank[i] + d + 1); rem0[i] += up; }
            else if (rank[i] > d) { rank[i] = (short)(rank[i] - (d + 1)); rem0[i] -= up; }
        }
        /* barycentric coordinates */
        for (int i = 0; i <= d + 1; ++i) bc[i] = 0.f;
        for (int i = 0; i <= d; ++i) {
            const float v = (elev[i] - rem0[i]) * down;
            bc[d - rank[i]] += v;
            bc[d - rank[i] + 1] -= v;
        }
        bc[0] += 1.0f + bc[d + 1];
        for (int r = 0; r <= d; ++r) {
            for (int i = 0; i < d; ++i) key[i] = (short)((int)rem0[i] + canon[r * (d + 1) + rank[i]]);
            L->offset[(size_t)p * (d + 1) + r] = ht_find(&ht, key, 1);
            L->bary[(size_t)p * (d + 1) + r] = bc[r];
        }
    }
    const int m = ht.filled;
    L->m = m;
    L->nb1 = (int*)malloc((size_t)(d + 1) * m * sizeof(int));
    L->nb2 = (int*)malloc((size_t)(d + 1) * m * sizeof(int));
    short* n1 = (short*)malloc((d + 1) * sizeof(short));
    short* n2 = (short*)malloc((d + 1) * sizeof(short));
    for (int j = 0; j <= d; ++j)
        for (int i = 0; i < m; ++i) {
            const short* k = ht.keys + (size_t)i * d;
            for (int c = 0; c < d; ++c) { n1[c] = (short)(k[c] - 1); n2[c] = (short)(k[c] + 1); }
            if (j < d) { n1[j] = (short)(k[j] + d); n2[j] = (short)(k[j] - d); }      /* j == d: the implied coordinate moves */
            L->nb1[(size_t)j * m + i] = ht_find(&ht, n1, 0);
            L->nb2[(size_t)j * m + i] = ht_find(&ht, n2, 0);
        }
    free(n1); free(n2); free(scale); free(elev); free(rem0); free(bc); free(rank); free(canon); free(key);
    free(ht.keys); free(ht.table);
    return L;
}

static void lattice_free(lattice* L) {
    if (!L) return;
    free(L->offset); free(L->bary); free(L->nb1); free(L->nb2); free(L);
}

/* out[p][c] = Filter(in)[p][c], values interleaved per point (vs channels) */
static void lattice_filter(const lattice* L, const float* in, float* out, int vs) {
    const int d = L->d, n = L->n, m = L->m;
    float* val = (float*)calloc((size_t)(m + 1) * vs, sizeof(float));      /* slot 0 = "no neighbour" (zero) */
    float* nxt = (float*)calloc((size_t)(m + 1) * vs, sizeof(float));
    for (int p = 0; p < n; ++p)
        for (int r = 0; r <= d; ++r) {
            const int o = L->offset[(size_t)p * (d + 1) + r] + 1;
            const float w = L->bary[(size_t)p * (d + 1) + r];
            for (int c = 0; c < vs; ++c) val[(size_t)o * vs + c] += w * in[(size_t)p * vs + c];
        }
    for (int j = 0; j <= d; ++j) {
        for (int i = 0; i < m; ++i) {
            const float* a = val + (size_t)(L->nb1[(size_t)j * m + i] + 1) * vs;
            const float* b = val + (size_t)(L->nb2[(size_t)j * m + i] + 1) * vs;
            const float* o = val + (size_t)(i + 1) * vs;
            float* q = nxt + (size_t)(i + 1) * vs;
            for (int c = 0; c < vs; ++c) q[c] = o[c] + 0.5f * (a[c] + b[c]);
        }
        float* t = val; val = nxt; nxt = t;
        memset(val, 0, (size_t)vs * sizeof(float));                          /* keep the zero slot zero */
    }
    const float alpha = 1.0f / (1.0f + powf(2.0f, -(float)d));
    for (int p = 0; p < n; ++p) {
        for (int c = 0; c < vs; ++c) out[(size_t)p * vs + c] = 0.f;
        for (int r = 0; r <= d; ++r) {
            const int o = L->offset[(size_t)p * (d + 1) + r] + 1;
            const float w = L->bary[(size_t)p * (d + 1) + r];
            for (int c = 0; c < vs; ++c) out[(size_t)p * vs + c] += w * val[(size_t)o * vs + c] * alpha;
        }
    }
    free(val); free(nxt);
}

/* public: the bare filter (tests compare it with the exact Gaussian sum) */
int msc_ref_lattice_filter(const float* feat, int n, int d, const float* in, float* out, int vs) {
    if (!feat || !in || !out || n <= 0 || d <= 0 || d > 15 || vs <= 0) return -1;
    lattice* L = lattice_build(feat, n, d);
    lattice_filter(L, in, out, vs);
    lattice_free(L);
    return 0;
}

static void softmax_cols(const float* e, float* q, int n, int M) {      /* q = softmax over the M labels of e[M][n] */
    for (int p = 0; p < n; ++p) {
        float mx = e[p];
        for (int c = 1; c < M; ++c) mx = fmaxf(mx, e[(size_t)c * n + p]);
        float s = 0.f;
        for (int c = 0; c < M; ++c) { q[(size_t)c * n + p] = expf(e[(size_t)c * n + p] - mx); s += q[(size_t)c * n + p]; }
        for (int c = 0; c < M; ++c) q[(size_t)c * n + p] /= s;
    }
}

/* Mean field of src/postprocessing.py:183-225 with lattice filters.  unary f32[M][H][W] (energies), rgb u8[H][W][3],
 * out f32[M][H][W].  Two kernels: Gaussian (x/sxy_g, y/sxy_g) weight w_g; bilateral (x/sxy_b, y/sxy_b, rgb/srgb) weight w_b. */
int msc_ref_dense_crf_lattice(const float* unary, const uint8_t* rgb, int M, int H, int W, float w_g, float sxy_g, float w_b,
                              float sxy_b, float srgb, int iterations, float* out) {
    if (!unary || !rgb || !out || M <= 0 || H <= 0 || W <= 0) return -1;
    const int n = H * W;
    float* f2 = (float*)malloc((size_t)n * 2 * sizeof(float));
    float* f5 = (float*)malloc((size_t)n * 5 * sizeof(float));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int p = y * W + x;
            f2[p * 2] = (float)x / sxy_g; f2[p * 2 + 1] = (float)y / sxy_g;
            f5[p * 5] = (float)x / sxy_b; f5[p * 5 + 1] = (float)y / sxy_b;
            for (int c = 0; c < 3; ++c) f5[p * 5 + 2 + c] = (float)rgb[(size_t)p * 3 + c] / srgb;
        }
    lattice* Ls[2] = {lattice_build(f2, n, 2), lattice_build(f5, n, 5)};
    const float wts[2] = {w_g, w_b};
    float* norm[2];
    float* ones = (float*)malloc((size_t)n * sizeof(float));
    for (int p = 0; p < n; ++p) ones[p] = 1.f;
    for (int k = 0; k < 2; ++k) {
        norm[k] = (float*)malloc((size_t)n * sizeof(float));
        lattice_filter(Ls[k], ones, norm[k], 1);
        for (int p = 0; p < n; ++p) norm[k][p] = 1.0f / sqrtf(norm[k][p] + 1e-20f);
    }
    float* q = (float*)malloc((size_t)M * n * sizeof(float));
    float* e = (float*)malloc((size_t)M * n * sizeof(float));
    float* tin = (float*)malloc((size_t)M * n * sizeof(float));      /* [n][M] interleaved */
    float* tout = (float*)malloc((size_t)M * n * sizeof(float));
    for (size_t i = 0; i < (size_t)M * n; ++i) e[i] = -unary[i];
    softmax_cols(e, q, n, M);
    for (int it = 0; it < iterations; ++it) {
        for (size_t i = 0; i < (size_t)M * n; ++i) e[i] = -unary[i];
        for (int k = 0; k < 2; ++k) {
            for (int p = 0; p < n; ++p)
                for (int c = 0; c < M; ++c) tin[(size_t)p * M + c] = q[(size_t)c * n + p] * norm[k][p];
            lattice_filter(Ls[k], tin, tout, M);
            for (int p = 0; p < n; ++p)
                for (int c = 0; c < M; ++c) e[(size_t)c * n + p] += wts[k] * tout[(size_t)p * M + c] * norm[k][p];
        }
        softmax_cols(e, q, n, M);
    }
    memcpy(out, q, (size_t)M * n * sizeof(float));
    free(q); free(e); free(tin); free(tout); free(ones); free(norm[0]); free(norm[1]); free(f2); free(f5);
    lattice_free(Ls[0]); lattice_free(Ls[1]);
    return 0;
}
