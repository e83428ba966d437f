/* ORACLE — test infrastructure only (never linked into, imported by or called from the product).
 *
 * Plain-C CPU restatement of the reference's per-tile HoVer-Net post-processing:
 *   DetectionCellPostProcessor.post_process_cell_segmentation / __proc_np_hv
 *     /root/reference/cell_segmentation/utils/post_proc_cellvit.py:67-153, 155-249
 *   get_bounding_box / remove_small_objects
 *     /root/reference/cell_segmentation/utils/tools.py:24-34, 61-101
 *
 * The arithmetic of that path lives in third-party packages that are NOT under /root/reference and
 * of which OpenCV cannot be imported in the development container:
 *   opencv_python_headless==4.5.4.58 (requirements.txt:12)  normalize, Sobel, GaussianBlur,
 *                                                            morphologyEx, moments, findContours
 *   scipy<1.8.2 (requirements.txt:23)                        ndimage.label, binary_fill_holes
 *   scikit-image==0.19.3 (requirements.txt:20)               segmentation.watershed
 * Their published algorithms are restated below, one function per call site.
 *
 * PARITY PIN STATUS: the scipy primitives (label, fill-holes) and the skimage watershed are pinned
 * by black-box comparison against the installed packages (tests/test_oracle_postproc.py; skimage
 * through golden vectors generated with /opt/conda/bin/python3.9, tools/make_golden_postproc.py).
 * The OpenCV-derived stages (min-max normalise, 21-tap Sobel, 3x3 blur, 5x5 ellipse opening,
 * moments, Suzuki-Abe contours) are "PARITY UNPINNED": no cv2 oracle exists here and the reference
 * has no tests at this boundary; they are pinned only by known-answer tests of the documented
 * OpenCV semantics.  Floating-point contraction is OFF (-ffp-contract=off); the two places where
 * OpenCV's SIMD convertTo uses a fused multiply-add are written as explicit fmaf()/fma().
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CVO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------- */
/* scipy.ndimage.label, default structure = 4-connectivity; ids 1..n in raster order of each     */
/* component's first pixel (post_proc_cellvit.py:181, 244).  Returns n.                          */
/* ------------------------------------------------------------------------------------------- */
static int32_t uf_find(int32_t* p, int32_t x) {
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

CVO_API int32_t cvo_label4(const int32_t* in, int H, int W, int32_t* out) {
    const int64_t N = (int64_t)H * W;
    int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) parent[i] = (int32_t)i;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t i = (int64_t)y * W + x;
            if (!in[i]) continue;
            if (x > 0 && in[i - 1]) {
                int32_t a = uf_find(parent, (int32_t)i), b = uf_find(parent, (int32_t)(i - 1));
                if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
            }
            if (y > 0 && in[i - W]) {
                int32_t a = uf_find(parent, (int32_t)i), b = uf_find(parent, (int32_t)(i - W));
                if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
            }
        }
    /* the root of every component is its raster-first pixel (union keeps the smaller index) */
    int32_t n = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (!in[i]) { out[i] = 0; continue; }
        const int32_t r = uf_find(parent, (int32_t)i);
        if (r == i) out[i] = ++n;          /* first visit of a component is always its root */
        else out[i] = out[r];
    }
    free(parent);
    return n;
}

/* tools.py:61-101 — zero every label whose pixel count is < min_size; ids are NOT compacted. */
CVO_API void cvo_remove_small(int32_t* lab, int H, int W, int32_t nlab, int min_size) {
    if (min_size == 0) return;
    const int64_t N = (int64_t)H * W;
    int64_t* cnt = (int64_t*)calloc((size_t)nlab + 1, sizeof(int64_t));
    for (int64_t i = 0; i < N; ++i) cnt[lab[i]]++;
    for (int64_t i = 0; i < N; ++i)
        if (cnt[lab[i]] < min_size) lab[i] = 0;   /* label 0 (background) is rewritten to 0: no-op */
    free(cnt);
}

/* ------------------------------------------------------------------------------------------- */
/* cv2.normalize(src, None, 0, 1, NORM_MINMAX, dtype=CV_32F)  (post_proc:185-200, 208-227).      */
/* OpenCV 4.5: scale = (1-0) * (max-min > DBL_EPSILON ? 1/(max-min) : 0); for a CV_32F result     */
/* scale and shift are rounded to float first (shift = (float)0 - (float)(min*scale)); then         */
/* src.convertTo(dst, CV_32F, scale, shift).  convertTo f32->f32 works in float, f64->f32 in       */
/* double, both with a fused multiply-add on FMA3 builds (the PyPI wheels).  [recalled]            */
/* ------------------------------------------------------------------------------------------- */
static void minmax_params(double mn, double mx, double* scale, double* shift) {
    double s = (mx - mn > DBL_EPSILON) ? 1.0 / (mx - mn) : 0.0;
    s = (double)(float)s;
    *scale = s;
    *shift = (double)((float)0.0f - (float)(mn * s));
}

CVO_API void cvo_normalize_f32(const float* src, int64_t n, float* dst) {
    float mn = src[0], mx = src[0];
    for (int64_t i = 1; i < n; ++i) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double sc, sh;
    minmax_params((double)mn, (double)mx, &sc, &sh);
    const float a = (float)sc, b = (float)sh;
    for (int64_t i = 0; i < n; ++i) dst[i] = fmaf(src[i], a, b);
}

CVO_API void cvo_normalize_f64(const double* src, int64_t n, float* dst) {
    double mn = src[0], mx = src[0];
    for (int64_t i = 1; i < n; ++i) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double sc, sh;
    minmax_params(mn, mx, &sc, &sh);
    for (int64_t i = 0; i < n; ++i) dst[i] = (float)fma(src[i], sc, sh);
}

/* ------------------------------------------------------------------------------------------- */
/* cv2.Sobel(src_f32, CV_64F, dx, dy, ksize)  (post_proc:205-206), ksize > 3: getSobelKernels      */
/* (integer binomial recurrences), separable correlation in double, BORDER_REFLECT_101.           */
/* Row pass = generic RowFilter (s = k[0]*S[0]; s += k[j]*S[j], j ascending); column pass =        */
/* SymmColumnFilter: symmetric  s = k0*S0; s += kj*(S[+j] + S[-j]);                                */
/*                   asymmetric s = 0;     s += kj*(S[+j] - S[-j]).                      [recalled] */
/* ------------------------------------------------------------------------------------------- */
CVO_API void cvo_sobel_kernel(int ksize, int order, double* out) {
    int64_t* k = (int64_t*)calloc((size_t)ksize + 1, sizeof(int64_t));
    k[0] = 1;
    for (int i = 0; i < ksize - order - 1; ++i) {
        int64_t oldv = k[0];
        for (int j = 1; j <= ksize; ++j) { const int64_t nv = k[j] + k[j - 1]; k[j - 1] = oldv; oldv = nv; }
    }
    for (int i = 0; i < order; ++i) {
        int64_t oldv = -k[0];
        for (int j = 1; j <= ksize; ++j) { const int64_t nv = k[j - 1] - k[j]; k[j - 1] = oldv; oldv = nv; }
    }
    for (int j = 0; j < ksize; ++j) out[j] = (double)k[j];
    free(k);
}

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

/* dx = 1: derivative along x (columns) + smoothing along y; dx = 0: the transpose (dy = 1). */
CVO_API void cvo_sobel(const float* src, int H, int W, int ksize, int dx, double* dst) {
    const int r = ksize / 2;
    double* kxr = (double*)malloc(sizeof(double) * (size_t)ksize);
    double* kyc = (double*)malloc(sizeof(double) * (size_t)ksize);
    cvo_sobel_kernel(ksize, dx ? 1 : 0, kxr);   /* row (x) kernel    */
    cvo_sobel_kernel(ksize, dx ? 0 : 1, kyc);   /* column (y) kernel */
    double* tmp = (double*)malloc(sizeof(double) * (size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s = kxr[0] * (double)src[(int64_t)y * W + reflect101(x - r, W)];
            for (int j = 1; j < ksize; ++j) s += kxr[j] * (double)src[(int64_t)y * W + reflect101(x - r + j, W)];
            tmp[(int64_t)y * W + x] = s;
        }
    const int col_symm = dx ? 1 : 0;   /* smoothing kernel is symmetric, derivative kernel anti-symmetric */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double s;
            if (col_symm) {
                s = kyc[r] * tmp[(int64_t)y * W + x];
                for (int j = 1; j <= r; ++j)
                    s += kyc[r + j] * (tmp[(int64_t)reflect101(y + j, H) * W + x] + tmp[(int64_t)reflect101(y - j, H) * W + x]);
            } else {
                s = 0.0;
                for (int j = 1; j <= r; ++j)
                    s += kyc[r + j] * (tmp[(int64_t)reflect101(y + j, H) * W + x] - tmp[(int64_t)reflect101(y - j, H) * W + x]);
            }
            dst[(int64_t)y * W + x] = s;
        }
    free(tmp); free(kxr); free(kyc);
}

/* cv2.GaussianBlur(src_f64, (3,3), 0): fixed kernel [1/4, 1/2, 1/4] per axis, REFLECT_101,
 * symmetric small filters: row D = S0*k0 + (S-1 + S+1)*k1, column likewise.  (post_proc:235) [recalled] */
CVO_API void cvo_blur3(const double* src, int H, int W, double* dst) {
    double* tmp = (double*)malloc(sizeof(double) * (size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double a = src[(int64_t)y * W + reflect101(x - 1, W)], b = src[(int64_t)y * W + x],
                         c = src[(int64_t)y * W + reflect101(x + 1, W)];
            tmp[(int64_t)y * W + x] = b * 0.5 + (a + c) * 0.25;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double a = tmp[(int64_t)reflect101(y - 1, H) * W + x], b = tmp[(int64_t)y * W + x],
                         c = tmp[(int64_t)reflect101(y + 1, H) * W + x];
            dst[(int64_t)y * W + x] = b * 0.5 + (a + c) * 0.25;
        }
    free(tmp);
}

/* scipy.ndimage.binary_fill_holes (post_proc:241): background pixels not 4-connected to the image
 * border become foreground. */
CVO_API void cvo_fill_holes(const uint8_t* in, int H, int W, uint8_t* out) {
    const int64_t N = (int64_t)H * W;
    uint8_t* outside = (uint8_t*)calloc((size_t)N, 1);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int64_t sp = 0;
#define PUSH_IF(i) do { if (!in[i] && !outside[i]) { outside[i] = 1; stack[sp++] = (int32_t)(i); } } while (0)
    for (int x = 0; x < W; ++x) { PUSH_IF(x); PUSH_IF((int64_t)(H - 1) * W + x); }
    for (int y = 0; y < H; ++y) { PUSH_IF((int64_t)y * W); PUSH_IF((int64_t)y * W + W - 1); }
    while (sp) {
        const int32_t i = stack[--sp];
        const int y = i / W, x = i - y * W;
        if (x > 0) PUSH_IF(i - 1);
        if (x + 1 < W) PUSH_IF(i + 1);
        if (y > 0) PUSH_IF(i - W);
        if (y + 1 < H) PUSH_IF(i + W);
    }
#undef PUSH_IF
    for (int64_t i = 0; i < N; ++i) out[i] = (in[i] || !outside[i]) ? 1 : 0;
    free(outside); free(stack);
}

/* cv2.morphologyEx(marker, MORPH_OPEN, getStructuringElement(MORPH_ELLIPSE, (5,5)))  (post_proc:242-243).
 * Element rows 00100/11111/11111/11111/00100; erosion then dilation, anchor at the centre, the
 * default constant border never erodes / never dilates from outside the image.          [recalled] */
static const int8_t ELL_DX0[5] = {0, -2, -2, -2, 0}, ELL_DX1[5] = {0, 2, 2, 2, 0};

CVO_API void cvo_open5(const uint8_t* in, int H, int W, uint8_t* out) {
    uint8_t* er = (uint8_t*)malloc((size_t)H * W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = 1;
            for (int dy = -2; dy <= 2 && v; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int ddx = ELL_DX0[dy + 2]; ddx <= ELL_DX1[dy + 2]; ++ddx) {
                    const int xx = x + ddx;
                    if (xx < 0 || xx >= W) continue;
                    if (!in[(int64_t)yy * W + xx]) { v = 0; break; }
                }
            }
            er[(int64_t)y * W + x] = v;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v = 0;
            for (int dy = -2; dy <= 2 && !v; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int ddx = ELL_DX0[dy + 2]; ddx <= ELL_DX1[dy + 2]; ++ddx) {
                    const int xx = x + ddx;
                    if (xx < 0 || xx >= W) continue;
                    if (er[(int64_t)yy * W + xx]) { v = 1; break; }
                }
            }
            out[(int64_t)y * W + x] = v;
        }
    free(er);
}

/* ------------------------------------------------------------------------------------------- */
/* skimage.segmentation.watershed(dist, markers, mask)  (post_proc:247): connectivity 1,          */
/* compactness 0, no watershed line.  Priority flood: all marker pixels are queued first (age 0);  */
/* pop the smallest (value, age); every unlabeled in-mask 4-neighbour (order -W, -1, +1, +W) takes  */
/* the popper's label AT PUSH TIME and is queued with value image[nbr], age = ++counter.           */
/* Ties between age-0 entries of equal value are resolved by flat index here (skimage: heap-layout  */
/* dependent) — the only known divergence, measure zero for real-valued dist.                      */
/* ------------------------------------------------------------------------------------------- */
typedef struct { double v; uint32_t age; int32_t idx; } heap_el;

static inline int el_less(const heap_el* a, const heap_el* b) {
    if (a->v != b->v) return a->v < b->v;
    if (a->age != b->age) return a->age < b->age;
    return a->idx < b->idx;
}

static void heap_push(heap_el* h, int64_t* n, heap_el e) {
    int64_t i = (*n)++;
    while (i > 0) {
        const int64_t p = (i - 1) >> 1;
        if (!el_less(&e, &h[p])) break;
        h[i] = h[p]; i = p;
    }
    h[i] = e;
}

static heap_el heap_pop(heap_el* h, int64_t* n) {
    const heap_el top = h[0];
    const heap_el last = h[--(*n)];
    int64_t i = 0;
    for (;;) {
        int64_t c = 2 * i + 1;
        if (c >= *n) break;
        if (c + 1 < *n && el_less(&h[c + 1], &h[c])) ++c;
        if (!el_less(&h[c], &last)) break;
        h[i] = h[c]; i = c;
    }
    if (*n > 0) h[i] = last;
    return top;
}

CVO_API void cvo_watershed(const double* image, const int32_t* markers, const int32_t* mask, int H, int W,
                           int32_t* out) {
    const int64_t N = (int64_t)H * W;
    heap_el* heap = (heap_el*)malloc(sizeof(heap_el) * (size_t)(N + 1));
    int64_t hn = 0;
    uint32_t age = 0;
    for (int64_t i = 0; i < N; ++i) {
        out[i] = mask[i] ? markers[i] : 0;            /* markers * mask (_validate_inputs) */
        if (out[i] != 0) { heap_el e = {image[i], 0u, (int32_t)i}; heap_push(heap, &hn, e); }
    }
    const int dxs[4] = {0, -1, 1, 0}, dys[4] = {-1, 0, 0, 1};
    while (hn) {
        const heap_el e = heap_pop(heap, &hn);
        const int y = e.idx / W, x = e.idx - y * W;
        for (int k = 0; k < 4; ++k) {
            const int yy = y + dys[k], xx = x + dxs[k];
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t j = (int64_t)yy * W + xx;
            if (!mask[j] || out[j] != 0) continue;
            out[j] = out[e.idx];
            heap_el ne = {image[j], ++age, (int32_t)j};
            heap_push(heap, &hn, ne);
        }
    }
    free(heap);
}

/* ------------------------------------------------------------------------------------------- */
/* __proc_np_hv (post_proc:155-249).  pred channels: binary argmax, hv0 (h_dir), hv1 (v_dir).      */
/* Optional stage dumps (may be NULL): blb [int32], dist [f64], marker [int32].                    */
/* ------------------------------------------------------------------------------------------- */
CVO_API int cvo_proc_np_hv(const uint8_t* bin_map, const float* hv0, const float* hv1, int H, int W,
                           int object_size, int ksize, int32_t* inst_out, int32_t* dbg_blb, double* dbg_dist,
                           int32_t* dbg_marker) {
    const int64_t N = (int64_t)H * W;
    int32_t* blb = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* lab = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) blb[i] = ((float)bin_map[i] >= 0.5f) ? 1 : 0;          /* :179 */
    int32_t n = cvo_label4(blb, H, W, lab);                                                 /* :181 */
    cvo_remove_small(lab, H, W, n, 10);                                                     /* :182 (hard-wired 10) */
    for (int64_t i = 0; i < N; ++i) blb[i] = lab[i] > 0 ? 1 : 0;                            /* :183 */

    float* hn = (float*)malloc(sizeof(float) * (size_t)N);
    float* vn = (float*)malloc(sizeof(float) * (size_t)N);
    cvo_normalize_f32(hv0, N, hn);                                                          /* :185-192 */
    cvo_normalize_f32(hv1, N, vn);                                                          /* :193-200 */
    double* sh = (double*)malloc(sizeof(double) * (size_t)N);
    double* sv = (double*)malloc(sizeof(double) * (size_t)N);
    cvo_sobel(hn, H, W, ksize, 1, sh);                                                      /* :205 */
    cvo_sobel(vn, H, W, ksize, 0, sv);                                                      /* :206 */
    cvo_normalize_f64(sh, N, hn);                                                           /* :208-217 */
    cvo_normalize_f64(sv, N, vn);                                                           /* :218-227 */
    double* overall = sh;   /* reuse */
    double* dist = sv;
    for (int64_t i = 0; i < N; ++i) {
        const float a = 1.0f - hn[i], b = 1.0f - vn[i];          /* float32 arithmetic (1 - f32 array) */
        const float m = a > b ? a : b;                            /* np.maximum            :229 */
        double o = (double)m - (double)(1 - blb[i]);              /* f32 - int32 -> f64    :230 */
        if (o < 0) o = 0;                                         /*                       :231 */
        overall[i] = o;
    }
    double* d0 = (double*)malloc(sizeof(double) * (size_t)N);
    for (int64_t i = 0; i < N; ++i) d0[i] = (1.0 - overall[i]) * (double)blb[i];           /* :233 */
    uint8_t* mk = (uint8_t*)malloc((size_t)N);
    uint8_t* mk2 = (uint8_t*)malloc((size_t)N);
    for (int64_t i = 0; i < N; ++i) {
        const int ob = overall[i] >= 0.4 ? 1 : 0;                                           /* :237 */
        int m = blb[i] - ob; if (m < 0) m = 0;                                              /* :239-240 */
        mk[i] = (uint8_t)m;
    }
    cvo_blur3(d0, H, W, dist);                                                              /* :235 */
    for (int64_t i = 0; i < N; ++i) dist[i] = -dist[i];
    cvo_fill_holes(mk, H, W, mk2);                                                          /* :241 */
    cvo_open5(mk2, H, W, mk);                                                               /* :242-243 */
    for (int64_t i = 0; i < N; ++i) lab[i] = mk[i];
    int32_t* marker = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    n = cvo_label4(lab, H, W, marker);                                                      /* :244 */
    cvo_remove_small(marker, H, W, n, object_size);                                         /* :245 */
    cvo_watershed(dist, marker, blb, H, W, inst_out);                                       /* :247 */
    if (dbg_blb) memcpy(dbg_blb, blb, sizeof(int32_t) * (size_t)N);
    if (dbg_dist) memcpy(dbg_dist, dist, sizeof(double) * (size_t)N);
    if (dbg_marker) memcpy(dbg_marker, marker, sizeof(int32_t) * (size_t)N);
    free(blb); free(lab); free(hn); free(vn); free(sh); free(sv); free(d0); free(mk); free(mk2); free(marker);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Per-instance records (post_proc:95-151).  Contour: cv2.findContours(RETR_TREE,                  */
/* CHAIN_APPROX_SIMPLE)[0][0] of the instance's bbox crop = Suzuki-Abe outer border of the first    */
/* component in raster order (icvFetchContour): start at the raster-first pixel, initial clockwise   */
/* search from W, then counter-clockwise neighbour search; a point is emitted whenever the chain     */
/* direction changes.  moments m10/m00, m01/m00 -> centroid.                              [recalled] */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t id;
    int32_t rmin, cmin, rmax, cmax;      /* bbox, max exclusive (tools.py:24-34) */
    int32_t npix;
    int32_t type;
    int32_t contour_off, contour_len;    /* into the (x, y) int32 contour arena; len < 3 => skipped by the reference */
    double cx, cy;                       /* centroid (x, y) in tile coordinates */
    double type_prob;
} cvo_instance;

static const int DIRX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DIRY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* trace the outer border of label `id`; writes (x, y) pairs if pts != NULL; returns the point count */
static int trace_contour(const int32_t* inst, int H, int W, int32_t id, int x0, int y0, int32_t* pts) {
#define ON(xx, yy) ((xx) >= 0 && (xx) < W && (yy) >= 0 && (yy) < H && inst[(int64_t)(yy) * W + (xx)] == id)
    int n = 0;
    int s = 4, s_end = 4;
    int x1, y1;
    do {
        s = (s - 1) & 7;
        x1 = x0 + DIRX[s]; y1 = y0 + DIRY[s];
    } while (!ON(x1, y1) && s != s_end);
    if (s == s_end) {                        /* isolated pixel */
        if (pts) { pts[0] = x0; pts[1] = y0; }
        return 1;
    }
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (;;) {
        int x4, y4;
        s_end = s;
        for (;;) {
            ++s;
            x4 = x3 + DIRX[s & 7]; y4 = y3 + DIRY[s & 7];
            if (ON(x4, y4)) break;
        }
        s &= 7;
        if (s != prev_s) {
            if (pts) { pts[2 * n] = x3; pts[2 * n + 1] = y3; }
            ++n;
            prev_s = s;
        }
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
#undef ON
    (void)s_end;
    return n;
}

/* Returns the number of records written (<= max_inst); *n_pts = contour points written (<= max_pts). */
CVO_API int cvo_instances(const int32_t* inst, const uint8_t* type_map, int H, int W, int nr_types,
                          cvo_instance* recs, int max_inst, int32_t* contour_xy, int max_pts, int* n_pts) {
    const int64_t N = (int64_t)H * W;
    int32_t maxid = 0;
    int has_zero = 0;
    for (int64_t i = 0; i < N; ++i) { if (inst[i] > maxid) maxid = inst[i]; if (inst[i] == 0) has_zero = 1; }
    int32_t* slot = (int32_t*)malloc(sizeof(int32_t) * ((size_t)maxid + 1));
    for (int32_t i = 0; i <= maxid; ++i) slot[i] = -1;
    int64_t* sx = (int64_t*)calloc((size_t)maxid + 1, sizeof(int64_t));
    int64_t* sy = (int64_t*)calloc((size_t)maxid + 1, sizeof(int64_t));
    int32_t* firstpix = (int32_t*)malloc(sizeof(int32_t) * ((size_t)maxid + 1));
    int64_t* hist = (int64_t*)calloc(((size_t)maxid + 1) * (size_t)(nr_types > 0 ? nr_types : 1), sizeof(int64_t));
    /* pass 1: discover ids (ascending id order == np.unique), bbox, sums */
    int32_t* rmin = (int32_t*)malloc(sizeof(int32_t) * ((size_t)maxid + 1) * 5);
    int32_t *cmin = rmin + (maxid + 1), *rmax = cmin + (maxid + 1), *cmax = rmax + (maxid + 1), *cnt = cmax + (maxid + 1);
    for (int32_t i = 0; i <= maxid; ++i) { rmin[i] = H; cmin[i] = W; rmax[i] = -1; cmax[i] = -1; cnt[i] = 0; firstpix[i] = -1; }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int32_t id = inst[(int64_t)y * W + x];
            if (id <= 0) continue;
            if (firstpix[id] < 0) firstpix[id] = y * W + x;
            if (y < rmin[id]) rmin[id] = y;
            if (y > rmax[id]) rmax[id] = y;
            if (x < cmin[id]) cmin[id] = x;
            if (x > cmax[id]) cmax[id] = x;
            cnt[id]++; sx[id] += x; sy[id] += y;
            if (nr_types > 0) hist[(size_t)id * nr_types + type_map[(int64_t)y * W + x]]++;
        }
    int nrec = 0, npt = 0;
    int first = 1;
    for (int32_t id = 1; id <= maxid; ++id) {
        if (cnt[id] == 0) continue;
        /* np.unique(pred_inst)[1:] drops the SMALLEST value: label 0 if present, else the smallest id (quirk 1) */
        if (first && !has_zero) { first = 0; continue; }
        first = 0;
        if (nrec >= max_inst) break;
        cvo_instance* r = &recs[nrec];
        r->id = id; r->rmin = rmin[id]; r->cmin = cmin[id]; r->rmax = rmax[id] + 1; r->cmax = cmax[id] + 1;
        r->npix = cnt[id];
        /* cv2.moments on the crop: m10/m00 + bbox origin (post_proc:117-125) */
        const double m00 = (double)cnt[id];
        const double m10 = (double)(sx[id] - (int64_t)cnt[id] * cmin[id]);
        const double m01 = (double)(sy[id] - (int64_t)cnt[id] * rmin[id]);
        r->cx = m10 / m00 + (double)cmin[id];
        r->cy = m01 / m00 + (double)rmin[id];
        const int fx = firstpix[id] % W, fy = firstpix[id] / W;
        const int len = trace_contour(inst, H, W, id, fx, fy, NULL);
        r->contour_off = npt; r->contour_len = len;
        if (contour_xy && npt + len <= max_pts) trace_contour(inst, H, W, id, fx, fy, contour_xy + 2 * (size_t)npt);
        npt += len;
        /* type vote (post_proc:134-151): stable sort by count desc over ascending type ids */
        r->type = 0; r->type_prob = 0.0;
        if (nr_types > 0) {
            const int64_t* hcnt = &hist[(size_t)id * nr_types];
            int best = -1, second = -1, present = 0;
            for (int t = 0; t < nr_types; ++t) {
                if (!hcnt[t]) continue;
                ++present;
                if (best < 0 || hcnt[t] > hcnt[best]) { second = best; best = t; }
                else if (second < 0 || hcnt[t] > hcnt[second]) second = t;
            }
            /* `second` must be the runner-up in the stable order: recompute exactly */
            second = -1;
            for (int t = 0; t < nr_types; ++t) {
                if (!hcnt[t] || t == best) continue;
                if (second < 0 || hcnt[t] > hcnt[second]) second = t;
            }
            int ty = best;
            if (ty == 0 && present > 1) ty = second;
            r->type = ty;
            r->type_prob = (double)hcnt[ty] / ((double)cnt[id] + 1.0e-6);
        }
        ++nrec;
    }
    if (n_pts) *n_pts = npt;
    free(slot); free(sx); free(sy); free(firstpix); free(hist); free(rmin);
    return nrec;
}

/* post_process_cell_segmentation (post_proc:67-153) on one tile: instance map + records. */
CVO_API int cvo_postprocess_tile(const uint8_t* type_map, const uint8_t* bin_map, const float* hv0,
                                 const float* hv1, int H, int W, int magnification, int nr_types,
                                 int32_t* inst_out, cvo_instance* recs, int max_inst, int32_t* contour_xy,
                                 int max_pts, int* n_pts) {
    int object_size, ksize;
    if (magnification == 40) { object_size = 10; ksize = 21; }
    else if (magnification == 20) { object_size = 3; ksize = 11; }
    else return -1;                                     /* NotImplementedError("Unknown magnification") :61-62 */
    cvo_proc_np_hv(bin_map, hv0, hv1, H, W, object_size, ksize, inst_out, NULL, NULL, NULL);
    return cvo_instances(inst_out, type_map, H, W, nr_types, recs, max_inst, contour_xy, max_pts, n_pts);
}
