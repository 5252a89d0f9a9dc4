/*
 * orc_seam.c -- CPU ORACLE (test infrastructure): cv.detail_DpSeamFinder('COLOR' | 'COLOR_GRAD').
 *
 * The reference's default seam finder (stitching_detailed_enhanced.py:243-249, called at :1618 on the float32 seam-scale
 * warps of :1601-1604 and the warped masks of :1591-1599).  The algorithm lives in OpenCV 4.6.0
 * modules/stitching/src/seam_finders.cpp (class DpSeamFinder), which is NOT under /root/reference and not installed here;
 * this file restates the published algorithm function by function (names below are OpenCV's):
 *
 *   find              all pairs (i < j), std::sort by the squared distance of the image centres, reversed (far pairs first)
 *   process           union canvas of the two masks, contour masks, findComponents, findEdges, resolveConflicts
 *   findComponents    4-connected components of {both, first only, second only}, numbered in raster order of their first pixel
 *   findEdges         component adjacency (4-neighbourhood of the contour pixels)
 *   resolveConflicts  while an INTERSection component touches a component of the "wrong" image: give it away whole (one
 *                     neighbour) or cut it along a minimum-cost path (getSeamTips, estimateSeam, updateLabelsUsingSeam)
 *   computeGradients  cvtColor(BGR2GRAY) + Sobel(CV_32F, 3x3) of both images (COLOR_GRAD)
 *   computeCosts      per pixel-edge cost: mean squared colour difference across the edge, divided by 1 + the sum of the
 *                     four gradient magnitudes next to it (COLOR_GRAD)
 *   estimateSeam      dynamic programme over the component's bounding box, three predecessors per cell
 *
 * Pinning: no cv2 binary is at hand.  tests/test_seam_dp.py pins this file (a) on hand-checkable cases, (b) through the
 * properties the algorithm guarantees (masks only shrink, the cut masks of a pair are disjoint on their common area and
 * together still cover it), and (c) against the recorded run's 21 seamed masks
 * (the JPEGs under `..._06_masks_warped_seamed/`, shrunk and JPEG coded: agreement is measured as area overlap, not bit for bit).
 * Details that cannot be read off the recorded masks and are restated from OpenCV's source as remembered are marked [CV-U]:
 * the float association of cvtColor / Sobel, and libstdc++'s std::sort order among equal keys (restated below: introsort with
 * median-of-three pivots and a final insertion sort, threshold 16).
 */
#include "orc_internal.h"

enum { ST_FIRST = 1, ST_SECOND = 2, ST_INTERS = 4 };

typedef struct { int x, y; } pt_t;
typedef struct { pt_t *p; int n, cap; } ptvec_t;

static void pv_push(ptvec_t *v, int x, int y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 64;
        v->p = (pt_t *)realloc(v->p, sizeof(pt_t) * (size_t)v->cap);
    }
    v->p[v->n].x = x; v->p[v->n].y = y; ++v->n;
}

/* ---- std::sort of libstdc++ on (i, j) pairs with ImagePairLess -------------------------------------------------------------- */
typedef struct { int a, b; } pair_t;
typedef struct { const int *corners, *sizes; } pairless_t;

static int pair_dist(const pairless_t *c, pair_t p)
{
    const int c1x = c->corners[2 * p.a] + c->sizes[2 * p.a] / 2, c1y = c->corners[2 * p.a + 1] + c->sizes[2 * p.a + 1] / 2;
    const int c2x = c->corners[2 * p.b] + c->sizes[2 * p.b] / 2, c2y = c->corners[2 * p.b + 1] + c->sizes[2 * p.b + 1] / 2;
    return (c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y);
}
static int pair_less(const pairless_t *c, pair_t l, pair_t r) { return pair_dist(c, l) < pair_dist(c, r); }
static void pair_swap(pair_t *a, pair_t *b) { pair_t t = *a; *a = *b; *b = t; }

static void gnu_unguarded_linear_insert(pair_t *last, const pairless_t *c)
{
    pair_t val = *last, *next = last - 1;
    while (pair_less(c, val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void gnu_insertion_sort(pair_t *first, pair_t *last, const pairless_t *c)
{
    if (first == last) return;
    for (pair_t *i = first + 1; i != last; ++i) {
        if (pair_less(c, *i, *first)) {
            pair_t val = *i;
            memmove(first + 1, first, sizeof(pair_t) * (size_t)(i - first));
            *first = val;
        } else
            gnu_unguarded_linear_insert(i, c);
    }
}
static void gnu_adjust_heap(pair_t *first, long hole, long len, pair_t value, const pairless_t *c)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (pair_less(c, first[child], first[child - 1])) --child;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;   /* __push_heap */
    while (hole > top && pair_less(c, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void gnu_heap_sort(pair_t *first, pair_t *last, const pairless_t *c)   /* partial_sort(first, last, last) */
{
    const long len = last - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; --parent) {
            gnu_adjust_heap(first, parent, len, first[parent], c);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        pair_t value = *last;
        *last = *first;
        gnu_adjust_heap(first, 0, last - first, value, c);
    }
}
static void gnu_introsort_loop(pair_t *first, pair_t *last, long depth, const pairless_t *c)
{
    while (last - first > 16) {
        if (depth == 0) { gnu_heap_sort(first, last, c); return; }
        --depth;
        /* __move_median_to_first(first, first + 1, mid, last - 1) */
        pair_t *a = first + 1, *b = first + (last - first) / 2, *d = last - 1;
        if (pair_less(c, *a, *b)) {
            if (pair_less(c, *b, *d)) pair_swap(first, b);
            else if (pair_less(c, *a, *d)) pair_swap(first, d);
            else pair_swap(first, a);
        } else if (pair_less(c, *a, *d)) pair_swap(first, a);
        else if (pair_less(c, *b, *d)) pair_swap(first, d);
        else pair_swap(first, b);
        /* __unguarded_partition(first + 1, last, first) */
        pair_t *lo = first + 1, *hi = last;
        for (;;) {
            while (pair_less(c, *lo, *first)) ++lo;
            --hi;
            while (pair_less(c, *first, *hi)) --hi;
            if (!(lo < hi)) break;
            pair_swap(lo, hi);
            ++lo;
        }
        gnu_introsort_loop(lo, last, depth, c);
        last = lo;
    }
}
static void gnu_sort(pair_t *first, pair_t *last, const pairless_t *c)
{
    if (first == last) return;
    long n = last - first, lg = 0;
    while ((n >> (lg + 1)) > 0) ++lg;
    gnu_introsort_loop(first, last, 2 * lg, c);
    if (last - first > 16) {
        gnu_insertion_sort(first, first + 16, c);
        for (pair_t *i = first + 16; i != last; ++i) gnu_unguarded_linear_insert(i, c);
    } else
        gnu_insertion_sort(first, last, c);
}

/* ---- the per-pair state (DpSeamFinder's members) ------------------------------------------------------------------------------ */
typedef struct {
    int cost_func;                  /* 0 COLOR, 1 COLOR_GRAD */
    int uw, uh, utlx, utly;         /* unionSize_, unionTl_ */
    uint8_t *mask1, *mask2, *cont1, *cont2;
    int *labels;
    int ncomps;
    int *states;
    pt_t *tls, *brs;
    ptvec_t *contours;
    uint8_t *edges;                 /* ncomps x ncomps adjacency: the std::set<pair<int,int>> edges_ */
    const float *gx1, *gy1, *gx2, *gy2;
    int w1, h1, w2, h2;
} dp_t;

#define LBL(s, y, x) ((s)->labels[(size_t)(y) * (s)->uw + (x)])

/* cv::floodFill(image CV_32S, seed, newVal), 4-connected, loDiff = upDiff = 0: the component of equal values */
static void flood_fill_i32(int *img, int w, int h, int sx, int sy, int new_val, int *stack)
{
    const int old = img[(size_t)sy * w + sx];
    if (old == new_val) return;
    int sp = 0;
    stack[sp++] = sy * w + sx;
    img[(size_t)sy * w + sx] = new_val;
    while (sp) {
        const int o = stack[--sp], y = o / w, x = o - y * w;
        if (x > 0 && img[o - 1] == old) { img[o - 1] = new_val; stack[sp++] = o - 1; }
        if (x < w - 1 && img[o + 1] == old) { img[o + 1] = new_val; stack[sp++] = o + 1; }
        if (y > 0 && img[o - w] == old) { img[o - w] = new_val; stack[sp++] = o - w; }
        if (y < h - 1 && img[o + w] == old) { img[o + w] = new_val; stack[sp++] = o + w; }
    }
}

static int is_contour(const dp_t *s, int y, int x, int l)
{
    return (x == 0 || LBL(s, y, x - 1) != l) || (x == s->uw - 1 || LBL(s, y, x + 1) != l) || (y == 0 || LBL(s, y - 1, x) != l) ||
           (y == s->uh - 1 || LBL(s, y + 1, x) != l);
}

static void find_components(dp_t *s)
{
    const int W = s->uw, H = s->uh;
    int cap = 16;
    s->ncomps = 0;
    s->states = (int *)malloc(sizeof(int) * (size_t)cap);
    s->tls = (pt_t *)malloc(sizeof(pt_t) * (size_t)cap);
    s->brs = (pt_t *)malloc(sizeof(pt_t) * (size_t)cap);
    s->contours = (ptvec_t *)calloc((size_t)cap, sizeof(ptvec_t));
    int *stack = (int *)malloc(sizeof(int) * (size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t o = (size_t)y * W + x;
            s->labels[o] = (s->mask1[o] && s->mask2[o]) ? INT_MAX : s->mask1[o] ? INT_MAX - 1 : s->mask2[o] ? INT_MAX - 2 : 0;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (LBL(s, y, x) >= INT_MAX - 2) {
                if (s->ncomps == cap) {
                    cap *= 2;
                    s->states = (int *)realloc(s->states, sizeof(int) * (size_t)cap);
                    s->tls = (pt_t *)realloc(s->tls, sizeof(pt_t) * (size_t)cap);
                    s->brs = (pt_t *)realloc(s->brs, sizeof(pt_t) * (size_t)cap);
                    s->contours = (ptvec_t *)realloc(s->contours, sizeof(ptvec_t) * (size_t)cap);
                    memset(s->contours + cap / 2, 0, sizeof(ptvec_t) * (size_t)(cap / 2));
                }
                const int v = LBL(s, y, x);
                s->states[s->ncomps] = v == INT_MAX ? ST_INTERS : v == INT_MAX - 1 ? ST_FIRST : ST_SECOND;
                flood_fill_i32(s->labels, W, H, x, y, s->ncomps + 1, stack);
                s->tls[s->ncomps].x = x; s->tls[s->ncomps].y = y;
                s->brs[s->ncomps].x = x + 1; s->brs[s->ncomps].y = y + 1;
                ++s->ncomps;
            }
            if (LBL(s, y, x)) {
                const int l = LBL(s, y, x), ci = l - 1;
                if (x < s->tls[ci].x) s->tls[ci].x = x;
                if (y < s->tls[ci].y) s->tls[ci].y = y;
                if (x + 1 > s->brs[ci].x) s->brs[ci].x = x + 1;
                if (y + 1 > s->brs[ci].y) s->brs[ci].y = y + 1;
                if (is_contour(s, y, x, l)) pv_push(&s->contours[ci], x, y);
            }
        }
    free(stack);
}

static void find_edges(dp_t *s)
{
    const int n = s->ncomps;
    s->edges = (uint8_t *)calloc((size_t)n * n + 1, 1);
    for (int ci = 0; ci < n; ++ci)
        for (int i = 0; i < s->contours[ci].n; ++i) {
            const int x = s->contours[ci].p[i].x, y = s->contours[ci].p[i].y, l = ci + 1;
            int o;
            if (x > 0 && (o = LBL(s, y, x - 1)) && o != l) s->edges[(size_t)ci * n + o - 1] = s->edges[(size_t)(o - 1) * n + ci] = 1;
            if (y > 0 && (o = LBL(s, y - 1, x)) && o != l) s->edges[(size_t)ci * n + o - 1] = s->edges[(size_t)(o - 1) * n + ci] = 1;
            if (x < s->uw - 1 && (o = LBL(s, y, x + 1)) && o != l) s->edges[(size_t)ci * n + o - 1] = s->edges[(size_t)(o - 1) * n + ci] = 1;
            if (y < s->uh - 1 && (o = LBL(s, y + 1, x)) && o != l) s->edges[(size_t)ci * n + o - 1] = s->edges[(size_t)(o - 1) * n + ci] = 1;
        }
}

static int has_only_one_neighbor(const dp_t *s, int comp)
{
    int cnt = 0;
    for (int j = 0; j < s->ncomps; ++j) cnt += s->edges[(size_t)comp * s->ncomps + j];
    return cnt == 1;
}

static int close_to_contour(const dp_t *s, int y, int x, const uint8_t *cm)
{
    const int rad = 2;
    for (int dy = -rad; dy <= rad; ++dy)
        if (y + dy >= 0 && y + dy < s->uh)
            for (int dx = -rad; dx <= rad; ++dx)
                if (x + dx >= 0 && x + dx < s->uw && cm[(size_t)(y + dy) * s->uw + (x + dx)]) return 1;
    return 0;
}

static int touches(const dp_t *s, int y, int x, int l2)
{
    return (x > 0 && LBL(s, y, x - 1) == l2) || (y > 0 && LBL(s, y - 1, x) == l2) || (x < s->uw - 1 && LBL(s, y, x + 1) == l2) ||
           (y < s->uh - 1 && LBL(s, y + 1, x) == l2);
}

/* cvRound(double) */
static double round_half_even(double v) { return rint(v); }

static int get_seam_tips(const dp_t *s, int comp1, int comp2, pt_t *p1, pt_t *p2)
{
    ptvec_t sp = {0, 0, 0};
    const int l2 = comp2 + 1;
    for (int i = 0; i < s->contours[comp1].n; ++i) {
        const int x = s->contours[comp1].p[i].x, y = s->contours[comp1].p[i].y;
        if (close_to_contour(s, y, x, s->cont1) && close_to_contour(s, y, x, s->cont2) && touches(s, y, x, l2)) pv_push(&sp, x, y);
    }
    if (sp.n < 2) { free(sp.p); return 0; }
    /* cv::partition(specialPoints, labels, ClosePoints(10)): classes of the transitive closure of |p - q|^2 < 100, numbered in
     * the order their first member appears */
    const int N = sp.n;
    int *parent = (int *)malloc(sizeof(int) * (size_t)N), *cls = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; ++i) parent[i] = i;
    for (int i = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j) {
            const int dx = sp.p[i].x - sp.p[j].x, dy = sp.p[i].y - sp.p[j].y;
            if (dx * dx + dy * dy < 100) {
                int a = i, b = j;
                while (parent[a] != a) a = parent[a];
                while (parent[b] != b) b = parent[b];
                if (a != b) parent[b > a ? b : a] = b > a ? a : b;
            }
        }
    int nlabels = 0;
    int *root_cls = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; ++i) root_cls[i] = -1;
    for (int i = 0; i < N; ++i) {
        int r = i;
        while (parent[r] != r) r = parent[r];
        if (root_cls[r] < 0) root_cls[r] = nlabels++;
        cls[i] = root_cls[r];
    }
    free(root_cls); free(parent);
    if (nlabels < 2) { free(cls); free(sp.p); return 0; }
    long *sumx = (long *)calloc((size_t)nlabels, sizeof(long)), *sumy = (long *)calloc((size_t)nlabels, sizeof(long));
    int *cnt = (int *)calloc((size_t)nlabels, sizeof(int));
    for (int i = 0; i < N; ++i) { sumx[cls[i]] += sp.p[i].x; sumy[cls[i]] += sp.p[i].y; ++cnt[cls[i]]; }
    int idx[2] = {-1, -1};
    double max_dist = -1.7976931348623157e308;
    for (int i = 0; i < nlabels - 1; ++i)
        for (int j = i + 1; j < nlabels; ++j) {
            const double cx1 = round_half_even(sumx[i] / (double)cnt[i]), cy1 = round_half_even(sumy[i] / (double)cnt[i]);
            const double cx2 = round_half_even(sumx[j] / (double)cnt[j]), cy2 = round_half_even(sumy[j] / (double)cnt[j]);
            const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
            if (dist > max_dist) { max_dist = dist; idx[0] = i; idx[1] = j; }
        }
    pt_t p[2];
    for (int k = 0; k < 2; ++k) {
        const double cx = round_half_even(sumx[idx[k]] / (double)cnt[idx[k]]), cy = round_half_even(sumy[idx[k]] / (double)cnt[idx[k]]);
        double min_dist = 1.7976931348623157e308;
        p[k].x = p[k].y = 0;
        for (int i = 0; i < N; ++i) {
            if (cls[i] != idx[k]) continue;
            const double d = (sp.p[i].x - cx) * (sp.p[i].x - cx) + (sp.p[i].y - cy) * (sp.p[i].y - cy);
            if (d < min_dist) { min_dist = d; p[k] = sp.p[i]; }
        }
    }
    *p1 = p[0]; *p2 = p[1];
    free(sumx); free(sumy); free(cnt); free(cls); free(sp.p);
    return 1;
}

static float sqr_f(float v) { return v * v; }
/* diffL2Square3<float> */
static float diff3(const float *im1, int w1, int y1, int x1, const float *im2, int w2, int y2, int x2)
{
    const float *a = im1 + ((size_t)y1 * w1 + x1) * 3, *b = im2 + ((size_t)y2 * w2 + x2) * 3;
    return sqr_f(a[0] - b[0]) + sqr_f(a[1] - b[1]) + sqr_f(a[2] - b[2]);
}

static void compute_costs(const dp_t *s, const float *im1, const float *im2, int tl1x, int tl1y, int tl2x, int tl2y, int comp, float *costV, float *costH)
{
    const int l = comp + 1;
    const int rx = s->tls[comp].x, ry = s->tls[comp].y, rw = s->brs[comp].x - rx, rh = s->brs[comp].y - ry;
    const int dx1 = s->utlx - tl1x, dy1 = s->utly - tl1y, dx2 = s->utlx - tl2x, dy2 = s->utly - tl2y;
    const float bad = 255.f * 255.f + 255.f * 255.f + 255.f * 255.f;   /* normL2(Point3f(255,255,255), 0): the SQUARED norm */
    const int w1 = s->w1, w2 = s->w2;
    for (int y = ry; y < ry + rh; ++y)
        for (int x = rx; x < rx + rw + 1; ++x) {
            float *out = &costV[(size_t)(y - ry) * (rw + 1) + (x - rx)];
            if (x > 0 && x < s->uw && LBL(s, y, x) == l && LBL(s, y, x - 1) == l) {
                const float cc = (diff3(im1, w1, y + dy1, x + dx1 - 1, im2, w2, y + dy2, x + dx2) + diff3(im1, w1, y + dy1, x + dx1, im2, w2, y + dy2, x + dx2 - 1)) / 2;
                if (!s->cost_func) *out = cc;
                else {
                    const float cg = fabsf(s->gx1[(size_t)(y + dy1) * w1 + x + dx1]) + fabsf(s->gx1[(size_t)(y + dy1) * w1 + x + dx1 - 1]) +
                                     fabsf(s->gx2[(size_t)(y + dy2) * w2 + x + dx2]) + fabsf(s->gx2[(size_t)(y + dy2) * w2 + x + dx2 - 1]) + 1.f;
                    *out = cc / cg;
                }
            } else
                *out = bad;
        }
    for (int y = ry; y < ry + rh + 1; ++y)
        for (int x = rx; x < rx + rw; ++x) {
            float *out = &costH[(size_t)(y - ry) * rw + (x - rx)];
            if (y > 0 && y < s->uh && LBL(s, y, x) == l && LBL(s, y - 1, x) == l) {
                const float cc = (diff3(im1, w1, y + dy1 - 1, x + dx1, im2, w2, y + dy2, x + dx2) + diff3(im1, w1, y + dy1, x + dx1, im2, w2, y + dy2 - 1, x + dx2)) / 2;
                if (!s->cost_func) *out = cc;
                else {
                    const float cg = fabsf(s->gy1[(size_t)(y + dy1) * w1 + x + dx1]) + fabsf(s->gy1[(size_t)(y + dy1 - 1) * w1 + x + dx1]) +
                                     fabsf(s->gy2[(size_t)(y + dy2) * w2 + x + dx2]) + fabsf(s->gy2[(size_t)(y + dy2 - 1) * w2 + x + dx2]) + 1.f;
                    *out = cc / cg;
                }
            } else
                *out = bad;
        }
}

/* min_element over std::pair<float, int>: smaller cost, then smaller step code */
static void take_step(float c, int code, float *best, int *best_code, int *n)
{
    if (*n == 0 || c < *best || (!(*best < c) && code < *best_code)) { *best = c; *best_code = code; }
    ++*n;
}

static int estimate_seam(const dp_t *s, const float *im1, const float *im2, int tl1x, int tl1y, int tl2x, int tl2y, int comp, pt_t p1, pt_t p2, ptvec_t *seam,
                         int *is_horizontal)
{
    const int rx = s->tls[comp].x, ry = s->tls[comp].y, rw = s->brs[comp].x - rx, rh = s->brs[comp].y - ry, l = comp + 1;
    float *costV = (float *)malloc(sizeof(float) * (size_t)rh * (rw + 1)), *costH = (float *)malloc(sizeof(float) * (size_t)(rh + 1) * rw);
    compute_costs(s, im1, im2, tl1x, tl1y, tl2x, tl2y, comp, costV, costH);
    pt_t src = {p1.x - rx, p1.y - ry}, dst = {p2.x - rx, p2.y - ry};
    int swapped = 0;
    *is_horizontal = abs(dst.x - src.x) > abs(dst.y - src.y);
    if (*is_horizontal) {
        if (src.x > dst.x) { pt_t t = src; src = dst; dst = t; swapped = 1; }
    } else if (src.y > dst.y) { pt_t t = src; src = dst; dst = t; swapped = 1; }
    uint8_t *control = (uint8_t *)calloc((size_t)rw * rh, 1), *reach = (uint8_t *)calloc((size_t)rw * rh, 1);
    float *cost = (float *)calloc((size_t)rw * rh, sizeof(float));
#define CV_(y, x) costV[(size_t)(y) * (rw + 1) + (x)]
#define CH_(y, x) costH[(size_t)(y) * rw + (x)]
#define AT(a, y, x) a[(size_t)(y) * rw + (x)]
    AT(reach, src.y, src.x) = 1;
    AT(cost, src.y, src.x) = 0.f;
    if (*is_horizontal) {
        for (int x = src.x + 1; x <= dst.x; ++x)
            for (int y = 0; y < rh; ++y) {
                int n = 0, code = 0;
                float best = 0.f;
                if (LBL(s, y + ry, x + rx) == l) {
                    if (AT(reach, y, x - 1)) take_step(AT(cost, y, x - 1) + CH_(y, x - 1), 1, &best, &code, &n);
                    if (y > 0 && AT(reach, y - 1, x - 1)) take_step(AT(cost, y - 1, x - 1) + CH_(y - 1, x - 1) + CV_(y - 1, x), 2, &best, &code, &n);
                    if (y < rh - 1 && AT(reach, y + 1, x - 1)) take_step(AT(cost, y + 1, x - 1) + CH_(y + 1, x - 1) + CV_(y, x), 3, &best, &code, &n);
                }
                if (n) { AT(cost, y, x) = best; AT(control, y, x) = (uint8_t)code; AT(reach, y, x) = 255; }
            }
    } else {
        for (int y = src.y + 1; y <= dst.y; ++y)
            for (int x = 0; x < rw; ++x) {
                int n = 0, code = 0;
                float best = 0.f;
                if (LBL(s, y + ry, x + rx) == l) {
                    if (AT(reach, y - 1, x)) take_step(AT(cost, y - 1, x) + CV_(y - 1, x), 1, &best, &code, &n);
                    if (x > 0 && AT(reach, y - 1, x - 1)) take_step(AT(cost, y - 1, x - 1) + CV_(y - 1, x - 1) + CH_(y, x - 1), 2, &best, &code, &n);
                    if (x < rw - 1 && AT(reach, y - 1, x + 1)) take_step(AT(cost, y - 1, x + 1) + CV_(y - 1, x + 1) + CH_(y, x), 3, &best, &code, &n);
                }
                if (n) { AT(cost, y, x) = best; AT(control, y, x) = (uint8_t)code; AT(reach, y, x) = 255; }
            }
    }
    int ok = AT(reach, dst.y, dst.x) != 0;
    if (ok) {
        pt_t p = dst;
        seam->n = 0;
        pv_push(seam, p.x + rx, p.y + ry);
        if (*is_horizontal) {
            while (p.x != src.x) {
                const int c = AT(control, p.y, p.x);
                if (c == 2) p.y--; else if (c == 3) p.y++;
                p.x--;
                pv_push(seam, p.x + rx, p.y + ry);
            }
        } else {
            while (p.y != src.y) {
                const int c = AT(control, p.y, p.x);
                if (c == 2) p.x--; else if (c == 3) p.x++;
                p.y--;
                pv_push(seam, p.x + rx, p.y + ry);
            }
        }
        if (!swapped)
            for (int i = 0, j = seam->n - 1; i < j; ++i, --j) { pt_t t = seam->p[i]; seam->p[i] = seam->p[j]; seam->p[j] = t; }
        /* CV_Assert(seam.front() == p1 && seam.back() == p2) */
        if (seam->p[0].x != p1.x || seam->p[0].y != p1.y || seam->p[seam->n - 1].x != p2.x || seam->p[seam->n - 1].y != p2.y) ok = -1;
    }
#undef CV_
#undef CH_
#undef AT
    free(costV); free(costH); free(control); free(reach); free(cost);
    return ok;
}

static void update_labels_using_seam(dp_t *s, int comp1, int comp2, const ptvec_t *seam, int is_horizontal)
{
    const int tx = s->tls[comp1].x, ty = s->tls[comp1].y, mw = s->brs[comp1].x - tx, mh = s->brs[comp1].y - ty;
    int *mask = (int *)calloc((size_t)mw * mh, sizeof(int)), *stack = (int *)malloc(sizeof(int) * (size_t)mw * mh);
#define M(y, x) mask[(size_t)(y) * mw + (x)]
    const ptvec_t *ct = &s->contours[comp1];
    for (int i = 0; i < ct->n; ++i) M(ct->p[i].y - ty, ct->p[i].x - tx) = 255;
    for (int i = 0; i < seam->n; ++i) M(seam->p[i].y - ty, seam->p[i].x - tx) = 255;
    const int l1 = comp1 + 1, l2 = comp2 + 1;
    int ncomps = 0;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x)
            if (!M(y, x) && LBL(s, y + ty, x + tx) == l1) flood_fill_i32(mask, mw, mh, x, y, ++ncomps, stack);   /* a 255th part would alias the marker, as in OpenCV */
    for (int i = 0; i < ct->n; ++i) {
        const int x = ct->p[i].x - tx, y = ct->p[i].y - ty;
        static const int dx[] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy[] = {0, 0, -1, +1, -1, -1, +1, +1};
        int ok = 0;
        for (int j = 0; j < 8; ++j) {
            const int c = x + dx[j], r = y + dy[j];
            if (c >= 0 && c < mw && r >= 0 && r < mh && M(r, c) && M(r, c) != 255) { ok = 1; M(y, x) = M(r, c); }
        }
        if (!ok) M(y, x) = 0;
    }
    for (int i = 0; i < seam->n; ++i) {
        const int x = seam->p[i].x - tx, y = seam->p[i].y - ty;
        if (is_horizontal) {
            if (y < mh - 1 && M(y + 1, x) && M(y + 1, x) != 255) M(y, x) = M(y + 1, x);
            else M(y, x) = 0;
        } else {
            if (x < mw - 1 && M(y, x + 1) && M(y, x + 1) != 255) M(y, x) = M(y, x + 1);
            else M(y, x) = 0;
        }
    }
    /* new components connected with the second component, and with components other than the two at work (std::map keys
     * 0 .. ncomps: key 0 collects contour pixels that lost their part and never marks anything) */
    const int nk = ncomps + 1;
    int *connect2 = (int *)calloc((size_t)nk, sizeof(int)), *connect_other = (int *)calloc((size_t)nk, sizeof(int));
    for (int i = 0; i < ct->n; ++i) {
        const int x = ct->p[i].x, y = ct->p[i].y, m = M(y - ty, x - tx);
        if (m < 0 || m >= nk) continue;
        if (touches(s, y, x, l2)) connect2[m]++;
        if ((x > 0 && LBL(s, y, x - 1) != l1 && LBL(s, y, x - 1) != l2) || (y > 0 && LBL(s, y - 1, x) != l1 && LBL(s, y - 1, x) != l2) ||
            (x < s->uw - 1 && LBL(s, y, x + 1) != l1 && LBL(s, y, x + 1) != l2) || (y < s->uh - 1 && LBL(s, y + 1, x) != l1 && LBL(s, y + 1, x) != l2))
            connect_other[m]++;
    }
    uint8_t *adj = (uint8_t *)calloc((size_t)nk, 1);
    const double len = (double)ct->n;
    for (int k = 1; k < nk; ++k)
        if (connect2[k] / len > 0.05 && connect_other[k] / len < 0.1) adj[k] = 1;
    for (int y = 0; y < mh; ++y)
        for (int x = 0; x < mw; ++x)
            if (M(y, x) > 0 && M(y, x) < nk && adj[M(y, x)]) LBL(s, y + ty, x + tx) = l2;
#undef M
    free(mask); free(stack); free(connect2); free(connect_other); free(adj);
}

static void refresh_component(dp_t *s, int c)
{
    const int l = c + 1, x0 = s->tls[c].x, x1 = s->brs[c].x, y0 = s->tls[c].y, y1 = s->brs[c].y;
    s->tls[c].x = s->tls[c].y = INT_MAX;
    s->brs[c].x = s->brs[c].y = INT_MIN;
    s->contours[c].n = 0;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x)
            if (LBL(s, y, x) == l) {
                if (x < s->tls[c].x) s->tls[c].x = x;
                if (y < s->tls[c].y) s->tls[c].y = y;
                if (x + 1 > s->brs[c].x) s->brs[c].x = x + 1;
                if (y + 1 > s->brs[c].y) s->brs[c].y = y + 1;
                if (is_contour(s, y, x, l)) pv_push(&s->contours[c], x, y);
            }
}

static int resolve_conflicts(dp_t *s, const float *im1, const float *im2, int tl1x, int tl1y, int tl2x, int tl2y, uint8_t *m1, uint8_t *m2)
{
    const int n = s->ncomps;
    int rc = 0;
    ptvec_t seam = {0, 0, 0};
    for (;;) {
        int c1 = 0, c2 = 0, conflict = 0;
        /* std::set<pair<int,int>>: lexicographic order */
        for (int a = 0; a < n && !conflict; ++a)
            for (int b = 0; b < n; ++b)
                if (s->edges[(size_t)a * n + b] && (s->states[a] & ST_INTERS) && (s->states[a] & ~ST_INTERS) != s->states[b]) { c1 = a; c2 = b; conflict = 1; break; }
        if (!conflict) break;
        const int l1 = c1 + 1, l2 = c2 + 1;
        if (has_only_one_neighbor(s, c1)) {
            for (int y = s->tls[c1].y; y < s->brs[c1].y; ++y)
                for (int x = s->tls[c1].x; x < s->brs[c1].x; ++x)
                    if (LBL(s, y, x) == l1) LBL(s, y, x) = l2;
            s->states[c1] = s->states[c2] == ST_FIRST ? ST_SECOND : ST_FIRST;
        } else {
            pt_t p1, p2;
            if (get_seam_tips(s, c1, c2, &p1, &p2)) {
                int horiz = 0;
                const int ok = estimate_seam(s, im1, im2, tl1x, tl1y, tl2x, tl2y, c1, p1, p2, &seam, &horiz);
                if (ok < 0) { orc_set_error("DpSeamFinder: the restored seam does not join its tips"); rc = -1; break; }
                if (ok) update_labels_using_seam(s, c1, c2, &seam, horiz);
            }
            s->states[c1] = s->states[c2] == ST_FIRST ? (ST_INTERS | ST_SECOND) : (ST_INTERS | ST_FIRST);
        }
        refresh_component(s, c1);
        refresh_component(s, c2);
        s->edges[(size_t)c1 * n + c2] = s->edges[(size_t)c2 * n + c1] = 0;
    }
    free(seam.p);
    if (rc) return rc;
    /* update masks */
    const int dx1 = s->utlx - tl1x, dy1 = s->utly - tl1y, dx2 = s->utlx - tl2x, dy2 = s->utly - tl2y;
    for (int y = 0; y < s->h2; ++y)
        for (int x = 0; x < s->w2; ++x) {
            const int l = LBL(s, y - dy2, x - dx2), y1 = y - dy2 + dy1, x1 = x - dx2 + dx1;
            if (l > 0 && (s->states[l - 1] & ST_FIRST) && y1 >= 0 && y1 < s->h1 && x1 >= 0 && x1 < s->w1 && m1[(size_t)y1 * s->w1 + x1]) m2[(size_t)y * s->w2 + x] = 0;
        }
    for (int y = 0; y < s->h1; ++y)
        for (int x = 0; x < s->w1; ++x) {
            const int l = LBL(s, y - dy1, x - dx1), y2 = y - dy1 + dy2, x2 = x - dx1 + dx2;
            if (l > 0 && (s->states[l - 1] & ST_SECOND) && y2 >= 0 && y2 < s->h2 && x2 >= 0 && x2 < s->w2 && m2[(size_t)y2 * s->w2 + x2]) m1[(size_t)y * s->w1 + x] = 0;
        }
    return 0;
}

static int process_pair(int cost_func, const float *im1, const float *im2, const float *g1[2], const float *g2[2], int tl1x, int tl1y, int w1, int h1, int tl2x,
                        int tl2y, int w2, int h2, uint8_t *m1, uint8_t *m2)
{
    const int ix0 = tl1x > tl2x ? tl1x : tl2x, iy0 = tl1y > tl2y ? tl1y : tl2y;
    const int ix1 = tl1x + w1 < tl2x + w2 ? tl1x + w1 : tl2x + w2, iy1 = tl1y + h1 < tl2y + h2 ? tl1y + h1 : tl2y + h2;
    if (ix0 >= ix1 || iy0 >= iy1) return 0;
    dp_t s;
    memset(&s, 0, sizeof s);
    s.cost_func = cost_func;
    s.utlx = tl1x < tl2x ? tl1x : tl2x; s.utly = tl1y < tl2y ? tl1y : tl2y;
    s.uw = (tl1x + w1 > tl2x + w2 ? tl1x + w1 : tl2x + w2) - s.utlx;
    s.uh = (tl1y + h1 > tl2y + h2 ? tl1y + h1 : tl2y + h2) - s.utly;
    s.w1 = w1; s.h1 = h1; s.w2 = w2; s.h2 = h2;
    s.gx1 = g1[0]; s.gy1 = g1[1]; s.gx2 = g2[0]; s.gy2 = g2[1];
    const size_t un = (size_t)s.uw * s.uh;
    s.mask1 = (uint8_t *)calloc(un, 1); s.mask2 = (uint8_t *)calloc(un, 1);
    s.cont1 = (uint8_t *)calloc(un, 1); s.cont2 = (uint8_t *)calloc(un, 1);
    s.labels = (int *)malloc(sizeof(int) * un);
    for (int y = 0; y < h1; ++y) memcpy(s.mask1 + (size_t)(y + tl1y - s.utly) * s.uw + (tl1x - s.utlx), m1 + (size_t)y * w1, (size_t)w1);
    for (int y = 0; y < h2; ++y) memcpy(s.mask2 + (size_t)(y + tl2y - s.utly) * s.uw + (tl2x - s.utlx), m2 + (size_t)y * w2, (size_t)w2);
    for (int y = 0; y < s.uh; ++y)
        for (int x = 0; x < s.uw; ++x) {
            const size_t o = (size_t)y * s.uw + x;
            const uint8_t *mm[2] = {s.mask1, s.mask2};
            uint8_t *cc[2] = {s.cont1, s.cont2};
            for (int k = 0; k < 2; ++k)
                if (mm[k][o] && ((x == 0 || !mm[k][o - 1]) || (x == s.uw - 1 || !mm[k][o + 1]) || (y == 0 || !mm[k][o - s.uw]) || (y == s.uh - 1 || !mm[k][o + s.uw])))
                    cc[k][o] = 255;
        }
    find_components(&s);
    find_edges(&s);
    const int rc = resolve_conflicts(&s, im1, im2, tl1x, tl1y, tl2x, tl2y, m1, m2);
    for (int i = 0; i < s.ncomps; ++i) free(s.contours[i].p);
    free(s.contours); free(s.states); free(s.tls); free(s.brs); free(s.edges);
    free(s.mask1); free(s.mask2); free(s.cont1); free(s.cont2); free(s.labels);
    return rc;
}

/* computeGradients: cvtColor(BGR2GRAY) on float (0.114 B + 0.587 G + 0.299 R, summed left to right [CV-U]) and Sobel(CV_32F, ksize 3,
 * BORDER_REFLECT_101) as sepFilter2D runs it: the row filter first, then the column filter; (1 2 1) is a + b*2 + c, (-1 0 1) is c - a. */
void orc_seam_dp_gradients(const float *img, int w, int h, float *gx, float *gy)
{
    float *gray = (float *)malloc(sizeof(float) * (size_t)w * h), *rd = (float *)malloc(sizeof(float) * (size_t)w * h), *rs = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) gray[i] = img[3 * i] * 0.114f + img[3 * i + 1] * 0.587f + img[3 * i + 2] * 0.299f;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xp = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);
            const float a = gray[(size_t)y * w + xm], b = gray[(size_t)y * w + x], c = gray[(size_t)y * w + xp];
            rd[(size_t)y * w + x] = c - a;
            rs[(size_t)y * w + x] = a + b * 2 + c;
        }
    for (int y = 0; y < h; ++y) {
        const int ym = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yp = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
        for (int x = 0; x < w; ++x) {
            gx[(size_t)y * w + x] = rd[(size_t)ym * w + x] + rd[(size_t)y * w + x] * 2 + rd[(size_t)yp * w + x];
            gy[(size_t)y * w + x] = rs[(size_t)yp * w + x] - rs[(size_t)ym * w + x];
        }
    }
    free(gray); free(rd); free(rs);
}

/* DpSeamFinder::find.  images: float32 BGR (sde.py:1601-1604), sizes (w, h) = mask sizes, masks cut in place.
 * order_out (optional, 2 * n(n-1)/2 ints): the pairs in the order they were processed. */
int orc_seam_dp(int n, const int *corners, const int *sizes, const float *const *images, uint8_t *const *masks, int cost_func, int *order_out)
{
    if (n <= 0) return 0;
    const int np = n * (n - 1) / 2;
    pair_t *pairs = (pair_t *)malloc(sizeof(pair_t) * (size_t)(np > 0 ? np : 1));
    int k = 0;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) { pairs[k].a = i; pairs[k].b = j; ++k; }
    pairless_t cmp = {corners, sizes};
    gnu_sort(pairs, pairs + np, &cmp);
    for (int i = 0, j = np - 1; i < j; ++i, --j) pair_swap(&pairs[i], &pairs[j]);
    float **gx = (float **)calloc((size_t)n, sizeof(float *)), **gy = (float **)calloc((size_t)n, sizeof(float *));
    if (cost_func)
        for (int i = 0; i < n; ++i) {
            const size_t px = (size_t)sizes[2 * i] * sizes[2 * i + 1];
            gx[i] = (float *)malloc(sizeof(float) * (px ? px : 1));
            gy[i] = (float *)malloc(sizeof(float) * (px ? px : 1));
            orc_seam_dp_gradients(images[i], sizes[2 * i], sizes[2 * i + 1], gx[i], gy[i]);
        }
    int rc = 0;
    for (int q = 0; q < np && !rc; ++q) {
        const int a = pairs[q].a, b = pairs[q].b;
        if (order_out) { order_out[2 * q] = a; order_out[2 * q + 1] = b; }
        const float *g1[2] = {gx[a], gy[a]}, *g2[2] = {gx[b], gy[b]};
        rc = process_pair(cost_func, images[a], images[b], g1, g2, corners[2 * a], corners[2 * a + 1], sizes[2 * a], sizes[2 * a + 1], corners[2 * b], corners[2 * b + 1],
                          sizes[2 * b], sizes[2 * b + 1], masks[a], masks[b]);
    }
    for (int i = 0; i < n; ++i) { free(gx[i]); free(gy[i]); }
    free(gx); free(gy); free(pairs);
    return rc;
}
