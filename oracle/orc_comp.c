/*
 * orc_comp.c -- CPU ORACLE (test infrastructure): exposure compensators.
 *
 * Restates OpenCV 4.6.0 stitching/exposure_compensate.cpp as reached from
 * stitching_detailed_enhanced.py:
 *   :649-665  get_compensator(): createDefault(type) / ChannelsCompensator / BlocksChannelsCompensator
 *   :1613     compensator.feed(corners, images_warped (seam scale, u8c3), masks_warped)
 *   :1754     compensator.apply(idx, corner, image_warped, mask_warped)   (in place)
 * (SURVEY.md 8(a) rows C1/C2, Appendix A.5).
 */
#include "orc_internal.h"

typedef struct {
    const uint8_t *img; /* first pixel of the view */
    const uint8_t *mask;
    int w, h, cn;
    size_t istep, mstep; /* bytes per row */
    int cx, cy;          /* corner */
} view_t;

struct orc_comp {
    int type, bl_w, bl_h, nr_feeds, nr_filter;
    int n;
    double *gains; /* n (GAIN) or n*3 (CHANNELS) */
    int *gm_w, *gm_h, gm_cn;
    float **gmap;
};

orc_comp *orc_comp_create(int type, int bl_w, int bl_h, int nr_feeds, int nr_filter)
{
    orc_comp *c = (orc_comp *)calloc(1, sizeof *c);
    c->type = type;
    c->bl_w = bl_w > 0 ? bl_w : 32;
    c->bl_h = bl_h > 0 ? bl_h : 32;
    c->nr_feeds = nr_feeds > 0 ? nr_feeds : 1;
    c->nr_filter = nr_filter >= 0 ? nr_filter : 2;
    return c;
}
static void comp_clear(orc_comp *c)
{
    free(c->gains);
    c->gains = NULL;
    if (c->gmap) for (int i = 0; i < c->n; ++i) free(c->gmap[i]);
    free(c->gmap); free(c->gm_w); free(c->gm_h);
    c->gmap = NULL; c->gm_w = NULL; c->gm_h = NULL;
    c->n = 0;
}
void orc_comp_destroy(orc_comp *c) { if (c) { comp_clear(c); free(c); } }
int orc_comp_num_images(const orc_comp *c) { return c->n; }

/* cv::solve(A, b, x, DECOMP_LU) for doubles: hal::LU64f (partial pivoting), eps = DBL_EPSILON*100.
 * OPEN ITEM [CV-U]: this is the branch GainCompensator::singleFeed takes when OpenCV is built WITHOUT Eigen.  A build with
 * HAVE_EIGEN maps A and b into Eigen float matrices and solves with a single-precision LLT (Cholesky) instead; the gains then
 * differ from the double LU ones in about the 7th digit, i.e. at most 1 grey level after apply()'s saturating multiply (a
 * product within 1e-5 of x.5).  Which branch the reference's pinned opencv-python 4.6.0.66 wheel was built with cannot be read
 * from /root/reference and no cv2 is installed here; the recorded runs leave exposure compensation at its no-op setting in their
 * compose step, so they do not tell either.  tests/test_oracle_pixels.py (_gain_solve_precision_bound) pins the <= 1 LSB bound
 * by running the same system through a float32 Cholesky. */
static int lu_solve(double *A, double *b, int m)
{
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (fabs(A[(size_t)j * m + i]) > fabs(A[(size_t)k * m + i])) k = j;
        if (fabs(A[(size_t)k * m + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; ++j) { double t = A[(size_t)i * m + j]; A[(size_t)i * m + j] = A[(size_t)k * m + j]; A[(size_t)k * m + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[(size_t)i * m + i];
        for (int j = i + 1; j < m; ++j) {
            double alpha = A[(size_t)j * m + i] * d;
            for (int q = i + 1; q < m; ++q) A[(size_t)j * m + q] += alpha * A[(size_t)i * m + q];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; --i) {
        double s = b[i];
        for (int q = i + 1; q < m; ++q) s -= A[(size_t)i * m + q] * b[q];
        b[i] = s / A[(size_t)i * m + i];
    }
    return 1;
}

/* GainCompensator::singleFeed over a list of views; gains[n] out */
static void gain_single_feed(const view_t *v, int n, double *gains)
{
    int *N = (int *)calloc((size_t)n * n, sizeof(int));
    double *I = (double *)calloc((size_t)n * n, sizeof(double));
    uint8_t *skip = (uint8_t *)malloc(n);
    memset(skip, 1, n);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            int x_tl = v[i].cx > v[j].cx ? v[i].cx : v[j].cx, y_tl = v[i].cy > v[j].cy ? v[i].cy : v[j].cy;
            int x_br = v[i].cx + v[i].w < v[j].cx + v[j].w ? v[i].cx + v[i].w : v[j].cx + v[j].w;
            int y_br = v[i].cy + v[i].h < v[j].cy + v[j].h ? v[i].cy + v[i].h : v[j].cy + v[j].h;
            if (!(x_tl < x_br && y_tl < y_br)) continue;
            int rw = x_br - x_tl, rh = y_br - y_tl;
            int cnt = 0;
            double Isum1 = 0, Isum2 = 0;
            for (int y = 0; y < rh; ++y) {
                const uint8_t *m1 = v[i].mask + (size_t)(y_tl - v[i].cy + y) * v[i].mstep + (x_tl - v[i].cx);
                const uint8_t *m2 = v[j].mask + (size_t)(y_tl - v[j].cy + y) * v[j].mstep + (x_tl - v[j].cx);
                const uint8_t *r1 = v[i].img + (size_t)(y_tl - v[i].cy + y) * v[i].istep + (size_t)(x_tl - v[i].cx) * v[i].cn;
                const uint8_t *r2 = v[j].img + (size_t)(y_tl - v[j].cy + y) * v[j].istep + (size_t)(x_tl - v[j].cx) * v[j].cn;
                for (int x = 0; x < rw; ++x) {
                    if (m1[x] == 255 && m2[x] == 255) {
                        ++cnt;
                        if (v[i].cn == 3) {
                            const uint8_t *p = r1 + x * 3, *q = r2 + x * 3;
                            Isum1 += sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
                            Isum2 += sqrt((double)q[0] * q[0] + (double)q[1] * q[1] + (double)q[2] * q[2]);
                        } else {
                            Isum1 += r1[x];
                            Isum2 += r2[x];
                        }
                    }
                }
            }
            cnt = cnt > 1 ? cnt : 1;
            N[(size_t)i * n + j] = N[(size_t)j * n + i] = cnt;
            if (i != j) { skip[i] = 0; skip[j] = 0; }
            I[(size_t)i * n + j] = Isum1 / cnt;
            I[(size_t)j * n + i] = Isum2 / cnt;
        }
    const double alpha = 0.01, beta = 100;
    int num_eq = 0;
    for (int i = 0; i < n; ++i) { gains[i] = 1.0; if (!skip[i]) ++num_eq; }
    if (num_eq > 0) {
        double *A = (double *)calloc((size_t)num_eq * num_eq, sizeof(double)), *b = (double *)calloc(num_eq, sizeof(double));
        for (int i = 0, ki = 0; i < n; ++i) {
            if (skip[i]) continue;
            for (int j = 0, kj = 0; j < n; ++j) {
                if (skip[j]) continue;
                int Nij = N[(size_t)i * n + j];
                b[ki] += beta * Nij;
                A[(size_t)ki * num_eq + ki] += beta * Nij;
                if (j != i) {
                    A[(size_t)ki * num_eq + ki] += 2 * alpha * I[(size_t)i * n + j] * I[(size_t)i * n + j] * Nij;
                    A[(size_t)ki * num_eq + kj] -= 2 * alpha * I[(size_t)i * n + j] * I[(size_t)j * n + i] * Nij;
                }
                ++kj;
            }
            ++ki;
        }
        if (!lu_solve(A, b, num_eq)) memset(b, 0, sizeof(double) * num_eq); /* cv::solve zeroes x when singular */
        for (int i = 0, j = 0; i < n; ++i)
            if (!skip[i]) gains[i] = b[j++];
        free(A); free(b);
    }
    free(N); free(I); free(skip);
}

/* cv::multiply(u8 image, double scalar): work type float, saturate_cast<uchar>(cvRound) */
static void scale_u8(uint8_t *p, size_t step, int w, int h, int cn, const float *g /* per channel */)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < cn; ++c) {
                uint8_t *q = p + (size_t)y * step + (size_t)x * cn + c;
                *q = orc_sat_u8(orc_cv_round((float)*q * g[c]));
            }
}

/* GainCompensator::feed with nr_feeds iterations; views' pixels are private copies when nr_feeds > 1 */
static void gain_feed(view_t *v, int n, int nr_feeds, double *gains)
{
    double *acc = (double *)malloc(sizeof(double) * n);
    for (int it = 0; it < nr_feeds; ++it) {
        if (it > 0)
            for (int i = 0; i < n; ++i) {
                float g[3] = {(float)gains[i], (float)gains[i], (float)gains[i]};
                scale_u8((uint8_t *)v[i].img, v[i].istep, v[i].w, v[i].h, v[i].cn, g);
            }
        gain_single_feed(v, n, gains);
        for (int i = 0; i < n; ++i) acc[i] = it == 0 ? gains[i] : acc[i] * gains[i];
    }
    memcpy(gains, acc, sizeof(double) * n);
    free(acc);
}

/* ChannelsCompensator::feed: split BGR, GainCompensator per channel -> gains[n*3] */
static void channels_feed(const view_t *v, int n, int nr_feeds, double *gains3)
{
    double *g = (double *)malloc(sizeof(double) * n);
    view_t *cv = (view_t *)malloc(sizeof(view_t) * n);
    uint8_t **planes = (uint8_t **)malloc(sizeof(uint8_t *) * n);
    for (int i = 0; i < n; ++i) planes[i] = (uint8_t *)malloc((size_t)v[i].w * v[i].h);
    for (int c = 0; c < 3; ++c) {
        for (int i = 0; i < n; ++i) {
            for (int y = 0; y < v[i].h; ++y)
                for (int x = 0; x < v[i].w; ++x) planes[i][(size_t)y * v[i].w + x] = v[i].img[(size_t)y * v[i].istep + (size_t)x * 3 + c];
            cv[i] = v[i];
            cv[i].img = planes[i];
            cv[i].cn = 1;
            cv[i].istep = (size_t)v[i].w;
        }
        gain_feed(cv, n, nr_feeds, g);
        for (int i = 0; i < n; ++i) gains3[(size_t)i * 3 + c] = g[i];
    }
    for (int i = 0; i < n; ++i) free(planes[i]);
    free(planes); free(cv); free(g);
}

/* sepFilter2D(gain_map, CV_32F, [.25 .5 .25], [.25 .5 .25]) with BORDER_REFLECT_101 */
static void filter_gain_map(float *m, int w, int h, int cn)
{
    float *t = (float *)malloc((size_t)w * h * cn * sizeof(float));
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int xl = orc_border(x - 1, w, ORC_BORDER_REFLECT_101), xr = orc_border(x + 1, w, ORC_BORDER_REFLECT_101);
            for (int c = 0; c < cn; ++c)
                t[((size_t)y * w + x) * cn + c] =
                    m[((size_t)y * w + x) * cn + c] * 0.5f + (m[((size_t)y * w + xl) * cn + c] + m[((size_t)y * w + xr) * cn + c]) * 0.25f;
        }
    for (int y = 0; y < h; ++y) {
        int yu = orc_border(y - 1, h, ORC_BORDER_REFLECT_101), yd = orc_border(y + 1, h, ORC_BORDER_REFLECT_101);
        for (int x = 0; x < w * cn; ++x)
            m[(size_t)y * w * cn + x] = t[(size_t)y * w * cn + x] * 0.5f + (t[(size_t)yu * w * cn + x] + t[(size_t)yd * w * cn + x]) * 0.25f;
    }
    free(t);
}

int orc_comp_feed(orc_comp *c, int n, const int *corners, const int *sizes, const uint8_t *const *images,
                  const uint8_t *const *masks)
{
    comp_clear(c);
    c->n = n;
    if (c->type == ORC_COMP_NO || n == 0) return 0;
    const int blocks = c->type == ORC_COMP_GAIN_BLOCKS || c->type == ORC_COMP_CHANNELS_BLOCKS;
    const int channels = c->type == ORC_COMP_CHANNELS || c->type == ORC_COMP_CHANNELS_BLOCKS;

    /* private copies of the images: feed must not modify the caller's arrays (the Python binding
     * copies ndarray -> UMat), but nr_feeds > 1 applies gains in place between feeds */
    uint8_t **imgs = (uint8_t **)malloc(sizeof(uint8_t *) * n);
    for (int i = 0; i < n; ++i) {
        size_t bytes = (size_t)sizes[2 * i] * sizes[2 * i + 1] * 3;
        imgs[i] = (uint8_t *)malloc(bytes);
        memcpy(imgs[i], images[i], bytes);
    }

    int nv = 0;
    int *blw = (int *)calloc(n, sizeof(int)), *blh = (int *)calloc(n, sizeof(int));
    if (blocks)
        for (int i = 0; i < n; ++i) {
            blw[i] = (sizes[2 * i] + c->bl_w - 1) / c->bl_w;
            blh[i] = (sizes[2 * i + 1] + c->bl_h - 1) / c->bl_h;
            nv += blw[i] * blh[i];
        }
    else
        nv = n;
    view_t *v = (view_t *)malloc(sizeof(view_t) * (size_t)nv);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        int W = sizes[2 * i], H = sizes[2 * i + 1];
        if (!blocks) {
            v[k].img = imgs[i]; v[k].mask = masks[i]; v[k].w = W; v[k].h = H; v[k].cn = 3;
            v[k].istep = (size_t)W * 3; v[k].mstep = (size_t)W; v[k].cx = corners[2 * i]; v[k].cy = corners[2 * i + 1];
            ++k;
            continue;
        }
        int bw = (W + blw[i] - 1) / blw[i], bh = (H + blh[i] - 1) / blh[i];
        for (int by = 0; by < blh[i]; ++by)
            for (int bx = 0; bx < blw[i]; ++bx) {
                int tx = bx * bw, ty = by * bh;
                int brx = tx + bw < W ? tx + bw : W, bry = ty + bh < H ? ty + bh : H;
                v[k].img = imgs[i] + ((size_t)ty * W + tx) * 3;
                v[k].mask = masks[i] + (size_t)ty * W + tx;
                v[k].w = brx - tx; v[k].h = bry - ty; v[k].cn = 3;
                v[k].istep = (size_t)W * 3; v[k].mstep = (size_t)W;
                v[k].cx = corners[2 * i] + tx; v[k].cy = corners[2 * i + 1] + ty;
                ++k;
            }
    }
    const int gcn = channels ? 3 : 1;
    double *g = (double *)malloc(sizeof(double) * (size_t)nv * gcn);
    if (channels) channels_feed(v, nv, c->nr_feeds, g);
    else gain_feed(v, nv, c->nr_feeds, g);

    if (!blocks) {
        c->gains = g;
    } else {
        c->gm_cn = gcn;
        c->gmap = (float **)calloc(n, sizeof(float *));
        c->gm_w = (int *)malloc(sizeof(int) * n);
        c->gm_h = (int *)malloc(sizeof(int) * n);
        int idx = 0;
        for (int i = 0; i < n; ++i) {
            int gw = blw[i], gh = blh[i];
            c->gm_w[i] = gw; c->gm_h[i] = gh;
            c->gmap[i] = (float *)malloc((size_t)gw * gh * gcn * sizeof(float));
            for (int q = 0; q < gw * gh; ++q, ++idx)
                for (int ch = 0; ch < gcn; ++ch) c->gmap[i][(size_t)q * gcn + ch] = (float)g[(size_t)idx * gcn + ch];
            for (int it = 0; it < c->nr_filter; ++it) filter_gain_map(c->gmap[i], gw, gh, gcn);
        }
        free(g);
    }
    for (int i = 0; i < n; ++i) free(imgs[i]);
    free(imgs); free(v); free(blw); free(blh);
    return 0;
}

int orc_comp_apply(orc_comp *c, int index, uint8_t *image, int w, int h)
{
    if (c->type == ORC_COMP_NO) return 0;
    if (index < 0 || index >= c->n) { orc_set_error("apply: index %d out of range", index); return -1; }
    if (c->type == ORC_COMP_GAIN) {
        float g[3] = {(float)c->gains[index], (float)c->gains[index], (float)c->gains[index]};
        scale_u8(image, (size_t)w * 3, w, h, 3, g);
        return 0;
    }
    if (c->type == ORC_COMP_CHANNELS) {
        float g[3] = {(float)c->gains[index * 3], (float)c->gains[index * 3 + 1], (float)c->gains[index * 3 + 2]};
        scale_u8(image, (size_t)w * 3, w, h, 3, g);
        return 0;
    }
    /* BlocksCompensator::apply: resize(gain_map, image size, INTER_LINEAR) unless equal; multiply -> u8 */
    int gw = c->gm_w[index], gh = c->gm_h[index], gcn = c->gm_cn;
    const float *gm = c->gmap[index];
    float *up = NULL;
    if (gw != w || gh != h) {
        up = (float *)malloc((size_t)w * h * gcn * sizeof(float));
        orc_resize_linear_f32(gm, gw, gh, gcn, up, w, h);
        gm = up;
    }
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int ch = 0; ch < 3; ++ch) {
            float gv = gcn == 3 ? gm[i * 3 + ch] : gm[i];
            image[i * 3 + ch] = orc_sat_u8(orc_cv_round((float)image[i * 3 + ch] * gv));
        }
    free(up);
    return 0;
}

int orc_comp_gains(const orc_comp *c, double *out)
{
    if (!c->gains) return -1;
    int m = (c->type == ORC_COMP_CHANNELS) ? c->n * 3 : c->n;
    memcpy(out, c->gains, sizeof(double) * m);
    return m;
}
int orc_comp_gain_map_size(const orc_comp *c, int index, int *w, int *h, int *cn)
{
    if (!c->gmap || index < 0 || index >= c->n) return -1;
    *w = c->gm_w[index]; *h = c->gm_h[index]; *cn = c->gm_cn;
    return 0;
}
int orc_comp_gain_map(const orc_comp *c, int index, float *out)
{
    if (!c->gmap || index < 0 || index >= c->n) return -1;
    memcpy(out, c->gmap[index], sizeof(float) * (size_t)c->gm_w[index] * c->gm_h[index] * c->gm_cn);
    return 0;
}
