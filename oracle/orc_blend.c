/*
 * orc_blend.c -- CPU ORACLE (test infrastructure): pyramids, blenders, mask helpers.
 *
 * Restates OpenCV 4.6.0 imgproc/pyramids.cpp, stitching/blenders.cpp, imgproc/distransform.cpp,
 * morph (3x3 dilate) and resize (LINEAR_EXACT / LINEAR f32 / AREA) as reached from
 * stitching_detailed_enhanced.py:
 *   :1806-1820  Blender_createDefault / detail_MultiBandBlender / detail_FeatherBlender / prepare
 *   :1886       blender.feed(int16 image, mask, corner)
 *   :1930       blender.blend()
 *   :1760-1772  cv.dilate, cv.resize(INTER_LINEAR_EXACT), cv.bitwise_and
 *   :1701       cv.resize(INTER_AREA)
 * (SURVEY.md 8(a) rows G1, B1-B6, Appendix A.3/A.4/A.6).
 */
#include "orc_internal.h"
#include <float.h>

/* ================================ pyramids ================================ */
/* pyrDown: 5-tap [1 4 6 4 1] both axes, BORDER_REFLECT_101, dst ((n+1)/2). */
#define PYR_DOWN_IMPL(NAME, T, WT, CAST)                                                                  \
    void NAME(const T *src, int w, int h, int cn, T *dst)                                                 \
    {                                                                                                     \
        int dw = (w + 1) / 2, dh = (h + 1) / 2;                                                           \
        ORC_PAR                                                                                           \
        {                                                                                                 \
        WT *rows = (WT *)malloc((size_t)5 * dw * cn * sizeof(WT));                                        \
        ORC_FOR                                                                                           \
        for (int y = 0; y < dh; ++y) {                                                                    \
            for (int k = 0; k < 5; ++k) {                                                                 \
                int sy = orc_border(2 * y - 2 + k, h, ORC_BORDER_REFLECT_101);                            \
                const T *s = src + (size_t)sy * w * cn;                                                   \
                WT *row = rows + (size_t)k * dw * cn;                                                     \
                for (int x = 0; x < dw; ++x) {                                                            \
                    int x0 = orc_border(2 * x - 2, w, ORC_BORDER_REFLECT_101) * cn;                       \
                    int x1 = orc_border(2 * x - 1, w, ORC_BORDER_REFLECT_101) * cn;                       \
                    int x2 = orc_border(2 * x, w, ORC_BORDER_REFLECT_101) * cn;                           \
                    int x3 = orc_border(2 * x + 1, w, ORC_BORDER_REFLECT_101) * cn;                       \
                    int x4 = orc_border(2 * x + 2, w, ORC_BORDER_REFLECT_101) * cn;                       \
                    for (int c = 0; c < cn; ++c)                                                          \
                        row[x * cn + c] = s[x2 + c] * 6 + (s[x1 + c] + s[x3 + c]) * 4 + s[x0 + c] + s[x4 + c]; \
                }                                                                                         \
            }                                                                                             \
            const WT *r0 = rows, *r1 = rows + (size_t)dw * cn, *r2 = r1 + (size_t)dw * cn,                \
                     *r3 = r2 + (size_t)dw * cn, *r4 = r3 + (size_t)dw * cn;                              \
            T *d = dst + (size_t)y * dw * cn;                                                             \
            for (int x = 0; x < dw * cn; ++x) d[x] = CAST(r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x]); \
        }                                                                                                 \
        free(rows);                                                                                       \
        }                                                                                                 \
    }
#define CAST_S16_8(v) ((int16_t)(((v) + 128) >> 8))
#define CAST_F32_8(v) ((v) * (1.f / 256))
PYR_DOWN_IMPL(orc_pyr_down_s16, int16_t, int, CAST_S16_8)
PYR_DOWN_IMPL(orc_pyr_down_f32, float, float, CAST_F32_8)

/* pyrUp: x2 with even taps [1 6 1], odd taps [4 4] per axis; index -1 -> 1 (reflect-101),
 * index n -> n-1 (replicate); extra odd row/column copies as in pyramids.cpp */
#define PYR_UP_IMPL(NAME, T, WT, CAST)                                                                    \
    void NAME(const T *src, int w, int h, int cn, T *dst, int dw, int dh)                                 \
    {                                                                                                     \
        size_t rl = (size_t)(dw + 1) * cn;                                                                \
        ORC_PAR                                                                                           \
        {                                                                                                 \
        WT *buf = (WT *)malloc(3 * rl * sizeof(WT));                                                      \
        ORC_FOR                                                                                           \
        for (int y = 0; y < h; ++y) {                                                                     \
            WT *rr[3];                                                                                    \
            for (int k = 0; k < 3; ++k) {                                                                 \
                int sy = y - 1 + k;                                                                       \
                int _sy = orc_border(sy * 2, h * 2, ORC_BORDER_REFLECT_101) / 2;                          \
                const T *s = src + (size_t)_sy * w * cn;                                                  \
                WT *row = buf + k * rl;                                                                   \
                rr[k] = row;                                                                              \
                if (w == 1) {                                                                             \
                    for (int c = 0; c < cn; ++c) row[c] = row[c + cn] = s[c] * 8;                         \
                    continue;                                                                             \
                }                                                                                         \
                for (int c = 0; c < cn; ++c) {                                                            \
                    row[c] = s[c] * 6 + s[c + cn] * 2;                                                    \
                    row[c + cn] = (s[c] + s[c + cn]) * 4;                                                 \
                    int sx = (w - 1) * cn + c, dx = (w - 1) * 2 * cn + c;                                 \
                    row[dx] = s[sx - cn] + s[sx] * 7;                                                     \
                    row[dx + cn] = s[sx] * 8;                                                             \
                    if (dw > w * 2) row[(dw - 1) * cn + c] = row[dx + cn];                                \
                }                                                                                         \
                for (int x = 1; x < w - 1; ++x)                                                           \
                    for (int c = 0; c < cn; ++c) {                                                        \
                        int sx = x * cn + c, dx = x * 2 * cn + c;                                         \
                        row[dx] = s[sx - cn] + s[sx] * 6 + s[sx + cn];                                    \
                        row[dx + cn] = (s[sx] + s[sx + cn]) * 4;                                          \
                    }                                                                                     \
            }                                                                                             \
            T *d0 = dst + (size_t)(y * 2) * dw * cn;                                                      \
            int y1 = y * 2 + 1 < dh - 1 ? y * 2 + 1 : dh - 1;                                             \
            T *d1 = dst + (size_t)y1 * dw * cn;                                                           \
            for (int x = 0; x < dw * cn; ++x) {                                                           \
                T t1 = CAST((rr[1][x] + rr[2][x]) * 4);                                                   \
                T t0 = CAST(rr[0][x] + rr[1][x] * 6 + rr[2][x]);                                          \
                d1[x] = t1;                                                                               \
                d0[x] = t0;                                                                               \
            }                                                                                             \
        }                                                                                                 \
        free(buf);                                                                                        \
        }                                                                                                 \
        if (dh > h * 2) {                                                                                 \
            const T *d0 = dst + (size_t)(h * 2 - 2) * dw * cn;                                            \
            T *d2 = dst + (size_t)(h * 2) * dw * cn;                                                      \
            for (int x = 0; x < dw * cn; ++x) d2[x] = d0[x];                                              \
        }                                                                                                 \
    }
#define CAST_S16_6(v) ((int16_t)(((v) + 32) >> 6))
#define CAST_F32_6(v) ((v) * (1.f / 64))
PYR_UP_IMPL(orc_pyr_up_s16, int16_t, int, CAST_S16_6)
PYR_UP_IMPL(orc_pyr_up_f32, float, float, CAST_F32_6)

/* ================================ helpers ================================ */
/* cv.dilate(mask, None): 3x3 rectangle, one iteration, outside pixels ignored */
void orc_dilate3x3_u8(const uint8_t *src, int w, int h, uint8_t *dst)
{
    ORC_PAR_FOR
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int m = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    if (src[(size_t)yy * w + xx] > m) m = src[(size_t)yy * w + xx];
                }
            dst[(size_t)y * w + x] = (uint8_t)m;
        }
}

/* cv.resize(..., INTER_LINEAR_EXACT) for 8UC1: 8.8 fixed-point coefficients, 16.16 vertical sum,
 * round half up (resize.cpp interpolationLinear + ufixedpoint16/32) */
static void lin_exact_coeffs(int ssize, int dsize, int *ofs, int *c1)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; ++d) {
        double fval = scale * ((double)d + 0.5) - 0.5;
        int ival = (int)floor(fval);
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[d] = ival;
                c1[d] = orc_cv_round_d((fval - (double)ival) * 256.0);
            } else {
                ofs[d] = ssize - 1;
                c1[d] = -1; /* right/bottom edge: copy */
            }
        } else {
            ofs[d] = 0;
            c1[d] = -1; /* left/top edge: copy */
        }
    }
}
void orc_resize_linear_exact_u8(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    int *xo = (int *)malloc(sizeof(int) * dw), *xc = (int *)malloc(sizeof(int) * dw);
    int *yo = (int *)malloc(sizeof(int) * dh), *yc = (int *)malloc(sizeof(int) * dh);
    lin_exact_coeffs(sw, dw, xo, xc);
    lin_exact_coeffs(sh, dh, yo, yc);
    ORC_PAR_FOR
    for (int y = 0; y < dh; ++y) {
        const uint8_t *r0 = src + (size_t)yo[y] * sw;
        const uint8_t *r1 = yc[y] >= 0 ? r0 + sw : r0;
        uint32_t cy1 = yc[y] >= 0 ? (uint32_t)yc[y] : 0, cy0 = 256 - cy1;
        for (int x = 0; x < dw; ++x) {
            uint32_t h0, h1;
            if (xc[x] >= 0) {
                uint32_t cx1 = (uint32_t)xc[x], cx0 = 256 - cx1;
                h0 = r0[xo[x]] * cx0 + r0[xo[x] + 1] * cx1;
                h1 = r1[xo[x]] * cx0 + r1[xo[x] + 1] * cx1;
            } else {
                h0 = (uint32_t)r0[xo[x]] << 8;
                h1 = (uint32_t)r1[xo[x]] << 8;
            }
            uint32_t v = h0 * cy0 + h1 * cy1;
            dst[(size_t)y * dw + x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
    free(xo); free(xc); free(yo); free(yc);
}

/* cv.resize(f32, INTER_LINEAR): pixel-centre mapping, float weights, horizontal then vertical pass */
static void lin_f32_coeffs(int ssize, int dsize, int *ofs, float *a1)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        a1[d] = f;
    }
}
void orc_resize_linear_f32(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    int *xo = (int *)malloc(sizeof(int) * dw), *yo = (int *)malloc(sizeof(int) * dh);
    float *xa = (float *)malloc(sizeof(float) * dw), *ya = (float *)malloc(sizeof(float) * dh);
    lin_f32_coeffs(sw, dw, xo, xa);
    lin_f32_coeffs(sh, dh, yo, ya);
    ORC_PAR_FOR
    for (int y = 0; y < dh; ++y) {
        int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        float b1 = ya[y], b0 = 1.f - b1;
        for (int x = 0; x < dw; ++x) {
            int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            float a1 = xa[x], a0 = 1.f - a1;
            for (int c = 0; c < cn; ++c) {
                float t0 = src[((size_t)y0 * sw + x0) * cn + c] * a0 + src[((size_t)y0 * sw + x1) * cn + c] * a1;
                float t1 = src[((size_t)y1 * sw + x0) * cn + c] * a0 + src[((size_t)y1 * sw + x1) * cn + c] * a1;
                dst[((size_t)y * dw + x) * cn + c] = t0 * b0 + t1 * b1;
            }
        }
    }
    free(xo); free(yo); free(xa); free(ya);
}

/* cv.resize(u8, fx=fy<1, INTER_AREA): fractional-coverage box filter (resizeArea_), float accumulate,
 * saturate_cast<uchar>(sum) = round half to even.  [next-row (f).1, sde.py:1701] */
typedef struct { int si, di; float alpha; } area_tab_t;
static int area_tab(int ssize, int dsize, double scale, area_tab_t *tab)
{
    int k = 0;
    for (int dx = 0; dx < dsize; ++dx) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell); }
        for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell); }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx2;
            double a = fsx2 - sx2; if (a > 1.0) a = 1.0; if (a > cell) a = cell;
            tab[k++].alpha = (float)(a / cell);
        }
    }
    return k;
}
/* cv.resize(src, None, fx=fx, fy=fy, interpolation=INTER_AREA) for decimation (sde.py:1701-1707): dsize = cvRound(size*f),
 * scale = 1/f (the caller's factor, not the size ratio), fractional-coverage tables, float accumulation in table order,
 * saturate_cast<uchar> (round half to even).  lut (256 entries) is applied to the result when not NULL: that is
 * adjust_black_and_white_point (image_processors.py:32-41), which the reference applies right after the resize (sde.py:1711). */
void orc_resize_area_u8_scale(const uint8_t *src, int sw, int sh, int cn, double fx, double fy, const uint8_t *lut, uint8_t *dst, int dw, int dh)
{
    double sx = 1.0 / fx, sy = 1.0 / fy;
    int isx = orc_cv_round_d(sx), isy = orc_cv_round_d(sy);
    if (fabs(sx - isx) < DBL_EPSILON && fabs(sy - isy) < DBL_EPSILON) {
        /* integer factors take resizeAreaFast_: int sums; 2x2 -> (s+2)>>2, otherwise cvRound(sum * (1.f/area)); destination
         * cells whose block sticks out of the source average the pixels that exist: cvRound((float)sum / count) */
        int area = isx * isy, wfull = sw / isx;
        float scale = 1.f / (float)area;
        for (int dy = 0; dy < dh; ++dy) {
            int sy0 = dy * isy;
            uint8_t *D = dst + (size_t)dy * dw * cn;
            int w = sy0 + isy <= sh ? wfull : 0;
            for (int dx = 0; dx < dw; ++dx)
                for (int c = 0; c < cn; ++c) {
                    int sum = 0, count = 0, sx0 = dx * isx;
                    for (int yy = 0; yy < isy && sy0 + yy < sh; ++yy)
                        for (int xx = 0; xx < isx && sx0 + xx < sw; ++xx) { sum += src[((size_t)(sy0 + yy) * sw + sx0 + xx) * cn + c]; ++count; }
                    uint8_t v;
                    if (sy0 >= sh || count == 0) v = 0;
                    else if (dx < w) v = (isx == 2 && isy == 2) ? (uint8_t)((sum + 2) >> 2) : orc_sat_u8(orc_cv_round((float)sum * scale));
                    else v = orc_sat_u8(orc_cv_round((float)sum / (float)count));
                    D[(size_t)dx * cn + c] = lut ? lut[v] : v;
                }
        }
        return;
    }
    area_tab_t *xt = (area_tab_t *)malloc(sizeof(area_tab_t) * (size_t)(sw * 2 + 2));
    area_tab_t *yt = (area_tab_t *)malloc(sizeof(area_tab_t) * (size_t)(sh * 2 + 2));
    int xn = area_tab(sw, dw, sx, xt), yn = area_tab(sh, dh, sy, yt);
    size_t rl = (size_t)dw * cn;
    float *buf = (float *)malloc(rl * sizeof(float)), *sum = (float *)calloc(rl, sizeof(float));
    int prev_dy = yn ? yt[0].di : 0;
    for (int j = 0; j < yn; ++j) {
        int dy = yt[j].di;
        float beta = yt[j].alpha;
        const uint8_t *S = src + (size_t)yt[j].si * sw * cn;
        for (size_t i = 0; i < rl; ++i) buf[i] = 0;
        for (int k = 0; k < xn; ++k)
            for (int c = 0; c < cn; ++c) buf[(size_t)xt[k].di * cn + c] += S[(size_t)xt[k].si * cn + c] * xt[k].alpha;
        if (dy != prev_dy) {
            uint8_t *D = dst + (size_t)prev_dy * rl;
            for (size_t i = 0; i < rl; ++i) { D[i] = orc_sat_u8(orc_cv_round(sum[i])); sum[i] = beta * buf[i]; }
            prev_dy = dy;
        } else
            for (size_t i = 0; i < rl; ++i) sum[i] += beta * buf[i];
    }
    if (yn) {
        uint8_t *D = dst + (size_t)prev_dy * rl;
        for (size_t i = 0; i < rl; ++i) D[i] = orc_sat_u8(orc_cv_round(sum[i]));
    }
    if (lut)
        for (size_t i = 0; i < (size_t)dh * rl; ++i) dst[i] = lut[dst[i]];
    free(xt); free(yt); free(buf); free(sum);
}
void orc_resize_area_u8(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh)
{
    orc_resize_area_u8_scale(src, sw, sh, cn, (double)dw / sw, (double)dh / sh, NULL, dst, dw, dh);
}

/* adjust_black_and_white_point (image_processors.py:32-41): ((clip(v, bp, wp) - bp) * (255 / (wp - bp))).astype(uint8),
 * evaluated by numpy in float64 and truncated */
void orc_bw_point_lut(int black, int white, uint8_t lut[256])
{
    double k = 255.0 / (double)(white - black);
    for (int v = 0; v < 256; ++v) {
        int c = v < black ? black : (v > white ? white : v);
        double r = (double)(c - black) * k;
        lut[v] = (uint8_t)r;
    }
}

/* distanceTransform(mask, DIST_L1, 3) -> exact city-block distance to the nearest zero pixel;
 * pixels outside the image are NOT zeros (distransform.cpp 3x3 chamfer with metrics 1/2) */
void orc_distance_l1(const uint8_t *mask, int w, int h, float *dist)
{
    const unsigned HV = 1u << 16, DIAG = 2u << 16, DMAX = 0xffffffffu - DIAG;
    size_t step = (size_t)w + 2;
    unsigned *tmp = (unsigned *)malloc(step * (size_t)(h + 2) * sizeof(unsigned));
    for (size_t i = 0; i < step * (size_t)(h + 2); ++i) tmp[i] = DMAX;
    for (int i = 0; i < h; ++i) {
        unsigned *t = tmp + (size_t)(i + 1) * step + 1;
        for (int j = 0; j < w; ++j) {
            if (!mask[(size_t)i * w + j]) { t[j] = 0; continue; }
            unsigned t0 = t[j - (long)step - 1] + DIAG, q = t[j - (long)step] + HV;
            if (t0 > q) t0 = q;
            q = t[j - (long)step + 1] + DIAG; if (t0 > q) t0 = q;
            q = t[j - 1] + HV; if (t0 > q) t0 = q;
            t[j] = t0 > DMAX ? DMAX : t0;
        }
    }
    const float scale = 1.f / (1 << 16);
    for (int i = h - 1; i >= 0; --i) {
        unsigned *t = tmp + (size_t)(i + 1) * step + 1;
        for (int j = w - 1; j >= 0; --j) {
            unsigned t0 = t[j];
            if (t0 > HV) {
                unsigned q = t[j + step + 1] + DIAG; if (t0 > q) t0 = q;
                q = t[j + step] + HV; if (t0 > q) t0 = q;
                q = t[j + step - 1] + DIAG; if (t0 > q) t0 = q;
                q = t[j + 1] + HV; if (t0 > q) t0 = q;
                t[j] = t0;
            }
            t0 = t0 > DMAX ? DMAX : t0;
            dist[(size_t)i * w + j] = (float)(t0 * scale);
        }
    }
    free(tmp);
}

/* ================================ blenders ================================ */
#define WEIGHT_EPS 1e-5f

struct orc_blender {
    int type, want_bands, num_bands, float_mode;
    float sharpness;
    int roi[4];       /* working dst roi (padded for multiband) */
    int final_roi[4]; /* as passed to prepare */
    /* Blender base / feather */
    int16_t *dst;
    float *dstf;
    uint8_t *dst_mask;
    float *dst_weight;
    /* multiband */
    int lw[16], lh[16];
    int16_t *lap[16];
    float *lapf[16];
    float *wgt[16];
};

orc_blender *orc_blender_create(int type)
{
    orc_blender *b = (orc_blender *)calloc(1, sizeof *b);
    b->type = type;
    b->want_bands = 5;      /* MultiBandBlender(try_gpu=false, num_bands=5) */
    b->sharpness = 0.02f;   /* FeatherBlender(sharpness=0.02f) */
    return b;
}
static void blender_free_state(orc_blender *b)
{
    free(b->dst); free(b->dstf); free(b->dst_mask); free(b->dst_weight);
    b->dst = NULL; b->dstf = NULL; b->dst_mask = NULL; b->dst_weight = NULL;
    for (int i = 0; i < 16; ++i) {
        /* lap[0]/lapf[0] alias dst/dstf */
        if (i > 0) { free(b->lap[i]); free(b->lapf[i]); }
        free(b->wgt[i]);
        b->lap[i] = NULL; b->lapf[i] = NULL; b->wgt[i] = NULL;
    }
}
void orc_blender_destroy(orc_blender *b) { if (b) { blender_free_state(b); free(b); } }
void orc_blender_set_num_bands(orc_blender *b, int n) { b->want_bands = n; }
int orc_blender_num_bands(const orc_blender *b) { return b->type == ORC_BLEND_MULTIBAND ? (b->dst || b->dstf ? b->num_bands : b->want_bands) : 0; }
void orc_blender_set_sharpness(orc_blender *b, float s) { b->sharpness = s; }
void orc_blender_set_float_mode(orc_blender *b, int on) { b->float_mode = on; }

int orc_blender_prepare(orc_blender *b, int x, int y, int w, int h)
{
    blender_free_state(b);
    b->final_roi[0] = x; b->final_roi[1] = y; b->final_roi[2] = w; b->final_roi[3] = h;
    if (b->type == ORC_BLEND_MULTIBAND) {
        double max_len = (double)(w > h ? w : h);
        int lim = (int)ceil(log(max_len) / log(2.0));
        b->num_bands = b->want_bands < lim ? b->want_bands : lim;
        int m = 1 << b->num_bands;
        w += (m - w % m) % m;
        h += (m - h % m) % m;
    }
    b->roi[0] = x; b->roi[1] = y; b->roi[2] = w; b->roi[3] = h;
    size_t n = (size_t)w * h;
    if (b->float_mode) b->dstf = (float *)calloc(n * 3, sizeof(float));
    else b->dst = (int16_t *)calloc(n * 3, sizeof(int16_t));
    b->dst_mask = (uint8_t *)calloc(n, 1);
    if (b->type == ORC_BLEND_FEATHER) b->dst_weight = (float *)calloc(n, sizeof(float));
    if (b->type == ORC_BLEND_MULTIBAND) {
        b->lw[0] = w; b->lh[0] = h;
        b->lap[0] = b->dst; b->lapf[0] = b->dstf;
        b->wgt[0] = (float *)calloc(n, sizeof(float));
        for (int i = 1; i <= b->num_bands; ++i) {
            b->lw[i] = (b->lw[i - 1] + 1) / 2;
            b->lh[i] = (b->lh[i - 1] + 1) / 2;
            size_t m = (size_t)b->lw[i] * b->lh[i];
            if (b->float_mode) b->lapf[i] = (float *)calloc(m * 3, sizeof(float));
            else b->lap[i] = (int16_t *)calloc(m * 3, sizeof(int16_t));
            b->wgt[i] = (float *)calloc(m, sizeof(float));
        }
    }
    return 0;
}

/* copyMakeBorder */
static void make_border_s16(const int16_t *src, int w, int h, int cn, int top, int bottom, int left, int right, int btype, int16_t *dst)
{
    int W = w + left + right, H = h + top + bottom;
    ORC_PAR_FOR
    for (int y = 0; y < H; ++y) {
        int sy = orc_border(y - top, h, btype);
        for (int x = 0; x < W; ++x) {
            int sx = orc_border(x - left, w, btype);
            for (int c = 0; c < cn; ++c)
                dst[((size_t)y * W + x) * cn + c] = (sx < 0 || sy < 0) ? 0 : src[((size_t)sy * w + sx) * cn + c];
        }
    }
}
static void make_border_f32(const float *src, int w, int h, int cn, int top, int bottom, int left, int right, int btype, float *dst)
{
    int W = w + left + right, H = h + top + bottom;
    ORC_PAR_FOR
    for (int y = 0; y < H; ++y) {
        int sy = orc_border(y - top, h, btype);
        for (int x = 0; x < W; ++x) {
            int sx = orc_border(x - left, w, btype);
            for (int c = 0; c < cn; ++c)
                dst[((size_t)y * W + x) * cn + c] = (sx < 0 || sy < 0) ? 0.f : src[((size_t)sy * w + sx) * cn + c];
        }
    }
}

static int feed_multiband(orc_blender *b, const void *img_, const uint8_t *mask, int iw, int ih, int tlx, int tly)
{
    const int nb = b->num_bands, m = 1 << nb;
    const int rx = b->roi[0], ry = b->roi[1], rbx = rx + b->roi[2], rby = ry + b->roi[3];
    int gap = 3 * (1 << nb);
    int tnx = rx > tlx - gap ? rx : tlx - gap, tny = ry > tly - gap ? ry : tly - gap;
    int bnx = rbx < tlx + iw + gap ? rbx : tlx + iw + gap, bny = rby < tly + ih + gap ? rby : tly + ih + gap;
    tnx = rx + (((tnx - rx) >> nb) << nb);
    tny = ry + (((tny - ry) >> nb) << nb);
    int width = bnx - tnx, height = bny - tny;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    bnx = tnx + width;
    bny = tny + height;
    int dy = bny - rby > 0 ? bny - rby : 0, dx = bnx - rbx > 0 ? bnx - rbx : 0;
    tnx -= dx; bnx -= dx; tny -= dy; bny -= dy;
    int top = tly - tny, left = tlx - tnx, bottom = bny - tly - ih, right = bnx - tlx - iw;
    if (top < 0 || left < 0 || bottom < 0 || right < 0) {
        orc_set_error("multiband feed: image (%d,%d %dx%d) outside the prepared roi", tlx, tly, iw, ih);
        return -1;
    }
    int W = width, H = height;
    int pw[17], ph[17];
    pw[0] = W; ph[0] = H;
    for (int i = 1; i <= nb; ++i) { pw[i] = (pw[i - 1] + 1) / 2; ph[i] = (ph[i - 1] + 1) / 2; }

    /* weight map: mask/255 (convertTo CV_32F, scale 1/255. applied in float), zero border */
    float *wsrc = (float *)malloc((size_t)iw * ih * sizeof(float));
    const float inv255 = (float)(1. / 255.);
    for (size_t i = 0; i < (size_t)iw * ih; ++i) wsrc[i] = (float)mask[i] * inv255;
    float *wp[17];
    wp[0] = (float *)malloc((size_t)W * H * sizeof(float));
    make_border_f32(wsrc, iw, ih, 1, top, bottom, left, right, ORC_BORDER_CONSTANT, wp[0]);
    free(wsrc);
    for (int i = 0; i < nb; ++i) {
        wp[i + 1] = (float *)malloc((size_t)pw[i + 1] * ph[i + 1] * sizeof(float));
        orc_pyr_down_f32(wp[i], pw[i], ph[i], 1, wp[i + 1]);
    }

    int x_tl = tnx - rx, y_tl = tny - ry, x_br = bnx - rx, y_br = bny - ry;
    if (!b->float_mode) {
        int16_t *gp[17];
        gp[0] = (int16_t *)malloc((size_t)W * H * 3 * sizeof(int16_t));
        make_border_s16((const int16_t *)img_, iw, ih, 3, top, bottom, left, right, ORC_BORDER_REFLECT, gp[0]);
        for (int i = 0; i < nb; ++i) {
            gp[i + 1] = (int16_t *)malloc((size_t)pw[i + 1] * ph[i + 1] * 3 * sizeof(int16_t));
            orc_pyr_down_s16(gp[i], pw[i], ph[i], 3, gp[i + 1]);
        }
        /* createLaplacePyr: L_i = saturate(G_i - pyrUp(G_{i+1})) */
        for (int i = 0; i < nb; ++i) {
            size_t n = (size_t)pw[i] * ph[i] * 3;
            int16_t *up = (int16_t *)malloc(n * sizeof(int16_t));
            orc_pyr_up_s16(gp[i + 1], pw[i + 1], ph[i + 1], 3, up, pw[i], ph[i]);
            ORC_PAR_FOR
            for (long long k = 0; k < (long long)n; ++k) gp[i][k] = orc_sat_s16((int)gp[i][k] - (int)up[k]);
            free(up);
        }
        for (int i = 0; i <= nb; ++i) {
            int rw = x_br - x_tl, rh = y_br - y_tl;
            ORC_PAR_FOR
            for (int y = 0; y < rh; ++y)
                for (int x = 0; x < rw; ++x) {
                    size_t di = (size_t)(y_tl + y) * b->lw[i] + (x_tl + x);
                    size_t si = (size_t)y * pw[i] + x;
                    float wv = wp[i][si];
                    for (int c = 0; c < 3; ++c) {
                        int16_t add = orc_trunc_s16((float)gp[i][si * 3 + c] * wv);
                        b->lap[i][di * 3 + c] = (int16_t)(b->lap[i][di * 3 + c] + add);
                    }
                    b->wgt[i][di] += wv;
                }
            x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
            free(gp[i]);
        }
    } else {
        /* f32 pyramid variant (BASELINE config 5): same structure, float arithmetic, no truncation */
        float *gp[17];
        gp[0] = (float *)malloc((size_t)W * H * 3 * sizeof(float));
        make_border_f32((const float *)img_, iw, ih, 3, top, bottom, left, right, ORC_BORDER_REFLECT, gp[0]);
        for (int i = 0; i < nb; ++i) {
            gp[i + 1] = (float *)malloc((size_t)pw[i + 1] * ph[i + 1] * 3 * sizeof(float));
            orc_pyr_down_f32(gp[i], pw[i], ph[i], 3, gp[i + 1]);
        }
        for (int i = 0; i < nb; ++i) {
            size_t n = (size_t)pw[i] * ph[i] * 3;
            float *up = (float *)malloc(n * sizeof(float));
            orc_pyr_up_f32(gp[i + 1], pw[i + 1], ph[i + 1], 3, up, pw[i], ph[i]);
            ORC_PAR_FOR
            for (long long k = 0; k < (long long)n; ++k) gp[i][k] = gp[i][k] - up[k];
            free(up);
        }
        for (int i = 0; i <= nb; ++i) {
            int rw = x_br - x_tl, rh = y_br - y_tl;
            ORC_PAR_FOR
            for (int y = 0; y < rh; ++y)
                for (int x = 0; x < rw; ++x) {
                    size_t di = (size_t)(y_tl + y) * b->lw[i] + (x_tl + x);
                    size_t si = (size_t)y * pw[i] + x;
                    float wv = wp[i][si];
                    for (int c = 0; c < 3; ++c) b->lapf[i][di * 3 + c] += gp[i][si * 3 + c] * wv;
                    b->wgt[i][di] += wv;
                }
            x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
            free(gp[i]);
        }
    }
    for (int i = 0; i <= nb; ++i) free(wp[i]);
    return 0;
}

int orc_blender_feed(orc_blender *b, const void *img_, const uint8_t *mask, int w, int h, int tlx, int tly)
{
    if (!b->dst && !b->dstf) { orc_set_error("feed before prepare"); return -1; }
    if (b->type == ORC_BLEND_MULTIBAND) return feed_multiband(b, img_, mask, w, h, tlx, tly);
    const int16_t *img = (const int16_t *)img_;
    int dx = tlx - b->roi[0], dy = tly - b->roi[1], W = b->roi[2];
    if (dx < 0 || dy < 0 || dx + w > b->roi[2] || dy + h > b->roi[3]) { orc_set_error("feed: image outside roi"); return -1; }
    if (b->type == ORC_BLEND_NO) {
        /* Blender::feed: where mask: dst = img; dst_mask |= mask */
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t si = (size_t)y * w + x, di = (size_t)(dy + y) * W + dx + x;
                if (mask[si]) for (int c = 0; c < 3; ++c) b->dst[di * 3 + c] = img[si * 3 + c];
                b->dst_mask[di] |= mask[si];
            }
        return 0;
    }
    /* FeatherBlender::feed: weight = min(1, sharpness * L1dist) */
    float *wm = (float *)malloc((size_t)w * h * sizeof(float));
    orc_distance_l1(mask, w, h, wm);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float t = wm[i] * b->sharpness;
        wm[i] = t > 1.f ? 1.f : t;
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            size_t si = (size_t)y * w + x, di = (size_t)(dy + y) * W + dx + x;
            for (int c = 0; c < 3; ++c) {
                int16_t add = orc_trunc_s16((float)img[si * 3 + c] * wm[si]);
                b->dst[di * 3 + c] = (int16_t)(b->dst[di * 3 + c] + add);
            }
            b->dst_weight[di] += wm[si];
        }
    free(wm);
    return 0;
}

int orc_blender_blend(orc_blender *b, void *dst_, uint8_t *dst_mask)
{
    if (!b->dst && !b->dstf) { orc_set_error("blend before prepare"); return -1; }
    const int W = b->roi[2], fw = b->final_roi[2], fh = b->final_roi[3];
    if (b->type == ORC_BLEND_FEATHER) {
        size_t n = (size_t)W * b->roi[3];
        for (size_t i = 0; i < n; ++i) {
            float d = b->dst_weight[i] + WEIGHT_EPS;
            for (int c = 0; c < 3; ++c) b->dst[i * 3 + c] = orc_trunc_s16((float)b->dst[i * 3 + c] / d);
            b->dst_mask[i] = b->dst_weight[i] > WEIGHT_EPS ? 255 : 0;
        }
    } else if (b->type == ORC_BLEND_MULTIBAND) {
        const int nb = b->num_bands;
        for (int i = 0; i <= nb; ++i) {
            size_t n = (size_t)b->lw[i] * b->lh[i];
            ORC_PAR_FOR
            for (long long k = 0; k < (long long)n; ++k) {
                float d = b->wgt[i][k] + WEIGHT_EPS;
                for (int c = 0; c < 3; ++c) {
                    if (b->float_mode) b->lapf[i][k * 3 + c] = b->lapf[i][k * 3 + c] / d;
                    else b->lap[i][k * 3 + c] = orc_trunc_s16((float)b->lap[i][k * 3 + c] / d);
                }
            }
        }
        /* restoreImageFromLaplacePyr */
        for (int i = nb; i > 0; --i) {
            size_t n = (size_t)b->lw[i - 1] * b->lh[i - 1] * 3;
            if (b->float_mode) {
                float *up = (float *)malloc(n * sizeof(float));
                orc_pyr_up_f32(b->lapf[i], b->lw[i], b->lh[i], 3, up, b->lw[i - 1], b->lh[i - 1]);
                ORC_PAR_FOR
                for (long long k = 0; k < (long long)n; ++k) b->lapf[i - 1][k] = up[k] + b->lapf[i - 1][k];
                free(up);
            } else {
                int16_t *up = (int16_t *)malloc(n * sizeof(int16_t));
                orc_pyr_up_s16(b->lap[i], b->lw[i], b->lh[i], 3, up, b->lw[i - 1], b->lh[i - 1]);
                ORC_PAR_FOR
                for (long long k = 0; k < (long long)n; ++k) b->lap[i - 1][k] = orc_sat_s16((int)up[k] + (int)b->lap[i - 1][k]);
                free(up);
            }
        }
        for (int y = 0; y < fh; ++y)
            for (int x = 0; x < fw; ++x) b->dst_mask[(size_t)y * W + x] = b->wgt[0][(size_t)y * W + x] > WEIGHT_EPS ? 255 : 0;
    }
    /* Blender::blend: dst.setTo(0, dst_mask == 0); crop to the final roi */
    ORC_PAR_FOR
    for (int y = 0; y < fh; ++y)
        for (int x = 0; x < fw; ++x) {
            size_t si = (size_t)y * W + x, di = (size_t)y * fw + x;
            uint8_t mk = b->dst_mask[si];
            dst_mask[di] = mk;
            for (int c = 0; c < 3; ++c) {
                if (b->float_mode) ((float *)dst_)[di * 3 + c] = mk ? b->dstf[si * 3 + c] : 0.f;
                else ((int16_t *)dst_)[di * 3 + c] = mk ? b->dst[si * 3 + c] : 0;
            }
        }
    blender_free_state(b); /* OpenCV releases dst_/dst_mask_: a second blend() is invalid */
    return 0;
}

int orc_blender_level_size(const orc_blender *b, int level, int *w, int *h)
{
    if (b->type != ORC_BLEND_MULTIBAND || level > b->num_bands) return -1;
    *w = b->lw[level]; *h = b->lh[level];
    return 0;
}
const int16_t *orc_blender_level_lap(const orc_blender *b, int level) { return b->lap[level]; }
const float *orc_blender_level_weight(const orc_blender *b, int level) { return b->wgt[level]; }
/* multi-GPU semantics: partial Laplacian sums travel as int32 and are added modulo 2^16, weights as f32 */
int orc_blender_add_partial(orc_blender *b, int level, const int32_t *lap, const float *wgt)
{
    if (b->type != ORC_BLEND_MULTIBAND || level > b->num_bands || b->float_mode) return -1;
    size_t n = (size_t)b->lw[level] * b->lh[level];
    for (size_t k = 0; k < n * 3; ++k) b->lap[level][k] = (int16_t)(uint16_t)((uint32_t)(uint16_t)b->lap[level][k] + (uint32_t)lap[k]);
    for (size_t k = 0; k < n; ++k) b->wgt[level][k] += wgt[k];
    return 0;
}

/* ================================ Voronoi seam finder ================================ */
/* cv.detail.SeamFinder_createDefault(SeamFinder_VORONOI_SEAM).find(images, corners, masks) (sde.py:243-249, :1618):
 * PairwiseSeamFinder::run visits every pair i < j whose rectangles overlap, in order, and VoronoiSeamFinder::findInPair
 * (stitching/src/seam_finders.cpp) splits the overlap by the L1 distance to the parts only one image covers.
 * masks are modified in place (later pairs see the earlier cuts). */
void orc_seam_voronoi(int n, const int *corners, const int *sizes, uint8_t *const *masks)
{
    const int gap = 10;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int x1 = corners[2 * i], y1 = corners[2 * i + 1], w1 = sizes[2 * i], h1 = sizes[2 * i + 1];
            const int x2 = corners[2 * j], y2 = corners[2 * j + 1], w2 = sizes[2 * j], h2 = sizes[2 * j + 1];
            const int rx = x1 > x2 ? x1 : x2, ry = y1 > y2 ? y1 : y2;
            const int rbx = x1 + w1 < x2 + w2 ? x1 + w1 : x2 + w2, rby = y1 + h1 < y2 + h2 ? y1 + h1 : y2 + h2;
            if (!(rx < rbx && ry < rby)) continue;
            const int rw = rbx - rx, rh = rby - ry, sw = rw + 2 * gap, sh = rh + 2 * gap;
            uint8_t *z1 = (uint8_t *)malloc((size_t)sw * sh), *z2 = (uint8_t *)malloc((size_t)sw * sh);
            float *d1 = (float *)malloc(sizeof(float) * (size_t)sw * sh), *d2 = (float *)malloc(sizeof(float) * (size_t)sw * sh);
            for (int y = -gap; y < rh + gap; ++y)
                for (int x = -gap; x < rw + gap; ++x) {
                    const int ya = ry - y1 + y, xa = rx - x1 + x, yb = ry - y2 + y, xb = rx - x2 + x;
                    const uint8_t s1 = (ya >= 0 && xa >= 0 && ya < h1 && xa < w1) ? masks[i][(size_t)ya * w1 + xa] : 0;
                    const uint8_t s2 = (yb >= 0 && xb >= 0 && yb < h2 && xb < w2) ? masks[j][(size_t)yb * w2 + xb] : 0;
                    const int collision = s1 && s2;
                    /* unique = submask with the collision removed; distanceTransform runs on (unique == 0) */
                    const size_t o = (size_t)(y + gap) * sw + (x + gap);
                    z1[o] = (s1 && !collision) ? 0 : 255;
                    z2[o] = (s2 && !collision) ? 0 : 255;
                }
            orc_distance_l1(z1, sw, sh, d1);
            orc_distance_l1(z2, sw, sh, d2);
            for (int y = 0; y < rh; ++y)
                for (int x = 0; x < rw; ++x) {
                    const size_t o = (size_t)(y + gap) * sw + (x + gap);
                    if (d1[o] < d2[o]) masks[j][(size_t)(ry - y2 + y) * w2 + (rx - x2 + x)] = 0;
                    else masks[i][(size_t)(ry - y1 + y) * w1 + (rx - x1 + x)] = 0;
                }
            free(z1); free(z2); free(d1); free(d2);
        }
}
