/*
 * orc_warp.c -- CPU ORACLE (test infrastructure): warpers + remap.
 *
 * Restates OpenCV 4.6.0 modules/stitching warpers.cpp / warpers_inl.hpp and imgproc remap as
 * reached from stitching_detailed_enhanced.py:
 *   :1545/:1684  cv.PyRotationWarper(type, scale)          -> orc_warper_create
 *   :1696        warper.warpRoi(sz, K, R)                  -> orc_warper_roi
 *   :1557/:1731  warper.warp(img, K, R, LINEAR|AREA, REFLECT)  -> orc_warper_warp
 *   :1591/:1740  warper.warp(mask, K, R, NEAREST, CONSTANT)    -> orc_warper_warp
 * (SURVEY.md 8(a) rows W1-W5, Appendix A.1/A.2).  All arithmetic is binary32 in OpenCV's operation
 * order, compiled with -ffp-contract=off (OpenCV's baseline x86-64 build has no FMA contraction).
 */
#include <stdarg.h>

#include "orc_internal.h"

static __thread char g_err[512];
void orc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *orc_last_error(void) { return g_err; }
int orc_uses_libm(void)
{
#ifdef SSP_ORACLE_LIBM
    return 1;
#else
    return 0;
#endif
}

enum {
    P_PLANE = 0, P_AFFINE, P_CYLINDRICAL, P_SPHERICAL, P_FISHEYE, P_STEREOGRAPHIC, P_COMPRESSED,
    P_COMPRESSED_PORTRAIT, P_PANINI, P_PANINI_PORTRAIT, P_MERCATOR, P_TRANSVERSE_MERCATOR
};

struct orc_warper {
    int type;
    float scale, a, b;
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];
};

/* PyRotationWarper::PyRotationWarper(String type, float scale)  [W1] */
orc_warper *orc_warper_create(const char *type, float scale)
{
    static const struct { const char *name; int type; float a, b; } tab[] = {
        {"plane", P_PLANE, 0, 0}, {"affine", P_AFFINE, 0, 0}, {"cylindrical", P_CYLINDRICAL, 0, 0},
        {"spherical", P_SPHERICAL, 0, 0}, {"fisheye", P_FISHEYE, 0, 0}, {"stereographic", P_STEREOGRAPHIC, 0, 0},
        {"compressedPlaneA2B1", P_COMPRESSED, 2.0f, 1.0f}, {"compressedPlaneA1.5B1", P_COMPRESSED, 1.5f, 1.0f},
        {"compressedPlanePortraitA2B1", P_COMPRESSED_PORTRAIT, 2.0f, 1.0f},
        {"compressedPlanePortraitA1.5B1", P_COMPRESSED_PORTRAIT, 1.5f, 1.0f},
        {"paniniA2B1", P_PANINI, 2.0f, 1.0f}, {"paniniA1.5B1", P_PANINI, 1.5f, 1.0f},
        {"paniniPortraitA2B1", P_PANINI_PORTRAIT, 2.0f, 1.0f}, {"paniniPortraitA1.5B1", P_PANINI_PORTRAIT, 1.5f, 1.0f},
        {"mercator", P_MERCATOR, 0, 0}, {"transverseMercator", P_TRANSVERSE_MERCATOR, 0, 0},
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; ++i) {
        if (strcmp(type, tab[i].name) == 0) {
            orc_warper *w = (orc_warper *)calloc(1, sizeof *w);
            w->type = tab[i].type;
            w->scale = scale;
            w->a = tab[i].a;
            w->b = tab[i].b;
            return w;
        }
    }
    orc_set_error("unknown warper :%s", type);
    return NULL;
}
void orc_warper_destroy(orc_warper *w) { free(w); }

/* 3x3 float product as cv::gemm's small-matrix path: ((a0*b0 + a1*b1) + a2*b2) in binary32 */
static void mat3_mul(const float *a, const float *b, float *d)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j];
            s = s + a[i * 3 + 2] * b[2 * 3 + j];
            d[i * 3 + j] = s;
        }
}
/* cv::invert for a 3x3 float matrix: cofactors and determinant in double, result cast to float */
static int mat3_inv(const float *m, float *d)
{
#define M(r, c) ((double)m[(r) * 3 + (c)])
    double det = M(0, 0) * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) - M(0, 1) * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0)) +
                 M(0, 2) * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
    if (det == 0.0) {
        memset(d, 0, 9 * sizeof(float));
        return 0;
    }
    double id = 1.0 / det;
    d[0] = (float)((M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) * id);
    d[1] = (float)((M(0, 2) * M(2, 1) - M(0, 1) * M(2, 2)) * id);
    d[2] = (float)((M(0, 1) * M(1, 2) - M(0, 2) * M(1, 1)) * id);
    d[3] = (float)((M(1, 2) * M(2, 0) - M(1, 0) * M(2, 2)) * id);
    d[4] = (float)((M(0, 0) * M(2, 2) - M(0, 2) * M(2, 0)) * id);
    d[5] = (float)((M(0, 2) * M(1, 0) - M(0, 0) * M(1, 2)) * id);
    d[6] = (float)((M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0)) * id);
    d[7] = (float)((M(0, 1) * M(2, 0) - M(0, 0) * M(2, 1)) * id);
    d[8] = (float)((M(0, 0) * M(1, 1) - M(0, 1) * M(1, 0)) * id);
#undef M
    return 1;
}

/* ProjectorBase::setCameraParams(K, R, T)  [W2]; AffineWarper::getRTfromHomogeneous for "affine" */
int orc_warper_set_camera(orc_warper *w, const float K[9], const float Rin[9])
{
    float R[9], T[3] = {0, 0, 0};
    memcpy(R, Rin, sizeof R);
    if (w->type == P_AFFINE) {
        /* R <- (H with tx,ty zeroed)^T ; T <- -(R * (tx,ty,0)) */
        float tx = R[2], ty = R[5];
        R[2] = 0.f;
        R[5] = 0.f;
        float Rt[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = R[j * 3 + i];
        memcpy(R, Rt, sizeof R);
        for (int i = 0; i < 3; ++i) {
            float s = R[i * 3 + 0] * tx + R[i * 3 + 1] * ty;
            s = s + R[i * 3 + 2] * 0.f;
            T[i] = s * -1.f;
        }
    }
    float Kinv[9];
    memcpy(w->k, K, sizeof w->k);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) w->rinv[i * 3 + j] = R[j * 3 + i];
    mat3_inv(K, Kinv);
    mat3_mul(R, Kinv, w->r_kinv);
    mat3_mul(K, w->rinv, w->k_rinv);
    memcpy(w->t, T, sizeof T);
    return 0;
}

void orc_warper_get_projector(const orc_warper *w, float k[9], float rinv[9], float r_kinv[9], float k_rinv[9], float t[3])
{
    memcpy(k, w->k, 36);
    memcpy(rinv, w->rinv, 36);
    memcpy(r_kinv, w->r_kinv, 36);
    memcpy(k_rinv, w->k_rinv, 36);
    memcpy(t, w->t, 12);
}

/* ---- mapForward (warpers_inl.hpp) --------------------------------------------------------------- */
void orc_warper_map_forward(const orc_warper *w, float x, float y, float *pu, float *pv)
{
    const float *rk = w->r_kinv;
    const float scale = w->scale, a = w->a, b = w->b;
    float x_, y_, z_;
    if (w->type == P_COMPRESSED_PORTRAIT || w->type == P_PANINI_PORTRAIT) {
        y_ = rk[0] * x + rk[1] * y + rk[2];
        x_ = rk[3] * x + rk[4] * y + rk[5];
    } else {
        x_ = rk[0] * x + rk[1] * y + rk[2];
        y_ = rk[3] * x + rk[4] * y + rk[5];
    }
    z_ = rk[6] * x + rk[7] * y + rk[8];
    float u, v;
    switch (w->type) {
    case P_PLANE:
    case P_AFFINE:
        x_ = w->t[0] + x_ / z_ * (1 - w->t[2]);
        y_ = w->t[1] + y_ / z_ * (1 - w->t[2]);
        u = scale * x_;
        v = scale * y_;
        break;
    case P_SPHERICAL: {
        u = scale * M_ATAN2(x_, z_);
        float ww = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        v = scale * (SSP_PI_F - M_ACOS(ww == ww ? ww : 0));
        break;
    }
    case P_CYLINDRICAL:
        u = scale * M_ATAN2(x_, z_);
        v = scale * y_ / sqrtf(x_ * x_ + z_ * z_);
        break;
    case P_FISHEYE: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = SSP_PI_F - M_ACOS(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = scale * v_ * M_COS(u_);
        v = scale * v_ * M_SIN(u_);
        break;
    }
    case P_STEREOGRAPHIC: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = SSP_PI_F - M_ACOS(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float r = M_SIN(v_) / (1 - M_COS(v_));
        u = scale * r * M_COS(u_);
        v = scale * r * M_SIN(u_);
        break;
    }
    case P_COMPRESSED: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = M_ASIN(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = scale * a * M_TAN(u_ / a);
        v = scale * b * M_TAN(v_) / M_COS(u_);
        break;
    }
    case P_COMPRESSED_PORTRAIT: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = M_ASIN(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = -scale * a * M_TAN(u_ / a);
        v = scale * b * M_TAN(v_) / M_COS(u_);
        break;
    }
    case P_PANINI:
    case P_PANINI_PORTRAIT: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = M_ASIN(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float tg = a * M_TAN(u_ / a);
        u = (w->type == P_PANINI ? scale : -scale) * tg;
        float sinu = M_SIN(u_);
        if (fabsf(sinu) < 1E-7f)
            v = scale * b * M_TAN(v_);
        else
            v = scale * b * tg * M_TAN(v_) / sinu;
        break;
    }
    case P_MERCATOR: {
        float u_ = M_ATAN2(x_, z_);
        float v_ = M_ASIN(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = scale * u_;
        v = scale * M_LOG(M_TAN((float)(SSP_PI_D / 4) + v_ / 2));
        break;
    }
    default: { /* P_TRANSVERSE_MERCATOR */
        float u_ = M_ATAN2(x_, z_);
        float v_ = M_ASIN(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float B = M_COS(v_) * M_SIN(u_);
        u = scale / 2 * M_LOG((1 + B) / (1 - B));
        v = scale * M_ATAN2(M_TAN(v_), M_COS(u_));
        break;
    }
    }
    *pu = u;
    *pv = v;
}

/* ---- mapBackward --------------------------------------------------------------------------------- */
void orc_warper_map_backward(const orc_warper *w, float u, float v, float *px, float *py)
{
    const float *kr = w->k_rinv;
    const float scale = w->scale, a = w->a, b = w->b;
    float x_, y_, z_, x, y, z;
    if (w->type == P_PLANE || w->type == P_AFFINE) {
        u = u / scale - w->t[0];
        v = v / scale - w->t[1];
        x = kr[0] * u + kr[1] * v + kr[2] * (1 - w->t[2]);
        y = kr[3] * u + kr[4] * v + kr[5] * (1 - w->t[2]);
        z = kr[6] * u + kr[7] * v + kr[8] * (1 - w->t[2]);
        x /= z;
        y /= z;
        *px = x;
        *py = y;
        return;
    }
    if (w->type == P_COMPRESSED_PORTRAIT || w->type == P_PANINI_PORTRAIT)
        u /= -scale;
    else
        u /= scale;
    v /= scale;
    switch (w->type) {
    case P_SPHERICAL: {
        float sinv = M_SIN(SSP_PI_F - v);
        x_ = sinv * M_SIN(u);
        y_ = M_COS(SSP_PI_F - v);
        z_ = sinv * M_COS(u);
        break;
    }
    case P_CYLINDRICAL:
        x_ = M_SIN(u);
        y_ = v;
        z_ = M_COS(u);
        break;
    case P_FISHEYE: {
        float u_ = M_ATAN2(v, u);
        float v_ = sqrtf(u * u + v * v);
        float sinv = M_SIN(SSP_PI_F - v_);
        x_ = sinv * M_SIN(u_);
        y_ = M_COS(SSP_PI_F - v_);
        z_ = sinv * M_COS(u_);
        break;
    }
    case P_STEREOGRAPHIC: {
        float u_ = M_ATAN2(v, u);
        float r = sqrtf(u * u + v * v);
        float v_ = 2 * M_ATAN(1.f / r);
        float sinv = M_SIN(SSP_PI_F - v_);
        x_ = sinv * M_SIN(u_);
        y_ = M_COS(SSP_PI_F - v_);
        z_ = sinv * M_COS(u_);
        break;
    }
    case P_COMPRESSED:
    case P_COMPRESSED_PORTRAIT: {
        float aatg = a * M_ATAN(u / a);
        float u_ = aatg;
        float v_ = M_ATAN(v * M_COS(aatg) / b);
        float cosv = M_COS(v_);
        float p = cosv * M_SIN(u_), q = M_SIN(v_);
        if (w->type == P_COMPRESSED) { x_ = p; y_ = q; } else { y_ = p; x_ = q; }
        z_ = cosv * M_COS(u_);
        break;
    }
    case P_PANINI:
    case P_PANINI_PORTRAIT: {
        float lamda = a * M_ATAN(u / a);
        float u_ = lamda;
        float v_;
        if (fabsf(lamda) > 1E-7f)
            v_ = M_ATAN(v * M_SIN(lamda) / (b * a * M_TAN(lamda / a)));
        else
            v_ = M_ATAN(v / b);
        float cosv = M_COS(v_);
        float p = cosv * M_SIN(u_), q = M_SIN(v_);
        if (w->type == P_PANINI) { x_ = p; y_ = q; } else { y_ = p; x_ = q; }
        z_ = cosv * M_COS(u_);
        break;
    }
    case P_MERCATOR: {
        float v_ = M_ATAN(M_SINH(v));
        float u_ = u;
        float cosv = M_COS(v_);
        x_ = cosv * M_SIN(u_);
        y_ = M_SIN(v_);
        z_ = cosv * M_COS(u_);
        break;
    }
    default: { /* P_TRANSVERSE_MERCATOR */
        float v_ = M_ASIN(M_SIN(v) / M_COSH(u));
        float u_ = M_ATAN2(M_SINH(u), M_COS(v));
        float cosv = M_COS(v_);
        x_ = cosv * M_SIN(u_);
        y_ = M_SIN(v_);
        z_ = cosv * M_COS(u_);
        break;
    }
    }
    x = kr[0] * x_ + kr[1] * y_ + kr[2] * z_;
    y = kr[3] * x_ + kr[4] * y_ + kr[5] * z_;
    z = kr[6] * x_ + kr[7] * y_ + kr[8] * z_;
    if (z > 0) {
        x /= z;
        y /= z;
    } else
        x = y = -1;
    *px = x;
    *py = y;
}

/* ---- detectResultRoi family  [W3] ----------------------------------------------------------------- */
typedef struct { float tl_u, tl_v, br_u, br_v; } mm_t;
static inline void mm_feed(const orc_warper *w, mm_t *m, float x, float y)
{
    float u, v;
    orc_warper_map_forward(w, x, y, &u, &v);
    /* std::min(a,b) = (b < a) ? b : a  -> NaN never replaces the running value */
    m->tl_u = (u < m->tl_u) ? u : m->tl_u;
    m->tl_v = (v < m->tl_v) ? v : m->tl_v;
    m->br_u = (m->br_u < u) ? u : m->br_u;
    m->br_v = (m->br_v < v) ? v : m->br_v;
}

static void detect_roi(const orc_warper *w, int W, int H, int tl[2], int br[2])
{
    mm_t m = {3.402823466e+38f, 3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    if (w->type == P_PLANE || w->type == P_AFFINE) {
        mm_feed(w, &m, 0.f, 0.f);
        mm_feed(w, &m, 0.f, (float)(H - 1));
        mm_feed(w, &m, (float)(W - 1), 0.f);
        mm_feed(w, &m, (float)(W - 1), (float)(H - 1));
    } else if (w->type == P_SPHERICAL || w->type == P_CYLINDRICAL) {
        for (int x = 0; x < W; ++x) {
            mm_feed(w, &m, (float)x, 0.f);
            mm_feed(w, &m, (float)x, (float)(H - 1));
        }
        for (int y = 0; y < H; ++y) {
            mm_feed(w, &m, 0.f, (float)y);
            mm_feed(w, &m, (float)(W - 1), (float)y);
        }
    } else {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) mm_feed(w, &m, (float)x, (float)y);
    }
    tl[0] = (int)m.tl_u;
    tl[1] = (int)m.tl_v;
    br[0] = (int)m.br_u;
    br[1] = (int)m.br_v;

    if (w->type == P_SPHERICAL) {
        /* SphericalWarper::detectResultRoi pole fix-up (after the int truncation above) */
        float tl_uf = (float)tl[0], tl_vf = (float)tl[1], br_uf = (float)br[0], br_vf = (float)br[1];
        for (int pass = 0; pass < 2; ++pass) {
            float x = w->rinv[1];
            float y = pass == 0 ? w->rinv[4] : -w->rinv[4];
            float z = w->rinv[7];
            if (y > 0.f) {
                float x_ = (w->k[0] * x + w->k[1] * y) / z + w->k[2];
                float y_ = w->k[4] * y / z + w->k[5];
                if (x_ > 0.f && x_ < W && y_ > 0.f && y_ < H) {
                    float pv = pass == 0 ? (float)(SSP_PI_D * w->scale) : 0.f;
                    tl_uf = (0.f < tl_uf) ? 0.f : tl_uf;
                    tl_vf = (pv < tl_vf) ? pv : tl_vf;
                    br_uf = (br_uf < 0.f) ? 0.f : br_uf;
                    br_vf = (br_vf < pv) ? pv : br_vf;
                }
            }
        }
        tl[0] = (int)tl_uf;
        tl[1] = (int)tl_vf;
        br[0] = (int)br_uf;
        br[1] = (int)br_vf;
    }
}

int orc_warper_roi(orc_warper *w, int W, int H, const float K[9], const float R[9], int roi[4])
{
    int tl[2], br[2];
    orc_warper_set_camera(w, K, R);
    detect_roi(w, W, H, tl, br);
    roi[0] = tl[0];
    roi[1] = tl[1];
    roi[2] = br[0] - tl[0] + 1;
    roi[3] = br[1] - tl[1] + 1;
    return 0;
}

int orc_warper_build_maps(orc_warper *w, int W, int H, const float K[9], const float R[9], float *xmap, float *ymap,
                          int roi[4])
{
    orc_warper_roi(w, W, H, K, R, roi);
    const int dw = roi[2], dh = roi[3];
    ORC_PAR_FOR
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u) {
            float x, y;
            orc_warper_map_backward(w, (float)(u + roi[0]), (float)(v + roi[1]), &x, &y);
            xmap[(size_t)v * dw + u] = x;
            ymap[(size_t)v * dw + u] = y;
        }
    return 0;
}

/* ---- cv::remap  [W4/W5, Appendix A.2] --------------------------------------------------------------- */
int orc_remap(const void *src_, int W, int H, int cn, int depth, const float *xmap, const float *ymap, int dw, int dh,
              int interp, int border, void *dst_)
{
    if (interp == ORC_INTER_AREA) interp = ORC_INTER_LINEAR; /* remap replaces AREA by LINEAR */
    if (!(depth == ORC_U8 || depth == ORC_F32) || cn < 1 || cn > 4) {
        orc_set_error("remap: unsupported type");
        return -1;
    }
    const uint8_t *s8 = (const uint8_t *)src_;
    const float *sf = (const float *)src_;
    uint8_t *d8 = (uint8_t *)dst_;
    float *df = (float *)dst_;
    ORC_PAR_FOR
    for (int dy = 0; dy < dh; ++dy)
        for (int dx = 0; dx < dw; ++dx) {
            size_t di = ((size_t)dy * dw + dx) * cn;
            float fx = xmap[(size_t)dy * dw + dx], fy = ymap[(size_t)dy * dw + dx];
            if (interp == ORC_INTER_NEAREST) {
                int sx = orc_sat_s16(orc_cv_round(fx)), sy = orc_sat_s16(orc_cv_round(fy));
                if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H) {
                    size_t si = ((size_t)sy * W + sx) * cn;
                    for (int c = 0; c < cn; ++c)
                        if (depth == ORC_U8) d8[di + c] = s8[si + c]; else df[di + c] = sf[si + c];
                } else if (border == ORC_BORDER_CONSTANT) {
                    for (int c = 0; c < cn; ++c)
                        if (depth == ORC_U8) d8[di + c] = 0; else df[di + c] = 0.f;
                } else {
                    sx = orc_border(sx, W, border);
                    sy = orc_border(sy, H, border);
                    size_t si = ((size_t)sy * W + sx) * cn;
                    for (int c = 0; c < cn; ++c)
                        if (depth == ORC_U8) d8[di + c] = s8[si + c]; else df[di + c] = sf[si + c];
                }
                continue;
            }
            /* INTER_LINEAR: 1/32-pixel quantised coordinates */
            int isx = orc_cv_round(fx * 32), isy = orc_cv_round(fy * 32);
            int sx = orc_sat_s16(isx >> 5), sy = orc_sat_s16(isy >> 5);
            int ax = isx & 31, ay = isy & 31;
            int x0 = sx, x1 = sx + 1, y0 = sy, y1 = sy + 1;
            int in00, in01, in10, in11; /* tap validity (only matters for BORDER_CONSTANT) */
            if (border == ORC_BORDER_CONSTANT) {
                int vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
                int vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
                in00 = vx0 && vy0; in01 = vx1 && vy0; in10 = vx0 && vy1; in11 = vx1 && vy1;
                if (!vx0) x0 = 0;
                if (!vx1) x1 = 0;
                if (!vy0) y0 = 0;
                if (!vy1) y1 = 0;
            } else {
                in00 = in01 = in10 = in11 = 1;
                x0 = orc_border(x0, W, border);
                x1 = orc_border(x1, W, border);
                y0 = orc_border(y0, H, border);
                y1 = orc_border(y1, H, border);
            }
            size_t i00 = ((size_t)y0 * W + x0) * cn, i01 = ((size_t)y0 * W + x1) * cn;
            size_t i10 = ((size_t)y1 * W + x0) * cn, i11 = ((size_t)y1 * W + x1) * cn;
            if (depth == ORC_U8) {
                /* initInterTab2D fixed-point table: w = saturate_cast<short>(vy*vx*32768) -> exact ints */
                int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32;
                int w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
                for (int c = 0; c < cn; ++c) {
                    int p00 = in00 ? s8[i00 + c] : 0, p01 = in01 ? s8[i01 + c] : 0;
                    int p10 = in10 ? s8[i10 + c] : 0, p11 = in11 ? s8[i11 + c] : 0;
                    int t = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
                    d8[di + c] = orc_sat_u8((t + (1 << 14)) >> 15);
                }
            } else {
                /* float table: tab[k1*2+k2] = vy[k1] * vx[k2], with v = {1 - f/32, f/32} */
                float vx1 = (float)ax * (1.f / 32), vx0 = 1.f - vx1;
                float vy1 = (float)ay * (1.f / 32), vy0 = 1.f - vy1;
                float w00 = vy0 * vx0, w01 = vy0 * vx1, w10 = vy1 * vx0, w11 = vy1 * vx1;
                for (int c = 0; c < cn; ++c) {
                    float p00 = in00 ? sf[i00 + c] : 0.f, p01 = in01 ? sf[i01 + c] : 0.f;
                    float p10 = in10 ? sf[i10 + c] : 0.f, p11 = in11 ? sf[i11 + c] : 0.f;
                    df[di + c] = p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11;
                }
            }
        }
    return 0;
}

/* RotationWarperBase<P>::warp = buildMaps + remap; dst is (br-tl+1) sized, returns tl in roi[0..1] */
int orc_warper_warp(orc_warper *w, const void *src, int W, int H, int cn, int depth, const float K[9], const float R[9],
                    int interp, int border, void *dst, int roi[4])
{
    orc_warper_roi(w, W, H, K, R, roi);
    size_t n = (size_t)roi[2] * roi[3];
    float *xm = (float *)malloc(n * sizeof(float)), *ym = (float *)malloc(n * sizeof(float));
    if (!xm || !ym) {
        free(xm);
        free(ym);
        orc_set_error("warp: out of memory for %d x %d maps", roi[2], roi[3]);
        return -1;
    }
    orc_warper_build_maps(w, W, H, K, R, xm, ym, roi);
    int rc = orc_remap(src, W, H, cn, depth, xm, ym, roi[2], roi[3], interp, border, dst);
    free(xm);
    free(ym);
    return rc;
}

/* PyRotationWarper::warpBackward(src, K, R, interp, border, dst_size) (RotationWarperBase<P>::warpBackward, warpers_inl.hpp):
 * src must have the size of warpRoi(dst_size, K, R); every destination pixel (x, y) of the original frame reads src at
 * mapForward(x, y) - roi.tl.  Not on the reference's path (nice-to-have). */
int orc_warper_warp_backward(orc_warper *w, const void *src, int sw, int sh, int cn, int depth, const float K[9], const float R[9], int interp, int border,
                             int dst_w, int dst_h, void *dst)
{
    int roi[4];
    orc_warper_roi(w, dst_w, dst_h, K, R, roi);
    if (roi[2] != sw || roi[3] != sh) {
        orc_set_error("warpBackward: src is %dx%d but warpRoi(dst_size) is %dx%d", sw, sh, roi[2], roi[3]);
        return -1;
    }
    size_t n = (size_t)dst_w * dst_h;
    float *xm = (float *)malloc(n * sizeof(float)), *ym = (float *)malloc(n * sizeof(float));
    ORC_PAR_FOR
    for (int y = 0; y < dst_h; ++y)
        for (int x = 0; x < dst_w; ++x) {
            float u, v;
            orc_warper_map_forward(w, (float)x, (float)y, &u, &v);
            xm[(size_t)y * dst_w + x] = u - (float)roi[0];
            ym[(size_t)y * dst_w + x] = v - (float)roi[1];
        }
    int rc = orc_remap(src, sw, sh, cn, depth, xm, ym, dst_w, dst_h, interp, border, dst);
    free(xm);
    free(ym);
    return rc;
}

/* thread count of the OpenMP flavour (1 in the serial libraries) */
int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

void orc_result_roi(int n, const int *corners, const int *sizes, int roi[4])
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        if (corners[2 * i] < tlx) tlx = corners[2 * i];
        if (corners[2 * i + 1] < tly) tly = corners[2 * i + 1];
        if (corners[2 * i] + sizes[2 * i] > brx) brx = corners[2 * i] + sizes[2 * i];
        if (corners[2 * i + 1] + sizes[2 * i + 1] > bry) bry = corners[2 * i + 1] + sizes[2 * i + 1];
    }
    roi[0] = tlx;
    roi[1] = tly;
    roi[2] = brx - tlx;
    roi[3] = bry - tly;
}
