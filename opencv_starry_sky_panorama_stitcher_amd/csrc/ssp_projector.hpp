// ssp_projector.hpp -- the 16 projections of cv.PyRotationWarper as host+device functions.
//
// Replaces OpenCV's ProjectorBase::setCameraParams and the per-projection mapForward / mapBackward
// (detail/warpers_inl.hpp) that sde.py reaches through warper.warpRoi (:1696) and warper.warp
// (:1557, :1591, :1731, :1740).  binary32 arithmetic in OpenCV's operation order, no FMA contraction
// (the translation units are built with -ffp-contract=off); transcendentals through include/ssp_math.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ssp_math.h"

namespace ssp {

enum ProjKind : int {
    PK_PLANE = 0,
    PK_AFFINE,
    PK_CYLINDRICAL,
    PK_SPHERICAL,
    PK_FISHEYE,
    PK_STEREOGRAPHIC,
    PK_COMPRESSED,
    PK_COMPRESSED_PORTRAIT,
    PK_PANINI,
    PK_PANINI_PORTRAIT,
    PK_MERCATOR,
    PK_TRANSVERSE_MERCATOR,
    PK_COUNT
};

struct Projector {
    int kind;
    float scale, a, b;
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];
};

// how detectResultRoi scans the source frame
enum RoiScan : int { SCAN_FULL = 0, SCAN_BORDER = 1, SCAN_CORNERS = 2 };
__host__ __device__ inline int roi_scan_mode(int kind)
{
    if (kind == PK_PLANE || kind == PK_AFFINE) return SCAN_CORNERS;
    if (kind == PK_SPHERICAL || kind == PK_CYLINDRICAL) return SCAN_BORDER;
    return SCAN_FULL;
}
// separable projections: mapBackward's trigonometry depends on u only through the column and on v only
// through the row, so it is tabulated per output column / row (2(w+h) evaluations instead of 4wh)
__host__ __device__ inline bool is_separable(int kind) { return kind == PK_SPHERICAL || kind == PK_CYLINDRICAL || kind == PK_MERCATOR; }
__host__ __device__ inline bool is_portrait(int kind) { return kind == PK_COMPRESSED_PORTRAIT || kind == PK_PANINI_PORTRAIT; }

struct Ray { float x, y, z; };

// direction on the unit sphere (or cylinder / plane) for panorama coordinate (u, v) already divided by scale
__host__ __device__ inline Ray backward_ray(const Projector &p, float u, float v)
{
    Ray r;
    switch (p.kind) {
    case PK_SPHERICAL: {
        float sv = ssp_sinf(SSP_PI_F - v);
        r.x = sv * ssp_sinf(u);
        r.y = ssp_cosf(SSP_PI_F - v);
        r.z = sv * ssp_cosf(u);
        break;
    }
    case PK_CYLINDRICAL:
        r.x = ssp_sinf(u);
        r.y = v;
        r.z = ssp_cosf(u);
        break;
    case PK_FISHEYE: {
        float az = ssp_atan2f(v, u);
        float rad = sqrtf(u * u + v * v);
        float sv = ssp_sinf(SSP_PI_F - rad);
        r.x = sv * ssp_sinf(az);
        r.y = ssp_cosf(SSP_PI_F - rad);
        r.z = sv * ssp_cosf(az);
        break;
    }
    case PK_STEREOGRAPHIC: {
        float az = ssp_atan2f(v, u);
        float rad = sqrtf(u * u + v * v);
        float pol = 2 * ssp_atanf(1.f / rad);
        float sv = ssp_sinf(SSP_PI_F - pol);
        r.x = sv * ssp_sinf(az);
        r.y = ssp_cosf(SSP_PI_F - pol);
        r.z = sv * ssp_cosf(az);
        break;
    }
    case PK_COMPRESSED:
    case PK_COMPRESSED_PORTRAIT: {
        float lon = p.a * ssp_atanf(u / p.a);
        float lat = ssp_atanf(v * ssp_cosf(lon) / p.b);
        float cl = ssp_cosf(lat);
        float h = cl * ssp_sinf(lon), q = ssp_sinf(lat);
        r.x = p.kind == PK_COMPRESSED ? h : q;
        r.y = p.kind == PK_COMPRESSED ? q : h;
        r.z = cl * ssp_cosf(lon);
        break;
    }
    case PK_PANINI:
    case PK_PANINI_PORTRAIT: {
        float lon = p.a * ssp_atanf(u / p.a);
        float lat;
        if (fabsf(lon) > 1E-7f)
            lat = ssp_atanf(v * ssp_sinf(lon) / (p.b * p.a * ssp_tanf(lon / p.a)));
        else
            lat = ssp_atanf(v / p.b);
        float cl = ssp_cosf(lat);
        float h = cl * ssp_sinf(lon), q = ssp_sinf(lat);
        r.x = p.kind == PK_PANINI ? h : q;
        r.y = p.kind == PK_PANINI ? q : h;
        r.z = cl * ssp_cosf(lon);
        break;
    }
    case PK_MERCATOR: {
        float lat = ssp_atanf(ssp_sinhf(v));
        float cl = ssp_cosf(lat);
        r.x = cl * ssp_sinf(u);
        r.y = ssp_sinf(lat);
        r.z = cl * ssp_cosf(u);
        break;
    }
    default: {  // PK_TRANSVERSE_MERCATOR
        float lat = ssp_asinf(ssp_sinf(v) / ssp_coshf(u));
        float lon = ssp_atan2f(ssp_sinhf(u), ssp_cosf(v));
        float cl = ssp_cosf(lat);
        r.x = cl * ssp_sinf(lon);
        r.y = ssp_sinf(lat);
        r.z = cl * ssp_cosf(lon);
        break;
    }
    }
    return r;
}

// source-pixel coordinates of a ray: K * R^T * ray, perspective divide; behind the camera -> (-1,-1)
__host__ __device__ inline void project_ray(const float kr[9], Ray r, float &x, float &y)
{
    float X = kr[0] * r.x + kr[1] * r.y + kr[2] * r.z;
    float Y = kr[3] * r.x + kr[4] * r.y + kr[5] * r.z;
    float Z = kr[6] * r.x + kr[7] * r.y + kr[8] * r.z;
    if (Z > 0) {
        x = X / Z;
        y = Y / Z;
    } else {
        x = -1;
        y = -1;
    }
}

__host__ __device__ inline void map_backward(const Projector &p, float u, float v, float &x, float &y)
{
    if (p.kind == PK_PLANE || p.kind == PK_AFFINE) {
        u = u / p.scale - p.t[0];
        v = v / p.scale - p.t[1];
        float w = 1 - p.t[2];
        float X = p.k_rinv[0] * u + p.k_rinv[1] * v + p.k_rinv[2] * w;
        float Y = p.k_rinv[3] * u + p.k_rinv[4] * v + p.k_rinv[5] * w;
        float Z = p.k_rinv[6] * u + p.k_rinv[7] * v + p.k_rinv[8] * w;
        x = X / Z;  // the plane projector has no z > 0 test
        y = Y / Z;
        return;
    }
    if (is_portrait(p.kind)) u /= -p.scale; else u /= p.scale;
    v /= p.scale;
    project_ray(p.k_rinv, backward_ray(p, u, v), x, y);
}

// K * R^T * ray before the perspective divide (the tile records and the coordinate planes of the batched warp test Z themselves); for the
// rotation projections map_backward(u, v) == (Z > 0 ? (X / Z, Y / Z) : (-1, -1)), for plane / affine (X / Z, Y / Z) whatever the sign of Z
__host__ __device__ inline void map_backward_xyz(const Projector &p, float u, float v, float &X, float &Y, float &Z)
{
    float a, b, c;
    if (p.kind == PK_PLANE || p.kind == PK_AFFINE) {
        a = u / p.scale - p.t[0];
        b = v / p.scale - p.t[1];
        c = 1 - p.t[2];
    } else {
        if (is_portrait(p.kind)) u /= -p.scale; else u /= p.scale;
        v /= p.scale;
        const Ray r = backward_ray(p, u, v);
        a = r.x; b = r.y; c = r.z;
    }
    X = p.k_rinv[0] * a + p.k_rinv[1] * b + p.k_rinv[2] * c;
    Y = p.k_rinv[3] * a + p.k_rinv[4] * b + p.k_rinv[5] * c;
    Z = p.k_rinv[6] * a + p.k_rinv[7] * b + p.k_rinv[8] * c;
}

__host__ __device__ inline void map_forward(const Projector &p, float x, float y, float &u, float &v)
{
    const float *m = p.r_kinv;
    float r0 = m[0] * x + m[1] * y + m[2];
    float r1 = m[3] * x + m[4] * y + m[5];
    float z_ = m[6] * x + m[7] * y + m[8];
    float x_ = is_portrait(p.kind) ? r1 : r0;
    float y_ = is_portrait(p.kind) ? r0 : r1;
    const float s = p.scale;
    switch (p.kind) {
    case PK_PLANE:
    case PK_AFFINE:
        x_ = p.t[0] + x_ / z_ * (1 - p.t[2]);
        y_ = p.t[1] + y_ / z_ * (1 - p.t[2]);
        u = s * x_;
        v = s * y_;
        return;
    case PK_CYLINDRICAL:
        u = s * ssp_atan2f(x_, z_);
        v = s * y_ / sqrtf(x_ * x_ + z_ * z_);
        return;
    default:
        break;
    }
    float lon = ssp_atan2f(x_, z_);
    float sinlat = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
    switch (p.kind) {
    case PK_SPHERICAL:
        u = s * lon;
        v = s * (SSP_PI_F - ssp_acosf(sinlat == sinlat ? sinlat : 0));
        break;
    case PK_FISHEYE: {
        float pol = SSP_PI_F - ssp_acosf(sinlat);
        u = s * pol * ssp_cosf(lon);
        v = s * pol * ssp_sinf(lon);
        break;
    }
    case PK_STEREOGRAPHIC: {
        float pol = SSP_PI_F - ssp_acosf(sinlat);
        float rad = ssp_sinf(pol) / (1 - ssp_cosf(pol));
        u = s * rad * ssp_cosf(lon);
        v = s * rad * ssp_sinf(lon);
        break;
    }
    case PK_COMPRESSED:
    case PK_COMPRESSED_PORTRAIT: {
        float lat = ssp_asinf(sinlat);
        float su = p.kind == PK_COMPRESSED ? s : -s;
        u = su * p.a * ssp_tanf(lon / p.a);
        v = s * p.b * ssp_tanf(lat) / ssp_cosf(lon);
        break;
    }
    case PK_PANINI:
    case PK_PANINI_PORTRAIT: {
        float lat = ssp_asinf(sinlat);
        float tg = p.a * ssp_tanf(lon / p.a);
        u = (p.kind == PK_PANINI ? s : -s) * tg;
        float sl = ssp_sinf(lon);
        if (fabsf(sl) < 1E-7f)
            v = s * p.b * ssp_tanf(lat);
        else
            v = s * p.b * tg * ssp_tanf(lat) / sl;
        break;
    }
    case PK_MERCATOR: {
        float lat = ssp_asinf(sinlat);
        u = s * lon;
        v = s * ssp_logf(ssp_tanf((float)(SSP_PI_D / 4) + lat / 2));
        break;
    }
    default: {  // PK_TRANSVERSE_MERCATOR
        float lat = ssp_asinf(sinlat);
        float B = ssp_cosf(lat) * ssp_sinf(lon);
        u = s / 2 * ssp_logf((1 + B) / (1 - B));
        v = s * ssp_atan2f(ssp_tanf(lat), ssp_cosf(lon));
        break;
    }
    }
}

}  // namespace ssp
