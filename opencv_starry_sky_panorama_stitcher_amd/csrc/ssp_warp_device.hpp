// ssp_warp_device.hpp -- device-side cv::remap arithmetic (imgproc/imgwarp.cpp semantics) for gfx950.
//
// INTER_LINEAR: coordinates quantised to 1/32 px with round-half-even, integer base saturated to int16,
// 8-bit samples blended with the 15-bit fixed-point table (weights 32*a*b, sum 2^15) and rounded as
// (t + 2^14) >> 15; float samples with float weights.  INTER_NEAREST: round-half-even, saturate to int16.
// Borders through cv::borderInterpolate (closed form instead of OpenCV's reflection loop).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ssp.h"

namespace ssp {

struct SrcView {
    const uint8_t *data;
    size_t pitch;  // bytes
    int w, h;
};

__device__ inline int cv_round(float v)
{
    // cvRound: round half to even; a value that does not fit int32 (or NaN) gives INT_MIN, as cvtss2si does
    float r = __builtin_rintf(v);
    return (r >= -2147483648.0f && r < 2147483648.0f) ? (int)r : INT32_MIN;
}
__device__ inline int sat_s16(int v) { return min(max(v, -32768), 32767); }

__device__ inline int border_index(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == SSP_BORDER_REPLICATE) return p < 0 ? 0 : len - 1;
    if (type == SSP_BORDER_REFLECT) {
        if (len == 1) return 0;
        int period = 2 * len;
        int m = p % period;
        if (m < 0) m += period;
        return m < len ? m : period - 1 - m;
    }
    if (type == SSP_BORDER_REFLECT_101) {
        if (len == 1) return 0;
        int period = 2 * len - 2;
        int m = p % period;
        if (m < 0) m += period;
        return m < len ? m : period - m;
    }
    if (type == SSP_BORDER_WRAP) {
        int m = p % len;
        if (m < 0) m += len;
        return m;
    }
    return -1;  // BORDER_CONSTANT
}

typedef uint32_t u32x2_unaligned __attribute__((ext_vector_type(2), aligned(1)));

// fixed-point bilinear of one 8UC3 pixel; returns B | G<<8 | R<<16.
// (t + 2^14) >> 15 with w = 32*a*b equals (V + 512) >> 10 with V = (p00*(32-ax) + p01*ax)*(32-ay) + (p10*(32-ax) + p11*ax)*ay.
__device__ inline uint32_t blend_12(uint32_t b00, uint32_t g00, uint32_t r00, uint32_t b01, uint32_t g01, uint32_t r01, uint32_t b10, uint32_t g10,
                                    uint32_t r10, uint32_t b11, uint32_t g11, uint32_t r11, uint32_t ax, uint32_t ay)
{
    const uint32_t bx = 32 - ax, by = 32 - ay;
    uint32_t vb = (b00 * bx + b01 * ax) * by + (b10 * bx + b11 * ax) * ay;
    uint32_t vg = (g00 * bx + g01 * ax) * by + (g10 * bx + g11 * ax) * ay;
    uint32_t vr = (r00 * bx + r01 * ax) * by + (r10 * bx + r11 * ax) * ay;
    return ((vb + 512) >> 10) | (((vg + 512) >> 10) << 8) | (((vr + 512) >> 10) << 16);
}

// taps from two 8-byte row reads starting at pixel (ix, iy): q.x = B0 G0 R0 B1, q.y = G1 R1 x x
__device__ inline uint32_t blend_taps_u8c3(u32x2_unaligned q0, u32x2_unaligned q1, uint32_t ax, uint32_t ay)
{
    return blend_12(q0.x & 0xff, (q0.x >> 8) & 0xff, (q0.x >> 16) & 0xff, q0.x >> 24, q0.y & 0xff, (q0.y >> 8) & 0xff, q1.x & 0xff, (q1.x >> 8) & 0xff,
                    (q1.x >> 16) & 0xff, q1.x >> 24, q1.y & 0xff, (q1.y >> 8) & 0xff, ax, ay);
}

// general form: quantised coordinates already known; any border mode; byte loads
__device__ inline uint32_t bilinear_u8c3_at(const SrcView &s, int ix, int iy, uint32_t ax, uint32_t ay, int border)
{
    int x0 = ix, x1 = ix + 1, y0 = iy, y1 = iy + 1;
    bool v00 = true, v01 = true, v10 = true, v11 = true;
    if (border == SSP_BORDER_CONSTANT) {
        bool vx0 = (unsigned)x0 < (unsigned)s.w, vx1 = (unsigned)x1 < (unsigned)s.w;
        bool vy0 = (unsigned)y0 < (unsigned)s.h, vy1 = (unsigned)y1 < (unsigned)s.h;
        v00 = vx0 && vy0; v01 = vx1 && vy0; v10 = vx0 && vy1; v11 = vx1 && vy1;
        x0 = vx0 ? x0 : 0; x1 = vx1 ? x1 : 0; y0 = vy0 ? y0 : 0; y1 = vy1 ? y1 : 0;
    } else {
        x0 = border_index(x0, s.w, border); x1 = border_index(x1, s.w, border);
        y0 = border_index(y0, s.h, border); y1 = border_index(y1, s.h, border);
    }
    const uint8_t *p00 = s.data + (size_t)y0 * s.pitch + (size_t)x0 * 3, *p01 = s.data + (size_t)y0 * s.pitch + (size_t)x1 * 3;
    const uint8_t *p10 = s.data + (size_t)y1 * s.pitch + (size_t)x0 * 3, *p11 = s.data + (size_t)y1 * s.pitch + (size_t)x1 * 3;
    uint32_t z = 0;
    return blend_12(v00 ? p00[0] : z, v00 ? p00[1] : z, v00 ? p00[2] : z, v01 ? p01[0] : z, v01 ? p01[1] : z, v01 ? p01[2] : z, v10 ? p10[0] : z,
                    v10 ? p10[1] : z, v10 ? p10[2] : z, v11 ? p11[0] : z, v11 ? p11[1] : z, v11 ? p11[2] : z, ax, ay);
}

__device__ inline uint32_t bilinear_u8c3(const SrcView &s, float fx, float fy, int border)
{
    const int isx = cv_round(fx * 32.f), isy = cv_round(fy * 32.f);
    const int ix = sat_s16(isx >> 5), iy = sat_s16(isy >> 5);
    const uint32_t ax = isx & 31, ay = isy & 31;
    // the 8-byte row reads touch bytes [3*ix, 3*ix+8): keep them inside the row
    if (ix >= 0 && ix <= s.w - 3 && iy >= 0 && iy <= s.h - 2) {
        const uint8_t *p0 = s.data + (size_t)iy * s.pitch + (size_t)ix * 3;
        u32x2_unaligned q0 = *(const u32x2_unaligned *)p0;
        u32x2_unaligned q1 = *(const u32x2_unaligned *)(p0 + s.pitch);
        return blend_taps_u8c3(q0, q1, ax, ay);
    }
    return bilinear_u8c3_at(s, ix, iy, ax, ay, border);
}

// ---- fast forms used by the fused kernel (bit-identical to the plain ones above) -------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// x/z and y/z for two pixels at once, correctly rounded, sharing the reciprocal refinement.  This is the instruction
// sequence hipcc emits for an IEEE float division (v_rcp, two fma to refine it, q = n*y, two residual corrections) without
// v_div_scale / v_div_fixup, which only act when operands or quotient leave the normal range -- the caller guarantees
// 2^-60 < z < 2^60 and |x|, |y| < 2^60 (otherwise it takes the plain '/' path).  v_pk_fma_f32 does both pixels per instruction.
__device__ inline void div2_exact(f32x2 x, f32x2 y, f32x2 z, f32x2 &qx, f32x2 &qy)
{
    f32x2 r = {__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
    const f32x2 one = {1.f, 1.f};
    f32x2 e = __builtin_elementwise_fma(-z, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    f32x2 q = x * r;
    f32x2 t = __builtin_elementwise_fma(-z, q, x);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(-z, q, x);
    qx = __builtin_elementwise_fma(t, r, q);
    q = y * r;
    t = __builtin_elementwise_fma(-z, q, y);
    q = __builtin_elementwise_fma(t, r, q);
    t = __builtin_elementwise_fma(-z, q, y);
    qy = __builtin_elementwise_fma(t, r, q);
}

// cvRound with one compare: v_cvt_i32_f32 saturates and maps NaN to 0, the x86 result for both is INT_MIN
__device__ inline int cv_round_fast(float v)
{
    float r = __builtin_rintf(v);
    int i = (int)r;
    return __builtin_fabsf(r) < 2147483648.0f ? i : INT32_MIN;
}

// fixed-point bilinear from two 8-byte row reads: horizontal taps with v_dot4_u32_u8 (both taps of a channel brought into
// one register by v_alignbit), vertical with v_dot2_u32_u16 including the +512 rounding term; (V + 512) >> 10.
__device__ inline uint32_t blend_taps_dot(u32x2_unaligned q0, u32x2_unaligned q1, uint32_t ax, uint32_t ay)
{
    const uint32_t wx = (32u - ax) | (ax << 24);                       // weights for bytes 0 and 3
    const u16x2 wy = __builtin_bit_cast(u16x2, (32u - ay) | (ay << 16));
    // row 0: [B0 G0 R0 B1] [G1 R1 . .] -> B taps in bytes 0/3 of q.x, G taps in bytes 0/3 of q >> 8, R taps of q >> 16
    const uint32_t g0 = __builtin_amdgcn_alignbit(q0.y, q0.x, 8), r0 = __builtin_amdgcn_alignbit(q0.y, q0.x, 16);
    const uint32_t g1 = __builtin_amdgcn_alignbit(q1.y, q1.x, 8), r1 = __builtin_amdgcn_alignbit(q1.y, q1.x, 16);
    const uint32_t hb0 = __builtin_amdgcn_udot4(q0.x, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(q1.x, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(r0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1, wx, 0u, false);
    const uint32_t vb = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hb0 | (hb1 << 16)), wy, 512u, false);
    const uint32_t vg = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hg0 | (hg1 << 16)), wy, 512u, false);
    const uint32_t vr = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, hr0 | (hr1 << 16)), wy, 512u, false);
    return (vb >> 10) | ((vg >> 10) << 8) | ((vr >> 10) << 16);
}

template <typename T> __device__ inline T zero_of() { return (T)0; }

// generic cv::remap of one output pixel
template <typename T, int CN>
__device__ inline void remap_pixel(const SrcView &s, float fx, float fy, int interp, int border, T *out)
{
    if (interp == SSP_INTER_NEAREST) {
        int sx = sat_s16(cv_round(fx)), sy = sat_s16(cv_round(fy));
        if (!((unsigned)sx < (unsigned)s.w && (unsigned)sy < (unsigned)s.h)) {
            if (border == SSP_BORDER_CONSTANT) {
#pragma unroll
                for (int c = 0; c < CN; ++c) out[c] = zero_of<T>();
                return;
            }
            sx = border_index(sx, s.w, border);
            sy = border_index(sy, s.h, border);
        }
        const T *p = (const T *)(s.data + (size_t)sy * s.pitch) + (size_t)sx * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) out[c] = p[c];
        return;
    }
    const int isx = cv_round(fx * 32.f), isy = cv_round(fy * 32.f);
    const int ix = sat_s16(isx >> 5), iy = sat_s16(isy >> 5);
    const int ax = isx & 31, ay = isy & 31;
    int x0 = ix, x1 = ix + 1, y0 = iy, y1 = iy + 1;
    bool v00 = true, v01 = true, v10 = true, v11 = true;
    if (border == SSP_BORDER_CONSTANT) {
        bool vx0 = (unsigned)x0 < (unsigned)s.w, vx1 = (unsigned)x1 < (unsigned)s.w;
        bool vy0 = (unsigned)y0 < (unsigned)s.h, vy1 = (unsigned)y1 < (unsigned)s.h;
        v00 = vx0 && vy0; v01 = vx1 && vy0; v10 = vx0 && vy1; v11 = vx1 && vy1;
        x0 = vx0 ? x0 : 0; x1 = vx1 ? x1 : 0; y0 = vy0 ? y0 : 0; y1 = vy1 ? y1 : 0;
    } else {
        x0 = border_index(x0, s.w, border); x1 = border_index(x1, s.w, border);
        y0 = border_index(y0, s.h, border); y1 = border_index(y1, s.h, border);
    }
    const T *p00 = (const T *)(s.data + (size_t)y0 * s.pitch) + (size_t)x0 * CN, *p01 = (const T *)(s.data + (size_t)y0 * s.pitch) + (size_t)x1 * CN;
    const T *p10 = (const T *)(s.data + (size_t)y1 * s.pitch) + (size_t)x0 * CN, *p11 = (const T *)(s.data + (size_t)y1 * s.pitch) + (size_t)x1 * CN;
    if constexpr (sizeof(T) == 1) {
        const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            int a = v00 ? p00[c] : 0, b = v01 ? p01[c] : 0, d = v10 ? p10[c] : 0, e = v11 ? p11[c] : 0;
            int t = a * w00 + b * w01 + d * w10 + e * w11;
            out[c] = (T)min(max((t + (1 << 14)) >> 15, 0), 255);
        }
    } else {
        const float vx1 = (float)ax * (1.f / 32), vx0 = 1.f - vx1, vy1 = (float)ay * (1.f / 32), vy0 = 1.f - vy1;
        const float w00 = vy0 * vx0, w01 = vy0 * vx1, w10 = vy1 * vx0, w11 = vy1 * vx1;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            float a = v00 ? (float)p00[c] : 0.f, b = v01 ? (float)p01[c] : 0.f, d = v10 ? (float)p10[c] : 0.f, e = v11 ? (float)p11[c] : 0.f;
            float t = a * w00 + b * w01;
            t = t + d * w10;
            t = t + e * w11;
            out[c] = (T)t;
        }
    }
}

}  // namespace ssp
