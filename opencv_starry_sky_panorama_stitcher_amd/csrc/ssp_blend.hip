// ssp_blend.hip -- cv.detail.Blender / FeatherBlender / MultiBandBlender on gfx950.
//
// Replaces (stitching_detailed_enhanced.py):
//   :1806-1819  Blender_createDefault(NO) / detail_MultiBandBlender().setNumBands / detail_FeatherBlender().setSharpness
//   :1820       blender.prepare(resultRoi)
//   :1886/:1889 blender.feed(image_warped_s, mask_warped, corner)
//   :1930       blender.blend(None, None) -> (result int16, result_mask)
//
// MI355X-first restructuring of MultiBandBlender (same arithmetic, different schedule):
//   OpenCV feeds image by image: copyMakeBorder, Laplacian pyramid, then a read-modify-write of the pano-sized
//   accumulators dst_pyr_laplace_/dst_band_weights_ at every level (about 27 B per padded pixel per image), and
//   blend() re-reads them to normalise and collapse.  Here feed() only builds the image's Gaussian pyramids
//   (G_1..G_nb int16x3, W_1..W_nb f32; level 0 stays the warped frame itself, borders are index arithmetic).
//   blend() then runs ONE kernel per pano level, top level first: each output pixel gathers every image that covers
//   it (in feed order), forms the Laplacian sample G_l - pyrUp(G_{l+1}) on the fly, accumulates (short)(L*w) and w
//   in registers, normalises, adds pyrUp of the already collapsed parent level and stores the collapsed level once.
//   The pano-sized accumulators never exist in HBM.  Integer sums wrap mod 2^16 exactly as C "short +=" does and the
//   float weight sums are taken in feed order, so results are bit-identical to the sequential formulation.
#include <type_traits>

#include "ssp_internal.hpp"

using namespace ssp;

#define WEIGHT_EPS 1e-5f
#define MAX_BANDS 16

// ====================================================================================================================
// device helpers
// ====================================================================================================================
__device__ inline int reflect_idx(int p, int len)  // BORDER_REFLECT
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    int period = 2 * len, m = p % period;
    if (m < 0) m += period;
    return m < len ? m : period - 1 - m;
}
__device__ inline int reflect101_idx(int p, int len)  // BORDER_REFLECT_101
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    int period = 2 * len - 2, m = p % period;
    if (m < 0) m += period;
    return m < len ? m : period - m;
}
__device__ inline int sat16(int v) { return min(max(v, -32768), 32767); }
// static_cast<short>(float) as on x86-64: cvttss2si, then the low 16 bits
__device__ inline int trunc16(float f)
{
    int t = (f > -2147483648.0f && f < 2147483648.0f) ? (int)f : INT32_MIN;
    return (int)(int16_t)(uint16_t)(t & 0xffff);
}

// 3-channel pixel value types: integer path (int) and float path
template <bool FLT> struct Acc3;
template <> struct Acc3<false> { typedef int T; };
template <> struct Acc3<true> { typedef float T; };

// level-0 placement of a fed image inside its padded rectangle
struct Place {
    int left, top, iw, ih;
};

// Level-0 sample of the (virtually) bordered image: copyMakeBorder(BORDER_REFLECT) as index arithmetic
template <typename ST, typename VT>
__device__ inline void load_img0(const void *base, size_t pitch, const Place &pl, int x, int y, VT out[3])
{
    int sx = reflect_idx(x - pl.left, pl.iw), sy = reflect_idx(y - pl.top, pl.ih);
    const ST *p = (const ST *)((const char *)base + (size_t)sy * pitch) + (size_t)sx * 3;
    out[0] = (VT)p[0];
    out[1] = (VT)p[1];
    out[2] = (VT)p[2];
}
// Level-0 weight: mask/255 inside the image, 0 in the border (copyMakeBorder BORDER_CONSTANT)
__device__ inline float load_w0(const void *base, size_t pitch, const Place &pl, int x, int y)
{
    int sx = x - pl.left, sy = y - pl.top;
    if ((unsigned)sx >= (unsigned)pl.iw || (unsigned)sy >= (unsigned)pl.ih) return 0.f;
    const float inv255 = (float)(1. / 255.);
    return (float)((const uint8_t *)base + (size_t)sy * pitch)[sx] * inv255;
}
template <typename ST, typename VT>
__device__ inline void load_px(const void *base, size_t pitch, int x, int y, VT out[3])
{
    const ST *p = (const ST *)((const char *)base + (size_t)y * pitch) + (size_t)x * 3;
    out[0] = (VT)p[0];
    out[1] = (VT)p[1];
    out[2] = (VT)p[2];
}

// ====================================================================================================================
// pyrDown: 5-tap [1 4 6 4 1] both axes, BORDER_REFLECT_101, dst = (n+1)/2; image (3 channels) and weight together
// ====================================================================================================================
struct PyrDownArgs {
    // source level
    const void *g; size_t gp;   // image level (level 0: the fed image)
    const void *w; size_t wp;   // weight level (level 0: the u8 mask)
    int sw, sh;                 // (padded) source level size
    Place pl;                   // level 0 only
    // destination level
    void *dg; size_t dgp;
    float *dw; size_t dwp;
    int dwid, dhei;
};

typedef uint32_t u32x2_u1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// five consecutive 3-channel pixels starting at p (interior fast path; may read a few bytes past the fifth pixel)
template <typename ST, typename VT> __device__ inline void load5(const ST *p, VT s[5][3]);
template <> __device__ inline void load5<uint8_t, int>(const uint8_t *p, int s[5][3])
{
    u32x2_u1 a = *(const u32x2_u1 *)p, b = *(const u32x2_u1 *)(p + 8);
    const uint32_t w[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = 3 * k + c;
            s[k][c] = (int)((w[byte >> 2] >> (8 * (byte & 3))) & 0xff);
        }
}
template <> __device__ inline void load5<int16_t, int>(const int16_t *p, int s[5][3])
{
    u32x4_a4 a = *(const u32x4_a4 *)p, b = *(const u32x4_a4 *)(p + 8);
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int e = 3 * k + c;
            s[k][c] = (int)(int16_t)(uint16_t)(w[e >> 1] >> (16 * (e & 1)));
        }
}
template <> __device__ inline void load5<float, float>(const float *p, float s[5][3])
{
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[k][c] = p[3 * k + c];
}
template <> __device__ inline void load5<uint8_t, float>(const uint8_t *p, float s[5][3])
{
    int t[5][3];
    load5<uint8_t, int>(p, t);
    for (int k = 0; k < 5; ++k)
        for (int c = 0; c < 3; ++c) s[k][c] = (float)t[k][c];
}
template <> __device__ inline void load5<int16_t, float>(const int16_t *p, float s[5][3])
{
    int t[5][3];
    load5<int16_t, int>(p, t);
    for (int k = 0; k < 5; ++k)
        for (int c = 0; c < 3; ++c) s[k][c] = (float)t[k][c];
}
template <> __device__ inline void load5<float, int>(const float *p, int s[5][3])
{
    for (int k = 0; k < 5; ++k)
        for (int c = 0; c < 3; ++c) s[k][c] = (int)p[3 * k + c];
}

// generic (border-aware) evaluation of one output pixel
template <bool LEVEL0, typename ST, bool FLT>
__device__ inline void pyr_down_one(const PyrDownArgs &a, int x, int y)
{
    typedef typename Acc3<FLT>::T VT;
    int xs[5], ys[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xs[k] = reflect101_idx(2 * x - 2 + k, a.sw);
        ys[k] = reflect101_idx(2 * y - 2 + k, a.sh);
    }
    VT rowv[5][3];
    float roww[5];
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        VT s[5][3];
        float ws[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (LEVEL0) {
                load_img0<ST, VT>(a.g, a.gp, a.pl, xs[k], ys[r], s[k]);
                ws[k] = load_w0(a.w, a.wp, a.pl, xs[k], ys[r]);
            } else {
                load_px<ST, VT>(a.g, a.gp, xs[k], ys[r], s[k]);
                ws[k] = ((const float *)((const char *)a.w + (size_t)ys[r] * a.wp))[xs[k]];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            VT t = s[2][c] * 6 + (s[1][c] + s[3][c]) * 4;
            t = t + s[0][c];
            rowv[r][c] = t + s[4][c];
        }
        float tw = ws[2] * 6 + (ws[1] + ws[3]) * 4;
        tw = tw + ws[0];
        roww[r] = tw + ws[4];
    }
    if (FLT) {
        float *d = (float *)((char *)a.dg + (size_t)y * a.dgp) + (size_t)x * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float t = (float)rowv[2][c] * 6 + ((float)rowv[1][c] + (float)rowv[3][c]) * 4;
            t = t + (float)rowv[0][c];
            t = t + (float)rowv[4][c];
            d[c] = t * (1.f / 256);
        }
    } else {
        int16_t *d = (int16_t *)((char *)a.dg + (size_t)y * a.dgp) + (size_t)x * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int t = (int)rowv[2][c] * 6 + ((int)rowv[1][c] + (int)rowv[3][c]) * 4 + (int)rowv[0][c] + (int)rowv[4][c];
            d[c] = (int16_t)((t + 128) >> 8);
        }
    }
    float tw = roww[2] * 6 + (roww[1] + roww[3]) * 4;
    tw = tw + roww[0];
    tw = tw + roww[4];
    ((float *)((char *)a.dw + (size_t)y * a.dwp))[x] = tw * (1.f / 256);
}

// One lane = one output column x and PD_ROWS consecutive output rows; a 256-thread group covers 64 x (4*PD_ROWS) outputs.
// Interior lanes read 2*PD_ROWS+3 source rows once (5 taps each, wide unaligned loads) and keep the horizontal results
// in registers; a wave that touches a border falls back to the per-pixel border-aware form.
// Up to PD_MAXB images per launch (blockIdx.z); the descriptors travel by value in the kernel-argument segment: scalar
// loads, and every pointer is known to be global memory.  (The texture addresser is the bottleneck of these kernels:
// what counts is the number of vector memory instructions per wave, so nothing uniform may be loaded per lane.)
#define PD_MAXB 8
struct PyrDownBatch {
    PyrDownArgs a[PD_MAXB];
};

template <bool LEVEL0, typename ST, bool FLT, int PD_ROWS>
__global__ __launch_bounds__(256) void k_pyr_down(const PyrDownBatch batch)
{
    typedef typename Acc3<FLT>::T VT;
    const PyrDownArgs &a = batch.a[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PD_ROWS;
    if (x >= a.dwid || y0 >= a.dhei) return;
    constexpr int NR = 2 * PD_ROWS + 3;
    // source column/row of the first tap, in the coordinates of the stored buffer
    const int cx = LEVEL0 ? 2 * x - 2 - a.pl.left : 2 * x - 2;
    const int cy = LEVEL0 ? 2 * y0 - 2 - a.pl.top : 2 * y0 - 2;
    const int bw = LEVEL0 ? a.pl.iw : a.sw, bh = LEVEL0 ? a.pl.ih : a.sh;
    // fast path: all taps inside the stored buffer (with 3 spare pixels for the wide reads at level 0) and all rows exist
    const bool interior = cx >= 0 && cx + 8 <= bw && cy >= 0 && cy + NR <= bh && y0 + PD_ROWS <= a.dhei;
    if (__ballot(!interior) != 0ULL) {
#pragma unroll 1
        for (int j = 0; j < PD_ROWS; ++j)
            if (y0 + j < a.dhei) pyr_down_one<LEVEL0, ST, FLT>(a, x, y0 + j);
        return;
    }
    VT hv[NR][3];
    float hw[NR];
    const float inv255 = (float)(1. / 255.);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        VT s[5][3];
        load5<ST, VT>((const ST *)((const char *)a.g + (size_t)(cy + r) * a.gp) + (size_t)cx * 3, s);
        float ws[5];
        if (LEVEL0) {
            u32x2_u1 m = *(const u32x2_u1 *)((const uint8_t *)a.w + (size_t)(cy + r) * a.wp + cx);
            ws[0] = (float)(m.x & 0xff) * inv255; ws[1] = (float)((m.x >> 8) & 0xff) * inv255; ws[2] = (float)((m.x >> 16) & 0xff) * inv255;
            ws[3] = (float)(m.x >> 24) * inv255; ws[4] = (float)(m.y & 0xff) * inv255;
        } else {
            const float *wp = (const float *)((const char *)a.w + (size_t)(cy + r) * a.wp) + cx;
#pragma unroll
            for (int k = 0; k < 5; ++k) ws[k] = wp[k];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            VT t = s[2][c] * 6 + (s[1][c] + s[3][c]) * 4;
            t = t + s[0][c];
            hv[r][c] = t + s[4][c];
        }
        float tw = ws[2] * 6 + (ws[1] + ws[3]) * 4;
        tw = tw + ws[0];
        hw[r] = tw + ws[4];
    }
#pragma unroll
    for (int j = 0; j < PD_ROWS; ++j) {
        const int r0 = 2 * j, y = y0 + j;
        if (FLT) {
            float *d = (float *)((char *)a.dg + (size_t)y * a.dgp) + (size_t)x * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = (float)hv[r0 + 2][c] * 6 + ((float)hv[r0 + 1][c] + (float)hv[r0 + 3][c]) * 4;
                t = t + (float)hv[r0][c];
                t = t + (float)hv[r0 + 4][c];
                d[c] = t * (1.f / 256);
            }
        } else {
            int16_t *d = (int16_t *)((char *)a.dg + (size_t)y * a.dgp) + (size_t)x * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int t = (int)hv[r0 + 2][c] * 6 + ((int)hv[r0 + 1][c] + (int)hv[r0 + 3][c]) * 4 + (int)hv[r0][c] + (int)hv[r0 + 4][c];
                d[c] = (int16_t)((t + 128) >> 8);
            }
        }
        float tw = hw[r0 + 2] * 6 + (hw[r0 + 1] + hw[r0 + 3]) * 4;
        tw = tw + hw[r0];
        tw = tw + hw[r0 + 4];
        ((float *)((char *)a.dw + (size_t)y * a.dwp))[x] = tw * (1.f / 256);
    }
}

// ====================================================================================================================
// pyrUp sample: value of pyrUp(src)(X, Y) for a 2x upsampling; index -1 -> 1 (reflect-101), index n -> n-1 (replicate)
// ====================================================================================================================
template <bool FLT>
__device__ inline void pyr_up_at(const void *base, size_t pitch, int nw, int nh, int X, int Y, typename Acc3<FLT>::T out[3])
{
    typedef typename Acc3<FLT>::T VT;
    typedef typename std::conditional<FLT, float, int16_t>::type ST;
    const int sx = X >> 1, sy = Y >> 1;
    const bool ox = X & 1, oy = Y & 1;
    const int xm = sx - 1 < 0 ? min(1, nw - 1) : sx - 1, xp = sx + 1 >= nw ? nw - 1 : sx + 1;
    const int ym = sy - 1 < 0 ? min(1, nh - 1) : sy - 1, yp = sy + 1 >= nh ? nh - 1 : sy + 1;
    VT h[3][3];  // horizontal results for rows ym, sy, yp
    const int rows[3] = {ym, sy, yp};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r == 0 && oy) continue;  // odd output rows use rows sy and yp only
        VT a[3], b[3], c[3];
        load_px<ST, VT>(base, pitch, sx, rows[r], b);
        load_px<ST, VT>(base, pitch, xp, rows[r], c);
        if (!ox) {
            load_px<ST, VT>(base, pitch, xm, rows[r], a);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (FLT) {
                    // float path keeps pyramids.cpp's border expressions (they round differently)
                    if (nw == 1) h[r][q] = b[q] * 8;
                    else if (sx == 0) h[r][q] = b[q] * 6 + c[q] * 2;
                    else if (sx == nw - 1) h[r][q] = a[q] + b[q] * 7;
                    else { VT t = a[q] + b[q] * 6; h[r][q] = t + c[q]; }
                } else {
                    h[r][q] = a[q] + b[q] * 6 + c[q];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (FLT && (nw == 1 || sx == nw - 1)) h[r][q] = b[q] * 8;
                else h[r][q] = (b[q] + c[q]) * 4;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        VT v;
        if (!oy) { VT t = h[0][q] + h[1][q] * 6; v = t + h[2][q]; }
        else v = (h[1][q] + h[2][q]) * 4;
        if (FLT) out[q] = v * (1.f / 64);
        else out[q] = ((int)v + 32) >> 6;
    }
}

// ====================================================================================================================
// blend level kernel (tile-centric gather over the fed images)
// ====================================================================================================================
struct LevelImg {
    const void *g; size_t gp;    // G_l   (level 0: the fed image)
    const void *gn; size_t gnp;  // G_{l+1}
    const void *w; size_t wp;    // W_l   (level 0: the u8 mask)
    int rx, ry, pw, ph;          // rectangle of this image at level l, in pano level coordinates
    int pwn, phn;                // size of level l+1
    Place pl;                    // level 0 only
    int src_depth;               // level 0 only: SSP_U8 / SSP_S16 / SSP_F32
};

struct LevelArgs {
    const LevelImg *imgs;        // descriptors in global memory: uniform index -> the compiler already uses scalar loads
    int n_imgs;
    int lw, lh;                  // pano level size (padded): border rules refer to it
    int cx0, cy0, cw, ch;        // region of the level that is computed (whole level, or a sub-rectangle for multi-GPU)
    int top;                     // 1: top level (no Laplacian subtraction, no parent)
    const void *parent; size_t pp; int pw, ph;   // collapsed level l+1: full level size (border rules) ...
    int px0, py0, prw, prh;                       // ... and the region of it that exists in memory (buffer origin)
    void *out; size_t op;        // collapsed level l (int16x3 / f32x3) for the region, origin (cx0, cy0); null at level 0
    // optional partial sums imported from other GPUs (full level size)
    const void *ext_lap; size_t elp;
    const float *ext_w; size_t ewp;
    // level-0 outputs: images whose pixel (0,0) is pano pixel (ox0, oy0); nothing is written beyond (fw, fh)
    int fw, fh, ox0, oy0;
    void *result; size_t rp;     // int16x3 / f32x3 or null
    uint8_t *rmask; size_t rmp;  // u8 or null
    uint8_t *mosaic; size_t mp;  // u8x3 or null
    // export mode (multi-GPU): write the un-normalised sums of the region instead of collapsing
    int export_mode;
    void *exp_lap; float *exp_w;
};

// ---- per-pixel form: top level, and export of any level ---------------------------------------------------------------
template <bool LEVEL0, bool FLT>
__global__ __launch_bounds__(256) void k_blend_level(LevelArgs a)
{
    typedef typename Acc3<FLT>::T VT;
    const int X = a.cx0 + blockIdx.x * 64 + (threadIdx.x & 63), Y = a.cy0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool inside = X < a.cx0 + a.cw && Y < a.cy0 + a.ch;
    VT acc[3] = {0, 0, 0};
    float ws = 0.f;
    const int bx0 = a.cx0 + blockIdx.x * 64, by0 = a.cy0 + blockIdx.y * 4;
    for (int i = 0; i < a.n_imgs; ++i) {
        const LevelImg &im = a.imgs[i];
        if (bx0 + 64 <= im.rx || bx0 >= im.rx + im.pw || by0 + 4 <= im.ry || by0 >= im.ry + im.ph) continue;
        const int lx = X - im.rx, ly = Y - im.ry;
        const bool in = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
        float w = 0.f;
        if (in) {
            if (LEVEL0) w = load_w0(im.w, im.wp, im.pl, lx, ly);
            else w = ((const float *)((const char *)im.w + (size_t)ly * im.wp))[lx];
        }
        // a wave whose weights are all zero contributes (short)(L*0) = 0 and w + 0: skip the image loads
        if (__ballot(in && w != 0.f) == 0ULL) continue;
        if (in) {
            VT g[3];
            if (LEVEL0) {
                if (im.src_depth == SSP_U8) load_img0<uint8_t, VT>(im.g, im.gp, im.pl, lx, ly, g);
                else if (im.src_depth == SSP_S16) load_img0<int16_t, VT>(im.g, im.gp, im.pl, lx, ly, g);
                else load_img0<float, VT>(im.g, im.gp, im.pl, lx, ly, g);
            } else {
                if (FLT) load_px<float, VT>(im.g, im.gp, lx, ly, g);
                else load_px<int16_t, VT>(im.g, im.gp, lx, ly, g);
            }
            if (!a.top) {
                VT up[3];
                pyr_up_at<FLT>(im.gn, im.gnp, im.pwn, im.phn, lx, ly, up);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (FLT) g[c] = g[c] - up[c];
                    else g[c] = (VT)sat16((int)g[c] - (int)up[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (FLT) acc[c] = acc[c] + g[c] * w;
                else acc[c] = (VT)((int)acc[c] + trunc16((float)g[c] * w));
            }
            ws += w;
        }
    }
    if (!inside) return;
    if (a.export_mode) {
        // multi-GPU export: this GPU's own partial sums only (imported ones are never re-exported)
        const int ex = X - a.cx0, ey = Y - a.cy0;
        if (FLT) {
            float *d = (float *)a.exp_lap + ((size_t)ey * a.cw + ex) * 3;
            for (int c = 0; c < 3; ++c) d[c] = (float)acc[c];
        } else {
            int16_t *d = (int16_t *)a.exp_lap + ((size_t)ey * a.cw + ex) * 3;
            for (int c = 0; c < 3; ++c) d[c] = (int16_t)(uint16_t)((int)acc[c] & 0xffff);
        }
        a.exp_w[(size_t)ey * a.cw + ex] = ws;
        return;
    }
    if (a.ext_lap) {
        if (FLT) {
            const float *e = (const float *)((const char *)a.ext_lap + (size_t)Y * a.elp) + (size_t)X * 3;
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + e[c];
        } else {
            const int16_t *e = (const int16_t *)((const char *)a.ext_lap + (size_t)Y * a.elp) + (size_t)X * 3;
            for (int c = 0; c < 3; ++c) acc[c] = (VT)((int)acc[c] + (int)e[c]);
        }
        ws += ((const float *)((const char *)a.ext_w + (size_t)Y * a.ewp))[X];
    }
    // normalizeUsingWeightMap for the top level (it is its own collapsed level)
    const float den = ws + WEIGHT_EPS;
    if (FLT) {
        float *d = (float *)((char *)a.out + (size_t)(Y - a.cy0) * a.op) + (size_t)(X - a.cx0) * 3;
        for (int c = 0; c < 3; ++c) d[c] = (float)acc[c] / den;
    } else {
        int16_t *d = (int16_t *)((char *)a.out + (size_t)(Y - a.cy0) * a.op) + (size_t)(X - a.cx0) * 3;
        for (int c = 0; c < 3; ++c) d[c] = (int16_t)trunc16((float)(int16_t)(uint16_t)((int)acc[c] & 0xffff) / den);
    }
}

// ---- 2x2 quad form: every level below the top ----------------------------------------------------------------------------
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef uint16_t u16_q1 __attribute__((aligned(1)));
typedef uint32_t u32_q2 __attribute__((aligned(2)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

// pyrUp of the 3x3 parent neighbourhood around (sx, sy) -> the 2x2 outputs (2sx..2sx+1, 2sy..2sy+1).
// out[0]=(even x, even y) out[1]=(odd x, even y) out[2]=(even x, odd y) out[3]=(odd x, odd y).
// (nw, nh): parent level size for the border rules (-1 -> 1, n -> n-1).  (rx0, ry0, rw, rh): the part of the level that is
// in memory (base points at its first pixel); indices are clamped into it, which only matters for sub-rectangle blends.
template <bool FLT>
__device__ inline void pyr_up_quad(const void *base, size_t pitch, int nw, int nh, int rx0, int ry0, int rw, int rh, int sx, int sy,
                                   typename Acc3<FLT>::T out[4][3])
{
    typedef typename Acc3<FLT>::T VT;
    typedef typename std::conditional<FLT, float, int16_t>::type ST;
    int xm = sx - 1 < 0 ? min(1, nw - 1) : sx - 1, xp = sx + 1 >= nw ? nw - 1 : sx + 1;
    int ym = sy - 1 < 0 ? min(1, nh - 1) : sy - 1, yp = sy + 1 >= nh ? nh - 1 : sy + 1;
    const int xlo = rx0, xhi = rx0 + rw - 1, ylo = ry0, yhi = ry0 + rh - 1;
    const int xc = min(max(sx, xlo), xhi), yc = min(max(sy, ylo), yhi);
    xm = min(max(xm, xlo), xhi); xp = min(max(xp, xlo), xhi);
    ym = min(max(ym, ylo), yhi); yp = min(max(yp, ylo), yhi);
    const int rows[3] = {ym - ry0, yc - ry0, yp - ry0};
    VT he[3][3], ho[3][3];
    const bool contiguous = !FLT && xm == sx - 1 && xc == sx && xp == sx + 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        VT pa[3], pb[3], pc[3];
        const char *rowp = (const char *)base + (size_t)rows[r] * pitch;
        if (contiguous) {
            // 9 int16 = 18 bytes starting at pixel sx-1
            const char *p = rowp + (size_t)(sx - 1 - rx0) * 6;
            u32x4_a2 v = *(const u32x4_a2 *)p;
            uint32_t last = *(const uint16_t *)(p + 16);
            pa[0] = (VT)(int16_t)(v.x & 0xffff); pa[1] = (VT)(int16_t)(v.x >> 16); pa[2] = (VT)(int16_t)(v.y & 0xffff);
            pb[0] = (VT)(int16_t)(v.y >> 16); pb[1] = (VT)(int16_t)(v.z & 0xffff); pb[2] = (VT)(int16_t)(v.z >> 16);
            pc[0] = (VT)(int16_t)(v.w & 0xffff); pc[1] = (VT)(int16_t)(v.w >> 16); pc[2] = (VT)(int16_t)last;
        } else {
            const ST *q = (const ST *)rowp;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pa[c] = (VT)q[(size_t)(xm - rx0) * 3 + c];
                pb[c] = (VT)q[(size_t)(xc - rx0) * 3 + c];
                pc[c] = (VT)q[(size_t)(xp - rx0) * 3 + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (FLT) {
                // pyramids.cpp's border expressions round differently from the interior one: keep them
                if (nw == 1) { he[r][c] = pb[c] * 8; ho[r][c] = pb[c] * 8; }
                else if (sx == 0) { he[r][c] = pb[c] * 6 + pc[c] * 2; ho[r][c] = (pb[c] + pc[c]) * 4; }
                else if (sx == nw - 1) { he[r][c] = pa[c] + pb[c] * 7; ho[r][c] = pb[c] * 8; }
                else { VT t = pa[c] + pb[c] * 6; he[r][c] = t + pc[c]; ho[r][c] = (pb[c] + pc[c]) * 4; }
            } else {
                he[r][c] = pa[c] + pb[c] * 6 + pc[c];
                ho[r][c] = (pb[c] + pc[c]) * 4;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (FLT) {
            VT t = he[0][c] + he[1][c] * 6; out[0][c] = (t + he[2][c]) * (1.f / 64);
            VT u = ho[0][c] + ho[1][c] * 6; out[1][c] = (u + ho[2][c]) * (1.f / 64);
            out[2][c] = ((he[1][c] + he[2][c]) * 4) * (1.f / 64);
            out[3][c] = ((ho[1][c] + ho[2][c]) * 4) * (1.f / 64);
        } else {
            out[0][c] = ((int)he[0][c] + (int)he[1][c] * 6 + (int)he[2][c] + 32) >> 6;
            out[1][c] = ((int)ho[0][c] + (int)ho[1][c] * 6 + (int)ho[2][c] + 32) >> 6;
            out[2][c] = (((int)he[1][c] + (int)he[2][c]) * 4 + 32) >> 6;
            out[3][c] = (((int)ho[1][c] + (int)ho[2][c]) * 4 + 32) >> 6;
        }
    }
}

template <bool LEVEL0, bool FLT>
__global__ __launch_bounds__(256) void k_blend_quad(const LevelArgs a)
{
    typedef typename Acc3<FLT>::T VT;
    const int X0 = a.cx0 + 2 * (blockIdx.x * 32 + (threadIdx.x & 31)), Y0 = a.cy0 + 2 * (blockIdx.y * 8 + (threadIdx.x >> 5));
    const bool inside = X0 < a.cx0 + a.cw && Y0 < a.cy0 + a.ch;  // cw, ch, cx0, cy0 are even: a quad is inside or outside as a whole
    const int bx0 = a.cx0 + blockIdx.x * 64, by0 = a.cy0 + blockIdx.y * 16;
    VT acc[4][3];
    float ws[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q][0] = acc[q][1] = acc[q][2] = 0;
    const float inv255 = (float)(1. / 255.);
    for (int i = 0; i < a.n_imgs; ++i) {
        const LevelImg &im = a.imgs[i];
        if (bx0 + 64 <= im.rx || bx0 >= im.rx + im.pw || by0 + 16 <= im.ry || by0 >= im.ry + im.ph) continue;
        const int lx = X0 - im.rx, ly = Y0 - im.ry;  // even: the rectangle origin is a multiple of 2 below the top level
        const bool in = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        const int mx = lx - im.pl.left, my = ly - im.pl.top;  // level 0: coordinates inside the fed image
        if (in) {
            if (LEVEL0) {
                // weight = mask/255 inside the image, 0 in the border band around it
                if (mx >= 0 && mx + 2 <= im.pl.iw && my >= 0 && my + 2 <= im.pl.ih) {
                    const uint8_t *mp = (const uint8_t *)im.w + (size_t)my * im.wp + mx;
                    const uint32_t m0 = *(const u16_q1 *)mp, m1 = *(const u16_q1 *)(mp + im.wp);
                    w[0] = (float)(m0 & 0xff) * inv255; w[1] = (float)(m0 >> 8) * inv255;
                    w[2] = (float)(m1 & 0xff) * inv255; w[3] = (float)(m1 >> 8) * inv255;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int px = mx + (q & 1), py = my + (q >> 1);
                        if ((unsigned)px < (unsigned)im.pl.iw && (unsigned)py < (unsigned)im.pl.ih)
                            w[q] = (float)((const uint8_t *)im.w + (size_t)py * im.wp)[px] * inv255;
                    }
                }
            } else {
                const float2 w0 = *(const float2 *)((const char *)im.w + (size_t)ly * im.wp + (size_t)lx * 4);
                const float2 w1 = *(const float2 *)((const char *)im.w + (size_t)(ly + 1) * im.wp + (size_t)lx * 4);
                w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y;
            }
        }
        // a wave whose weights are all zero contributes (short)(L*0) = 0 and w + 0: skip the image loads
        const bool any = in && (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f);
        if (__ballot(any) == 0ULL) continue;
        if (in) {
            VT g[4][3];
            if (LEVEL0) {
                // where w != 0 the pixel lies inside the fed image, so no border reflection is ever needed here;
                // pixels outside it get g = 0 (their weight is 0, so the product is 0 whatever g is)
                if (im.src_depth == SSP_U8 && mx >= 0 && mx + 3 <= im.pl.iw && my >= 0 && my + 2 <= im.pl.ih) {
                    const uint8_t *p = (const uint8_t *)im.g + (size_t)my * im.gp + (size_t)mx * 3;
                    const u32x2_u1 r0 = *(const u32x2_u1 *)p, r1 = *(const u32x2_u1 *)(p + im.gp);
                    g[0][0] = (VT)(r0.x & 0xff); g[0][1] = (VT)((r0.x >> 8) & 0xff); g[0][2] = (VT)((r0.x >> 16) & 0xff);
                    g[1][0] = (VT)(r0.x >> 24); g[1][1] = (VT)(r0.y & 0xff); g[1][2] = (VT)((r0.y >> 8) & 0xff);
                    g[2][0] = (VT)(r1.x & 0xff); g[2][1] = (VT)((r1.x >> 8) & 0xff); g[2][2] = (VT)((r1.x >> 16) & 0xff);
                    g[3][0] = (VT)(r1.x >> 24); g[3][1] = (VT)(r1.y & 0xff); g[3][2] = (VT)((r1.y >> 8) & 0xff);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int px = mx + (q & 1), py = my + (q >> 1);
                        g[q][0] = g[q][1] = g[q][2] = 0;
                        if ((unsigned)px < (unsigned)im.pl.iw && (unsigned)py < (unsigned)im.pl.ih) {
                            if (im.src_depth == SSP_U8) load_px<uint8_t, VT>(im.g, im.gp, px, py, g[q]);
                            else if (im.src_depth == SSP_S16) load_px<int16_t, VT>(im.g, im.gp, px, py, g[q]);
                            else load_px<float, VT>(im.g, im.gp, px, py, g[q]);
                        }
                    }
                }
            } else if (FLT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) load_px<float, VT>(im.g, im.gp, lx + (q & 1), ly + (q >> 1), g[q]);
            } else {
                // two int16x3 pixels per row = 12 bytes, 4-byte aligned (lx is even)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const char *p = (const char *)im.g + (size_t)(ly + r) * im.gp + (size_t)lx * 6;
                    const u32x2_a4 v = *(const u32x2_a4 *)p;
                    const uint32_t t = *(const uint32_t *)(p + 8);
                    g[2 * r][0] = (VT)(int16_t)(v.x & 0xffff); g[2 * r][1] = (VT)(int16_t)(v.x >> 16); g[2 * r][2] = (VT)(int16_t)(v.y & 0xffff);
                    g[2 * r + 1][0] = (VT)(int16_t)(v.y >> 16); g[2 * r + 1][1] = (VT)(int16_t)(t & 0xffff); g[2 * r + 1][2] = (VT)(int16_t)(t >> 16);
                }
            }
            VT up[4][3];
            pyr_up_quad<FLT>(im.gn, im.gnp, im.pwn, im.phn, 0, 0, im.pwn, im.phn, lx >> 1, ly >> 1, up);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (FLT) acc[q][c] = acc[q][c] + (g[q][c] - up[q][c]) * w[q];
                    else acc[q][c] = (VT)((int)acc[q][c] + trunc16((float)sat16((int)g[q][c] - (int)up[q][c]) * w[q]));
                }
                ws[q] += w[q];
            }
        }
    }
    if (!inside) return;
    if (a.export_mode) {
        // multi-GPU export: this GPU's own partial sums only (imported ones are never re-exported)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ex = X0 - a.cx0 + (q & 1), ey = Y0 - a.cy0 + (q >> 1);
            if (FLT) {
                float *d = (float *)a.exp_lap + ((size_t)ey * a.cw + ex) * 3;
                for (int c = 0; c < 3; ++c) d[c] = (float)acc[q][c];
            } else {
                int16_t *d = (int16_t *)a.exp_lap + ((size_t)ey * a.cw + ex) * 3;
                for (int c = 0; c < 3; ++c) d[c] = (int16_t)(uint16_t)((int)acc[q][c] & 0xffff);
            }
            a.exp_w[(size_t)ey * a.cw + ex] = ws[q];
        }
        return;
    }
    if (a.ext_lap) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
            if (FLT) {
                const float *e = (const float *)((const char *)a.ext_lap + (size_t)Y * a.elp) + (size_t)X * 3;
                for (int c = 0; c < 3; ++c) acc[q][c] = acc[q][c] + e[c];
            } else {
                const int16_t *e = (const int16_t *)((const char *)a.ext_lap + (size_t)Y * a.elp) + (size_t)X * 3;
                for (int c = 0; c < 3; ++c) acc[q][c] = (VT)((int)acc[q][c] + (int)e[c]);
            }
            ws[q] += ((const float *)((const char *)a.ext_w + (size_t)Y * a.ewp))[X];
        }
    }
    // normalizeUsingWeightMap, then this level's step of restoreImageFromLaplacePyr
    VT up[4][3];
    pyr_up_quad<FLT>(a.parent, a.pp, a.pw, a.ph, a.px0, a.py0, a.prw, a.prh, X0 >> 1, Y0 >> 1, up);
    VT n[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float den = ws[q] + WEIGHT_EPS;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (FLT) n[q][c] = up[q][c] + acc[q][c] / den;
            else n[q][c] = (VT)sat16((int)up[q][c] + trunc16((float)(int16_t)(uint16_t)((int)acc[q][c] & 0xffff) / den));
        }
    }
    if (!LEVEL0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            char *p = (char *)a.out + (size_t)(Y0 - a.cy0 + r) * a.op;
            if (FLT) {
                float *d = (float *)p + (size_t)(X0 - a.cx0) * 3;
                for (int c = 0; c < 3; ++c) { d[c] = (float)n[2 * r][c]; d[3 + c] = (float)n[2 * r + 1][c]; }
            } else {
                uint32_t *d = (uint32_t *)(p + (size_t)(X0 - a.cx0) * 6);
                d[0] = ((uint32_t)(uint16_t)(int)n[2 * r][0]) | ((uint32_t)(uint16_t)(int)n[2 * r][1] << 16);
                d[1] = ((uint32_t)(uint16_t)(int)n[2 * r][2]) | ((uint32_t)(uint16_t)(int)n[2 * r + 1][0] << 16);
                d[2] = ((uint32_t)(uint16_t)(int)n[2 * r + 1][1]) | ((uint32_t)(uint16_t)(int)n[2 * r + 1][2] << 16);
            }
        }
        return;
    }
    // compare(dst_band_weights_0, WEIGHT_EPS, CMP_GT); dst.setTo(0, mask == 0); crop to dst_roi_final_
    int v8[4][3];
    bool valid[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        valid[q] = ws[q] > WEIGHT_EPS;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v;
            if (FLT) { float r = __builtin_rintf((float)n[q][c]); v = r < 0.f ? 0 : (r > 255.f ? 255 : (int)r); }
            else v = min(max((int)n[q][c], 0), 255);  // cv.imwrite's convertTo(CV_8U) saturation, sde.py:1938
            v8[q][c] = valid[q] ? v : 0;
        }
    }
    if (X0 + 2 <= a.fw && Y0 + 2 <= a.fh) {
        // whole quad inside: 2-pixel rows as one 2-byte (mask), 4+2-byte (mosaic) or 3x4-byte (int16 result) store
        const int ox = X0 - a.ox0, oy = Y0 - a.oy0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int q0 = 2 * r, q1 = 2 * r + 1;
            if (a.rmask) *(u16_q1 *)(a.rmask + (size_t)(oy + r) * a.rmp + ox) = (uint16_t)((valid[q0] ? 255u : 0u) | (valid[q1] ? 0xff00u : 0u));
            if (a.mosaic) {
                uint8_t *d = a.mosaic + (size_t)(oy + r) * a.mp + (size_t)ox * 3;  // ox even: 2-byte aligned
                *(u32_q2 *)d = (uint32_t)v8[q0][0] | ((uint32_t)v8[q0][1] << 8) | ((uint32_t)v8[q0][2] << 16) | ((uint32_t)v8[q1][0] << 24);
                *(uint16_t *)(d + 4) = (uint16_t)((uint32_t)v8[q1][1] | ((uint32_t)v8[q1][2] << 8));
            }
            if (a.result) {
                if (FLT) {
                    float *d = (float *)((char *)a.result + (size_t)(oy + r) * a.rp) + (size_t)ox * 3;
                    for (int c = 0; c < 3; ++c) { d[c] = valid[q0] ? (float)n[q0][c] : 0.f; d[3 + c] = valid[q1] ? (float)n[q1][c] : 0.f; }
                } else {
                    uint32_t *d = (uint32_t *)((char *)a.result + (size_t)(oy + r) * a.rp + (size_t)ox * 6);  // ox even: 4-byte aligned
                    const uint32_t a0 = valid[q0] ? (uint16_t)(int)n[q0][0] : 0u, a1 = valid[q0] ? (uint16_t)(int)n[q0][1] : 0u, a2 = valid[q0] ? (uint16_t)(int)n[q0][2] : 0u;
                    const uint32_t b0 = valid[q1] ? (uint16_t)(int)n[q1][0] : 0u, b1 = valid[q1] ? (uint16_t)(int)n[q1][1] : 0u, b2 = valid[q1] ? (uint16_t)(int)n[q1][2] : 0u;
                    d[0] = a0 | (a1 << 16);
                    d[1] = a2 | (b0 << 16);
                    d[2] = b1 | (b2 << 16);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
        if (X >= a.fw || Y >= a.fh) continue;
        const int ox = X - a.ox0, oy = Y - a.oy0;
        if (a.rmask) a.rmask[(size_t)oy * a.rmp + ox] = valid[q] ? 255 : 0;
        if (a.result) {
            if (FLT) {
                float *d = (float *)((char *)a.result + (size_t)oy * a.rp) + (size_t)ox * 3;
                for (int c = 0; c < 3; ++c) d[c] = valid[q] ? (float)n[q][c] : 0.f;
            } else {
                int16_t *d = (int16_t *)((char *)a.result + (size_t)oy * a.rp) + (size_t)ox * 3;
                for (int c = 0; c < 3; ++c) d[c] = valid[q] ? (int16_t)(int)n[q][c] : (int16_t)0;
            }
        }
        if (a.mosaic) {
            uint8_t *d = a.mosaic + (size_t)oy * a.mp + (size_t)ox * 3;
            for (int c = 0; c < 3; ++c) d[c] = (uint8_t)v8[q][c];
        }
    }
}

// ====================================================================================================================
// Blender(NO) and FeatherBlender kernels
// ====================================================================================================================
template <typename ST>
__global__ void k_feed_plain(const ST *img, size_t ip, const uint8_t *mask, size_t mp, int w, int h, int16_t *dst, size_t dp, uint8_t *dmask, size_t dmp, int dx,
                             int dy)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    uint8_t m = mask[(size_t)y * mp + x];
    const ST *s = (const ST *)((const char *)img + (size_t)y * ip) + (size_t)x * 3;
    int16_t *d = (int16_t *)((char *)dst + (size_t)(y + dy) * dp) + (size_t)(x + dx) * 3;
    if (m) { d[0] = (int16_t)s[0]; d[1] = (int16_t)s[1]; d[2] = (int16_t)s[2]; }
    dmask[(size_t)(y + dy) * dmp + x + dx] |= m;
}

// exact L1 distance to the nearest zero pixel (distanceTransform(DIST_L1, 3)): row scan, then column scan
#define DIST_INF 65534
__global__ void k_dist_rows(const uint8_t *mask, size_t mp, int w, int h, int *d, size_t dpitch)
{
    int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= h) return;
    const uint8_t *m = mask + (size_t)y * mp;
    int *r = (int *)((char *)d + (size_t)y * dpitch);
    int cur = DIST_INF;
    for (int x = 0; x < w; ++x) { cur = m[x] ? min(cur + 1, DIST_INF) : 0; r[x] = cur; }
    cur = DIST_INF;
    for (int x = w - 1; x >= 0; --x) { cur = m[x] ? min(cur + 1, DIST_INF) : 0; r[x] = min(r[x], cur); }
}
__global__ void k_dist_cols(int *d, size_t dpitch, int w, int h, float sharpness, float *wm, size_t wp)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    int cur = DIST_INF;
    for (int y = 0; y < h; ++y) {
        int *p = (int *)((char *)d + (size_t)y * dpitch) + x;
        cur = min(*p, min(cur + 1, DIST_INF));
        *p = cur;
    }
    cur = DIST_INF;
    for (int y = h - 1; y >= 0; --y) {
        int *p = (int *)((char *)d + (size_t)y * dpitch) + x;
        cur = min(*p, min(cur + 1, DIST_INF));
        float t = (float)cur * sharpness;  // multiply(weight, sharpness); threshold(THRESH_TRUNC, 1)
        ((float *)((char *)wm + (size_t)y * wp))[x] = t > 1.f ? 1.f : t;
    }
}
template <typename ST>
__global__ void k_feed_feather(const ST *img, size_t ip, const float *wm, size_t wp, int w, int h, int16_t *dst, size_t dp, float *dw, size_t dwp, int dx, int dy)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    float wv = ((const float *)((const char *)wm + (size_t)y * wp))[x];
    const ST *s = (const ST *)((const char *)img + (size_t)y * ip) + (size_t)x * 3;
    int16_t *d = (int16_t *)((char *)dst + (size_t)(y + dy) * dp) + (size_t)(x + dx) * 3;
    for (int c = 0; c < 3; ++c) d[c] = (int16_t)(uint16_t)(((int)d[c] + trunc16((float)s[c] * wv)) & 0xffff);
    ((float *)((char *)dw + (size_t)(y + dy) * dwp))[x + dx] += wv;
}
// Blender::blend / FeatherBlender::blend epilogue: normalise (feather), mask, zero unmasked, crop, 8-bit mosaic
__global__ void k_finish_plain(const int16_t *dst, size_t dp, const uint8_t *dmask, size_t dmp, const float *dw, size_t dwp, int fw, int fh, int16_t *res,
                               size_t rp, uint8_t *rmask, size_t rmp, uint8_t *mosaic, size_t mp)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= fw || y >= fh) return;
    const int16_t *s = (const int16_t *)((const char *)dst + (size_t)y * dp) + (size_t)x * 3;
    int v[3] = {s[0], s[1], s[2]};
    uint8_t m;
    if (dw) {
        float wv = ((const float *)((const char *)dw + (size_t)y * dwp))[x];
        float den = wv + WEIGHT_EPS;
        for (int c = 0; c < 3; ++c) v[c] = trunc16((float)v[c] / den);
        m = wv > WEIGHT_EPS ? 255 : 0;
    } else
        m = dmask[(size_t)y * dmp + x];
    if (!m) v[0] = v[1] = v[2] = 0;
    if (rmask) rmask[(size_t)y * rmp + x] = m;
    if (res) {
        int16_t *d = (int16_t *)((char *)res + (size_t)y * rp) + (size_t)x * 3;
        d[0] = (int16_t)v[0]; d[1] = (int16_t)v[1]; d[2] = (int16_t)v[2];
    }
    if (mosaic) {
        uint8_t *d = mosaic + (size_t)y * mp + (size_t)x * 3;
        for (int c = 0; c < 3; ++c) d[c] = (uint8_t)min(max(v[c], 0), 255);
    }
}

__global__ void k_add_partial(void *dl, size_t dlp, float *dw, size_t dwp, const void *sl, const float *sw, int x0, int y0, int w, int h, int flt)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    if (flt) {
        float *d = (float *)((char *)dl + (size_t)(y + y0) * dlp) + (size_t)(x + x0) * 3;
        const float *s = (const float *)sl + ((size_t)y * w + x) * 3;
        for (int c = 0; c < 3; ++c) d[c] += s[c];
    } else {
        int16_t *d = (int16_t *)((char *)dl + (size_t)(y + y0) * dlp) + (size_t)(x + x0) * 3;
        const int16_t *s = (const int16_t *)sl + ((size_t)y * w + x) * 3;
        for (int c = 0; c < 3; ++c) d[c] = (int16_t)(uint16_t)(((int)d[c] + (int)s[c]) & 0xffff);
    }
    ((float *)((char *)dw + (size_t)(y + y0) * dwp))[x + x0] += sw[(size_t)y * w + x];
}

// ====================================================================================================================
// host side
// ====================================================================================================================
struct FeedRec {
    ssp_image *img = nullptr, *mask = nullptr;
    Place pl;
    int pw[MAX_BANDS + 1], ph[MAX_BANDS + 1];  // padded level sizes
    int rx[MAX_BANDS + 1], ry[MAX_BANDS + 1];  // rectangle origin per level (pano level coordinates)
    void *G[MAX_BANDS + 1]; size_t gp[MAX_BANDS + 1];
    float *W[MAX_BANDS + 1]; size_t wp[MAX_BANDS + 1];
};

struct ssp_blender {
    int type = SSP_BLEND_NO;
    int want_bands = 5, num_bands = 0;
    float sharpness = 0.02f;
    bool float_mode = false;
    bool prepared = false;
    int roi[4] = {0, 0, 0, 0}, final_roi[4] = {0, 0, 0, 0};
    // NO / FEATHER accumulators
    ssp_image *dst = nullptr, *dst_mask = nullptr, *dst_weight = nullptr;
    // MULTIBAND
    int lw[MAX_BANDS + 1], lh[MAX_BANDS + 1];
    std::vector<FeedRec> feeds;
    ssp_image *ext_lap[MAX_BANDS + 1] = {nullptr}, *ext_w[MAX_BANDS + 1] = {nullptr};
    DescRing ring;  // per-level image descriptors (and batched pyrDown arguments)
};

namespace ssp {

static void release_feeds(ssp_blender *b)
{
    for (auto &f : b->feeds) {
        image_unref(f.img);
        image_unref(f.mask);
        for (int l = 1; l <= b->num_bands; ++l) { pool_free(f.G[l]); pool_free(f.W[l]); }
    }
    b->feeds.clear();
}
static void release_state(ssp_blender *b)
{
    release_feeds(b);
    image_unref(b->dst); image_unref(b->dst_mask); image_unref(b->dst_weight);
    b->dst = b->dst_mask = b->dst_weight = nullptr;
    for (int l = 0; l <= MAX_BANDS; ++l) { image_unref(b->ext_lap[l]); image_unref(b->ext_w[l]); b->ext_lap[l] = b->ext_w[l] = nullptr; }
    b->prepared = false;
}

// MultiBandBlender::feed geometry: grow by gap, clip to the pano, snap to multiples of 2^nb, shift back inside;
// allocates the Gaussian levels 1..nb of the image and of its weight map
static int make_feed_rec(ssp_blender *b, ssp_image *img, ssp_image *mask, int tlx, int tly, FeedRec &f)
{
    const int nb = b->num_bands, m = 1 << nb;
    const int rx = b->roi[0], ry = b->roi[1], rbx = rx + b->roi[2], rby = ry + b->roi[3];
    const int iw = img->w, ih = img->h;
    const int gap = 3 * (1 << nb);
    int tnx = std::max(rx, tlx - gap), tny = std::max(ry, tly - gap);
    int bnx = std::min(rbx, tlx + iw + gap), bny = std::min(rby, tly + ih + gap);
    tnx = rx + (((tnx - rx) >> nb) << nb);
    tny = ry + (((tny - ry) >> nb) << nb);
    int width = bnx - tnx, height = bny - tny;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    bnx = tnx + width;
    bny = tny + height;
    int dy = std::max(bny - rby, 0), dx = std::max(bnx - rbx, 0);
    tnx -= dx; bnx -= dx; tny -= dy; bny -= dy;
    const int top = tly - tny, left = tlx - tnx, bottom = bny - tly - ih, right = bnx - tlx - iw;
    SSP_REQUIRE(top >= 0 && left >= 0 && bottom >= 0 && right >= 0, "feed: image at (%d,%d) %dx%d does not fit the prepared roi (%d,%d %dx%d)", tlx, tly, iw, ih,
                rx, ry, b->roi[2], b->roi[3]);
    f.pl = {left, top, iw, ih};
    f.pw[0] = width; f.ph[0] = height;
    int x_tl = tnx - rx, y_tl = tny - ry;
    for (int l = 0; l <= nb; ++l) {
        if (l > 0) { f.pw[l] = (f.pw[l - 1] + 1) / 2; f.ph[l] = (f.ph[l - 1] + 1) / 2; }
        f.rx[l] = x_tl; f.ry[l] = y_tl;
        x_tl /= 2; y_tl /= 2;
        f.G[l] = nullptr; f.W[l] = nullptr; f.gp[l] = 0; f.wp[l] = 0;
    }
    const int esz = b->float_mode ? 4 : 2;
    for (int l = 1; l <= nb; ++l) {
        f.gp[l] = align_up((size_t)f.pw[l] * 3 * esz, 16);
        f.wp[l] = align_up((size_t)f.pw[l] * 4, 16);
        int rc = pool_alloc(f.gp[l] * f.ph[l], &f.G[l]);
        if (!rc) rc = pool_alloc(f.wp[l] * f.ph[l], (void **)&f.W[l]);
        if (rc) {
            for (int q = 1; q <= l; ++q) { pool_free(f.G[q]); pool_free(f.W[q]); }
            return rc;
        }
    }
    f.img = img;
    f.mask = mask;
    return 0;
}

static void fill_pyr_args(const ssp_blender *b, const FeedRec &f, int l, PyrDownArgs &a)
{
    a.g = l == 0 ? f.img->data : f.G[l]; a.gp = l == 0 ? f.img->pitch : f.gp[l];
    a.w = l == 0 ? f.mask->data : (void *)f.W[l]; a.wp = l == 0 ? f.mask->pitch : f.wp[l];
    a.sw = f.pw[l]; a.sh = f.ph[l];
    a.pl = f.pl;
    a.dg = f.G[l + 1]; a.dgp = f.gp[l + 1];
    a.dw = f.W[l + 1]; a.dwp = f.wp[l + 1];
    a.dwid = f.pw[l + 1]; a.dhei = f.ph[l + 1];
}

static double pyr_bytes(const ssp_blender *b, const FeedRec &f, int l)
{
    const int esz = b->float_mode ? 4 : 2;
    double dst_px = (double)f.pw[l + 1] * f.ph[l + 1];
    if (l == 0) return (double)f.img->w * f.img->h * (3.0 * depth_size(f.img->depth) + 1) + dst_px * (3 * esz + 4);
    return (double)f.pw[l] * f.ph[l] * (3 * esz + 4) + dst_px * (3 * esz + 4);
}

// launch one level of pyrDown for `count` images (descriptors by value, PD_MAXB per launch)
static void launch_pyr_down(const ssp_blender *b, int l, int src_depth, const PyrDownArgs *args, int count)
{
    static int force_rows = getenv("SSP_PD_ROWS") ? atoi(getenv("SSP_PD_ROWS")) : 0;
    // measured on MI355X: 1 row per lane wins at every level (more waves beats vertical reuse); 2 and 4 kept for tuning
    int rows = 1;
    if (force_rows == 1 || force_rows == 2 || force_rows == 4) rows = force_rows;
    for (int base = 0; base < count; base += PD_MAXB) {
        const int cnt = std::min(PD_MAXB, count - base);
        PyrDownBatch batch;
        memset(&batch, 0, sizeof batch);
        int max_w = 0, max_h = 0;
        for (int i = 0; i < cnt; ++i) {
            batch.a[i] = args[base + i];
            max_w = std::max(max_w, args[base + i].dwid);
            max_h = std::max(max_h, args[base + i].dhei);
        }
        dim3 grid((max_w + 63) / 64, (max_h + 4 * rows - 1) / (4 * rows), cnt), block(256);
#define PD_LAUNCH(L0, ST, FLT)                                                                                     \
    do {                                                                                                           \
        if (rows == 4) hipLaunchKernelGGL((k_pyr_down<L0, ST, FLT, 4>), grid, block, 0, stream(), batch);            \
        else if (rows == 2) hipLaunchKernelGGL((k_pyr_down<L0, ST, FLT, 2>), grid, block, 0, stream(), batch);       \
        else hipLaunchKernelGGL((k_pyr_down<L0, ST, FLT, 1>), grid, block, 0, stream(), batch);                      \
    } while (0)
        if (l == 0) {
            if (src_depth == SSP_U8) PD_LAUNCH(true, uint8_t, false);
            else if (src_depth == SSP_S16) PD_LAUNCH(true, int16_t, false);
            else PD_LAUNCH(true, float, true);
        } else {
            if (b->float_mode) PD_LAUNCH(false, float, true);
            else PD_LAUNCH(false, int16_t, false);
        }
#undef PD_LAUNCH
    }
}

static int feed_multiband(ssp_blender *b, ssp_image *img, ssp_image *mask, int tlx, int tly)
{
    FeedRec f;
    SSP_TRY(make_feed_rec(b, img, mask, tlx, tly, f));
    for (int l = 0; l < b->num_bands; ++l) {
        PyrDownArgs a;
        fill_pyr_args(b, f, l, a);
        ProfileScope ps(l == 0 ? "pyr_down_l0" : "pyr_down", pyr_bytes(b, f, l));
        launch_pyr_down(b, l, img->depth, &a, 1);
    }
    SSP_HIP(hipGetLastError());
    img->refs++;
    mask->refs++;
    b->feeds.push_back(f);
    return 0;
}

// same as n calls of feed(), but every pyramid level of all images is ONE launch (blockIdx.z = image)
static int feed_multiband_batch(ssp_blender *b, int n, ssp_image *const *imgs, ssp_image *const *masks, const int *tls)
{
    const int nb = b->num_bands;
    std::vector<FeedRec> recs(n);
    int rc = 0, made = 0;
    for (; made < n && !rc; ++made) rc = make_feed_rec(b, imgs[made], masks[made], tls[2 * made], tls[2 * made + 1], recs[made]);
    if (rc) {
        for (int i = 0; i < made - 1; ++i)
            for (int l = 1; l <= nb; ++l) { pool_free(recs[i].G[l]); pool_free(recs[i].W[l]); }
        return rc;
    }
    std::vector<PyrDownArgs> args(n);
    for (int l = 0; l < nb; ++l) {
        double bytes_l = 0;
        for (int i = 0; i < n; ++i) {
            fill_pyr_args(b, recs[i], l, args[i]);
            bytes_l += pyr_bytes(b, recs[i], l);
        }
        ProfileScope ps(l == 0 ? "pyr_down_l0" : "pyr_down", bytes_l);
        launch_pyr_down(b, l, imgs[0]->depth, args.data(), n);
    }
    SSP_HIP(hipGetLastError());
    for (int i = 0; i < n; ++i) {
        imgs[i]->refs++;
        masks[i]->refs++;
        b->feeds.push_back(recs[i]);
    }
    return 0;
}

// Run the per-level gather kernels.
//   region: level-0 rectangle (pano-relative, multiples of 2^nb) to compute, or null for the whole padded pano.
//   export_level >= 0: only write the raw sums of that level's part of the region into exp_lap/exp_w (tightly packed).
//   outputs (result/rmask/mosaic) have their pixel (0,0) at the region origin.
static int run_levels(ssp_blender *b, ssp_image *result, ssp_image *rmask, ssp_image *mosaic, int export_level, const int *region, void *exp_lap,
                      float *exp_w)
{
    const int nb = b->num_bands, n = (int)b->feeds.size();
    const int esz = b->float_mode ? 4 : 2;
    int reg[4] = {0, 0, b->lw[0], b->lh[0]};
    if (region) memcpy(reg, region, sizeof reg);
    const int m = 1 << nb;
    SSP_REQUIRE(reg[0] % m == 0 && reg[1] % m == 0 && reg[2] % m == 0 && reg[3] % m == 0 && reg[0] >= 0 && reg[1] >= 0 && reg[2] > 0 && reg[3] > 0 &&
                    reg[0] + reg[2] <= b->lw[0] && reg[1] + reg[3] <= b->lh[0],
                "blend region (%d,%d %dx%d) must be inside the padded pano and aligned to %d", reg[0], reg[1], reg[2], reg[3], m);
    // image descriptors for every level: pinned staging owned by the blender, uploaded asynchronously
    const size_t cnt = (size_t)std::max(1, n) * (nb + 1);
    int slot = 0;
    void *hv = nullptr, *dv = nullptr;
    SSP_TRY(b->ring.acquire(sizeof(LevelImg) * cnt, &hv, &dv, &slot));
    LevelImg *h_imgs = (LevelImg *)hv, *d_imgs = (LevelImg *)dv;
    for (int l = 0; l <= nb; ++l)
        for (int i = 0; i < n; ++i) {
            const FeedRec &f = b->feeds[i];
            LevelImg &li = h_imgs[(size_t)l * n + i];
            li.g = l == 0 ? f.img->data : f.G[l]; li.gp = l == 0 ? f.img->pitch : f.gp[l];
            li.gn = l < nb ? f.G[l + 1] : nullptr; li.gnp = l < nb ? f.gp[l + 1] : 0;
            li.w = l == 0 ? f.mask->data : (void *)f.W[l]; li.wp = l == 0 ? f.mask->pitch : f.wp[l];
            li.rx = f.rx[l]; li.ry = f.ry[l]; li.pw = f.pw[l]; li.ph = f.ph[l];
            li.pwn = l < nb ? f.pw[l + 1] : 0; li.phn = l < nb ? f.ph[l + 1] : 0;
            li.pl = f.pl;
            li.src_depth = f.img->depth;
        }
    SSP_TRY(b->ring.commit(slot, sizeof(LevelImg) * cnt));

    void *coll[MAX_BANDS + 1] = {nullptr};
    size_t cp[MAX_BANDS + 1] = {0};
    int rc = 0;
    const int l_first = export_level >= 0 ? export_level : nb, l_last = export_level >= 0 ? export_level : 0;
    for (int l = l_first; l >= l_last && !rc; --l) {
        LevelArgs a;
        memset(&a, 0, sizeof a);
        a.imgs = d_imgs + (size_t)l * n;
        a.n_imgs = n;
        a.lw = b->lw[l]; a.lh = b->lh[l];
        a.cx0 = reg[0] >> l; a.cy0 = reg[1] >> l; a.cw = reg[2] >> l; a.ch = reg[3] >> l;
        a.top = l == nb;
        if (export_level < 0) {
            if (l < nb) {
                a.parent = coll[l + 1]; a.pp = cp[l + 1]; a.pw = b->lw[l + 1]; a.ph = b->lh[l + 1];
                a.px0 = reg[0] >> (l + 1); a.py0 = reg[1] >> (l + 1); a.prw = reg[2] >> (l + 1); a.prh = reg[3] >> (l + 1);
            }
            if (l > 0) {
                cp[l] = align_up((size_t)a.cw * 3 * esz, 16);
                rc = pool_alloc(cp[l] * a.ch, &coll[l]);
                if (rc) break;
                a.out = coll[l]; a.op = cp[l];
            } else {
                a.fw = b->final_roi[2]; a.fh = b->final_roi[3];
                a.ox0 = reg[0]; a.oy0 = reg[1];
                if (result) { a.result = result->data; a.rp = result->pitch; }
                if (rmask) { a.rmask = (uint8_t *)rmask->data; a.rmp = rmask->pitch; }
                if (mosaic) { a.mosaic = (uint8_t *)mosaic->data; a.mp = mosaic->pitch; }
            }
        } else {
            a.export_mode = 1;
            a.exp_lap = exp_lap; a.exp_w = exp_w;
        }
        if (b->ext_lap[l]) { a.ext_lap = b->ext_lap[l]->data; a.elp = b->ext_lap[l]->pitch; a.ext_w = (const float *)b->ext_w[l]->data; a.ewp = b->ext_w[l]->pitch; }
        // algorithmic bytes: every covering image's level samples read once, parent level read once, outputs written once
        double cover = 0;
        for (int i = 0; i < n; ++i) cover += (double)b->feeds[i].pw[l] * b->feeds[i].ph[l];
        double px = (double)a.cw * a.ch;
        double in_b = l == 0 ? 0 : cover * (3 * esz + 4);
        if (l == 0) for (int i = 0; i < n; ++i) in_b += (double)b->feeds[i].img->w * b->feeds[i].img->h * (3.0 * depth_size(b->feeds[i].img->depth) + 1);
        if (l < nb) in_b += cover / 4 * 3 * esz + px / 4 * 3 * esz;
        double out_b = l > 0 ? px * 3 * esz : (double)std::min(a.fw, reg[0] + reg[2]) * std::min(a.fh, reg[1] + reg[3]) * ((result ? 3 * esz : 0) + (rmask ? 1 : 0) + (mosaic ? 3 : 0));
        ProfileScope ps(l == 0 ? "blend_level0" : "blend_level", in_b + out_b);
        if (l == nb) {
            // top level: per-pixel kernel (also used when nb == 0, where level 0 is the top)
            dim3 grid((a.cw + 63) / 64, (a.ch + 3) / 4), block(256);
            if (nb == 0 && !a.export_mode) {
                rc = set_error(SSP_ERR_ARG, "multiband blending with 0 bands is not supported on this path (use Blender_NO)");
                break;
            }
            if (l == 0) {
                if (b->float_mode) hipLaunchKernelGGL((k_blend_level<true, true>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_level<true, false>), grid, block, 0, stream(), a);
            } else {
                if (b->float_mode) hipLaunchKernelGGL((k_blend_level<false, true>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_level<false, false>), grid, block, 0, stream(), a);
            }
        } else {
            dim3 grid((a.cw + 63) / 64, (a.ch + 15) / 16), block(256);
            if (l == 0) {
                if (b->float_mode) hipLaunchKernelGGL((k_blend_quad<true, true>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_quad<true, false>), grid, block, 0, stream(), a);
            } else {
                if (b->float_mode) hipLaunchKernelGGL((k_blend_quad<false, true>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_quad<false, false>), grid, block, 0, stream(), a);
            }
        }
    }
    for (int l = 1; l <= nb; ++l) pool_free(coll[l]);
    SSP_TRY(b->ring.release(slot));  // the kernels above are the last readers of this slot
    if (rc) return rc;
    SSP_HIP(hipGetLastError());
    return 0;
}

}  // namespace ssp

// ---- C ABI ---------------------------------------------------------------------------------------------------------
SSP_API int ssp_blender_create(int type, ssp_blender **out)
{
    SSP_REQUIRE(out && type >= SSP_BLEND_NO && type <= SSP_BLEND_MULTIBAND, "blender: unknown type %d", type);
    ssp_blender *b = new ssp_blender();
    b->type = type;
    *out = b;
    return 0;
}
SSP_API int ssp_blender_destroy(ssp_blender *b)
{
    if (b) {
        release_state(b);
        b->ring.destroy();
        delete b;
    }
    return 0;
}
SSP_API int ssp_blender_set_num_bands(ssp_blender *b, int n)
{
    SSP_REQUIRE(b && n >= 0 && n <= MAX_BANDS - 1, "setNumBands: %d out of range", n);
    b->want_bands = n;
    return 0;
}
SSP_API int ssp_blender_get_num_bands(const ssp_blender *b, int *n)
{
    SSP_REQUIRE(b && n, "numBands: null");
    *n = b->type == SSP_BLEND_MULTIBAND ? (b->prepared ? b->num_bands : b->want_bands) : 0;
    return 0;
}
SSP_API int ssp_blender_set_sharpness(ssp_blender *b, float s) { SSP_REQUIRE(b, "null"); b->sharpness = s; return 0; }
SSP_API int ssp_blender_set_float_mode(ssp_blender *b, int on)
{
    SSP_REQUIRE(b, "null");
    SSP_REQUIRE(!on || b->type == SSP_BLEND_MULTIBAND, "float mode exists for the multiband blender only");
    b->float_mode = on != 0;
    return 0;
}

SSP_API int ssp_blender_prepare(ssp_blender *b, int x, int y, int w, int h)
{
    SSP_REQUIRE(b && w > 0 && h > 0, "prepare: empty roi %dx%d", w, h);
    SSP_TRY(ensure_init());
    release_state(b);
    b->final_roi[0] = x; b->final_roi[1] = y; b->final_roi[2] = w; b->final_roi[3] = h;
    if (b->type == SSP_BLEND_MULTIBAND) {
        double max_len = (double)std::max(w, h);
        b->num_bands = std::min(b->want_bands, (int)ceil(std::log(max_len) / std::log(2.0)));
        int m = 1 << b->num_bands;
        w += (m - w % m) % m;
        h += (m - h % m) % m;
        b->lw[0] = w; b->lh[0] = h;
        for (int l = 1; l <= b->num_bands; ++l) { b->lw[l] = (b->lw[l - 1] + 1) / 2; b->lh[l] = (b->lh[l - 1] + 1) / 2; }
    } else {
        SSP_TRY(image_new(w, h, 3, SSP_S16, &b->dst));
        SSP_TRY(ssp_image_fill(b->dst, 0));
        if (b->type == SSP_BLEND_NO) {
            SSP_TRY(image_new(w, h, 1, SSP_U8, &b->dst_mask));
            SSP_TRY(ssp_image_fill(b->dst_mask, 0));
        } else {
            SSP_TRY(image_new(w, h, 1, SSP_F32, &b->dst_weight));
            SSP_TRY(ssp_image_fill(b->dst_weight, 0));
        }
    }
    b->roi[0] = x; b->roi[1] = y; b->roi[2] = w; b->roi[3] = h;
    b->prepared = true;
    return 0;
}

SSP_API int ssp_blender_feed(ssp_blender *b, ssp_image *img, ssp_image *mask, int tlx, int tly)
{
    SSP_REQUIRE(b && img && mask, "feed: null argument");
    if (!b->prepared) SSP_FAIL(SSP_ERR_STATE, "feed called before prepare (or after blend)");
    SSP_REQUIRE(img->cn == 3, "feed: image must have 3 channels (CV_16SC3)");
    SSP_REQUIRE(mask->cn == 1 && mask->depth == SSP_U8, "feed: mask must be CV_8U");
    SSP_REQUIRE(mask->w == img->w && mask->h == img->h, "feed: mask %dx%d differs from image %dx%d", mask->w, mask->h, img->w, img->h);
    if (b->type == SSP_BLEND_MULTIBAND) {
        if (b->float_mode) SSP_REQUIRE(img->depth == SSP_F32, "feed: float mode needs CV_32FC3 images");
        else SSP_REQUIRE(img->depth == SSP_S16 || img->depth == SSP_U8, "feed: image must be CV_16SC3 or CV_8UC3");
        return feed_multiband(b, img, mask, tlx, tly);
    }
    SSP_REQUIRE(img->depth == SSP_S16 || img->depth == SSP_U8, "feed: image must be CV_16SC3 (or 8UC3 holding the same values)");
    int dx = tlx - b->roi[0], dy = tly - b->roi[1];
    SSP_REQUIRE(dx >= 0 && dy >= 0 && dx + img->w <= b->roi[2] && dy + img->h <= b->roi[3], "feed: image outside the prepared roi");
    dim3 grid((img->w + 255) / 256, img->h), block(256);
    if (b->type == SSP_BLEND_NO) {
        ProfileScope ps("feed_plain", (double)img->w * img->h * (3.0 * depth_size(img->depth) + 1 + 6 + 2));
        if (img->depth == SSP_S16)
            hipLaunchKernelGGL(k_feed_plain<int16_t>, grid, block, 0, stream(), (const int16_t *)img->data, img->pitch, (const uint8_t *)mask->data, mask->pitch, img->w,
                               img->h, (int16_t *)b->dst->data, b->dst->pitch, (uint8_t *)b->dst_mask->data, b->dst_mask->pitch, dx, dy);
        else
            hipLaunchKernelGGL(k_feed_plain<uint8_t>, grid, block, 0, stream(), (const uint8_t *)img->data, img->pitch, (const uint8_t *)mask->data, mask->pitch, img->w,
                               img->h, (int16_t *)b->dst->data, b->dst->pitch, (uint8_t *)b->dst_mask->data, b->dst_mask->pitch, dx, dy);
    } else {
        ssp_image *dist = nullptr, *wm = nullptr;
        SSP_TRY(image_new(img->w, img->h, 1, SSP_F32, &wm));
        int rc = image_new(img->w, img->h, 1, SSP_F32 /* int32 storage */, &dist);
        if (rc) { image_unref(wm); return rc; }
        {
            ProfileScope ps("feather_distance", (double)img->w * img->h * (1 + 4 * 4 + 4));
            hipLaunchKernelGGL(k_dist_rows, dim3((img->h + 63) / 64), dim3(64), 0, stream(), (const uint8_t *)mask->data, mask->pitch, img->w, img->h, (int *)dist->data,
                               dist->pitch);
            hipLaunchKernelGGL(k_dist_cols, dim3((img->w + 63) / 64), dim3(64), 0, stream(), (int *)dist->data, dist->pitch, img->w, img->h, b->sharpness,
                               (float *)wm->data, wm->pitch);
        }
        {
            ProfileScope ps("feed_feather", (double)img->w * img->h * (3.0 * depth_size(img->depth) + 4 + 12 + 8));
            if (img->depth == SSP_S16)
                hipLaunchKernelGGL(k_feed_feather<int16_t>, grid, block, 0, stream(), (const int16_t *)img->data, img->pitch, (const float *)wm->data, wm->pitch, img->w,
                                   img->h, (int16_t *)b->dst->data, b->dst->pitch, (float *)b->dst_weight->data, b->dst_weight->pitch, dx, dy);
            else
                hipLaunchKernelGGL(k_feed_feather<uint8_t>, grid, block, 0, stream(), (const uint8_t *)img->data, img->pitch, (const float *)wm->data, wm->pitch, img->w,
                                   img->h, (int16_t *)b->dst->data, b->dst->pitch, (float *)b->dst_weight->data, b->dst_weight->pitch, dx, dy);
        }
        image_unref(dist);
        image_unref(wm);
    }
    SSP_HIP(hipGetLastError());
    return 0;
}

SSP_API int ssp_blender_feed_batch(ssp_blender *b, int n, ssp_image *const *imgs, ssp_image *const *masks, const int *tls_xy)
{
    SSP_REQUIRE(b && n > 0 && imgs && masks && tls_xy, "feed_batch: bad arguments");
    if (!b->prepared) SSP_FAIL(SSP_ERR_STATE, "feed called before prepare (or after blend)");
    bool same = b->type == SSP_BLEND_MULTIBAND;
    for (int i = 0; i < n; ++i) {
        SSP_REQUIRE(imgs[i] && masks[i], "feed_batch: null image %d", i);
        if (imgs[i]->depth != imgs[0]->depth || imgs[i]->cn != 3 || masks[i]->cn != 1 || masks[i]->depth != SSP_U8 || masks[i]->w != imgs[i]->w ||
            masks[i]->h != imgs[i]->h)
            same = false;
    }
    if (same && b->float_mode && imgs[0]->depth != SSP_F32) same = false;
    if (same && !b->float_mode && imgs[0]->depth == SSP_F32) same = false;
    if (!same) {  // mixed types or another blender: the per-image path validates and reports
        for (int i = 0; i < n; ++i) SSP_TRY(ssp_blender_feed(b, imgs[i], masks[i], tls_xy[2 * i], tls_xy[2 * i + 1]));
        return 0;
    }
    return feed_multiband_batch(b, n, imgs, masks, tls_xy);
}

SSP_API int ssp_blender_blend(ssp_blender *b, ssp_image **result, ssp_image **result_mask, ssp_image **mosaic)
{
    SSP_REQUIRE(b, "blend: null blender");
    if (!b->prepared) SSP_FAIL(SSP_ERR_STATE, "blend called before prepare, or twice (the blender state is consumed by blend)");
    const int fw = b->final_roi[2], fh = b->final_roi[3];
    ssp_image *res = nullptr, *rm = nullptr, *mo = nullptr;
    int rc = 0;
    if (result) rc = image_new(fw, fh, 3, b->float_mode ? SSP_F32 : SSP_S16, &res);
    if (!rc && result_mask) rc = image_new(fw, fh, 1, SSP_U8, &rm);
    if (!rc && mosaic) rc = image_new(fw, fh, 3, SSP_U8, &mo);
    if (!rc) {
        if (b->type == SSP_BLEND_MULTIBAND) {
            rc = run_levels(b, res, rm, mo, -1, nullptr, nullptr, nullptr);
        } else {
            ProfileScope ps("blend_finish", (double)fw * fh * (6 + 4 + (res ? 6 : 0) + (rm ? 1 : 0) + (mo ? 3 : 0)));
            hipLaunchKernelGGL(k_finish_plain, dim3((fw + 255) / 256, fh), dim3(256), 0, stream(), (const int16_t *)b->dst->data, b->dst->pitch,
                               b->dst_mask ? (const uint8_t *)b->dst_mask->data : nullptr, b->dst_mask ? b->dst_mask->pitch : 0,
                               b->dst_weight ? (const float *)b->dst_weight->data : nullptr, b->dst_weight ? b->dst_weight->pitch : 0, fw, fh,
                               res ? (int16_t *)res->data : nullptr, res ? res->pitch : 0, rm ? (uint8_t *)rm->data : nullptr, rm ? rm->pitch : 0,
                               mo ? (uint8_t *)mo->data : nullptr, mo ? mo->pitch : 0);
            if (hipGetLastError() != hipSuccess) rc = set_error(SSP_ERR_DEVICE, "blend_finish launch failed");
        }
    }
    if (rc) { image_unref(res); image_unref(rm); image_unref(mo); return rc; }
    release_state(b);  // OpenCV releases dst_/dst_mask_ and the pyramids in blend()
    if (result) *result = res;
    if (result_mask) *result_mask = rm;
    if (mosaic) *mosaic = mo;
    return 0;
}

SSP_API int ssp_blender_level_info(const ssp_blender *b, int level, int *w, int *h)
{
    SSP_REQUIRE(b && b->prepared && b->type == SSP_BLEND_MULTIBAND && level >= 0 && level <= b->num_bands, "level_info: no such level");
    *w = b->lw[level]; *h = b->lh[level];
    return 0;
}

SSP_API int ssp_blender_export_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, void *lap, void *wgt)
{
    SSP_REQUIRE(b && lap && wgt, "export_partial: null argument");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "export_partial needs a prepared multiband blender");
    SSP_REQUIRE(level >= 0 && level <= b->num_bands, "export_partial: no level %d", level);
    int rect[4] = {x0, y0, w, h};
    return run_levels(b, nullptr, nullptr, nullptr, level, rect, lap, (float *)wgt);
}

SSP_API int ssp_blender_import_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, const void *lap, const void *wgt)
{
    SSP_REQUIRE(b && lap && wgt, "import_partial: null argument");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "import_partial needs a prepared multiband blender");
    const int m = 1 << b->num_bands;
    SSP_REQUIRE(level >= 0 && level <= b->num_bands && x0 >= 0 && y0 >= 0 && w > 0 && h > 0 && x0 % m == 0 && y0 % m == 0 && w % m == 0 && h % m == 0 &&
                    x0 + w <= b->lw[0] && y0 + h <= b->lh[0],
                "import_partial: region (%d,%d %dx%d) must be inside the padded pano and aligned to %d", x0, y0, w, h, m);
    if (!b->ext_lap[level]) {
        SSP_TRY(image_new(b->lw[level], b->lh[level], 3, b->float_mode ? SSP_F32 : SSP_S16, &b->ext_lap[level]));
        SSP_TRY(image_new(b->lw[level], b->lh[level], 1, SSP_F32, &b->ext_w[level]));
        SSP_TRY(ssp_image_fill(b->ext_lap[level], 0));
        SSP_TRY(ssp_image_fill(b->ext_w[level], 0));
    }
    const int lx = x0 >> level, ly = y0 >> level, lw = w >> level, lh = h >> level;
    hipLaunchKernelGGL(k_add_partial, dim3((lw + 255) / 256, lh), dim3(256), 0, stream(), b->ext_lap[level]->data, b->ext_lap[level]->pitch,
                       (float *)b->ext_w[level]->data, b->ext_w[level]->pitch, lap, (const float *)wgt, lx, ly, lw, lh, b->float_mode ? 1 : 0);
    SSP_HIP(hipGetLastError());
    return 0;
}

// blend only a sub-rectangle of the pano (multi-GPU: every GPU collapses the region its own frames cover)
SSP_API int ssp_blender_blend_region(ssp_blender *b, int x0, int y0, int w, int h, ssp_image **result, ssp_image **result_mask, ssp_image **mosaic)
{
    SSP_REQUIRE(b, "blend_region: null blender");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "blend_region needs a prepared multiband blender");
    const int ow = std::min(x0 + w, b->final_roi[2]) - x0, oh = std::min(y0 + h, b->final_roi[3]) - y0;
    SSP_REQUIRE(ow > 0 && oh > 0, "blend_region: region outside the final roi");
    ssp_image *res = nullptr, *rm = nullptr, *mo = nullptr;
    int rc = 0;
    if (result) rc = image_new(ow, oh, 3, b->float_mode ? SSP_F32 : SSP_S16, &res);
    if (!rc && result_mask) rc = image_new(ow, oh, 1, SSP_U8, &rm);
    if (!rc && mosaic) rc = image_new(ow, oh, 3, SSP_U8, &mo);
    int rect[4] = {x0, y0, w, h};
    if (!rc) rc = run_levels(b, res, rm, mo, -1, rect, nullptr, nullptr);
    if (rc) { image_unref(res); image_unref(rm); image_unref(mo); return rc; }
    release_state(b);
    if (result) *result = res;
    if (result_mask) *result_mask = rm;
    if (mosaic) *mosaic = mo;
    return 0;
}
