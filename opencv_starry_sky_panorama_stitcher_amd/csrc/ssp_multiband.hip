// ssp_multiband.hip -- cv.detail.MultiBandBlender on gfx950.
//
// Replaces (stitching_detailed_enhanced.py): :1813-1815 detail_MultiBandBlender().setNumBands, :1820 prepare,
// :1886 feed(image_warped_s, mask_warped, corner), :1930 blend(None, None).
//
// MI355X-first restructuring (same arithmetic, different schedule):
//   * OpenCV feeds image by image: copyMakeBorder, Laplacian pyramid, then a read-modify-write of the pano-sized
//     accumulators at every level, and blend() re-reads them.  Here feed() only builds the image's Gaussian pyramids
//     (G_1..G_nb int16x3, W_1..W_nb f32); blend() runs ONE kernel per pano level, top first: each 2x2 output quad gathers
//     every image that covers it (feed order), forms G_l - pyrUp(G_{l+1}) on the fly, accumulates (short)(L*w) and w in
//     registers, normalises, adds pyrUp of the collapsed parent level and stores the collapsed level once.
//   * Borders are materialised once instead of being re-derived per tap: level 0 is stored with the border
//     copyMakeBorder would add (the warp kernel writes the interior in place, k_border0 fills the rest) and every
//     level carries a 4-pixel BORDER_REFLECT_101 apron, so the pyramid kernels have no border logic.  These kernels
//     are bound by the texture addresser (about 36 cycles per vector memory instruction of a wave, whatever its width),
//     so they are organised around few, wide loads: one lane = a 2x2 block of outputs from 7 rows x 8 pixels.
//   Integer sums wrap mod 2^16 exactly as C "short +=" does and the float weight sums are taken in feed order, so the
//   results are bit-identical to the sequential formulation.
#include <type_traits>

#include "ssp_blender.hpp"

#include <algorithm>

using namespace ssp;

#define WEIGHT_EPS 1e-5f
#define MAX_BANDS SSP_MAX_BANDS
#define APRON SSP_APRON

// This file is compiled twice (Makefile).  ssp_multiband.o: everything.  ssp_multiband_f32.o (-DSSP_MB_F32_TU -fno-slp-vectorize): the kernels only, with
// internal linkage, and the launchers of the float-pyramid kernels at the end of the file.  The SLP vectoriser packs the float kernels' arithmetic into
// v_pk_*_f32 (half rate on gfx950, plus the moves that pair the operands): config 5's level-0 blend 2 195 -> 2 081 us without it -- while the integer
// kernels, whose weight sums it packs, are 6 % faster WITH it.  One flag per translation unit, so the float kernels get a unit of their own.
#ifdef SSP_MB_F32_TU
namespace {
#endif

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

// three numerators over one denominator, correctly rounded: the instruction sequence of an IEEE float division (v_rcp, two
// fma to refine the reciprocal, q = n*y, two residual corrections) with the reciprocal shared.  Valid without
// v_div_scale / v_div_fixup because 1e-5 <= den < 2^24 and |n| <= 32768 keep every intermediate in the normal range.
__device__ inline void div3_exact(float n0, float n1, float n2, float den, float q[3])
{
    float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.f);
    r = __builtin_fmaf(e, r, r);
    const float nn[3] = {n0, n1, n2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = nn[c] * r;
        float t = __builtin_fmaf(-den, v, nn[c]);
        v = __builtin_fmaf(t, r, v);
        t = __builtin_fmaf(-den, v, nn[c]);
        q[c] = __builtin_fmaf(t, r, v);
    }
}

// 3-channel pixel load
template <typename ST, typename VT>
__device__ inline void load_px(const void *base, size_t pitch, int x, int y, VT out[3])
{
    const ST *p = (const ST *)((const char *)base + (ptrdiff_t)y * (ptrdiff_t)pitch) + (ptrdiff_t)x * 3;
    out[0] = (VT)p[0];
    out[1] = (VT)p[1];
    out[2] = (VT)p[2];
}

typedef uint32_t u32x2_u1 __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_u1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x3_u1 __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint16_t u16_q1 __attribute__((aligned(1)));
typedef uint32_t u32_q2 __attribute__((aligned(2)));
typedef uint32_t u32_u1 __attribute__((aligned(1)));

// ====================================================================================================================
// level-0 border: everything of the bordered plane outside the image interior
// ====================================================================================================================
struct Border0Desc {
    char *g; size_t gp;        // level-0 image plane, element (0,0) = padded-rectangle origin
    uint8_t *m; size_t mp;     // level-0 mask plane
    int iw, ih, left, top, pw, ph, depth;
};
#define MB_MAXB 24
// Batched launches run over a 1-D grid of tiles: tile t belongs to the image z with start[z] <= t < start[z+1], its position inside
// the image is (l % tx[z], l / tx[z]) with l = t - start[z].  Images of different sizes (own frames and the strips of other GPUs'
// frames) then cost exactly their own tiles -- a (max_w, max_h, n) grid launches mostly empty blocks for the small ones.
struct TileMap { int cnt; int start[MB_MAXB + 1]; int tx[MB_MAXB]; };
__device__ inline void tile_locate(const TileMap &m, int t, int &z, int &bx, int &by)
{
    z = 0;
    while (z + 1 < m.cnt && t >= m.start[z + 1]) ++z;   // wave-uniform, <= 15 steps
    const int l = t - m.start[z];
    by = l / m.tx[z];
    bx = l - by * m.tx[z];
}
struct Border0Batch { Border0Desc d[MB_MAXB]; TileMap tm; };

template <typename ST>
__device__ inline void border0_pixel(const Border0Desc &d, int X, int Y)
{
    // apron: BORDER_REFLECT_101 of the padded rectangle; band: BORDER_REFLECT of the image, weight 0
    int xi = reflect101_idx(X, d.pw) - d.left, yi = reflect101_idx(Y, d.ph) - d.top;
    const bool inside = (unsigned)xi < (unsigned)d.iw && (unsigned)yi < (unsigned)d.ih;
    const int sx = reflect_idx(xi, d.iw) + d.left, sy = reflect_idx(yi, d.ih) + d.top;
    const ST *s = (const ST *)(d.g + (ptrdiff_t)sy * (ptrdiff_t)d.gp) + (ptrdiff_t)sx * 3;
    ST *t = (ST *)(d.g + (ptrdiff_t)Y * (ptrdiff_t)d.gp) + (ptrdiff_t)X * 3;
    t[0] = s[0]; t[1] = s[1]; t[2] = s[2];
    d.m[(ptrdiff_t)Y * (ptrdiff_t)d.mp + X] = inside ? d.m[(ptrdiff_t)sy * (ptrdiff_t)d.mp + sx] : (uint8_t)0;
}

__global__ __launch_bounds__(256) void k_border0(const Border0Batch batch)
{
    const Border0Desc &d = batch.d[blockIdx.z];
    const int A = APRON;
    const int wt = d.pw + 2 * A, n_top = A + d.top, n_bot = d.ph + A - (d.top + d.ih), n_left = A + d.left, n_right = d.pw + A - (d.left + d.iw);
    const long long s0 = (long long)wt * n_top, s1 = s0 + (long long)wt * n_bot, s2 = s1 + (long long)d.ih * n_left, s3 = s2 + (long long)d.ih * n_right;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s3) return;
    int X, Y;
    if (t < s0) { Y = (int)(t / wt) - A; X = (int)(t % wt) - A; }
    else if (t < s1) { t -= s0; Y = d.top + d.ih + (int)(t / wt); X = (int)(t % wt) - A; }
    else if (t < s2) { t -= s1; Y = d.top + (int)(t / n_left); X = (int)(t % n_left) - A; }
    else { t -= s2; Y = d.top + (int)(t / n_right); X = d.left + d.iw + (int)(t % n_right); }
    if (d.depth == SSP_U8) border0_pixel<uint8_t>(d, X, Y);
    else if (d.depth == SSP_S16) border0_pixel<int16_t>(d, X, Y);
    else border0_pixel<float>(d, X, Y);
}

// float planes (config 5): a pixel per lane like k_border0, with 32-bit index arithmetic (the host takes it when the item count fits) and the three
// channels as ONE 12-byte load and store -- the generic kernel's 64-bit divisions and per-channel accesses were what its 373 us on twelve 8K frames
// were made of (texture addresser: three times the instructions for the same bytes)
typedef float f32x3_b4 __attribute__((ext_vector_type(3), aligned(4)));
__global__ __launch_bounds__(256) void k_border0_f32(const Border0Batch batch)
{
    const Border0Desc &d = batch.d[blockIdx.z];
    const int A = APRON;
    const uint32_t wt = (uint32_t)(d.pw + 2 * A), n_top = (uint32_t)(A + d.top), n_bot = (uint32_t)(d.ph + A - (d.top + d.ih)), n_left = (uint32_t)(A + d.left),
                   n_right = (uint32_t)(d.pw + A - (d.left + d.iw));
    const uint32_t s0 = wt * n_top, s1 = s0 + wt * n_bot, s2 = s1 + (uint32_t)d.ih * n_left, s3 = s2 + (uint32_t)d.ih * n_right;
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= s3) return;
    int X, Y;
    if (t < s0) { const uint32_t q = t / wt; Y = (int)q - A; X = (int)(t - q * wt) - A; }
    else if (t < s1) { t -= s0; const uint32_t q = t / wt; Y = d.top + d.ih + (int)q; X = (int)(t - q * wt) - A; }
    else if (t < s2) { t -= s1; const uint32_t q = t / n_left; Y = d.top + (int)q; X = (int)(t - q * n_left) - A; }
    else { t -= s2; const uint32_t q = t / n_right; Y = d.top + (int)q; X = d.left + d.iw + (int)(t - q * n_right); }
    // apron: BORDER_REFLECT_101 of the padded rectangle; band: BORDER_REFLECT of the image, weight 0 (border0_pixel)
    const int xi = reflect101_idx(X, d.pw) - d.left, yi = reflect101_idx(Y, d.ph) - d.top;
    const bool inside = (unsigned)xi < (unsigned)d.iw && (unsigned)yi < (unsigned)d.ih;
    const int sx = reflect_idx(xi, d.iw) + d.left, sy = reflect_idx(yi, d.ih) + d.top;
    const f32x3_b4 v = *(const f32x3_b4 *)((const float *)(d.g + (ptrdiff_t)sy * (ptrdiff_t)d.gp) + (ptrdiff_t)sx * 3);
    *(f32x3_b4 *)((float *)(d.g + (ptrdiff_t)Y * (ptrdiff_t)d.gp) + (ptrdiff_t)X * 3) = v;
    d.m[(ptrdiff_t)Y * (ptrdiff_t)d.mp + X] = inside ? d.m[(ptrdiff_t)sy * (ptrdiff_t)d.mp + sx] : (uint8_t)0;
}

// 8-bit planes, 4 pixels per lane: every destination group is 4-byte aligned (plane origin + multiples of 4 columns); its sources
// are 4 consecutive image pixels in forward order (rows above / below the image) or in reverse order (columns left / right of
// it), read with one 12-byte load and put in order with v_perm_b32.  Groups that contain a reflection point, or that straddle
// the image interior, go pixel by pixel.
typedef uint32_t u32x3_b1 __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x3_b4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32_b1 __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void k_border0_u8x4(const Border0Batch batch)
{
    int z, bxl, byl;
    tile_locate(batch.tm, blockIdx.x, z, bxl, byl);
    const Border0Desc &d = batch.d[z];
    const int A = APRON;
    const int gw = (d.pw + 2 * A) / 4, n_top = A + d.top, n_bot = d.ph + A - (d.top + d.ih);
    const int gl = (d.left + A + 3) / 4, r0 = (d.left + d.iw) & ~3, gr = (d.pw + A - r0) / 4;
    // 32-bit group indices (the host takes this kernel only when the group count fits): the 64-bit divisions of the first version were most
    // of its 314 VALU instructions per wave
    const uint32_t s0 = (uint32_t)gw * (uint32_t)n_top, s1 = s0 + (uint32_t)gw * (uint32_t)n_bot, s2 = s1 + (uint32_t)d.ih * (uint32_t)gl, s3 = s2 + (uint32_t)d.ih * (uint32_t)gr;
    uint32_t t = ((uint32_t)bxl + (uint32_t)byl) * 256u + threadIdx.x;   // tx = 1 block per "row": bxl is always 0, byl the block index
    if (t >= s3) return;
    int X0, Y;
    if (t < s0) { const uint32_t q = t / (uint32_t)gw; Y = (int)q - A; X0 = (int)(t - q * (uint32_t)gw) * 4 - A; }
    else if (t < s1) { t -= s0; const uint32_t q = t / (uint32_t)gw; Y = d.top + d.ih + (int)q; X0 = (int)(t - q * (uint32_t)gw) * 4 - A; }
    else if (t < s2) { t -= s1; const uint32_t q = t / (uint32_t)gl; Y = d.top + (int)q; X0 = (int)(t - q * (uint32_t)gl) * 4 - A; }
    else { t -= s2; const uint32_t q = t / (uint32_t)gr; Y = d.top + (int)q; X0 = r0 + (int)(t - q * (uint32_t)gr) * 4; }
    const int yr = reflect101_idx(Y, d.ph) - d.top;
    const bool iny = (unsigned)yr < (unsigned)d.ih, rowin = Y >= d.top && Y < d.top + d.ih;
    const int sy = reflect_idx(yr, d.ih) + d.top;
    const uint8_t *srow = (const uint8_t *)d.g + (ptrdiff_t)sy * (ptrdiff_t)d.gp, *mrow = d.m + (ptrdiff_t)sy * (ptrdiff_t)d.mp;
    uint8_t *drow = (uint8_t *)d.g + (ptrdiff_t)Y * (ptrdiff_t)d.gp, *dmrow = d.m + (ptrdiff_t)Y * (ptrdiff_t)d.mp;
    // the common groups without any index reflection arithmetic: 4 columns inside the padded rectangle that all lie over the image
    // columns (rows above / below the image: forward copy), or all one reflection left / right of it (reversed copy)
    if (!rowin && X0 >= 0 && X0 + 3 < d.pw) {
        const int xi0 = X0 - d.left;
        if (xi0 >= 0 && xi0 + 3 < d.iw) {
            const u32x3_b1 v = *(const u32x3_b1 *)(srow + (ptrdiff_t)X0 * 3);
            u32x3_b4 o; o.x = v.x; o.y = v.y; o.z = v.z;
            *(u32x3_b4 *)(drow + (ptrdiff_t)X0 * 3) = o;
            *(uint32_t *)(dmrow + X0) = iny ? *(const u32_b1 *)(mrow + X0) : 0u;
            return;
        }
    }
    int sx[4];
    bool in[4], wr[4], all_wr = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X = X0 + k, xi = reflect101_idx(X, d.pw) - d.left;
        in[k] = iny && (unsigned)xi < (unsigned)d.iw;
        sx[k] = reflect_idx(xi, d.iw) + d.left;
        wr[k] = !(rowin && X >= d.left && X < d.left + d.iw);   // never touch the image interior
        all_wr = all_wr && wr[k];
    }
    const bool fwd = sx[3] - sx[0] == 3, rev = sx[0] - sx[3] == 3;   // the index maps have slope +-1: 3 apart means contiguous
    if (all_wr && (fwd || rev)) {
        const int base = fwd ? sx[0] : sx[3];
        const u32x3_b1 v = *(const u32x3_b1 *)(srow + (ptrdiff_t)base * 3);
        uint32_t mk = in[0] ? *(const u32_b1 *)(mrow + base) : 0u;   // inside-ness cannot change within a contiguous group
        u32x3_b4 o;
        if (fwd) { o.x = v.x; o.y = v.y; o.z = v.z; }
        else {
            // pixels P3 P2 P1 P0: bytes 9 10 11 6 | 7 8 3 4 | 5 0 1 2
            o.x = __builtin_amdgcn_perm(v.y, v.z, 0x06030201u);
            o.y = __builtin_amdgcn_perm(v.x, __builtin_amdgcn_perm(v.z, v.y, 0x000c0403u), 0x03070100u);
            o.z = __builtin_amdgcn_perm(v.y, v.x, 0x02010005u);
            mk = __builtin_amdgcn_perm(mk, mk, 0x00010203u);
        }
        *(u32x3_b4 *)(drow + (ptrdiff_t)X0 * 3) = o;
        *(uint32_t *)(dmrow + X0) = mk;
        return;
    }
    // groups with a reflection point inside, or that straddle the image's first / last column: one unaligned 4-byte read per pixel (its 3
    // bytes + one more that is masked off; rows carry slack behind them), assembled into the same 12-byte + 4-byte stores as above.
    // Pixels of the image interior that share the group are rewritten with their own values (their source index is themselves):
    // byte stores per pixel cost this kernel a third of its time, 16 store and 16 load instructions in every wave of the side bands.
    uint32_t px[4], mk = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        px[k] = *(const u32_b1 *)(srow + (ptrdiff_t)sx[k] * 3) & 0x00ffffffu;
        if (in[k]) mk |= (uint32_t)mrow[sx[k]] << (8 * k);
    }
    (void)wr; (void)all_wr;
    u32x3_b4 o;
    o.x = px[0] | (px[1] << 24);
    o.y = (px[1] >> 8) | (px[2] << 16);
    o.z = (px[2] >> 16) | (px[3] << 8);
    *(u32x3_b4 *)(drow + (ptrdiff_t)X0 * 3) = o;
    *(uint32_t *)(dmrow + X0) = mk;
}

// copy a fed image / mask into the interior of its bordered planes (object API; the composer's warp writes in place)
struct CopyDesc {
    const char *simg; size_t sip; const uint8_t *smask; size_t smp;
    char *dimg; size_t dip; uint8_t *dmask; size_t dmp;
    int w, h, bpp;  // bytes per pixel of the image (3 channels)
};
__global__ __launch_bounds__(256) void k_copy_interior(const CopyDesc c)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= c.w || y >= c.h) return;
    const char *s = c.simg + (size_t)y * c.sip + (size_t)x * c.bpp;
    char *d = c.dimg + (size_t)y * c.dip + (size_t)x * c.bpp;
    for (int k = 0; k < c.bpp; ++k) d[k] = s[k];
    c.dmask[(size_t)y * c.dmp + x] = c.smask[(size_t)y * c.smp + x];
}

// BORDER_REFLECT_101 apron of a pyramid level (levels >= 1): u8x3 / int16x3 / f32x3 image level and f32 weight level
struct ApronDesc { char *g; size_t gp; char *w; size_t wp; int pw, ph, gbytes; };
struct ApronBatch { ApronDesc d[MB_MAXB]; };
__global__ __launch_bounds__(256) void k_apron(const ApronBatch batch)
{
    const ApronDesc &d = batch.d[blockIdx.z];
    const int A = APRON, wt = d.pw + 2 * A;
    const long long s0 = (long long)wt * A, s1 = 2 * s0, s2 = s1 + (long long)d.ph * A, s3 = s2 + (long long)d.ph * A;
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s3) return;
    int X, Y;
    if (t < s0) { Y = (int)(t / wt) - A; X = (int)(t % wt) - A; }
    else if (t < s1) { t -= s0; Y = d.ph + (int)(t / wt); X = (int)(t % wt) - A; }
    else if (t < s2) { t -= s1; Y = (int)(t / A); X = (int)(t % A) - A; }
    else { t -= s2; Y = (int)(t / A); X = d.pw + (int)(t % A); }
    const int sx = reflect101_idx(X, d.pw), sy = reflect101_idx(Y, d.ph);
    const char *s = d.g + (ptrdiff_t)sy * (ptrdiff_t)d.gp + (ptrdiff_t)sx * d.gbytes;
    char *q = d.g + (ptrdiff_t)Y * (ptrdiff_t)d.gp + (ptrdiff_t)X * d.gbytes;
    if (d.gbytes & 1) for (int k = 0; k < d.gbytes; ++k) q[k] = s[k];                      // 8-bit levels: 3 bytes
    else for (int k = 0; k < d.gbytes; k += 2) *(uint16_t *)(q + k) = *(const uint16_t *)(s + k);
    *(float *)(d.w + (ptrdiff_t)Y * (ptrdiff_t)d.wp + (ptrdiff_t)X * 4) = *(const float *)(d.w + (ptrdiff_t)sy * (ptrdiff_t)d.wp + (ptrdiff_t)sx * 4);
}

// ====================================================================================================================
// pyrDown: 5-tap [1 4 6 4 1] both axes, dst = n/2; image (3 channels) and weight map together.  Every tap is in
// memory (border + apron), so there is no border logic.  SRC: 0 = level 0 from u8 image + u8 mask, 1 = level 0 from
// int16 image + u8 mask, 2 = int16 level + f32 weights, 3 = u8 level + f32 weights.
// A pyramid fed from an 8-bit frame keeps EVERY Gaussian level in 8 bits (SRC 0 and 3 write u8x3 levels): pyrDown's
// (sum of 256 weights x [0, 255] + 128) >> 8 stays within [0, 255], so the int16 OpenCV stores carries no more information -- and
// these kernels and the blend kernels that read the levels back are bound by bytes (3 + 4 per sample instead of 6 + 4).
#define PYR_SRC_U8(SRC) ((SRC) == 0 || (SRC) == 3)
// ====================================================================================================================
struct PyrDownArgs {
    const char *g; size_t gp;   // source image level (element (0,0))
    const char *w; size_t wp;   // source weight level / mask
    char *dg; size_t dgp;       // destination levels
    char *dw; size_t dwp;
    int dwid, dhei;
};
struct PyrDownBatch { PyrDownArgs a[MB_MAXB]; TileMap tm; };

__device__ inline float hpass_f(float s0, float s1, float s2, float s3, float s4)
{
    float t = s2 * 6 + (s1 + s3) * 4;
    t = t + s0;
    return t + s4;
}

// De-interleave 4 BGR pixels that start at byte 2 of `a` (a, b, c, d: consecutive words): the 8-bit rows are read from a
// 4-byte aligned address two bytes ahead of the first tap -- misaligned 16-byte loads cost 3x on gfx950 (tools/ta_microbench.hip).
// v_perm_b32(hi, lo, sel): selector values 0-3 take bytes of lo, 4-7 bytes of hi, 0x0c gives 0.
__device__ inline void deint4_off2(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &B, uint32_t &G, uint32_t &R)
{
    B = __builtin_amdgcn_perm(c, __builtin_amdgcn_perm(b, a, 0x0c0c0502u), 0x07040100u);                     // bytes 6, 9, 12, 15
    G = __builtin_amdgcn_perm(b, a, 0x0c0c0603u) | __builtin_amdgcn_perm(d, c, 0x04010c0cu);               // bytes 7, 10, 13, 16
    R = __builtin_amdgcn_perm(d, __builtin_amdgcn_perm(c, b, 0x0c060300u), 0x05020100u);                     // bytes 8, 11, 14, 17
}

// APR: also write the BORDER_REFLECT_101 apron of the destination level (needs dwid >= 5 and dhei >= 5)
template <int SRC, bool APR>
__global__ __launch_bounds__(256) void k_pyr_down_2x2(const PyrDownBatch batch)
{
    int z, bx, by;
    tile_locate(batch.tm, blockIdx.x, z, bx, by);
    const PyrDownArgs &a = batch.a[z];
    // lane -> outputs (x0, x0+1) x (y0, y0+1); a 256-thread group covers 128 x 8 outputs
    const int x0 = 2 * (bx * 64 + (threadIdx.x & 63)), y0 = 2 * (by * 4 + (threadIdx.x >> 6));
    if (x0 >= a.dwid || y0 >= a.dhei) return;
    const int cx = 2 * x0 - 2, cy = 2 * y0 - 2;  // first tap; 7 rows x 7 pixels feed the 2x2 outputs (an 8th pixel pads the reads)
    int hA[7][3], hB[7][3];
    float wA[7], wB[7];
    const float inv255 = (float)(1. / 255.);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        if (PYR_SRC_U8(SRC)) {
            // 7 taps from byte 6 of the aligned address of pixel cx - 2 (cx = 2 mod 4): words 1..6 of it
            const uint8_t *p = (const uint8_t *)a.g + (ptrdiff_t)(cy + r) * (ptrdiff_t)a.gp + (ptrdiff_t)(cx - 2) * 3 + 4;
            const u32x4_a4 v = *(const u32x4_a4 *)p;
            const u32x2_a4 t = *(const u32x2_a4 *)(p + 16);
            uint32_t b03, g03, r03, b47, g47, r47;
            deint4_off2(v.x, v.y, v.z, v.w, b03, g03, r03);
            deint4_off2(v.w, t.x, t.y, 0u, b47, g47, r47);   // the 8th pixel only meets zero weights
            // column A taps pixels 0..4 with (1 4 6 4 1), column B taps pixels 2..6
            const uint32_t kA0 = 0x04060401u, kA1 = 0x00000001u, kB0 = 0x04010000u, kB1 = 0x00010406u;
            hA[r][0] = (int)__builtin_amdgcn_udot4(b47, kA1, __builtin_amdgcn_udot4(b03, kA0, 0u, false), false);
            hB[r][0] = (int)__builtin_amdgcn_udot4(b47, kB1, __builtin_amdgcn_udot4(b03, kB0, 0u, false), false);
            hA[r][1] = (int)__builtin_amdgcn_udot4(g47, kA1, __builtin_amdgcn_udot4(g03, kA0, 0u, false), false);
            hB[r][1] = (int)__builtin_amdgcn_udot4(g47, kB1, __builtin_amdgcn_udot4(g03, kB0, 0u, false), false);
            hA[r][2] = (int)__builtin_amdgcn_udot4(r47, kA1, __builtin_amdgcn_udot4(r03, kA0, 0u, false), false);
            hB[r][2] = (int)__builtin_amdgcn_udot4(r47, kB1, __builtin_amdgcn_udot4(r03, kB0, 0u, false), false);
        } else {
            // int16x3 rows: 7 pixels = 42 bytes, 4-byte aligned (cx is even, planes start 4-byte aligned)
            const char *p = a.g + (ptrdiff_t)(cy + r) * (ptrdiff_t)a.gp + (ptrdiff_t)cx * 6;
            const u32x4_a4 v0 = *(const u32x4_a4 *)p, v1 = *(const u32x4_a4 *)(p + 16);
            const u32x3_a4 v2 = *(const u32x3_a4 *)(p + 32);
            const uint32_t w[11] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z};
            int s[7][3];
#pragma unroll
            for (int k = 0; k < 7; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int e = 3 * k + c;
                    s[k][c] = (int)(int16_t)(uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                hA[r][c] = s[2][c] * 6 + (s[1][c] + s[3][c]) * 4 + s[0][c] + s[4][c];
                hB[r][c] = s[4][c] * 6 + (s[3][c] + s[5][c]) * 4 + s[2][c] + s[6][c];
            }
        }
        if (SRC >= 2) {
            const float *wp = (const float *)(a.w + (ptrdiff_t)(cy + r) * (ptrdiff_t)a.wp) + cx;
            const f32x4_a4 f0 = *(const f32x4_a4 *)wp;
            const f32x3_a4 f1 = *(const f32x3_a4 *)(wp + 4);
            wA[r] = hpass_f(f0.x, f0.y, f0.z, f0.w, f1.x);
            wB[r] = hpass_f(f0.z, f0.w, f1.x, f1.y, f1.z);
        } else {
            // 7 mask samples from byte 2 of the aligned address of pixel cx - 2
            const u32x3_a4 mq = *(const u32x3_a4 *)((const uint8_t *)a.w + (ptrdiff_t)(cy + r) * (ptrdiff_t)a.wp + (cx - 2));
            const float m0 = (float)((mq.x >> 16) & 0xff) * inv255, m1 = (float)(mq.x >> 24) * inv255, m2 = (float)(mq.y & 0xff) * inv255,
                        m3 = (float)((mq.y >> 8) & 0xff) * inv255, m4 = (float)((mq.y >> 16) & 0xff) * inv255, m5 = (float)(mq.y >> 24) * inv255,
                        m6 = (float)(mq.z & 0xff) * inv255;
            wA[r] = hpass_f(m0, m1, m2, m3, m4);
            wB[r] = hpass_f(m2, m3, m4, m5, m6);
        }
    }
    const bool two_cols = x0 + 1 < a.dwid;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r0 = 2 * j, y = y0 + j;
        if (y >= a.dhei) break;
        int oa[3], ob[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            oa[c] = (hA[r0 + 2][c] * 6 + (hA[r0 + 1][c] + hA[r0 + 3][c]) * 4 + hA[r0][c] + hA[r0 + 4][c] + 128) >> 8;
            ob[c] = (hB[r0 + 2][c] * 6 + (hB[r0 + 1][c] + hB[r0 + 3][c]) * 4 + hB[r0][c] + hB[r0 + 4][c] + 128) >> 8;
        }
        const float fa = hpass_f(wA[r0], wA[r0 + 1], wA[r0 + 2], wA[r0 + 3], wA[r0 + 4]) * (1.f / 256);
        const float fb = hpass_f(wB[r0], wB[r0 + 1], wB[r0 + 2], wB[r0 + 3], wB[r0 + 4]) * (1.f / 256);
        // target rows: y itself and, with APR, the apron rows that mirror it (Y in [-4,-1] <- -Y, Y in [H, H+3] <- 2H-2-Y)
        const int H = a.dhei, W = a.dwid;
        const int ty[3] = {y, -y, 2 * H - 2 - y};
        const bool ton[3] = {true, APR && y >= 1 && y <= 4, APR && y >= H - 5 && y <= H - 2};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (!ton[t]) continue;
            char *drow = a.dg + (ptrdiff_t)ty[t] * (ptrdiff_t)a.dgp;
            float *wrow = (float *)(a.dw + (ptrdiff_t)ty[t] * (ptrdiff_t)a.dwp);
            constexpr bool D8 = PYR_SRC_U8(SRC);
            char *dg = drow + (ptrdiff_t)x0 * (D8 ? 3 : 6);
            float *dw = wrow + x0;
            if (two_cols) {
                if (D8) {
                    // two u8x3 pixels = 6 bytes, 2-byte aligned (x0 is even)
                    *(u32_q2 *)dg = (uint32_t)oa[0] | ((uint32_t)oa[1] << 8) | ((uint32_t)oa[2] << 16) | ((uint32_t)ob[0] << 24);
                    *(uint16_t *)(dg + 4) = (uint16_t)((uint32_t)ob[1] | ((uint32_t)ob[2] << 8));
                } else {
                    u32x3_a4 o;  // two int16x3 pixels = 12 bytes, 4-byte aligned (x0 is even)
                    o.x = (uint32_t)(uint16_t)oa[0] | ((uint32_t)(uint16_t)oa[1] << 16);
                    o.y = (uint32_t)(uint16_t)oa[2] | ((uint32_t)(uint16_t)ob[0] << 16);
                    o.z = (uint32_t)(uint16_t)ob[1] | ((uint32_t)(uint16_t)ob[2] << 16);
                    *(u32x3_a4 *)dg = o;
                }
                float2 wo = {fa, fb};
                *(float2 *)dw = wo;
            } else {
                if (D8) { uint8_t *d = (uint8_t *)dg; d[0] = (uint8_t)oa[0]; d[1] = (uint8_t)oa[1]; d[2] = (uint8_t)oa[2]; }
                else { int16_t *d = (int16_t *)dg; d[0] = (int16_t)oa[0]; d[1] = (int16_t)oa[1]; d[2] = (int16_t)oa[2]; }
                dw[0] = fa;
            }
            if (APR && (x0 <= 4 || x0 >= W - 6)) {
                // apron columns: X in [-4,-1] <- -X, X in [W, W+3] <- 2W-2-X
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int x = x0 + k;
                    if (x >= W) continue;
                    const int *ov = k ? ob : oa;
                    const float fv = k ? fb : fa;
                    if (x >= 1 && x <= 4) {
                        if (D8) { uint8_t *q = (uint8_t *)drow - (ptrdiff_t)x * 3; q[0] = (uint8_t)ov[0]; q[1] = (uint8_t)ov[1]; q[2] = (uint8_t)ov[2]; }
                        else { int16_t *q = (int16_t *)drow - (ptrdiff_t)x * 3; q[0] = (int16_t)ov[0]; q[1] = (int16_t)ov[1]; q[2] = (int16_t)ov[2]; }
                        wrow[-x] = fv;
                    }
                    if (x >= W - 5 && x <= W - 2) {
                        const int X = 2 * W - 2 - x;
                        if (D8) { uint8_t *q = (uint8_t *)drow + (ptrdiff_t)X * 3; q[0] = (uint8_t)ov[0]; q[1] = (uint8_t)ov[1]; q[2] = (uint8_t)ov[2]; }
                        else { int16_t *q = (int16_t *)drow + (ptrdiff_t)X * 3; q[0] = (int16_t)ov[0]; q[1] = (int16_t)ov[1]; q[2] = (int16_t)ov[2]; }
                        wrow[X] = fv;
                    }
                }
            }
        }
    }
}

// ---- strip form: a lane owns 4 output columns x R output rows and walks down the source rows -----------------------------
// Every source row is fetched and filtered horizontally once per strip (the 2x2 form does both 3.5 times); the five most
// recent horizontal results stay in registers.  Also writes the BORDER_REFLECT_101 apron of the destination level, so no
// separate apron launch is needed.  Requires dwid % 4 == 0, dwid >= 8, dhei >= 5 (host checks).
struct HRow { int v[4][3]; float w[4]; };

template <int SRC>
__device__ inline void pyr_hrow(const PyrDownArgs &a, int cx, int row, HRow &h)
{
    const float inv255 = (float)(1. / 255.);
    if (PYR_SRC_U8(SRC)) {
        // 11 BGR pixels = 33 bytes from byte 6 of the aligned address of pixel cx - 2: words 1..10 of it
        const uint8_t *p = (const uint8_t *)a.g + (ptrdiff_t)row * (ptrdiff_t)a.gp + (ptrdiff_t)(cx - 2) * 3 + 4;
        const u32x4_a4 v0 = *(const u32x4_a4 *)p, v1 = *(const u32x4_a4 *)(p + 16);
        const u32x2_a4 v2 = *(const u32x2_a4 *)(p + 32);
        const uint32_t w[10] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y};
        uint32_t ch[3][3];  // [channel][group of 4 pixels]
#pragma unroll
        for (int g = 0; g < 3; ++g) deint4_off2(w[3 * g], w[3 * g + 1], w[3 * g + 2], w[3 * g + 3], ch[0][g], ch[1][g], ch[2][g]);
        // outputs tap pixels 0..4, 2..6, 4..8, 6..10 with (1 4 6 4 1)
        const uint32_t kA0 = 0x04060401u, kA1 = 0x00000001u, kB0 = 0x04010000u, kB1 = 0x00010406u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            h.v[0][c] = (int)__builtin_amdgcn_udot4(ch[c][1], kA1, __builtin_amdgcn_udot4(ch[c][0], kA0, 0u, false), false);
            h.v[1][c] = (int)__builtin_amdgcn_udot4(ch[c][1], kB1, __builtin_amdgcn_udot4(ch[c][0], kB0, 0u, false), false);
            h.v[2][c] = (int)__builtin_amdgcn_udot4(ch[c][2], kA1, __builtin_amdgcn_udot4(ch[c][1], kA0, 0u, false), false);
            h.v[3][c] = (int)__builtin_amdgcn_udot4(ch[c][2], kB1, __builtin_amdgcn_udot4(ch[c][1], kB0, 0u, false), false);
        }
    } else {
        // 11 int16x3 pixels = 66 bytes, 4-byte aligned
        const char *p = a.g + (ptrdiff_t)row * (ptrdiff_t)a.gp + (ptrdiff_t)cx * 6;
        const u32x4_a4 q0 = *(const u32x4_a4 *)p, q1 = *(const u32x4_a4 *)(p + 16), q2 = *(const u32x4_a4 *)(p + 32), q3 = *(const u32x4_a4 *)(p + 48);
        const uint32_t q4 = *(const uint32_t *)(p + 64);
        const uint32_t w[17] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4};
        int sv[11][3];
#pragma unroll
        for (int k = 0; k < 11; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int e = 3 * k + c;
                sv[k][c] = (e & 1) ? ((int)w[e >> 1] >> 16) : (int)(int16_t)(uint16_t)(w[e >> 1] & 0xffffu);
            }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int c = 0; c < 3; ++c) h.v[o][c] = sv[2 * o + 2][c] * 6 + (sv[2 * o + 1][c] + sv[2 * o + 3][c]) * 4 + sv[2 * o][c] + sv[2 * o + 4][c];
    }
    float m[11];
    if (SRC < 2) {
        // interior of a frame: all 11 mask samples of every lane are 255 -> weights 1.0f, (1 4 6 4 1) gives exactly 16
        // the 11 samples are bytes 2..12 of the aligned 16-byte read at pixel cx - 2
        const u32x4_a4 mv = *(const u32x4_a4 *)((const uint8_t *)a.w + (ptrdiff_t)row * (ptrdiff_t)a.wp + (cx - 2));
        const bool full = ((mv.x | 0x0000ffffu) & mv.y & mv.z & (mv.w | 0xffffff00u)) == 0xffffffffu;
        const bool empty = ((mv.x & 0xffff0000u) | mv.y | mv.z | (mv.w & 0x000000ffu)) == 0u;
        if (__ballot(!full) == 0ULL) {
#pragma unroll
            for (int o = 0; o < 4; ++o) h.w[o] = 16.f;
            return;
        }
        if (__ballot(!empty) == 0ULL) {
#pragma unroll
            for (int o = 0; o < 4; ++o) h.w[o] = 0.f;
            return;
        }
        const uint32_t mw[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int k = 0; k < 11; ++k) m[k] = (float)((mw[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xffu) * inv255;
    } else {
        const float *wp = (const float *)(a.w + (ptrdiff_t)row * (ptrdiff_t)a.wp) + cx;
        const f32x4_a4 f0 = *(const f32x4_a4 *)wp, f1 = *(const f32x4_a4 *)(wp + 4);
        const f32x3_a4 f2 = *(const f32x3_a4 *)(wp + 8);
        m[0] = f0.x; m[1] = f0.y; m[2] = f0.z; m[3] = f0.w; m[4] = f1.x; m[5] = f1.y; m[6] = f1.z; m[7] = f1.w; m[8] = f2.x; m[9] = f2.y; m[10] = f2.z;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) h.w[o] = hpass_f(m[2 * o], m[2 * o + 1], m[2 * o + 2], m[2 * o + 3], m[2 * o + 4]);
}

// store 4 output pixels of one row, and the apron columns that mirror them (D8: an 8-bit level, values within [0, 255])
template <bool D8>
__device__ inline void pyr_store_row(const PyrDownArgs &a, int x0, int y, const int o[4][3], const float f[4])
{
    char *dg = a.dg + (ptrdiff_t)y * (ptrdiff_t)a.dgp;
    float *dw = (float *)(a.dw + (ptrdiff_t)y * (ptrdiff_t)a.dwp);
    if (D8) {
        // 12 bytes at a multiple of 12 from the 4-byte aligned plane origin
        u32x3_a4 s0;
        s0.x = (uint32_t)o[0][0] | ((uint32_t)o[0][1] << 8) | ((uint32_t)o[0][2] << 16) | ((uint32_t)o[1][0] << 24);
        s0.y = (uint32_t)o[1][1] | ((uint32_t)o[1][2] << 8) | ((uint32_t)o[2][0] << 16) | ((uint32_t)o[2][1] << 24);
        s0.z = (uint32_t)o[2][2] | ((uint32_t)o[3][0] << 8) | ((uint32_t)o[3][1] << 16) | ((uint32_t)o[3][2] << 24);
        *(u32x3_a4 *)(dg + (ptrdiff_t)x0 * 3) = s0;
    } else {
        uint32_t pk[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) pk[k] = (uint32_t)(uint16_t)o[(2 * k) / 3][(2 * k) % 3] | ((uint32_t)(uint16_t)o[(2 * k + 1) / 3][(2 * k + 1) % 3] << 16);
        u32x4_a4 s0; s0.x = pk[0]; s0.y = pk[1]; s0.z = pk[2]; s0.w = pk[3];
        u32x2_a4 s1; s1.x = pk[4]; s1.y = pk[5];
        *(u32x4_a4 *)(dg + (ptrdiff_t)x0 * 6) = s0;
        *(u32x2_a4 *)(dg + (ptrdiff_t)x0 * 6 + 16) = s1;
    }
    f32x4_a4 fw; fw.x = f[0]; fw.y = f[1]; fw.z = f[2]; fw.w = f[3];
    *(f32x4_a4 *)(dw + x0) = fw;
    // BORDER_REFLECT_101 apron columns: X in [-4, -1] mirrors -X, X in [W, W+3] mirrors 2W-2-X
    const int W = a.dwid;
    if (x0 <= 4 || x0 >= W - 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            if (x >= 1 && x <= 4) {
                if (D8) { uint8_t *t = (uint8_t *)dg - (ptrdiff_t)x * 3; t[0] = (uint8_t)o[k][0]; t[1] = (uint8_t)o[k][1]; t[2] = (uint8_t)o[k][2]; }
                else { int16_t *t = (int16_t *)dg - (ptrdiff_t)x * 3; t[0] = (int16_t)o[k][0]; t[1] = (int16_t)o[k][1]; t[2] = (int16_t)o[k][2]; }
                dw[-x] = f[k];
            }
            if (x >= W - 5 && x <= W - 2) {
                const int X = 2 * W - 2 - x;
                if (D8) { uint8_t *t = (uint8_t *)dg + (ptrdiff_t)X * 3; t[0] = (uint8_t)o[k][0]; t[1] = (uint8_t)o[k][1]; t[2] = (uint8_t)o[k][2]; }
                else { int16_t *t = (int16_t *)dg + (ptrdiff_t)X * 3; t[0] = (int16_t)o[k][0]; t[1] = (int16_t)o[k][1]; t[2] = (int16_t)o[k][2]; }
                dw[X] = f[k];
            }
        }
    }
}

template <int SRC, int R>
__global__ __launch_bounds__(256) void k_pyr_down_strip(const PyrDownBatch batch)
{
    int z, bx, by;
    tile_locate(batch.tm, blockIdx.x, z, bx, by);
    const PyrDownArgs &a = batch.a[z];
    const int x0 = 4 * (bx * 64 + (threadIdx.x & 63));
    const int y0 = __builtin_amdgcn_readfirstlane(R * (by * 4 + (threadIdx.x >> 6)));
    if (y0 >= a.dhei || x0 >= a.dwid) return;
    const int cx = 2 * x0 - 2, cy = 2 * y0 - 2;
    const int H = a.dhei;
    // Neighbouring strips share 3 source rows.  Odd strips sweep upwards, so a strip and its neighbour touch their common rows at
    // the same moment (both at the start or both at the end of their sweeps) and the second read hits in L2 instead of HBM.
    const bool up = ((threadIdx.x >> 6) & 1) && y0 + R <= H;
    const int first = up ? cy + 2 * R + 2 : cy, dir = up ? -1 : 1;
    HRow h[5];
    pyr_hrow<SRC>(a, cx, first, h[0]);
    pyr_hrow<SRC>(a, cx, first + dir, h[1]);
    pyr_hrow<SRC>(a, cx, first + 2 * dir, h[2]);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int y = up ? y0 + R - 1 - j : y0 + j;
        if (y >= H) break;
        // rows 2j .. 2j+4 of the sweep live in h[(2j + k) % 5]
        pyr_hrow<SRC>(a, cx, first + dir * (2 * j + 3), h[(2 * j + 3) % 5]);
        pyr_hrow<SRC>(a, cx, first + dir * (2 * j + 4), h[(2 * j + 4) % 5]);
        const HRow &r0 = h[(2 * j) % 5], &r1 = h[(2 * j + 1) % 5], &r2 = h[(2 * j + 2) % 5], &r3 = h[(2 * j + 3) % 5], &r4 = h[(2 * j + 4) % 5];
        int o[4][3];
        float f[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[k][c] = (r2.v[k][c] * 6 + (r1.v[k][c] + r3.v[k][c]) * 4 + r0.v[k][c] + r4.v[k][c] + 128) >> 8;
            // the float taps keep their top-to-bottom order whatever the sweep direction (float sums are order sensitive)
            f[k] = (up ? hpass_f(r4.w[k], r3.w[k], r2.w[k], r1.w[k], r0.w[k]) : hpass_f(r0.w[k], r1.w[k], r2.w[k], r3.w[k], r4.w[k])) * (1.f / 256);
        }
        pyr_store_row<PYR_SRC_U8(SRC)>(a, x0, y, o, f);
        // apron rows: Y in [-4, -1] mirrors -Y, Y in [H, H+3] mirrors 2H-2-Y (their apron columns included)
        if (y >= 1 && y <= 4) pyr_store_row<PYR_SRC_U8(SRC)>(a, x0, -y, o, f);
        if (y >= H - 5 && y <= H - 2) pyr_store_row<PYR_SRC_U8(SRC)>(a, x0, 2 * H - 2 - y, o, f);
    }
}

// ---- LDS-staged strip form of the level-0 pyrDown (8-bit frames + 8-bit masks) -----------------------------------------------------------------
// k_pyr_down_strip<0, 4> reads every lane's 11-pixel window straight from global memory: 16-byte loads at a 24-byte lane stride whose windows
// overlap, so a wave-level load touches ~24 cache lines and costs the texture addresser ~68 cycles (profiles/r03_pyramid_variants.txt: ta_busy
// 0.68) -- the address path, not HBM, bounds the kernel.  Here a wave copies the 1.5 KB + 0.5 KB its 64 windows cover into LDS with three
// coalesced LDS-DMA loads per source row (buffer_load ... lds: consecutive lanes, consecutive 16-byte chunks, no VGPR round trip) and the lanes
// take their windows from LDS (8-byte reads at a 24-byte stride: conflict free).  Rows are private to the wave: no barrier, the wave waits on
// its own vmcnt.  Two source rows are in flight while the previous two are filtered.  Arithmetic, tiling, sweep directions, apron stores: those
// of the strip form.
#define PL_GB 1568            // staged image bytes per row: 98 chunks of 16 (the 64 windows of 48 bytes at 24-byte steps)
#define PL_ROWB 2112          // + 528 mask bytes (33 chunks), rounded up
#define PL_NBUF 4

__device__ inline void pyr_hrow_u8_img(const uint32_t w[10], HRow &h)
{
    uint32_t ch[3][3];  // [channel][group of 4 pixels]
#pragma unroll
    for (int g = 0; g < 3; ++g) deint4_off2(w[3 * g], w[3 * g + 1], w[3 * g + 2], w[3 * g + 3], ch[0][g], ch[1][g], ch[2][g]);
    const uint32_t kA0 = 0x04060401u, kA1 = 0x00000001u, kB0 = 0x04010000u, kB1 = 0x00010406u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        h.v[0][c] = (int)__builtin_amdgcn_udot4(ch[c][1], kA1, __builtin_amdgcn_udot4(ch[c][0], kA0, 0u, false), false);
        h.v[1][c] = (int)__builtin_amdgcn_udot4(ch[c][1], kB1, __builtin_amdgcn_udot4(ch[c][0], kB0, 0u, false), false);
        h.v[2][c] = (int)__builtin_amdgcn_udot4(ch[c][2], kA1, __builtin_amdgcn_udot4(ch[c][1], kA0, 0u, false), false);
        h.v[3][c] = (int)__builtin_amdgcn_udot4(ch[c][2], kB1, __builtin_amdgcn_udot4(ch[c][1], kB0, 0u, false), false);
    }
}
__device__ inline void pyr_hrow_u8_words(const uint32_t w[10], const uint32_t mw[4], HRow &h)
{
    const float inv255 = (float)(1. / 255.);
    pyr_hrow_u8_img(w, h);
    // the 11 mask samples are bytes 2..12 of the 16; all 255 -> weights 1.0f and (1 4 6 4 1) gives exactly 16, all 0 -> 0
    const bool full = ((mw[0] | 0x0000ffffu) & mw[1] & mw[2] & (mw[3] | 0xffffff00u)) == 0xffffffffu;
    const bool empty = ((mw[0] & 0xffff0000u) | mw[1] | mw[2] | (mw[3] & 0x000000ffu)) == 0u;
    if (__ballot(!full) == 0ULL) {
#pragma unroll
        for (int o = 0; o < 4; ++o) h.w[o] = 16.f;
        return;
    }
    if (__ballot(!empty) == 0ULL) {
#pragma unroll
        for (int o = 0; o < 4; ++o) h.w[o] = 0.f;
        return;
    }
    float m[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) m[k] = (float)((mw[(k + 2) >> 2] >> (8 * ((k + 2) & 3))) & 0xffu) * inv255;
#pragma unroll
    for (int o = 0; o < 4; ++o) h.w[o] = hpass_f(m[2 * o], m[2 * o + 1], m[2 * o + 2], m[2 * o + 3], m[2 * o + 4]);
}

template <int R>
__global__ __launch_bounds__(256) void k_pyr_down_strip_lds(const PyrDownBatch batch)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[4][PL_NBUF][PL_ROWB];
    int z, bx, by;
    tile_locate(batch.tm, blockIdx.x, z, bx, by);
    const PyrDownArgs &a = batch.a[z];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0w = 256 * bx, x0 = x0w + 4 * lane;
    const int y0 = __builtin_amdgcn_readfirstlane(R * (by * 4 + wave));
    if (y0 >= a.dhei) return;                              // wave-uniform
    const bool act = x0 < a.dwid;                          // lanes past the level still carry chunks of the wave's copies
    const int cy = 2 * y0 - 2, H = a.dhei;
    const bool up = (wave & 1) && y0 + R <= H;             // odd strips sweep upwards (see k_pyr_down_strip)
    const int first = up ? cy + 2 * R + 2 : cy, dir = up ? -1 : 1;
    // buffer resources over the planes with their 4-sample aprons: the source rows start at (row + 4) * pitch; out-of-range chunks read as zeros
    const uint32_t gp = (uint32_t)a.gp, wp = (uint32_t)a.wp;
    const uint32_t rows_all = (uint32_t)(2 * a.dhei + 2 * APRON);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.g - (ptrdiff_t)APRON * (ptrdiff_t)a.gp - 3 * APRON), (short)0, (int)(rows_all * gp), 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.w - (ptrdiff_t)APRON * (ptrdiff_t)a.wp - APRON), (short)0, (int)(rows_all * wp), 0x00020000);
    typedef __attribute__((address_space(3))) void lds_void;
    auto stage_row = [&](int r) {
        uint8_t *dst = s_rows[wave][r & (PL_NBUF - 1)];
        const uint32_t go = (uint32_t)(r + APRON) * gp + 6u * (uint32_t)x0w + 16u * (uint32_t)lane;      // the window of lane 0 starts at pixel 2 x0w - 4 = byte 6 x0w - 12
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void *)dst, 16, go, 0, 0, 0);
        if (lane < PL_GB / 16 - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void *)(dst + 1024), 16, go + 1024u, 0, 0, 0);
        if (lane < (PL_ROWB - PL_GB) / 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_void *)(dst + PL_GB), 16, (uint32_t)(r + APRON) * wp + 2u * (uint32_t)x0w + 16u * (uint32_t)lane, 0, 0, 0);
    };
    // The window reads are inline assembly on purpose: for an LDS read that follows an LDS-DMA the compiler (SIInsertWaitcnts, no alias information)
    // inserts s_waitcnt vmcnt(0), i.e. it would also wait for the copies of the NEXT rows and for the previous row's stores.  The waits are placed by
    // hand instead: vmcnt(n) before the reads (below), lgkmcnt(0) behind them, inside the same statement.
    auto read_row = [&](int r, uint32_t w[10], uint32_t mw[4]) {
        const uint32_t src = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_rows[wave][r & (PL_NBUF - 1)];
        const uint32_t ga = src + 24u * (uint32_t)lane, ma = src + PL_GB + 8u * (uint32_t)lane;
        // ds_read_b64 at the 24-byte lane stride meets every bank once (6 k mod 64 distinct over a 32-lane half); ds_read2_b32 is banked mod 32: 2-way
        // (ONE statement, wait included: between separate statements the compiler is free to copy the reads' destination registers -- it does not
        // know they are still in flight -- and did so in an experimental variant of k_pyr_down_strip_lds_lv: stale data, caught by the 4K parity test)
        unsigned long long q0, q1, q2, q3, q4, q5, m0, m1;
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\tds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\t"
                     "ds_read_b64 %6, %9\n\tds_read_b64 %7, %9 offset:8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(m0), "=&v"(m1) : "v"(ga), "v"(ma));
        // words 0..11 of the window's aligned 48 bytes; 1..10 hold the 11 pixels
        w[0] = (uint32_t)(q0 >> 32); w[1] = (uint32_t)q1; w[2] = (uint32_t)(q1 >> 32); w[3] = (uint32_t)q2; w[4] = (uint32_t)(q2 >> 32); w[5] = (uint32_t)q3;
        w[6] = (uint32_t)(q3 >> 32); w[7] = (uint32_t)q4; w[8] = (uint32_t)(q4 >> 32); w[9] = (uint32_t)q5;
        mw[0] = (uint32_t)m0; mw[1] = (uint32_t)(m0 >> 32); mw[2] = (uint32_t)m1; mw[3] = (uint32_t)(m1 >> 32);
    };
    // rows are numbered along the sweep: row i of the sweep is source row first + dir * i
    HRow h[5];
    stage_row(first); stage_row(first + dir); stage_row(first + 2 * dir);
    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the wave's own copies have landed
    {
        uint32_t w[3][10], mw[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i) read_row(first + dir * i, w[i], mw[i]);
        stage_row(first + 3 * dir); stage_row(first + 4 * dir);
        if (act) {
#pragma unroll
            for (int i = 0; i < 3; ++i) pyr_hrow_u8_words(w[i], mw[i], h[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int y = up ? y0 + R - 1 - j : y0 + j;
        if (y >= H) break;
        uint32_t wa[10], ma[4], wb[10], mb[4];
        // rows 2j+3, 2j+4 (their six copies were issued one step ago) have landed.  vmcnt counts in issue order: behind those copies this wave
        // has issued only the previous row's two plain stores (12 bytes of pixels, 16 of weights; more on apron tiles), so "at most 2
        // outstanding" leaves the copies complete without waiting for the stores' acknowledgements
        if (j == 0) __builtin_amdgcn_s_waitcnt(0x0f70); else __builtin_amdgcn_s_waitcnt(0x0f72);
        read_row(first + dir * (2 * j + 3), wa, ma);
        read_row(first + dir * (2 * j + 4), wb, mb);
        if (j + 1 < R) { stage_row(first + dir * (2 * j + 5)); stage_row(first + dir * (2 * j + 6)); }     // into the slots of rows 2j+1, 2j+2
        if (act) {
            pyr_hrow_u8_words(wa, ma, h[(2 * j + 3) % 5]);
            pyr_hrow_u8_words(wb, mb, h[(2 * j + 4) % 5]);
        }
        if (act) {
            const HRow &r0 = h[(2 * j) % 5], &r1 = h[(2 * j + 1) % 5], &r2 = h[(2 * j + 2) % 5], &r3 = h[(2 * j + 3) % 5], &r4 = h[(2 * j + 4) % 5];
            int o[4][3];
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) o[k][c] = (r2.v[k][c] * 6 + (r1.v[k][c] + r3.v[k][c]) * 4 + r0.v[k][c] + r4.v[k][c] + 128) >> 8;
                f[k] = (up ? hpass_f(r4.w[k], r3.w[k], r2.w[k], r1.w[k], r0.w[k]) : hpass_f(r0.w[k], r1.w[k], r2.w[k], r3.w[k], r4.w[k])) * (1.f / 256);
            }
            pyr_store_row<true>(a, x0, y, o, f);
            if (y >= 1 && y <= 4) pyr_store_row<true>(a, x0, -y, o, f);
            if (y >= H - 5 && y <= H - 2) pyr_store_row<true>(a, x0, 2 * H - 2 - y, o, f);
        }
    }
}

// ---- the same for the levels >= 1 of 8-bit fed pyramids: u8x3 samples + f32 weights ------------------------------------------------------------
// Per source row a wave copies 1536 bytes of samples and 2032 bytes of weights (the 63 windows of 48 / 48 bytes at 24 / 32-byte steps) with four
// coalesced LDS-DMA loads instead of six windowed global loads per lane.  Two row slots per wave: a row pair is read into registers, then the
// next pair is copied into the same slots while this one is filtered.
#define LV_TW 252             // output columns per wave: 63 lanes x 4 (lane 63 only carries copy chunks): both copies of a row are then two full
                              // instructions (96 and 127 chunks of 16 bytes) -- 256 columns need a third for one more chunk of weights
#define LV_GB 1536            // 24 * 62 + 48 window bytes
#define LV_ROWB 3568          // + 2032 weight bytes (32 * 62 + 48)
__device__ inline uint32_t lds_addr_of(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)p; }

template <int R>
__global__ __launch_bounds__(256) void k_pyr_down_strip_lds_lv(const PyrDownBatch batch)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_rows[4][2][LV_ROWB];
    int z, bx, by;
    tile_locate(batch.tm, blockIdx.x, z, bx, by);
    const PyrDownArgs &a = batch.a[z];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0w = LV_TW * bx, x0 = x0w + 4 * lane;
    const int y0 = __builtin_amdgcn_readfirstlane(R * (by * 4 + wave));
    if (y0 >= a.dhei) return;
    const bool act = x0 < a.dwid && lane < LV_TW / 4;
    const int cy = 2 * y0 - 2, H = a.dhei;
    const bool up = (wave & 1) && y0 + R <= H;             // odd strips sweep upwards (see k_pyr_down_strip)
    const int first = up ? cy + 2 * R + 2 : cy, dir = up ? -1 : 1;
    const uint32_t gp = (uint32_t)a.gp, wp = (uint32_t)a.wp;
    const uint32_t rows_all = (uint32_t)(2 * a.dhei + 2 * APRON);
    // plane origins with their aprons: (0, 0) of a u8x3 level sits 12 bytes, of a weight level 16 bytes into its row
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.g - (ptrdiff_t)APRON * (ptrdiff_t)a.gp - 3 * APRON), (short)0, (int)(rows_all * gp), 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.w - (ptrdiff_t)APRON * (ptrdiff_t)a.wp - 4 * APRON), (short)0, (int)(rows_all * wp), 0x00020000);
    typedef __attribute__((address_space(3))) void lds_void;
    // row i of the sweep = source row first + dir * i, staged in slot i & 1
    auto stage_row = [&](int i) {
        const int r = first + dir * i;
        uint8_t *dst = s_rows[wave][i & 1];
        // samples: the window of lane 0 starts at pixel 2 x0w - 4 = byte 6 x0w of the apron-based row (see k_pyr_down_strip_lds)
        const uint32_t go = (uint32_t)(r + APRON) * gp + 6u * (uint32_t)x0w + 16u * (uint32_t)lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void *)dst, 16, go, 0, 0, 0);
        if (lane < LV_GB / 16 - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void *)(dst + 1024), 16, go + 1024u, 0, 0, 0);
        // weights: the copy starts AT lane 0's window (sample 2 x0w - 2 = byte 8 x0w + 8 of the apron-based row: 4-byte aligned in memory, which is
        // all a dwordx4 load needs) so that every lane's window is 16-byte aligned in LDS for ds_read_b128
        const uint32_t wo = (uint32_t)(r + APRON) * wp + 8u * (uint32_t)x0w + 8u + 16u * (uint32_t)lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_void *)(dst + LV_GB), 16, wo, 0, 0, 0);
        if (lane < (LV_ROWB - LV_GB) / 16 - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_void *)(dst + LV_GB + 1024), 16, wo + 1024u, 0, 0, 0);
    };
    // inline assembly for the same reason as in k_pyr_down_strip_lds: the waits are placed by hand
    typedef uint32_t asm_u32x4 __attribute__((ext_vector_type(4)));
    auto read_row = [&](int i, HRow &h) {
        const uint32_t src = lds_addr_of(s_rows[wave][i & 1]);
        const uint32_t wl = (uint32_t)min(lane, LV_TW / 4 - 1);        // lane 63 has no window of its own: it re-reads lane 62's (stays inside the slot)
        const uint32_t ga = src + 24u * wl, ma = src + LV_GB + 32u * wl;
        unsigned long long q0, q1, q2, q3, q4, q5;
        asm_u32x4 m0, m1, m2;
        asm volatile("ds_read_b64 %0, %9\n\tds_read_b64 %1, %9 offset:8\n\tds_read_b64 %2, %9 offset:16\n\tds_read_b64 %3, %9 offset:24\n\tds_read_b64 %4, %9 offset:32\n\tds_read_b64 %5, %9 offset:40\n\t"
                     "ds_read_b128 %6, %10\n\tds_read_b128 %7, %10 offset:16\n\tds_read_b128 %8, %10 offset:32\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(m0), "=&v"(m1), "=&v"(m2) : "v"(ga), "v"(ma));
        if (!act) return;
        // words 0..11 of the window's aligned 48 bytes; 1..10 hold the 11 pixels
        const uint32_t w[10] = {(uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32), (uint32_t)q2, (uint32_t)(q2 >> 32), (uint32_t)q3,
                                (uint32_t)(q3 >> 32), (uint32_t)q4, (uint32_t)(q4 >> 32), (uint32_t)q5};
        const float m[11] = {__uint_as_float(m0.x), __uint_as_float(m0.y), __uint_as_float(m0.z), __uint_as_float(m0.w), __uint_as_float(m1.x), __uint_as_float(m1.y),
                             __uint_as_float(m1.z), __uint_as_float(m1.w), __uint_as_float(m2.x), __uint_as_float(m2.y), __uint_as_float(m2.z)};
        pyr_hrow_u8_img(w, h);
#pragma unroll
        for (int o = 0; o < 4; ++o) h.w[o] = hpass_f(m[2 * o], m[2 * o + 1], m[2 * o + 2], m[2 * o + 3], m[2 * o + 4]);
    };
    HRow h[5];
    stage_row(0); stage_row(1);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    read_row(0, h[0]); read_row(1, h[1]);
    stage_row(2);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    read_row(2, h[2]);
    stage_row(3); stage_row(4);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int y = up ? y0 + R - 1 - j : y0 + j;
        if (y >= H) break;
        // the 8 copies of rows 2j+3, 2j+4 have landed; behind them only the previous row's two plain stores were issued (see k_pyr_down_strip_lds)
        if (j == 0) __builtin_amdgcn_s_waitcnt(0x0f70); else __builtin_amdgcn_s_waitcnt(0x0f72);
        read_row(2 * j + 3, h[(2 * j + 3) % 5]);
        read_row(2 * j + 4, h[(2 * j + 4) % 5]);
        if (j + 1 < R) { stage_row(2 * j + 5); stage_row(2 * j + 6); }      // the reads above have completed (lgkmcnt(0)): the slots are free
        if (act) {
            const HRow &r0 = h[(2 * j) % 5], &r1 = h[(2 * j + 1) % 5], &r2 = h[(2 * j + 2) % 5], &r3 = h[(2 * j + 3) % 5], &r4 = h[(2 * j + 4) % 5];
            int o[4][3];
            float f[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) o[k][c] = (r2.v[k][c] * 6 + (r1.v[k][c] + r3.v[k][c]) * 4 + r0.v[k][c] + r4.v[k][c] + 128) >> 8;
                f[k] = (up ? hpass_f(r4.w[k], r3.w[k], r2.w[k], r1.w[k], r0.w[k]) : hpass_f(r0.w[k], r1.w[k], r2.w[k], r3.w[k], r4.w[k])) * (1.f / 256);
            }
            pyr_store_row<true>(a, x0, y, o, f);
            if (y >= 1 && y <= 4) pyr_store_row<true>(a, x0, -y, o, f);
            if (y >= H - 5 && y <= H - 2) pyr_store_row<true>(a, x0, 2 * H - 2 - y, o, f);
        }
    }
}

// float pyramids (BASELINE config 5): one output per lane, scalar association of pyramids.cpp, no border logic either.
// A lane loads only ITS two source pixels of each of the five rows (24 contiguous bytes; the wave's loads tile 1.5 KB without
// overlap) and takes the pair to its left and the pixel to its right from the neighbouring lanes; lanes 0 and 63 are suppliers only
// (62 outputs per wave).  The former five-pixel window per lane issued twice the vector-memory instructions and was bound by the
// texture-address path at 2.9 TB/s, not by HBM.
// value of the previous / next lane of the wave through the DPP wave shifts of gfx9 (one VALU move each; the LDS permute they
// replace measured slower than the loads it saved).  Lanes 0 / 63 keep their own value; the callers substitute theirs.
__device__ inline int lane_up_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
__device__ inline int lane_down_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ inline float lane_up(float v) { return __int_as_float(lane_up_i(__float_as_int(v))); }
__device__ inline float lane_down(float v) { return __int_as_float(lane_down_i(__float_as_int(v))); }
#define PDF_OUT 62   // outputs per wave: lanes 1 .. 62; lanes 0 and 63 only supply their neighbours' taps
// APR: also write the BORDER_REFLECT_101 apron of the destination level (needs dwid >= 5 and dhei >= 5), as k_pyr_down_2x2 does: a separate apron
// launch per level was 6 x 12 us of config 5's step
template <bool LEVEL0, bool APR>
__global__ __launch_bounds__(256) void k_pyr_down_float(const PyrDownBatch batch)
{
    const PyrDownArgs &a = batch.a[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int x = (int)blockIdx.x * PDF_OUT + lane - 1, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= a.dhei || (int)blockIdx.x * PDF_OUT >= a.dwid) return;   // wave-uniform: every lane of a live wave takes part in the exchanges
    const bool live = lane >= 1 && lane <= PDF_OUT && x < a.dwid;
    const int xl = x < a.dwid ? x : a.dwid;                            // -1 .. dwid: pixels -2 .. 2*dwid + 1 lie inside the apron
    const float inv255 = (float)(1. / 255.);
    float rv[5][3], rw[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const char *row = a.g + (ptrdiff_t)(2 * y - 2 + r) * (ptrdiff_t)a.gp;
        const f32x4_a4 q0 = *(const f32x4_a4 *)(row + (ptrdiff_t)(2 * xl) * 12);
        const f32x2_a4 q1 = *(const f32x2_a4 *)(row + (ptrdiff_t)(2 * xl) * 12 + 16);
        const float own[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        float p[15];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            p[k] = lane_up(own[k]);
            p[6 + k] = own[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) p[12 + k] = lane_down(own[k]);
#pragma unroll
        for (int c = 0; c < 3; ++c) rv[r][c] = hpass_f(p[c], p[3 + c], p[6 + c], p[9 + c], p[12 + c]);
        if (LEVEL0) {
            // own two mask bytes at 2x; the neighbours' as for the image
            const uint8_t *mrow = (const uint8_t *)a.w + (ptrdiff_t)(2 * y - 2 + r) * (ptrdiff_t)a.wp;
            const uint32_t mo = *(const uint16_t *)(mrow + 2 * xl);
            const uint32_t ml = (uint32_t)lane_up_i((int)mo), mr = (uint32_t)lane_down_i((int)mo);
            const float m0 = (float)(ml & 0xffu) * inv255, m1 = (float)(ml >> 8) * inv255, m2 = (float)(mo & 0xffu) * inv255,
                        m3 = (float)(mo >> 8) * inv255, m4 = (float)(mr & 0xffu) * inv255;
            rw[r] = hpass_f(m0, m1, m2, m3, m4);
        } else {
            const char *wrow = a.w + (ptrdiff_t)(2 * y - 2 + r) * (ptrdiff_t)a.wp;
            const f32x2_a4 wo = *(const f32x2_a4 *)(wrow + (ptrdiff_t)(2 * xl) * 4);
            const float l0 = lane_up(wo.x), l1 = lane_up(wo.y), r0 = lane_down(wo.x);
            rw[r] = hpass_f(l0, l1, wo.x, wo.y, r0);
        }
    }
    if (!live) return;
    const f32x3_a4 o = {hpass_f(rv[0][0], rv[1][0], rv[2][0], rv[3][0], rv[4][0]) * (1.f / 256), hpass_f(rv[0][1], rv[1][1], rv[2][1], rv[3][1], rv[4][1]) * (1.f / 256),
                        hpass_f(rv[0][2], rv[1][2], rv[2][2], rv[3][2], rv[4][2]) * (1.f / 256)};
    const float ow = hpass_f(rw[0], rw[1], rw[2], rw[3], rw[4]) * (1.f / 256);
    auto put = [&](int X, int Y) {
        *(f32x3_a4 *)((float *)(a.dg + (ptrdiff_t)Y * (ptrdiff_t)a.dgp) + (ptrdiff_t)X * 3) = o;
        ((float *)(a.dw + (ptrdiff_t)Y * (ptrdiff_t)a.dwp))[X] = ow;
    };
    put(x, y);
    // the apron positions that mirror this sample: column -k <- k and W - 1 + k <- W - 1 - k (k = 1 .. APRON), rows alike (on levels narrower than
    // 2 APRON + 2 a sample mirrors to both sides).  One scalar test keeps the waves of the interior -- nearly all -- out of it: a wave is one row
    // of 62 columns
    const bool edge_wave = y <= APRON || y >= a.dhei - 1 - APRON || (int)blockIdx.x * PDF_OUT <= APRON || ((int)blockIdx.x + 1) * PDF_OUT >= a.dwid - 1 - APRON;
    if (APR && edge_wave) {
        const int W = a.dwid, H = a.dhei;
        const int xs[3] = {x, -x, 2 * W - 2 - x}, ys[3] = {y, -y, 2 * H - 2 - y};
        const bool xon[3] = {true, x >= 1 && x <= APRON, x >= W - 1 - APRON && x <= W - 2}, yon[3] = {true, y >= 1 && y <= APRON, y >= H - 1 - APRON && y <= H - 2};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if ((i || j) && xon[i] && yon[j]) put(xs[i], ys[j]);
    }
}

// ====================================================================================================================
// pyrUp sample: value of pyrUp(src)(X, Y) for a 2x upsampling; index -1 -> 1 (reflect-101), index n -> n-1 (replicate)
// ====================================================================================================================
template <bool FLT, typename ST = typename std::conditional<FLT, float, int16_t>::type>
__device__ inline void pyr_up_at(const void *base, size_t pitch, int nw, int nh, int X, int Y, typename Acc3<FLT>::T out[3])
{
    typedef typename Acc3<FLT>::T VT;
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
    int src_depth;               // level 0 only: SSP_U8 / SSP_S16 / SSP_F32
    int lvl8;                    // levels >= 1 of this image are u8x3 (fed 8-bit), not int16x3
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
    // small levels are bound by dependent round trips, not by bytes: load an image's samples together with its weights instead of
    // probing the weights first (2x2 and per-pixel kernels)
    int eager;
    int gx;      // tiles per row of the 1-D grid (4x2 kernel)
    // 4x2 kernel: for every 256 x 8 tile (logical index t = by * gx + bx, the whole padded grid) and every group of 32 images one word
    // whose bit i says "image 32 g + i's rectangle meets the tile".  Static geometry: built on the host with the descriptor table and
    // cached with it, so a wave probes the weights of the 2-4 images that can reach it instead of walking all of them in rounds.
    const uint32_t *tmask; int tgroups;
};

// Images whose level rectangle meets the tile (bx0, by0, tw, th), as a bit set over images g0 .. g0+63: lane k tests image g0+k, one vector
// round trip and a ballot instead of one scalar load + branch per fed image (the small levels are chains of dependent round trips).
__device__ inline uint64_t tile_candidates(const LevelArgs &a, int g0, int bx0, int by0, int tw, int th)
{
    const int i = g0 + (int)(threadIdx.x & 63u);
    bool hit = false;
    if (i < a.n_imgs) {
        const LevelImg &im = a.imgs[i];
        hit = !(bx0 + tw <= im.rx || bx0 >= im.rx + im.pw || by0 + th <= im.ry || by0 >= im.ry + im.ph);
    }
    return __ballot(hit);
}

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
    for (int g0 = 0; g0 < a.n_imgs; g0 += 64)
    for (uint64_t cand = tile_candidates(a, g0, bx0, by0, 64, 4); cand; cand &= cand - 1ULL) {   // feed order
        const LevelImg &im = a.imgs[g0 + __builtin_ctzll(cand)];
        const int lx = X - im.rx, ly = Y - im.ry;
        const bool in = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
        float w = 0.f;
        if (in) {
            if (LEVEL0) w = (float)((const uint8_t *)im.w + (size_t)ly * im.wp)[lx] * (float)(1. / 255.);
            else w = ((const float *)((const char *)im.w + (size_t)ly * im.wp))[lx];
        }
        // a wave whose weights are all zero contributes (short)(L*0) = 0 and w + 0: skip the image loads
        if (!a.eager && __ballot(in && w != 0.f) == 0ULL) continue;
        if (in) {
            VT g[3];
            if (LEVEL0) {
                if (im.src_depth == SSP_U8) load_px<uint8_t, VT>(im.g, im.gp, lx, ly, g);
                else if (im.src_depth == SSP_S16) load_px<int16_t, VT>(im.g, im.gp, lx, ly, g);
                else load_px<float, VT>(im.g, im.gp, lx, ly, g);
            } else {
                if (FLT) load_px<float, VT>(im.g, im.gp, lx, ly, g);
                else if (im.lvl8) load_px<uint8_t, VT>(im.g, im.gp, lx, ly, g);
                else load_px<int16_t, VT>(im.g, im.gp, lx, ly, g);
            }
            if (!a.top) {
                VT up[3];
                if (!FLT && im.lvl8) pyr_up_at<FLT, uint8_t>(im.gn, im.gnp, im.pwn, im.phn, lx, ly, up);
                else pyr_up_at<FLT>(im.gn, im.gnp, im.pwn, im.phn, lx, ly, up);
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
template <bool FLT, typename ST = typename std::conditional<FLT, float, int16_t>::type>
__device__ inline void pyr_up_quad(const void *base, size_t pitch, int nw, int nh, int rx0, int ry0, int rw, int rh, int sx, int sy,
                                   typename Acc3<FLT>::T out[4][3])
{
    typedef typename Acc3<FLT>::T VT;
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
        if (contiguous && sizeof(ST) == 1) {
            // 9 bytes starting at pixel sx-1 (12 are read: the rows carry slack)
            const u32x3_u1 v = *(const u32x3_u1 *)(rowp + (size_t)(sx - 1 - rx0) * 3);
            pa[0] = (VT)(v.x & 0xff); pa[1] = (VT)((v.x >> 8) & 0xff); pa[2] = (VT)((v.x >> 16) & 0xff);
            pb[0] = (VT)(v.x >> 24); pb[1] = (VT)(v.y & 0xff); pb[2] = (VT)((v.y >> 8) & 0xff);
            pc[0] = (VT)((v.y >> 16) & 0xff); pc[1] = (VT)(v.y >> 24); pc[2] = (VT)(v.z & 0xff);
        } else if (contiguous) {
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
    for (int g0 = 0; g0 < a.n_imgs; g0 += 64)
    for (uint64_t cand = tile_candidates(a, g0, bx0, by0, 64, 16); cand; cand &= cand - 1ULL) {   // feed order
        const LevelImg &im = a.imgs[g0 + __builtin_ctzll(cand)];
        const int lx = X0 - im.rx, ly = Y0 - im.ry;  // even: the rectangle origin is a multiple of 2 below the top level
        const bool in = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (in) {
            if (LEVEL0) {
                // level 0 is stored with its border: weight = mask/255 (0 in the border band)
                const uint8_t *mp = (const uint8_t *)im.w + (size_t)ly * im.wp + lx;
                const uint32_t m0 = *(const u16_q1 *)mp, m1 = *(const u16_q1 *)(mp + im.wp);
                w[0] = (float)(m0 & 0xff) * inv255; w[1] = (float)(m0 >> 8) * inv255;
                w[2] = (float)(m1 & 0xff) * inv255; w[3] = (float)(m1 >> 8) * inv255;
            } else {
                const float2 w0 = *(const float2 *)((const char *)im.w + (size_t)ly * im.wp + (size_t)lx * 4);
                const float2 w1 = *(const float2 *)((const char *)im.w + (size_t)(ly + 1) * im.wp + (size_t)lx * 4);
                w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y;
            }
        }
        // a wave whose weights are all zero contributes (short)(L*0) = 0 and w + 0: skip the image loads
        if (!a.eager) {
            const bool any = in && (w[0] != 0.f || w[1] != 0.f || w[2] != 0.f || w[3] != 0.f);
            if (__ballot(any) == 0ULL) continue;
        }
        if (in) {
            VT g[4][3];
            if (LEVEL0 || (!FLT && im.lvl8)) {
                if (!LEVEL0 || im.src_depth == SSP_U8) {
                    // two BGR pixels per row = 6 bytes: one 8-byte read (the plane rows carry slack)
                    const uint8_t *p = (const uint8_t *)im.g + (size_t)ly * im.gp + (size_t)lx * 3;
                    const u32x2_u1 r0 = *(const u32x2_u1 *)p, r1 = *(const u32x2_u1 *)(p + im.gp);
                    g[0][0] = (VT)(r0.x & 0xff); g[0][1] = (VT)((r0.x >> 8) & 0xff); g[0][2] = (VT)((r0.x >> 16) & 0xff);
                    g[1][0] = (VT)(r0.x >> 24); g[1][1] = (VT)(r0.y & 0xff); g[1][2] = (VT)((r0.y >> 8) & 0xff);
                    g[2][0] = (VT)(r1.x & 0xff); g[2][1] = (VT)((r1.x >> 8) & 0xff); g[2][2] = (VT)((r1.x >> 16) & 0xff);
                    g[3][0] = (VT)(r1.x >> 24); g[3][1] = (VT)(r1.y & 0xff); g[3][2] = (VT)((r1.y >> 8) & 0xff);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (im.src_depth == SSP_S16) load_px<int16_t, VT>(im.g, im.gp, lx + (q & 1), ly + (q >> 1), g[q]);
                        else load_px<float, VT>(im.g, im.gp, lx + (q & 1), ly + (q >> 1), g[q]);
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
            if (!FLT && im.lvl8) pyr_up_quad<FLT, uint8_t>(im.gn, im.gnp, im.pwn, im.phn, 0, 0, im.pwn, im.phn, lx >> 1, ly >> 1, up);
            else pyr_up_quad<FLT>(im.gn, im.gnp, im.pwn, im.phn, 0, 0, im.pwn, im.phn, lx >> 1, ly >> 1, up);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (FLT) acc[q][c] = acc[q][c] + (g[q][c] - up[q][c]) * w[q];
                    else acc[q][c] = (VT)((int)acc[q][c] + (int)((float)sat16((int)g[q][c] - (int)up[q][c]) * w[q]));  // (short) wrap deferred: sums are taken mod 2^16
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
        if (FLT) {
#pragma unroll
            for (int c = 0; c < 3; ++c) n[q][c] = up[q][c] + acc[q][c] / den;
        } else {
            float qn[3];
            div3_exact((float)(int16_t)(uint16_t)((int)acc[q][0] & 0xffff), (float)(int16_t)(uint16_t)((int)acc[q][1] & 0xffff),
                       (float)(int16_t)(uint16_t)((int)acc[q][2] & 0xffff), den, qn);
#pragma unroll
            for (int c = 0; c < 3; ++c) n[q][c] = (VT)sat16((int)up[q][c] + trunc16(qn[c]));
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

// ---- 4x2 form: integer pyramids, levels whose rectangles are multiples of 4 (l <= bands - 2) ----------------------------------
// The level kernels are bound by the number of vector memory instructions per wave, not by bytes: a lane that owns 4x2
// pixels issues the same number of (wider) loads as a 2x2 lane.  One wave = 256 pixels x 2 rows, so rows are wave-uniform.
typedef uint32_t u32x2_a2 __attribute__((ext_vector_type(2), aligned(2)));

// 4 int16x3 pixels = 6 words
__device__ inline void unpack_s16x12(const uint32_t w[6], int v[4][3])
{
    int f[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        f[2 * k] = (int)(int16_t)(uint16_t)(w[k] & 0xffffu);
        f[2 * k + 1] = (int)w[k] >> 16;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[q][c] = f[3 * q + c];
}

// 4 int16x3 pixels at p, where p - 6 is 4-byte aligned (pixel sx-1 of a row whose even pixels are aligned): two aligned 16-byte reads
__device__ inline void load_s16x12_off6(const char *p, int v[4][3])
{
    const u32x4_a4 a = *(const u32x4_a4 *)(p - 6);
    const u32x4_a4 b = *(const u32x4_a4 *)(p + 10);
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int e = 3 + 3 * q + c;
            v[q][c] = (e & 1) ? ((int)w[e >> 1] >> 16) : (int)(int16_t)(uint16_t)(w[e >> 1] & 0xffffu);
        }
}

__device__ inline void load_s16x12(const char *p, int v[4][3])
{
    const u32x4_a2 a = *(const u32x4_a2 *)p;
    const u32x2_a2 b = *(const u32x2_a2 *)(p + 16);
    const uint32_t w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    unpack_s16x12(w, v);
}

// pyrUp of parent pixels sx-1 .. sx+2 of rows sy-1, sy, sy+1 (p0..p2 point at pixel sx-1 of each row, border rows already
// substituted) -> outputs (2sx .. 2sx+3) x (2sy, 2sy+1), index = row*4 + column.  last_dup: pixel sx+2 is past the level's
// right edge, pyrUp repeats pixel sx+1 there.
// l8: the parent level is u8x3 (the rows then point at byte 3 (sx - 1))
__device__ inline void pyr_up_oct(const char *p0, const char *p1, const char *p2, bool last_dup, int up[8][3], bool l8 = false)
{
    int he[3][2][3], ho[3][2][3];
    const char *rp[3] = {p0, p1, p2};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int P[4][3];
        if (l8) {
            const u32x3_u1 v = *(const u32x3_u1 *)rp[r];
            const uint32_t wd[3] = {v.x, v.y, v.z};
#pragma unroll
            for (int k = 0; k < 12; ++k) P[k / 3][k % 3] = (int)((wd[k >> 2] >> (8 * (k & 3))) & 0xffu);
        }
        else load_s16x12_off6(rp[r], P);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int d = last_dup ? P[2][c] : P[3][c];
            he[r][0][c] = P[0][c] + P[1][c] * 6 + P[2][c];
            ho[r][0][c] = (P[1][c] + P[2][c]) * 4;
            he[r][1][c] = P[1][c] + P[2][c] * 6 + d;
            ho[r][1][c] = (P[2][c] + d) * 4;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            up[2 * k][c] = (he[0][k][c] + he[1][k][c] * 6 + he[2][k][c] + 32) >> 6;
            up[2 * k + 1][c] = (ho[0][k][c] + ho[1][k][c] * 6 + ho[2][k][c] + 32) >> 6;
            up[4 + 2 * k][c] = ((he[1][k][c] + he[2][k][c]) * 4 + 32) >> 6;
            up[4 + 2 * k + 1][c] = ((ho[1][k][c] + ho[2][k][c]) * 4 + 32) >> 6;
        }
}

// ---- packed int16 arithmetic (two samples per VALU instruction) -----------------------------------------------------------------
// Pyramids fed from 8-bit frames hold values in [0, 255] at every Gaussian level (pyrDown is a rounded convex combination),
// so every intermediate of pyrUp (at most 64*255 + 32) fits an int16 lane: v_pk_mad_u16 / v_pk_add_u16 / v_pk_ashrrev_i16.
// Saturating v_pk_sub_i16 is the Laplacian's sat16, wrapping v_pk_add_u16 is C's `short +=`.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) + __builtin_bit_cast(u16x2v, b))); }
__device__ inline uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) - __builtin_bit_cast(u16x2v, b))); }
__device__ inline uint32_t pk_mad6(uint32_t a, uint32_t b)
{
    const u16x2v six = {6, 6};
    return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) * six + __builtin_bit_cast(u16x2v, b)));
}
__device__ inline uint32_t pk_shl2(uint32_t a) { const u16x2v two = {2, 2}; return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) << two)); }
template <int N>
__device__ inline uint32_t pk_asr(uint32_t a) { const s16x2 n = {N, N}; return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) >> n)); }
__device__ inline uint32_t pk_sub_sat(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ inline uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }
__device__ inline uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }

// Octet layout of the packed path: per output row two 3-word vectors, E = columns (0, 2) and O = columns (1, 3), each
// holding 2 pixels x 3 channels = 6 int16 -- the order pyrUp produces them in (even columns from one expression, odd
// columns from the other).
struct OctPk { uint32_t e[2][3], o[2][3]; };

// 4 natural-order int16x3 pixels (6 words) -> E / O vectors
__device__ inline void oct_split_s16(const uint32_t g[6], uint32_t e[3], uint32_t o[3])
{
    e[0] = g[0];
    e[1] = __builtin_amdgcn_perm(g[3], g[1], 0x05040100u);
    e[2] = __builtin_amdgcn_alignbit(g[4], g[3], 16);
    o[0] = __builtin_amdgcn_alignbit(g[2], g[1], 16);
    o[1] = __builtin_amdgcn_perm(g[4], g[2], 0x07060302u);
    o[2] = g[5];
}
// 4 BGR byte pixels (3 words) -> E / O vectors of int16
__device__ inline void oct_split_u8(uint32_t x, uint32_t y, uint32_t z, uint32_t e[3], uint32_t o[3])
{
    e[0] = __builtin_amdgcn_perm(x, x, 0x0c010c00u);
    e[1] = __builtin_amdgcn_perm(y, x, 0x0c060c02u);
    e[2] = __builtin_amdgcn_perm(z, y, 0x0c040c03u);
    o[0] = __builtin_amdgcn_perm(y, x, 0x0c040c03u);
    o[1] = __builtin_amdgcn_perm(z, y, 0x0c050c01u);
    o[2] = __builtin_amdgcn_perm(z, z, 0x0c030c02u);
}

// packed pyrUp of parent pixels sx-1 .. sx+2 (all four exist) of three rows -> E / O vectors of output rows 2sy, 2sy+1
// last_dup: pixel sx+2 is past the level's right edge, pyrUp repeats pixel sx+1 there (only the last lane of an image row).
// The parent is a u8x3 level (8-bit fed pyramids keep 8-bit levels).  The rows point at pixel sx-1 (sx even): byte 3 sx - 3 of a 4-byte
// aligned row, i.e. 1 or 3 bytes past an aligned address (the same for the three rows: pitches are multiples of 16).  One aligned 16-byte
// read per row holds P0..P3 from that byte on; v_alignbyte brings them to byte 0, v_perm zero-extends the pairs pyrUp combines.
__device__ inline void pyr_up_oct_pk8(const char *p0, const char *p1, const char *p2, OctPk &up, bool last_dup = false)
{
    uint32_t he[3][3], ho[3][3];
    const char *rp[3] = {p0, p1, p2};
    const uint32_t o = (uint32_t)(uintptr_t)p1 & 3u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const u32x4_a4 d = *(const u32x4_a4 *)(rp[r] - o);
        // bytes b0..b11 = P0 P1 P2 P3
        const uint32_t e0 = __builtin_amdgcn_alignbyte(d.y, d.x, o), e1 = __builtin_amdgcn_alignbyte(d.z, d.y, o), e2 = __builtin_amdgcn_alignbyte(d.w, d.z, o);
        // V01 = (P0, P1) = b0..b5, V12 = (P1, P2) = b3..b8, V23 = (P2, P3) = b6..b11 (with last_dup (P2, P2) = b6 b7 b8 b6 b7 b8), two samples per word
        const uint32_t v01[3] = {__builtin_amdgcn_perm(e0, e0, 0x0c010c00u), __builtin_amdgcn_perm(e0, e0, 0x0c030c02u), __builtin_amdgcn_perm(e1, e1, 0x0c010c00u)};
        const uint32_t v12[3] = {__builtin_amdgcn_perm(e1, e0, 0x0c040c03u), __builtin_amdgcn_perm(e1, e1, 0x0c020c01u), __builtin_amdgcn_perm(e2, e1, 0x0c040c03u)};
        const uint32_t v23[3] = {__builtin_amdgcn_perm(e1, e1, 0x0c030c02u),
                                 last_dup ? __builtin_amdgcn_perm(e2, e1, 0x0c020c04u) : __builtin_amdgcn_perm(e2, e2, 0x0c010c00u),
                                 last_dup ? v12[2] : __builtin_amdgcn_perm(e2, e2, 0x0c030c02u)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            he[r][k] = pk_add(pk_mad6(v12[k], v01[k]), v23[k]);   // (P0 + 6 P1 + P2, P1 + 6 P2 + P3)
            ho[r][k] = pk_shl2(pk_add(v12[k], v23[k]));           // (4 (P1 + P2), 4 (P2 + P3))
        }
    }
    const uint32_t c32 = 0x00200020u, c8 = 0x00080008u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        up.e[0][k] = pk_asr<6>(pk_add(pk_add(pk_mad6(he[1][k], he[0][k]), he[2][k]), c32));
        up.o[0][k] = pk_asr<6>(pk_add(pk_add(pk_mad6(ho[1][k], ho[0][k]), ho[2][k]), c32));
        up.e[1][k] = pk_asr<4>(pk_add(pk_add(he[1][k], he[2][k]), c8));   // ((a + b) * 4 + 32) >> 6
        up.o[1][k] = pk_asr<4>(pk_add(pk_add(ho[1][k], ho[2][k]), c8));
    }
}

// sample (pixel q = row*4 + column, channel c) of an OctPk as a sign-extended int
__device__ inline int oct_get(const OctPk &v, int q, int c)
{
    const int r = q >> 2, col = q & 3, i = 3 * (col >> 1) + c;
    const uint32_t w = (col & 1) ? v.o[r][i >> 1] : v.e[r][i >> 1];
    return (i & 1) ? ((int)w >> 16) : (int)(int16_t)(uint16_t)(w & 0xffffu);
}

// pack 8 pixels x 3 channels of ints (low 16 bits each) into the E / O vectors
__device__ inline void oct_pack(const int t[8][3], OctPk &v)
{
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i0 = 2 * k, i1 = 2 * k + 1;
            v.e[r][k] = ((uint32_t)t[4 * r + 2 * (i0 / 3)][i0 % 3] & 0xffffu) | ((uint32_t)t[4 * r + 2 * (i1 / 3)][i1 % 3] << 16);
            v.o[r][k] = ((uint32_t)t[4 * r + 1 + 2 * (i0 / 3)][i0 % 3] & 0xffffu) | ((uint32_t)t[4 * r + 1 + 2 * (i1 / 3)][i1 % 3] << 16);
        }
}
// E / O vectors of one row -> 4 natural-order int16x3 pixels (6 words)
__device__ inline void oct_merge_row(const uint32_t e[3], const uint32_t o[3], uint32_t n[6])
{
    n[0] = e[0];
    n[1] = __builtin_amdgcn_perm(o[0], e[1], 0x05040100u);   // (c0 ch2, c1 ch0)
    n[2] = __builtin_amdgcn_alignbit(o[1], o[0], 16);        // (c1 ch1, c1 ch2)
    n[3] = __builtin_amdgcn_alignbit(e[2], e[1], 16);        // (c2 ch0, c2 ch1)
    n[4] = __builtin_amdgcn_perm(o[1], e[2], 0x07060302u);   // (c2 ch2, c3 ch0)
    n[5] = o[2];
}

// Kernel structure (the level kernels are bound by dependent memory round trips at modest occupancy, not by bytes):
//   phase A  the parent-level loads and the weight probes of up to 4 images are issued back to back; the probes only produce
//            two wave-uniform bit sets: images that contribute to this wave, and images whose weights are all exactly 1;
//   phase B  one round trip per contributing image (its samples, its parent neighbourhood and -- general path -- its
//            weights again, now L1/L2 hits);
//   all sums are kept packed (mod 2^16, like C's `short +=`).
template <bool LEVEL0, bool PK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PK ? (LEVEL0 ? 6 : 5) : 1, PK ? (LEVEL0 ? 6 : 5) : 8))) void k_blend_oct(const LevelArgs a)
{
    // 1-D grid of 256 x 8 tiles.  Work-groups are dealt round robin to the 8 XCDs (each with its own L2).  An XCD takes chunks of 4
    // tile rows, the chunks interleaved over the XCDs: vertically adjacent tiles, which share parent-level rows, mostly share an L2,
    // and every XCD still sees every part of the panorama (whole bands per XCD leave the XCDs with unequal work: measured slower).
    int t;
    {
        const int C = 4 * a.gx, xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        t = ((i / C) * 8 + xcd) * C + i % C;
    }
    const int by = t / a.gx, bx = t - by * a.gx;
    const int X0 = a.cx0 + 4 * (bx * 64 + (threadIdx.x & 63));
    const int Y0 = __builtin_amdgcn_readfirstlane(a.cy0 + 2 * (by * 4 + (threadIdx.x >> 6)));
    // cx0, cw are multiples of 4 and cy0, ch of 2 at these levels: an octet is inside or outside as a whole
    const bool inside = X0 < a.cx0 + a.cw && Y0 < a.cy0 + a.ch;
    const int bx0 = a.cx0 + bx * 256, by0 = a.cy0 + by * 8;
    const float inv255 = (float)(1. / 255.);
    OctPk pacc;
    float ws[8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) pacc.e[r][k] = pacc.o[r][k] = 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) ws[q] = 0.f;

    // ---- parent level (collapsed level l+1): region array without apron -> explicit border rules; loads issued now, used last
    const int psx = X0 >> 1, psy = Y0 >> 1;
    const int xlo = a.px0, xhi = a.px0 + a.prw - 1, ylo = a.py0, yhi = a.py0 + a.prh - 1;
    const bool par_fast = inside && psx - 2 >= xlo && psx + 3 <= xhi;   // the aligned 32-byte row reads start at pixel psx-2
    constexpr int NB = LEVEL0 ? 4 : 2;
    for (int g0 = 0; g0 < a.n_imgs; g0 += 32) {
        const int gend = min(a.n_imgs, g0 + 32);
        uint32_t contrib = 0u, allone = 0u;
        // images whose rectangle meets this tile (host-built table; without one: every image of the group, tested below)
        uint32_t cand = a.tmask ? a.tmask[(size_t)t * (size_t)a.tgroups + (size_t)(g0 >> 5)] : (gend - g0 >= 32 ? 0xffffffffu : (1u << (gend - g0)) - 1u);
        cand = __builtin_amdgcn_readfirstlane(cand);
        // ---- phase A: weight probes, NB images per round trip (level 0: 4 x two mask words; other levels: 2 x two float4 rows -- four
        // would hold 32 registers of weights and cost the kernel a wave per SIMD)
        while (cand) {
            uint32_t m0[NB], m1[NB];      // level 0: the two mask words
            f32x4_a4 f0[NB], f1[NB];      // other levels: the two weight rows
            bool inr[NB], edge[NB];
            int idx[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                idx[k] = cand ? g0 + __builtin_ctz(cand) : -1;
                cand &= cand - 1u;         // (0 stays 0)
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                inr[k] = false; edge[k] = false;
                const int i = idx[k];
                if (i < 0) continue;
                const LevelImg &im = a.imgs[i];
                if (!a.tmask && (bx0 + 256 <= im.rx || bx0 >= im.rx + im.pw || by0 + 8 <= im.ry || by0 >= im.ry + im.ph)) continue;
                const int lx = X0 - im.rx, ly = Y0 - im.ry;  // multiples of 4 and 2: rectangle origins are multiples of 2^(bands - l)
                inr[k] = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
                edge[k] = (lx >> 1) + 2 >= im.pwn;
                if (inr[k]) {
                    if (LEVEL0) {
                        const uint8_t *mp = (const uint8_t *)im.w + (size_t)ly * im.wp + lx;
                        m0[k] = *(const u32_u1 *)mp; m1[k] = *(const u32_u1 *)(mp + im.wp);
                    } else {
                        const char *wp_ = (const char *)im.w + (size_t)ly * im.wp + (size_t)lx * 4;
                        f0[k] = *(const f32x4_a4 *)wp_; f1[k] = *(const f32x4_a4 *)(wp_ + im.wp);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int i = idx[k];
                if (i < 0) continue;
                bool any = false, one = false;
                if (inr[k]) {
                    if (LEVEL0) {
                        any = (m0[k] | m1[k]) != 0u;
                        one = (m0[k] & m1[k]) == 0xffffffffu;
                    } else {
                        any = f0[k].x != 0.f || f0[k].y != 0.f || f0[k].z != 0.f || f0[k].w != 0.f || f1[k].x != 0.f || f1[k].y != 0.f || f1[k].z != 0.f || f1[k].w != 0.f;
                        one = f0[k].x == 1.f && f0[k].y == 1.f && f0[k].z == 1.f && f0[k].w == 1.f && f1[k].x == 1.f && f1[k].y == 1.f && f1[k].z == 1.f && f1[k].w == 1.f;
                    }
                }
                // a wave whose weights are all zero contributes (short)(L*0) = 0 and w + 0: the image is skipped
                if (__ballot(inr[k] && any) != 0ULL) contrib |= 1u << (i - g0);
                // every lane that touches the image has all 8 weights exactly 1 and a complete parent neighbourhood
                if (PK && __ballot(inr[k] && !(one && !edge[k])) == 0ULL) allone |= 1u << (i - g0);
            }
        }
        // ---- phase B: contributing images, feed order
        while (contrib) {
            const int bi = __builtin_ctz(contrib);
            contrib &= contrib - 1u;
            const LevelImg &im = a.imgs[g0 + bi];
            const int lx = X0 - im.rx, ly = Y0 - im.ry;
            const bool in = inside && (unsigned)lx < (unsigned)im.pw && (unsigned)ly < (unsigned)im.ph;
            if (!in) continue;   // no cross-lane operation below
            const int sx = lx >> 1, sy = ly >> 1;
            const bool l8 = PK || im.lvl8;     // (packed paths: every image was fed 8-bit and keeps 8-bit levels)
            const char *r1 = (const char *)im.gn + (ptrdiff_t)sy * (ptrdiff_t)im.gnp + (ptrdiff_t)(sx - 1) * (l8 ? 3 : 6);
            // row -1 is the reflect-101 apron (= row 1, pyrUp's rule); row phn is not: pyrUp repeats the last row
            const char *r0 = r1 - (ptrdiff_t)im.gnp, *r2 = sy + 1 >= im.phn ? r1 : r1 + (ptrdiff_t)im.gnp;
            if (PK && ((allone >> bi) & 1u)) {
                // (short)(L * 1.f) = L: the whole contribution is packed integer arithmetic
                uint32_t ge[2][3], go[2][3];
                {
                    // four BGR pixels per row = 12 bytes (level 0: the frame; other levels: 12-byte steps from a 4-byte aligned origin)
                    const uint8_t *p = (const uint8_t *)im.g + (size_t)ly * im.gp + (size_t)lx * 3;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const u32x3_u1 v = *(const u32x3_u1 *)(p + (size_t)r * im.gp);
                        oct_split_u8(v.x, v.y, v.z, ge[r], go[r]);
                    }
                }
                OctPk up;
                pyr_up_oct_pk8(r0, r1, r2, up);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        pacc.e[r][k] = pk_add(pacc.e[r][k], pk_sub_sat(ge[r][k], up.e[r][k]));
                        pacc.o[r][k] = pk_add(pacc.o[r][k], pk_sub_sat(go[r][k], up.o[r][k]));
                    }
#pragma unroll
                for (int q = 0; q < 8; ++q) ws[q] += 1.f;
                continue;
            }
            // general path
            float w[8];
            if (LEVEL0) {
                const uint8_t *mp = (const uint8_t *)im.w + (size_t)ly * im.wp + lx;
                const uint32_t m0 = *(const u32_u1 *)mp, m1 = *(const u32_u1 *)(mp + im.wp);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    w[k] = (float)((m0 >> (8 * k)) & 0xffu) * inv255;
                    w[4 + k] = (float)((m1 >> (8 * k)) & 0xffu) * inv255;
                }
            } else {
                const char *wp_ = (const char *)im.w + (size_t)ly * im.wp + (size_t)lx * 4;
                const f32x4_a4 w0 = *(const f32x4_a4 *)wp_, w1 = *(const f32x4_a4 *)(wp_ + im.wp);
                w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
                w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
            }
            if (PK) {
                // 8-bit fed pyramids: Laplacian packed as above, then (short)(L * w) sample by sample
                uint32_t ge[2][3], go[2][3];
                {
                    const uint8_t *p = (const uint8_t *)im.g + (size_t)ly * im.gp + (size_t)lx * 3;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const u32x3_u1 v = *(const u32x3_u1 *)(p + (size_t)r * im.gp);
                        oct_split_u8(v.x, v.y, v.z, ge[r], go[r]);
                    }
                }
                OctPk lap;
                pyr_up_oct_pk8(r0, r1, r2, lap, sx + 2 >= im.pwn);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        lap.e[r][k] = pk_sub_sat(ge[r][k], lap.e[r][k]);
                        lap.o[r][k] = pk_sub_sat(go[r][k], lap.o[r][k]);
                    }
                // word k of E holds samples i = 2k, 2k+1 of columns (0, 2): pixel 4r + 2 (i / 3); O the same for columns (1, 3)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float wel = w[4 * r + 2 * ((2 * k) / 3)], weh = w[4 * r + 2 * ((2 * k + 1) / 3)];
                        const float wol = w[4 * r + 1 + 2 * ((2 * k) / 3)], woh = w[4 * r + 1 + 2 * ((2 * k + 1) / 3)];
                        const uint32_t le = lap.e[r][k], lo = lap.o[r][k];
                        const int el = (int)((float)(int)(int16_t)(uint16_t)(le & 0xffffu) * wel), eh = (int)((float)((int)le >> 16) * weh);
                        const int ol = (int)((float)(int)(int16_t)(uint16_t)(lo & 0xffffu) * wol), oh = (int)((float)((int)lo >> 16) * woh);
                        pacc.e[r][k] = pk_add(pacc.e[r][k], ((uint32_t)el & 0xffffu) | ((uint32_t)eh << 16));
                        pacc.o[r][k] = pk_add(pacc.o[r][k], ((uint32_t)ol & 0xffffu) | ((uint32_t)oh << 16));
                    }
#pragma unroll
                for (int q = 0; q < 8; ++q) ws[q] += w[q];
                continue;
            }
            int g[8][3];
            if (LEVEL0 ? im.src_depth == SSP_U8 : (bool)im.lvl8) {
                // four BGR pixels per row = 12 bytes
                const uint8_t *p = (const uint8_t *)im.g + (size_t)ly * im.gp + (size_t)lx * 3;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const u32x3_u1 v = *(const u32x3_u1 *)(p + (size_t)r * im.gp);
                    const uint32_t wd[3] = {v.x, v.y, v.z};
#pragma unroll
                    for (int k = 0; k < 12; ++k) g[4 * r + k / 3][k % 3] = (int)((wd[k >> 2] >> (8 * (k & 3))) & 0xffu);
                }
            } else {
                const char *p = (const char *)im.g + (size_t)ly * im.gp + (size_t)lx * 6;
                load_s16x12(p, &g[0]);
                load_s16x12(p + im.gp, &g[4]);
            }
            int up[8][3];
            pyr_up_oct(r0, r1, r2, sx + 2 >= im.pwn, up, l8);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int c = 0; c < 3; ++c) g[q][c] = (int)((float)sat16(g[q][c] - up[q][c]) * w[q]);  // the (short) wrap is the packed add below
                ws[q] += w[q];
            }
            OctPk t;
            oct_pack(g, t);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pacc.e[r][k] = pk_add(pacc.e[r][k], t.e[r][k]);
                    pacc.o[r][k] = pk_add(pacc.o[r][k], t.o[r][k]);
                }
        }
    }
    if (!inside) return;
    if (a.ext_lap) {
        // partial sums imported from other GPUs
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const char *p = (const char *)a.ext_lap + (size_t)(Y0 + r) * a.elp + (size_t)X0 * 6;
            const u32x4_a2 v0 = *(const u32x4_a2 *)p;
            const u32x2_a2 v1 = *(const u32x2_a2 *)(p + 16);
            const uint32_t gw[6] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y};
            uint32_t e[3], o[3];
            oct_split_s16(gw, e, o);
#pragma unroll
            for (int k = 0; k < 3; ++k) { pacc.e[r][k] = pk_add(pacc.e[r][k], e[k]); pacc.o[r][k] = pk_add(pacc.o[r][k], o[k]); }
            const f32x4_a4 ew = *(const f32x4_a4 *)((const char *)a.ext_w + (size_t)(Y0 + r) * a.ewp + (size_t)X0 * 4);
            ws[4 * r] += ew.x; ws[4 * r + 1] += ew.y; ws[4 * r + 2] += ew.z; ws[4 * r + 3] += ew.w;
        }
    }
    // ---- this level's step of restoreImageFromLaplacePyr: pyrUp of the collapsed parent level
    // Packed form: when every parent sample a wave reads lies in [-512, 511] (a collapsed pyramid of 8-bit frames leaves [0, 255] only by the
    // overshoot at seams) every intermediate of pyrUp fits a signed 16-bit lane (64 * 511 + 32 < 2^15, -64 * 512 + 32 >= -2^15) and the two
    // passes run on two samples per instruction, straight in the E / O layout: samples 3..14 of the 16 read are P0..P3, so (P1, P2) are words
    // 3..5 as read and (P0, P1), (P2, P3) one v_alignbit each.  The horizontal pass runs row by row as the rows arrive (the raw words die at
    // once); a lane whose samples leave the range, like the first / last octet of a region row, takes the 2x2 form's 32-bit pyrUp instead.
    OctPk upp;
    const char *pbase = nullptr;
    int prow[3] = {0, 0, 0};
    bool pk_parent = par_fast;
    uint32_t phe[3][3] = {}, pho[3][3] = {};
    if (par_fast) {
        int ym = psy - 1 < 0 ? min(1, a.ph - 1) : psy - 1, yp = psy + 1 >= a.ph ? a.ph - 1 : psy + 1;
        ym = min(max(ym, ylo), yhi); yp = min(max(yp, ylo), yhi);
        pbase = (const char *)a.parent + (ptrdiff_t)(psx - 2 - a.px0) * 6;   // psx, px0 even: 4-byte aligned
        prow[0] = ym - a.py0; prow[1] = psy - a.py0; prow[2] = yp - a.py0;
        uint32_t acc = 0u;
        auto row = [&](int r, const u32x4_a4 v0, const u32x4_a4 v1) {
            const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int k = 1; k < 8; ++k) acc |= pk_add(w[k], 0x02000200u);
            const uint32_t v01[3] = {__builtin_amdgcn_alignbit(w[2], w[1], 16), __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(w[4], w[3], 16)};
            const uint32_t v23[3] = {__builtin_amdgcn_alignbit(w[5], w[4], 16), __builtin_amdgcn_alignbit(w[6], w[5], 16), __builtin_amdgcn_alignbit(w[7], w[6], 16)};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                phe[r][k] = pk_add(pk_mad6(w[3 + k], v01[k]), v23[k]);   // (P0 + 6 P1 + P2, P1 + 6 P2 + P3)
                pho[r][k] = pk_shl2(pk_add(w[3 + k], v23[k]));           // (4 (P1 + P2), 4 (P2 + P3))
            }
        };
        const char *p0 = pbase + (size_t)prow[0] * a.pp, *p1 = pbase + (size_t)prow[1] * a.pp, *p2 = pbase + (size_t)prow[2] * a.pp;
        const u32x4_a4 a0 = *(const u32x4_a4 *)p0, a1 = *(const u32x4_a4 *)(p0 + 16), b0 = *(const u32x4_a4 *)p1, b1 = *(const u32x4_a4 *)(p1 + 16);
        const u32x4_a4 c0 = *(const u32x4_a4 *)p2, c1 = *(const u32x4_a4 *)(p2 + 16);
        row(0, a0, a1);
        row(1, b0, b1);
        row(2, c0, c1);
        pk_parent = (acc & 0xfc00fc00u) == 0u;
    }
    {
        const uint32_t c32 = 0x00200020u, c8 = 0x00080008u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            upp.e[0][k] = pk_asr<6>(pk_add(pk_add(pk_mad6(phe[1][k], phe[0][k]), phe[2][k]), c32));
            upp.o[0][k] = pk_asr<6>(pk_add(pk_add(pk_mad6(pho[1][k], pho[0][k]), pho[2][k]), c32));
            upp.e[1][k] = pk_asr<4>(pk_add(pk_add(phe[1][k], phe[2][k]), c8));   // ((a + b) * 4 + 32) >> 6
            upp.o[1][k] = pk_asr<4>(pk_add(pk_add(pho[1][k], pho[2][k]), c8));
        }
    }
    if (!pk_parent) {
        // first / last octet of a region row, or a sample outside the packed range: the 2x2 form (clamped border rules, 32-bit integers)
        int q0[4][3], q1[4][3], up[8][3];
        pyr_up_quad<false>(a.parent, a.pp, a.pw, a.ph, a.px0, a.py0, a.prw, a.prh, psx, psy, q0);
        pyr_up_quad<false>(a.parent, a.pp, a.pw, a.ph, a.px0, a.py0, a.prw, a.prh, psx + 1, psy, q1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            up[0][c] = q0[0][c]; up[1][c] = q0[1][c]; up[2][c] = q1[0][c]; up[3][c] = q1[1][c];
            up[4][c] = q0[2][c]; up[5][c] = q0[3][c]; up[6][c] = q1[2][c]; up[7][c] = q1[3][c];
        }
        oct_pack(up, upp);   // pyrUp of int16 samples is an int16
    }
    // ---- normalizeUsingWeightMap + collapse
    OctPk res;
    bool unit = true, onetwo = true;
#pragma unroll
    for (int q = 0; q < 8; ++q) { unit = unit && ws[q] == 1.f; onetwo = onetwo && (ws[q] == 1.f || ws[q] == 2.f); }
    uint32_t vmask0 = 0xffffffffu, vmask1 = 0xffffffffu;  // level 0: bytes of the result mask (ws > WEIGHT_EPS)
    if (__ballot(!unit) == 0ULL) {
        // every weight sum of the wave is exactly 1: (short)(n / (1 + 1e-5f)) = n - sign(n) for every int16 n (the quotient
        // lies strictly between n - sign(n) and n because |n| * 1e-5 < 1, and truncation goes towards zero)
        const uint32_t one = 0x00010001u, mone = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t ne = pacc.e[r][k], no = pacc.o[r][k];
                const uint32_t qe = pk_sub(ne, pk_max(pk_min(ne, one), mone)), qo = pk_sub(no, pk_max(pk_min(no, one), mone));
                res.e[r][k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, upp.e[r][k]), __builtin_bit_cast(s16x2, qe)));
                res.o[r][k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, upp.o[r][k]), __builtin_bit_cast(s16x2, qo)));
            }
    } else if (__ballot(!onetwo) == 0ULL) {
        // every weight sum is exactly 1 or 2 (one image, or the interior of a two-image overlap): (short)(n / (2 + 1e-5f)) is
        // (n - sign(n)) / 2 truncated towards zero -- the quotient lies strictly between that value's neighbours because
        // |n| * 2.5e-6 < 1/2.  Both candidates are computed packed and selected per pixel.
        const uint32_t one = 0x00010001u, mone = 0xffffffffu;
        const u16x2v s15 = {15, 15};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t t0 = ws[4 * r] == 2.f ? 0xffffffffu : 0u, t1 = ws[4 * r + 1] == 2.f ? 0xffffffffu : 0u, t2 = ws[4 * r + 2] == 2.f ? 0xffffffffu : 0u,
                           t3 = ws[4 * r + 3] == 2.f ? 0xffffffffu : 0u;
            const uint32_t me[3] = {t0, (t0 & 0xffffu) | (t2 & 0xffff0000u), t2}, mo[3] = {t1, (t1 & 0xffffu) | (t3 & 0xffff0000u), t3};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t ne = pacc.e[r][k], no = pacc.o[r][k];
                const uint32_t q1e = pk_sub(ne, pk_max(pk_min(ne, one), mone)), q1o = pk_sub(no, pk_max(pk_min(no, one), mone));
                const uint32_t q2e = pk_asr<1>(pk_add(q1e, __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, q1e) >> s15))));
                const uint32_t q2o = pk_asr<1>(pk_add(q1o, __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, q1o) >> s15))));
                const uint32_t qe = (q2e & me[k]) | (q1e & ~me[k]), qo = (q2o & mo[k]) | (q1o & ~mo[k]);
                res.e[r][k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, upp.e[r][k]), __builtin_bit_cast(s16x2, qe)));
                res.o[r][k] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, upp.o[r][k]), __builtin_bit_cast(s16x2, qo)));
            }
        }
    } else {
        int n[8][3];
        vmask0 = vmask1 = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float den = ws[q] + WEIGHT_EPS;
            float qn[3];
            div3_exact((float)oct_get(pacc, q, 0), (float)oct_get(pacc, q, 1), (float)oct_get(pacc, q, 2), den, qn);
            // level 0: compare(dst_band_weights_0, WEIGHT_EPS, CMP_GT); dst.setTo(0, mask == 0)
            const bool valid = !LEVEL0 || ws[q] > WEIGHT_EPS;
#pragma unroll
            for (int c = 0; c < 3; ++c) n[q][c] = valid ? sat16(oct_get(upp, q, c) + trunc16(qn[c])) : 0;
            if (LEVEL0 && valid) { if (q < 4) vmask0 |= 0xffu << (8 * q); else vmask1 |= 0xffu << (8 * (q - 4)); }
        }
        oct_pack(n, res);
    }
    if (!LEVEL0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            char *d = (char *)a.out + (size_t)(Y0 - a.cy0 + r) * a.op + (size_t)(X0 - a.cx0) * 6;  // 24-byte steps on a 16-byte aligned row
            uint32_t o[6];
            oct_merge_row(res.e[r], res.o[r], o);
            u32x4_a2 o4; o4.x = o[0]; o4.y = o[1]; o4.z = o[2]; o4.w = o[3];
            u32x2_a2 o2; o2.x = o[4]; o2.y = o[5];
            *(u32x4_a2 *)d = o4;
            *(u32x2_a2 *)(d + 16) = o2;
        }
        return;
    }
    // crop to dst_roi_final_
    const int ox = X0 - a.ox0, oy = Y0 - a.oy0;
    if (X0 + 4 <= a.fw && Y0 + 2 <= a.fh) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            uint32_t o[6];
            oct_merge_row(res.e[r], res.o[r], o);
            if (a.rmask) __builtin_nontemporal_store(r ? vmask1 : vmask0, (u32_u1 *)(a.rmask + (size_t)(oy + r) * a.rmp + ox));   // final outputs: written once, never read by this pipeline
            if (a.mosaic) {
                // cv.imwrite's convertTo(CV_8U) saturation, sde.py:1938
                uint32_t b[3];
                const uint32_t zero = 0u, c255 = 0x00ff00ffu;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    b[k] = __builtin_amdgcn_perm(pk_min(pk_max(o[2 * k + 1], zero), c255), pk_min(pk_max(o[2 * k], zero), c255), 0x06040200u);
                u32x3_u1 m; m.x = b[0]; m.y = b[1]; m.z = b[2];
                __builtin_nontemporal_store(m, (u32x3_u1 *)(a.mosaic + (size_t)(oy + r) * a.mp + (size_t)ox * 3));
            }
            if (a.result) {
                char *d = (char *)a.result + (size_t)(oy + r) * a.rp + (size_t)ox * 6;
                u32x4_a2 o4; o4.x = o[0]; o4.y = o[1]; o4.z = o[2]; o4.w = o[3];
                u32x2_a2 o2; o2.x = o[4]; o2.y = o[5];
                *(u32x4_a2 *)d = o4;
                *(u32x2_a2 *)(d + 16) = o2;
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int X = X0 + (q & 3), Y = Y0 + (q >> 2);
        if (X >= a.fw || Y >= a.fh) continue;
        const int px = X - a.ox0, py = Y - a.oy0;
        const uint32_t vm = q < 4 ? vmask0 >> (8 * q) : vmask1 >> (8 * (q - 4));
        if (a.rmask) a.rmask[(size_t)py * a.rmp + px] = (uint8_t)vm;
        if (a.result) {
            int16_t *d = (int16_t *)((char *)a.result + (size_t)py * a.rp) + (size_t)px * 3;
            for (int c = 0; c < 3; ++c) d[c] = (int16_t)oct_get(res, q, c);
        }
        if (a.mosaic) {
            uint8_t *d = a.mosaic + (size_t)py * a.mp + (size_t)px * 3;
            for (int c = 0; c < 3; ++c) d[c] = (uint8_t)min(max(oct_get(res, q, c), 0), 255);
        }
    }
}

// ====================================================================================================================
// multi-GPU: add imported partial sums
// ====================================================================================================================
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


#ifdef SSP_MB_F32_TU
}  // namespace
#endif
namespace ssp {
// launchers of the float-pyramid kernels: defined in the unit that is compiled without the SLP vectoriser (see the top of the file)
// (the argument blocks travel as untyped pointers: in the float unit their types live in its anonymous namespace)
void mb_f32_pyr_down(bool level0, bool apr, dim3 grid, hipStream_t st, const void *pyr_down_batch);
void mb_f32_blend_level(bool level0, dim3 grid, dim3 block, hipStream_t st, const void *level_args);
void mb_f32_blend_quad(bool level0, dim3 grid, dim3 block, hipStream_t st, const void *level_args);
}  // namespace ssp
#ifdef SSP_MB_F32_TU
namespace ssp {
void mb_f32_pyr_down(bool level0, bool apr, dim3 grid, hipStream_t st, const void *pyr_down_batch)
{
    const PyrDownBatch &pb = *(const PyrDownBatch *)pyr_down_batch;
    if (level0 && apr) hipLaunchKernelGGL((k_pyr_down_float<true, true>), grid, dim3(256), 0, st, pb);
    else if (level0) hipLaunchKernelGGL((k_pyr_down_float<true, false>), grid, dim3(256), 0, st, pb);
    else if (apr) hipLaunchKernelGGL((k_pyr_down_float<false, true>), grid, dim3(256), 0, st, pb);
    else hipLaunchKernelGGL((k_pyr_down_float<false, false>), grid, dim3(256), 0, st, pb);
}
void mb_f32_blend_level(bool level0, dim3 grid, dim3 block, hipStream_t st, const void *level_args)
{
    const LevelArgs &a = *(const LevelArgs *)level_args;
    if (level0) hipLaunchKernelGGL((k_blend_level<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_blend_level<false, true>), grid, block, 0, st, a);
}
void mb_f32_blend_quad(bool level0, dim3 grid, dim3 block, hipStream_t st, const void *level_args)
{
    const LevelArgs &a = *(const LevelArgs *)level_args;
    if (level0) hipLaunchKernelGGL((k_blend_quad<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_blend_quad<false, true>), grid, block, 0, st, a);
}
}  // namespace ssp
#else
// ====================================================================================================================
// host side
// ====================================================================================================================
namespace ssp {

static size_t plane_pitch(int w, int bpp, int lead = 0) { return align_up((size_t)lead + (size_t)(w + 2 * APRON + 4) * bpp, 16); }
// a strip buffer: w*h*cn packed rows, or (strip_planes) the memory of a whole level-0 plane of that size, apron included
#define STRIP_SLACK 256   // bytes kept free before and after a plane inside a strip buffer (pool blocks have their bucket rounding instead)
size_t mb_strip_buffer_bytes(int w, int h, int bpp, bool planes)
{
    return planes ? plane_pitch(w, bpp) * (size_t)(h + 2 * APRON) + 2 * STRIP_SLACK : (size_t)w * h * bpp;
}
static int alloc_plane(int w, int h, int bpp, int lead, Plane &p)
{
    const int A = APRON;
    p.pitch = plane_pitch(w, bpp, lead);
    SSP_TRY(pool_alloc(p.pitch * (size_t)(h + 2 * A), &p.alloc));
    p.base = (char *)p.alloc + (size_t)A * p.pitch + lead + (size_t)A * bpp;
    return 0;
}
static void free_rec(const ssp_blender *b, FeedRec &f)
{
    for (int l = 0; l <= b->num_bands; ++l) {
        pool_free(f.G[l].alloc); pool_free(f.W[l].alloc);
        f.G[l] = Plane(); f.W[l] = Plane();
    }
}

void mb_release(ssp_blender *b)
{
    for (auto &f : b->feeds) free_rec(b, f);
    b->feeds.clear();
    b->pending = 0;
    b->obj_pending = false;
    b->border_done = false;
    for (int l = 0; l <= MAX_BANDS; ++l) { image_unref(b->ext_lap[l]); image_unref(b->ext_w[l]); b->ext_lap[l] = b->ext_w[l] = nullptr; }
}

// MultiBandBlender::feed geometry: grow by gap, clip to the pano, snap to multiples of 2^nb, shift back inside
static int make_feed_rec(ssp_blender *b, int iw, int ih, int tlx, int tly, int depth, FeedRec &f)
{
    const int nb = b->num_bands, m = 1 << nb;
    const int rx = b->roi[0], ry = b->roi[1], rbx = rx + b->roi[2], rby = ry + b->roi[3];
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
    f.iw = iw; f.ih = ih; f.left = left; f.top = top;
    f.g0_depth = depth;
    f.lvl8 = depth == SSP_U8 && !b->float_mode;
    f.pw[0] = width; f.ph[0] = height;
    int x_tl = tnx - rx, y_tl = tny - ry;
    for (int l = 0; l <= nb; ++l) {
        if (l > 0) { f.pw[l] = (f.pw[l - 1] + 1) / 2; f.ph[l] = (f.ph[l - 1] + 1) / 2; }
        f.rx[l] = x_tl; f.ry[l] = y_tl;
        x_tl /= 2; y_tl /= 2;
        f.G[l] = Plane(); f.W[l] = Plane();
    }
    const int A = APRON;
    // every plane OF A LOCALLY FED IMAGE is 4-byte aligned at its rectangle origin and at every multiple of 4 columns from it (16-byte row
    // pitch, 4-pixel apron); the warp kernel shifts its 4-column groups by (left mod 4) to store aligned, so all readers load aligned.
    // (Planes fed from another GPU's all-level strips -- mb_feed_level_strips -- start ((x0 - ox) >> l) * bytes-per-sample mod 16 bytes into
    // their rows: at the levels where that is not a multiple of 4 only the kernels that read through aligned(1) types run, k_blend_quad /
    // k_blend_level; the packed 4x2 kernel takes a level only when every image's plane origin is 4-byte aligned.  ADVICE r3.)
    const int bpp0 = 3 * depth_size(depth);
    const int lead_g = 0, lead_m = 0;
    (void)A;
    int rc = alloc_plane(width, height, bpp0, lead_g, f.G[0]);
    if (!rc) rc = alloc_plane(width, height, 1, lead_m, f.W[0]);
    for (int l = 1; l <= nb && !rc; ++l) {
        rc = alloc_plane(f.pw[l], f.ph[l], f.level_bytes(), 0, f.G[l]);
        if (!rc) rc = alloc_plane(f.pw[l], f.ph[l], 4, 0, f.W[l]);
    }
    if (rc) free_rec(b, f);
    return rc;
}

// The object API (ssp_blender_feed, one image per call) leaves the pyramids of its images pending: nothing of a fed image is observable before
// blend() (sde.py:1886 -> :1930), and built together every level of ALL fed images is one launch instead of one per image.  Whoever needs
// the pyramids first -- blend, an export, a batch of the composer -- builds them.
int mb_flush(ssp_blender *b)
{
    if (!b->obj_pending) return 0;
    b->obj_pending = false;
    return mb_feed_end(b);
}

int mb_feed_begin(ssp_blender *b, int n, const int *tls, const int *sizes, int depth, FeedSlot *slots, bool append)
{
    if (!append) SSP_TRY(mb_flush(b));
    if (b->pending && !append) SSP_FAIL(SSP_ERR_STATE, "feed: a previous batch was not finished");
    if (append && b->pending && (!b->obj_pending || b->border_done)) SSP_FAIL(SSP_ERR_STATE, "feed: a batch of another kind is pending");
    if (b->float_mode) SSP_REQUIRE(depth == SSP_F32, "feed: float mode needs CV_32FC3 images");
    else SSP_REQUIRE(depth == SSP_S16 || depth == SSP_U8, "feed: image must be CV_16SC3 or CV_8UC3");
    const size_t first = b->feeds.size();
    for (int i = 0; i < n; ++i) {
        FeedRec f;
        int rc = make_feed_rec(b, sizes[2 * i], sizes[2 * i + 1], tls[2 * i], tls[2 * i + 1], depth, f);
        if (rc) {
            while (b->feeds.size() > first) { free_rec(b, b->feeds.back()); b->feeds.pop_back(); }
            return rc;
        }
        slots[i].img = (uint8_t *)f.G[0].base + (size_t)f.top * f.G[0].pitch + (size_t)f.left * 3 * depth_size(depth);
        slots[i].ipitch = f.G[0].pitch;
        slots[i].mask = (uint8_t *)f.W[0].base + (size_t)f.top * f.W[0].pitch + f.left;
        slots[i].xshift = f.left & 3;
        slots[i].mpitch = f.W[0].pitch;
        b->feeds.push_back(f);
    }
    b->pending = append ? b->pending + n : n;
    if (append) b->obj_pending = true;
    return 0;
}

// border of level 0, then the Gaussian pyramids of the pending images; every stage is one launch per MB_MAXB images
// level-0 planes of the pending images: everything outside the image interiors (after this the planes can be exported)
int mb_feed_border(ssp_blender *b)
{
    const int n = b->pending;
    if (n == 0 || b->border_done) return 0;
    b->border_done = true;
    FeedRec *recs = &b->feeds[b->feeds.size() - n];
    const int A = APRON;
    for (int base = 0; base < n; base += MB_MAXB) {
        const int cnt = std::min(MB_MAXB, n - base);
        {
            Border0Batch bb;
            memset(&bb, 0, sizeof bb);
            long long items = 0;
            double bytes = 0;
            for (int i = 0; i < cnt; ++i) {
                const FeedRec &f = recs[base + i];
                Border0Desc &d = bb.d[i];
                d.g = f.G[0].base; d.gp = f.G[0].pitch; d.m = (uint8_t *)f.W[0].base; d.mp = f.W[0].pitch;
                d.iw = f.iw; d.ih = f.ih; d.left = f.left; d.top = f.top; d.pw = f.pw[0]; d.ph = f.ph[0]; d.depth = f.g0_depth;
                long long t = (long long)(f.pw[0] + 2 * A) * (f.ph[0] + 2 * A) - (long long)f.iw * f.ih;
                items = std::max(items, t);
                bytes += 2.0 * t * (3 * depth_size(f.g0_depth) + 1);
            }
            ProfileScope ps("border_l0", bytes);
            bool x4 = true;
            long long groups = 0;
            for (int i = 0; i < cnt; ++i) {
                const Border0Desc &d = bb.d[i];
                x4 = x4 && d.depth == SSP_U8 && d.pw % 4 == 0 && d.iw >= 4 && (double)(d.pw + 2 * A) * (d.ph + 2 * A) < 4.0e9;     // 32-bit group indices
                const long long gw = (d.pw + 2 * A) / 4;
                groups = std::max(groups, gw * (A + d.top) + gw * (d.ph + A - (d.top + d.ih)) + (long long)d.ih * ((d.left + A + 3) / 4) +
                                              (long long)d.ih * ((d.pw + A - ((d.left + d.iw) & ~3)) / 4));
            }
            if (x4) {
                // one block = 256 groups; per image exactly its own blocks
                bb.tm.cnt = cnt;
                int total = 0;
                for (int i = 0; i < cnt; ++i) {
                    const Border0Desc &d = bb.d[i];
                    const long long gw = (d.pw + 2 * A) / 4;
                    const long long g = gw * (A + d.top) + gw * (d.ph + A - (d.top + d.ih)) + (long long)d.ih * ((d.left + A + 3) / 4) +
                                        (long long)d.ih * ((d.pw + A - ((d.left + d.iw) & ~3)) / 4);
                    bb.tm.start[i] = total; bb.tm.tx[i] = 1;
                    total += (int)((g + 255) / 256);
                }
                bb.tm.start[cnt] = total;
                hipLaunchKernelGGL(k_border0_u8x4, dim3((unsigned)total), dim3(256), 0, stream(), bb);
            }
            else {
                bool f32 = items < (1LL << 31);
                for (int i = 0; i < cnt; ++i) f32 = f32 && bb.d[i].depth == SSP_F32;
                if (f32) hipLaunchKernelGGL(k_border0_f32, dim3((unsigned)((items + 255) / 256), 1, cnt), dim3(256), 0, stream(), bb);
                else hipLaunchKernelGGL(k_border0, dim3((unsigned)((items + 255) / 256), 1, cnt), dim3(256), 0, stream(), bb);
            }
        }
    }
    SSP_HIP(hipGetLastError());
    return 0;
}

// Gaussian pyramids of a list of fed images whose level-0 planes are complete (one launch per level and MB_MAXB images of one depth)
static int build_pyramids_same_depth(const ssp_blender *b, const std::vector<FeedRec *> &list)
{
    const int n = (int)list.size(), nb = b->num_bands;
    if (n == 0) return 0;
    FeedRec *const *recs = list.data();
    const int A = APRON;
    const bool lvl8 = recs[0]->lvl8;
    const int lb = recs[0]->level_bytes();
    for (int base = 0; base < n; base += MB_MAXB) {
        const int cnt = std::min(MB_MAXB, n - base);
        for (int l = 0; l < nb; ++l) {
            PyrDownBatch pb;
            memset(&pb, 0, sizeof pb);
            int mw = 0, mh = 0;
            double bytes = 0;
            for (int i = 0; i < cnt; ++i) {
                const FeedRec &f = *recs[base + i];
                PyrDownArgs &a = pb.a[i];
                a.g = f.G[l].base; a.gp = f.G[l].pitch; a.w = f.W[l].base; a.wp = f.W[l].pitch;
                a.dg = f.G[l + 1].base; a.dgp = f.G[l + 1].pitch; a.dw = f.W[l + 1].base; a.dwp = f.W[l + 1].pitch;
                a.dwid = f.pw[l + 1]; a.dhei = f.ph[l + 1];
                mw = std::max(mw, a.dwid); mh = std::max(mh, a.dhei);
                const double src_px = (double)f.pw[l] * f.ph[l], dst_px = (double)a.dwid * a.dhei;
                bytes += (l == 0 ? src_px * (3.0 * depth_size(f.g0_depth) + 1) : src_px * (lb + 4)) + dst_px * (lb + 4);
            }
            // the destination apron is written by the pyrDown kernel itself when every destination is at least 5 x 5; the strip
            // kernel additionally needs widths that are multiples of 4 (>= 8) and pays off on large levels only
            bool apr = true, strip = !b->float_mode && (l > 0 || recs[base]->g0_depth != SSP_F32);
            bool lds_ok = true;               // the LDS-staged forms address their planes with 32-bit byte offsets
            for (int i = 0; i < cnt; ++i) lds_ok = lds_ok && (double)(2.0 * pb.a[i].dhei + 2 * A) * (double)std::max(pb.a[i].gp, pb.a[i].wp) < 2147483648.0;
            for (int i = 0; i < cnt; ++i) {
                apr = apr && pb.a[i].dwid >= 5 && pb.a[i].dhei >= 5;
                strip = strip && pb.a[i].dwid % 4 == 0 && pb.a[i].dwid >= 8 && pb.a[i].dhei >= 5;
            }
            strip = strip && mh >= 512;       // (with the LDS-staged forms too: 256 / 128 / 64 measured equal or slower, profiles/r03_pyramid_variants.txt)
            {
                ProfileScope ps(l == 0 ? "pyr_down_l0" : "pyr_down", bytes);
                // 0: u8 frame + u8 mask -> u8 level, 1: int16 frame + u8 mask -> int16 level, 2: int16 level + f32 weights, 3: u8 level + f32 weights
                const int src = l == 0 ? (lvl8 ? 0 : 1) : (lvl8 ? 3 : 2);
                if (b->float_mode) {
                    dim3 grid((mw + PDF_OUT - 1) / PDF_OUT, (mh + 3) / 4, cnt);
                    mb_f32_pyr_down(l == 0, apr, grid, stream(), &pb);
                } else if (strip) {
                    // 4 columns x 4 rows per lane: tiles of 256 x 16 outputs.  (A strip re-reads 3 of its 11 source rows -- its neighbours'
                    // copies are long gone from L2 -- but taller strips, 8 or 16 rows, measured slower: too few, too long waves.)
                    pb.tm.cnt = cnt;
                    static const bool lds0 = !getenv("SSP_PYR_GLOBAL");      // (A/B switch of round 3; the global-load forms stay for pitches beyond 32 bits)
                    const int tw = src == 3 && lds0 && lds_ok ? LV_TW : 256; // the staged form of the 8-bit levels takes 252-column tiles
                    int total = 0;
                    for (int i = 0; i < cnt; ++i) {
                        pb.tm.start[i] = total; pb.tm.tx[i] = (pb.a[i].dwid + tw - 1) / tw;
                        total += pb.tm.tx[i] * ((pb.a[i].dhei + 15) / 16);
                    }
                    pb.tm.start[cnt] = total;
                    dim3 grid(total);
                    if (src == 0 && lds0 && lds_ok) hipLaunchKernelGGL((k_pyr_down_strip_lds<4>), grid, dim3(256), 0, stream(), pb);
                    else if (src == 0) hipLaunchKernelGGL((k_pyr_down_strip<0, 4>), grid, dim3(256), 0, stream(), pb);
                    else if (src == 1) hipLaunchKernelGGL((k_pyr_down_strip<1, 4>), grid, dim3(256), 0, stream(), pb);
                    else if (src == 3 && lds0 && lds_ok) hipLaunchKernelGGL((k_pyr_down_strip_lds_lv<4>), grid, dim3(256), 0, stream(), pb);
                    else if (src == 3) hipLaunchKernelGGL((k_pyr_down_strip<3, 4>), grid, dim3(256), 0, stream(), pb);
                    else hipLaunchKernelGGL((k_pyr_down_strip<2, 4>), grid, dim3(256), 0, stream(), pb);
                } else {
                    // tiles of 128 x 8 outputs
                    pb.tm.cnt = cnt;
                    int total = 0;
                    for (int i = 0; i < cnt; ++i) {
                        pb.tm.start[i] = total; pb.tm.tx[i] = (pb.a[i].dwid + 127) / 128;
                        total += pb.tm.tx[i] * ((pb.a[i].dhei + 7) / 8);
                    }
                    pb.tm.start[cnt] = total;
                    dim3 grid(total);
                    if (apr) {
                        if (src == 0) hipLaunchKernelGGL((k_pyr_down_2x2<0, true>), grid, dim3(256), 0, stream(), pb);
                        else if (src == 1) hipLaunchKernelGGL((k_pyr_down_2x2<1, true>), grid, dim3(256), 0, stream(), pb);
                        else if (src == 2) hipLaunchKernelGGL((k_pyr_down_2x2<2, true>), grid, dim3(256), 0, stream(), pb);
                        else hipLaunchKernelGGL((k_pyr_down_2x2<3, true>), grid, dim3(256), 0, stream(), pb);
                    } else {
                        if (src == 0) hipLaunchKernelGGL((k_pyr_down_2x2<0, false>), grid, dim3(256), 0, stream(), pb);
                        else if (src == 1) hipLaunchKernelGGL((k_pyr_down_2x2<1, false>), grid, dim3(256), 0, stream(), pb);
                        else if (src == 2) hipLaunchKernelGGL((k_pyr_down_2x2<2, false>), grid, dim3(256), 0, stream(), pb);
                        else hipLaunchKernelGGL((k_pyr_down_2x2<3, false>), grid, dim3(256), 0, stream(), pb);
                    }
                }
            }
            if (l + 1 < nb && !strip && !apr) {
                // the next pyrDown reads this level through its BORDER_REFLECT_101 apron
                ApronBatch ab;
                memset(&ab, 0, sizeof ab);
                long long items = 0;
                for (int i = 0; i < cnt; ++i) {
                    const FeedRec &f = *recs[base + i];
                    ab.d[i] = {f.G[l + 1].base, f.G[l + 1].pitch, f.W[l + 1].base, f.W[l + 1].pitch, f.pw[l + 1], f.ph[l + 1], lb};
                    items = std::max(items, (long long)2 * A * (f.pw[l + 1] + 2 * A) + (long long)2 * A * f.ph[l + 1]);
                }
                ProfileScope ps("pyr_apron", 0);
                hipLaunchKernelGGL(k_apron, dim3((unsigned)((items + 255) / 256), 1, cnt), dim3(256), 0, stream(), ab);
            }
        }
    }
    SSP_HIP(hipGetLastError());
    return 0;
}

// a launch takes images of one depth (8-bit frames keep 8-bit levels, int16 frames int16 levels): one chain of launches per depth present
static int build_pyramids(const ssp_blender *b, const std::vector<FeedRec *> &list)
{
    std::vector<FeedRec *> part[2];
    for (FeedRec *f : list) part[f->lvl8 ? 1 : 0].push_back(f);
    for (auto &p : part) SSP_TRY(build_pyramids_same_depth(b, p));
    return 0;
}


int mb_feed_end(ssp_blender *b)
{
    const int n = b->pending;
    if (n == 0) return 0;
    SSP_TRY(mb_feed_border(b));
    b->pending = 0;
    b->obj_pending = false;
    b->border_done = false;
    std::vector<FeedRec *> list;
    for (size_t i = b->feeds.size() - n; i < b->feeds.size(); ++i) list.push_back(&b->feeds[i]);
    return build_pyramids(b, list);
}

// the pending images of TWO blenders of the same kind in one chain of launches (multi-GPU double buffering: the strips received
// for panorama k and the own frames of panorama k+1 -- one pass through the latency-bound small levels instead of two)
int mb_feed_end_pair(ssp_blender *a, ssp_blender *b)
{
    a->obj_pending = false;
    if (b) b->obj_pending = false;
    if (!b || b == a) return mb_feed_end(a);
    SSP_REQUIRE(a->num_bands == b->num_bands && a->float_mode == b->float_mode, "feed_end_pair: the blenders differ in bands or pyramid type");
    std::vector<FeedRec *> list;
    for (ssp_blender *q : {a, b}) {
        const int n = q->pending;
        if (n == 0) continue;
        SSP_TRY(mb_feed_border(q));
        q->pending = 0;
        q->border_done = false;
        for (size_t i = q->feeds.size() - n; i < q->feeds.size(); ++i) list.push_back(&q->feeds[i]);
    }
    return build_pyramids(a, list);
}

// up to MB_MAXB rectangle copies per launch: 16-byte units when every row length allows, 4-byte units when rows are multiples of 4 bytes
// and 4-byte aligned, single bytes otherwise (the few-pixel top levels of all-level strips): the strips of one exchange step
struct RectCopy { const char *s; size_t sp; char *d; size_t dp; int wbytes, h; };
#define RC_MAXB MB_MAXB
struct RectCopyBatch { RectCopy r[RC_MAXB]; TileMap tm; };
template <int UNIT>
__global__ __launch_bounds__(256) void k_rect_copy(const RectCopyBatch batch)
{
    int z, bx, by;
    tile_locate(batch.tm, blockIdx.x, z, bx, by);
    const RectCopy &c = batch.r[z];
    if (UNIT == 16) {
        // 64 chunks x 16 rows per block, four rows per lane: four independent 16-byte loads in flight, then the stores (written once, read by
        // the transport or by a later kernel: non-temporal)
        const int x = (bx * 64 + (threadIdx.x & 63)) * 16, y0 = by * 16 + (threadIdx.x >> 6);
        if (x >= c.wbytes) return;
        typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        nt_u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (y0 + 4 * k < c.h) v[k] = *(const nt_u32x4 *)(c.s + (size_t)(y0 + 4 * k) * c.sp + x);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (y0 + 4 * k < c.h) __builtin_nontemporal_store(v[k], (nt_u32x4 *)(c.d + (size_t)(y0 + 4 * k) * c.dp + x));
        return;
    }
    const int x = (bx * 64 + (threadIdx.x & 63)) * UNIT, y = by * 4 + (threadIdx.x >> 6);
    if (x >= c.wbytes || y >= c.h) return;
    if (UNIT == 4) *(uint32_t *)(c.d + (size_t)y * c.dp + x) = *(const uint32_t *)(c.s + (size_t)y * c.sp + x);
    else c.d[(size_t)y * c.dp + x] = c.s[(size_t)y * c.sp + x];
}
static int rect_copy_unit(const RectCopy &r)
{
    // (16-byte accesses need 4-byte alignment only)
    const uintptr_t a = (uintptr_t)r.s | (uintptr_t)r.d | r.sp | r.dp | (uintptr_t)r.wbytes;
    return a % 4 != 0 ? 1 : r.wbytes % 16 == 0 ? 16 : 4;
}
static void rect_copy_launch(const std::vector<RectCopy> &list)
{
    // a launch copies in the unit all its rectangles allow: keep the few byte-granular ones (top levels of all-level strips) out of the launches
    // of the large 16-byte ones
    std::vector<RectCopy> v(list);
    std::stable_sort(v.begin(), v.end(), [](const RectCopy &a, const RectCopy &b) { return rect_copy_unit(a) > rect_copy_unit(b); });
    for (size_t base = 0; base < v.size(); base += RC_MAXB) {
        const int cnt = (int)std::min<size_t>(RC_MAXB, v.size() - base);
        RectCopyBatch b;
        memset(&b, 0, sizeof b);
        int unit = 16;
        for (int i = 0; i < cnt; ++i) { b.r[i] = v[base + i]; unit = std::min(unit, rect_copy_unit(b.r[i])); }
        const bool wide = unit == 16, words = unit == 4;
        b.tm.cnt = cnt;
        int total = 0;
        for (int i = 0; i < cnt; ++i) {
            b.tm.start[i] = total; b.tm.tx[i] = (b.r[i].wbytes / unit + 63) / 64;
            total += b.tm.tx[i] * (wide ? (b.r[i].h + 15) / 16 : (b.r[i].h + 3) / 4);
        }
        b.tm.start[cnt] = total;
        if (wide) hipLaunchKernelGGL(k_rect_copy<16>, dim3(total), dim3(256), 0, stream(), b);
        else if (words) hipLaunchKernelGGL(k_rect_copy<4>, dim3(total), dim3(256), 0, stream(), b);
        else hipLaunchKernelGGL(k_rect_copy<1>, dim3(total), dim3(256), 0, stream(), b);
    }
}

// ---- strips of other GPUs' frames (multi-GPU) -----------------------------------------------------------------------------------------
// A strip is a sub-rectangle of a fed image's bordered level-0 planes (u8x3 image incl. its BORDER_REFLECT band, u8 mask).  The
// receiver feeds it as an image that fills its rectangle exactly (no band of its own) and rebuilds the Gaussian pyramids from it.
int mb_export_strips(ssp_blender *b, int n, const int *feeds, const int *rects_xywh, void *const *imgs, void *const *masks)
{
    SSP_TRY(mb_flush(b));
    std::vector<RectCopy> v;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        const int feed = feeds[i], x0 = rects_xywh[4 * i], y0 = rects_xywh[4 * i + 1], w = rects_xywh[4 * i + 2], h = rects_xywh[4 * i + 3];
        SSP_REQUIRE(feed >= 0 && feed < (int)b->feeds.size(), "export_strip: no fed image %d", feed);
        const FeedRec &f = b->feeds[feed];
        SSP_REQUIRE(f.g0_depth == (b->float_mode ? SSP_F32 : SSP_U8), "export_strip: 8-bit frames (float frames in float mode) only");
        const int bpp = 3 * depth_size(f.g0_depth);      // 3 (u8) or 12 (f32) bytes per pixel; the mask is 8-bit in both
        const int lx = x0 - f.rx[0], ly = y0 - f.ry[0];
        SSP_REQUIRE(w > 0 && h > 0 && lx >= 0 && ly >= 0 && lx + w <= f.pw[0] && ly + h <= f.ph[0] && lx % 4 == 0 && w % 4 == 0,
                    "export_strip: (%d,%d %dx%d) must lie inside the image's padded rectangle (%d,%d %dx%d) on 4-pixel columns", x0, y0, w, h, f.rx[0], f.ry[0],
                    f.pw[0], f.ph[0]);
        SSP_REQUIRE(imgs[i] && masks[i], "export_strip: null buffer %d", i);
        if (b->strip_planes) {
            // the receiver uses the buffer as the plane itself: rows at the plane pitch, interior after the apron
            const size_t gp = plane_pitch(w, bpp), mp = plane_pitch(w, 1);
            SSP_REQUIRE(((uintptr_t)imgs[i] | (uintptr_t)masks[i]) % 16 == 0, "export_strip: plane-layout buffers must be 16-byte aligned");
            v.push_back({f.G[0].base + (size_t)ly * f.G[0].pitch + (size_t)lx * bpp, f.G[0].pitch, (char *)imgs[i] + STRIP_SLACK + (size_t)APRON * gp + (size_t)APRON * bpp, gp, w * bpp, h});
            v.push_back({f.W[0].base + (size_t)ly * f.W[0].pitch + lx, f.W[0].pitch, (char *)masks[i] + STRIP_SLACK + (size_t)APRON * mp + APRON, mp, w, h});
        } else {
            v.push_back({f.G[0].base + (size_t)ly * f.G[0].pitch + (size_t)lx * bpp, f.G[0].pitch, (char *)imgs[i], (size_t)w * bpp, w * bpp, h});
            v.push_back({f.W[0].base + (size_t)ly * f.W[0].pitch + lx, f.W[0].pitch, (char *)masks[i], (size_t)w, w, h});
        }
        bytes += 2.0 * (bpp + 1) * w * h;
    }
    ProfileScope ps("strip_export", bytes);
    rect_copy_launch(v);
    SSP_HIP(hipGetLastError());
    return 0;
}

int mb_feed_strips(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs, const void *const *masks, bool defer)
{
    SSP_TRY(mb_flush(b));
    if (b->pending) SSP_FAIL(SSP_ERR_STATE, "feed: a previous batch was not finished");
    const int nb = b->num_bands, m = 1 << nb;
    SSP_REQUIRE(m % 4 == 0, "feed_strip: needs at least 2 bands (strip rows are copied in 4-byte units)");
    // strips carry what the blender's own feeds hold at level 0: 8-bit pixels, or float32 pixels for the float pyramids
    const int depth = b->float_mode ? SSP_F32 : SSP_U8, bpp = 3 * depth_size(depth);
    const size_t first = b->feeds.size();
    std::vector<RectCopy> copies;
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        const int x0 = rects_xywh[4 * i], y0 = rects_xywh[4 * i + 1], w = rects_xywh[4 * i + 2], h = rects_xywh[4 * i + 3];
        int rc = 0;
        FeedRec f;
        if (!(w > 0 && h > 0 && x0 >= 0 && y0 >= 0 && x0 % m == 0 && y0 % m == 0 && w % m == 0 && h % m == 0 && x0 + w <= b->lw[0] && y0 + h <= b->lh[0]))
            rc = set_error(SSP_ERR_ARG, "feed_strip: (%d,%d %dx%d) must be inside the padded pano and aligned to %d", x0, y0, w, h, m);
        if (!rc) {
            f.iw = w; f.ih = h; f.left = 0; f.top = 0; f.g0_depth = depth; f.lvl8 = depth == SSP_U8;
            f.pw[0] = w; f.ph[0] = h;
            int x_tl = x0, y_tl = y0;
            for (int l = 0; l <= nb; ++l) {
                if (l > 0) { f.pw[l] = (f.pw[l - 1] + 1) / 2; f.ph[l] = (f.ph[l - 1] + 1) / 2; }
                f.rx[l] = x_tl; f.ry[l] = y_tl;
                x_tl /= 2; y_tl /= 2;
                f.G[l] = Plane(); f.W[l] = Plane();
            }
            if (b->strip_planes) {
                // zero copy: the receive buffers are the level-0 planes (not pool blocks: free_rec leaves them alone); they must
                // stay untouched until this panorama is blended
                if (((uintptr_t)imgs[i] | (uintptr_t)masks[i]) % 16 != 0) rc = set_error(SSP_ERR_ARG, "feed_strip: plane-layout buffers must be 16-byte aligned");
                f.G[0].alloc = nullptr; f.G[0].pitch = plane_pitch(w, bpp); f.G[0].base = (char *)imgs[i] + STRIP_SLACK + (size_t)APRON * f.G[0].pitch + (size_t)APRON * bpp;
                f.W[0].alloc = nullptr; f.W[0].pitch = plane_pitch(w, 1); f.W[0].base = (char *)masks[i] + STRIP_SLACK + (size_t)APRON * f.W[0].pitch + APRON;
            } else {
                rc = alloc_plane(w, h, bpp, 0, f.G[0]);
                if (!rc) rc = alloc_plane(w, h, 1, 0, f.W[0]);
            }
            for (int l = 1; l <= nb && !rc; ++l) {
                rc = alloc_plane(f.pw[l], f.ph[l], f.level_bytes(), 0, f.G[l]);
                if (!rc) rc = alloc_plane(f.pw[l], f.ph[l], 4, 0, f.W[l]);
            }
            if (rc) free_rec(b, f);
        }
        if (rc) {
            while (b->feeds.size() > first) { free_rec(b, b->feeds.back()); b->feeds.pop_back(); }
            return rc;
        }
        if (!b->strip_planes) {
            copies.push_back({(const char *)imgs[i], (size_t)w * bpp, f.G[0].base, f.G[0].pitch, w * bpp, h});
            copies.push_back({(const char *)masks[i], (size_t)w, f.W[0].base, f.W[0].pitch, w, h});
            bytes += 2.0 * (bpp + 1) * w * h;
        }
        b->feeds.push_back(f);
    }
    if (!copies.empty()) {
        ProfileScope ps("strip_import", bytes);
        rect_copy_launch(copies);
    }
    b->pending = n;
    // border_l0 only has the apron to fill here (the image fills its rectangle); then the pyramids (defer: by a later
    // mb_feed_end / mb_feed_end_pair)
    return defer ? mb_feed_border(b) : mb_feed_end(b);
}

// ---- all-level strips ----------------------------------------------------------------------------------------------------------------------
// The level-0 strip protocol above ships a wide rectangle (the receiver's region grown by the reach of a whole pyramid) and lets the receiver
// rebuild the Gaussian pyramids: an interior rank of 8 then builds pyramids over 1.5x its own frames' pixels.  The sender HAS those pyramids.
// An all-level strip is the same sub-rectangle of EVERY level of the fed image's planes (G_l and W_l, l = 0 .. bands, aprons included), cut to
// the receiver's region grown by 2^bands only (one pixel of the top level: what pyrUp reaches), in one buffer; the receiver uses the buffer
// as the planes themselves and launches nothing.  Bit-identical to recomputing: they are the same numbers.
struct LevelStripLayout { size_t g[MAX_BANDS + 1], w[MAX_BANDS + 1], gp[MAX_BANDS + 1], wp[MAX_BANDS + 1], total; };
static LevelStripLayout level_strip_layout(int nb, bool float_mode, int w, int h)
{
    LevelStripLayout L;
    memset(&L, 0, sizeof L);
    const int bpp0 = float_mode ? 12 : 3, lb = float_mode ? 12 : 3;
    size_t off = 0;
    for (int l = 0; l <= nb; ++l) {
        const int wl = w >> l, hl = h >> l;
        // (+32: the exporter copies whole 16-byte chunks of the sender's rows: up to 15 bytes before the strip's first column -- the receiver's
        // plane starts that many bytes into its rows, level_strip_phase -- and up to 15 past its last)
        L.gp[l] = plane_pitch(wl, l == 0 ? bpp0 : lb) + 32;
        L.wp[l] = plane_pitch(wl, l == 0 ? 1 : 4) + 32;
        L.g[l] = off + STRIP_SLACK; off = align_up(L.g[l] + L.gp[l] * (size_t)(hl + 2 * APRON) + STRIP_SLACK, 256);
        L.w[l] = off + STRIP_SLACK; off = align_up(L.w[l] + L.wp[l] * (size_t)(hl + 2 * APRON) + STRIP_SLACK, 256);
    }
    L.total = off;
    return L;
}
size_t mb_level_strip_buffer_bytes(int num_bands, bool float_mode, int w, int h) { return level_strip_layout(num_bands, float_mode, w, h).total; }

// byte offset, within a 16-byte chunk, of column x0 >> l of the image whose padded rectangle starts at pano column ox (a multiple of 2^bands):
// the sender's planes are 16-byte aligned at their first column (256-byte aligned blocks, pitches of 16 k, apron of 4 samples before)
static int level_strip_phase(int x0, int ox, int l, int bytes_per_sample) { return (int)(((size_t)((x0 - ox) >> l) * bytes_per_sample) % 16); }

static int level_strip_check(const ssp_blender *b, const char *what, int x0, int y0, int w, int h)
{
    const int m = 1 << b->num_bands;
    SSP_REQUIRE(w > 0 && h > 0 && x0 >= 0 && y0 >= 0 && x0 % m == 0 && y0 % m == 0 && w % m == 0 && h % m == 0 && x0 + w <= b->lw[0] && y0 + h <= b->lh[0],
                "%s: (%d,%d %dx%d) must be inside the padded pano and aligned to %d", what, x0, y0, w, h, m);
    return 0;
}

int mb_export_level_strips(ssp_blender *b, int n, const int *feeds, const int *rects_xywh, void *const *bufs)
{
    SSP_TRY(mb_flush(b));
    const int nb = b->num_bands, A = APRON;
    SSP_REQUIRE(b->pending == 0, "export_level_strips: the pyramids of the fed images are not built yet (feed_end first)");
    // ordered by level: the many small copies of the deep levels share launches
    std::vector<RectCopy> v;
    double bytes = 0;
    for (int l = 0; l <= nb; ++l)
        for (int i = 0; i < n; ++i) {
            const int feed = feeds[i], x0 = rects_xywh[4 * i], y0 = rects_xywh[4 * i + 1], w = rects_xywh[4 * i + 2], h = rects_xywh[4 * i + 3];
            SSP_REQUIRE(feed >= 0 && feed < (int)b->feeds.size(), "export_level_strips: no fed image %d", feed);
            const FeedRec &f = b->feeds[feed];
            if (l == 0) {
                SSP_TRY(level_strip_check(b, "export_level_strips", x0, y0, w, h));
                SSP_REQUIRE(f.g0_depth == (b->float_mode ? SSP_F32 : SSP_U8), "export_level_strips: 8-bit frames (float frames in float mode) only");
                SSP_REQUIRE(x0 >= f.rx[0] && y0 >= f.ry[0] && x0 + w <= f.rx[0] + f.pw[0] && y0 + h <= f.ry[0] + f.ph[0],
                            "export_level_strips: (%d,%d %dx%d) must lie inside the image's padded rectangle (%d,%d %dx%d)", x0, y0, w, h, f.rx[0], f.ry[0], f.pw[0], f.ph[0]);
                SSP_REQUIRE(bufs[i] && (uintptr_t)bufs[i] % 16 == 0, "export_level_strips: buffer %d must be 16-byte aligned", i);
            }
            const LevelStripLayout L = level_strip_layout(b->num_bands, b->float_mode, w, h);
            const int wl = w >> l, hl = h >> l, lx = (x0 >> l) - f.rx[l], ly = (y0 >> l) - f.ry[l];
            const int gb = l == 0 ? 3 * depth_size(f.g0_depth) : f.level_bytes(), wb = l == 0 ? 1 : 4;
            // rows -A .. hl + A, columns -A .. wl + A of the strip: inside the sender's plane, whose own apron holds the border rule where the strip
            // ends at the image's edge
            // Rows are copied in whole 16-byte chunks from the chunk that holds the strip's first apron column: the bytes before it and past the last
            // column come from the sender's own row and land in the slack of the destination pitch.
            const char *sg = f.G[l].base + (ptrdiff_t)(ly - A) * (ptrdiff_t)f.G[l].pitch + (ptrdiff_t)(lx - A) * gb;
            const char *sw = f.W[l].base + (ptrdiff_t)(ly - A) * (ptrdiff_t)f.W[l].pitch + (ptrdiff_t)(lx - A) * wb;
            const int pg = level_strip_phase(x0, f.rx[0], l, gb), pw = level_strip_phase(x0, f.rx[0], l, wb);
            if (((uintptr_t)(sg - pg) | (uintptr_t)(sw - pw) | f.G[l].pitch | f.W[l].pitch) % 16 != 0)
                SSP_FAIL(SSP_ERR_STATE, "export_level_strips: the planes of fed image %d are not 16-byte aligned at their first column", feed);
            v.push_back({sg - pg, f.G[l].pitch, (char *)bufs[i] + L.g[l], L.gp[l], (int)align_up((size_t)pg + (size_t)(wl + 2 * A) * gb, 16), hl + 2 * A});
            v.push_back({sw - pw, f.W[l].pitch, (char *)bufs[i] + L.w[l], L.wp[l], (int)align_up((size_t)pw + (size_t)(wl + 2 * A) * wb, 16), hl + 2 * A});
            bytes += 2.0 * (gb + wb) * (wl + 2 * A) * (hl + 2 * A);
        }
    ProfileScope ps("strip_export", bytes);
    rect_copy_launch(v);
    SSP_HIP(hipGetLastError());
    return 0;
}

// the received buffers ARE the planes of every level (not pool blocks: free_rec leaves them alone); they must stay untouched until this
// panorama is blended.  Nothing is launched.
int mb_feed_level_strips(ssp_blender *b, int n, const int *rects_xywh, const int *origins_x, const void *const *bufs)
{
    SSP_TRY(mb_flush(b));
    if (b->pending) SSP_FAIL(SSP_ERR_STATE, "feed: a previous batch was not finished");
    const int nb = b->num_bands, A = APRON;
    const int depth = b->float_mode ? SSP_F32 : SSP_U8;
    for (int i = 0; i < n; ++i) {
        const int x0 = rects_xywh[4 * i], y0 = rects_xywh[4 * i + 1], w = rects_xywh[4 * i + 2], h = rects_xywh[4 * i + 3];
        SSP_TRY(level_strip_check(b, "feed_level_strips", x0, y0, w, h));
        SSP_REQUIRE(bufs[i] && (uintptr_t)bufs[i] % 16 == 0, "feed_level_strips: buffer %d must be 16-byte aligned", i);
        SSP_REQUIRE(origins_x[i] >= 0 && origins_x[i] <= x0 && origins_x[i] % (1 << nb) == 0, "feed_level_strips: strip %d starts at column %d, left of its image's padded rectangle (%d)", i, x0, origins_x[i]);
    }
    for (int i = 0; i < n; ++i) {
        const int x0 = rects_xywh[4 * i], y0 = rects_xywh[4 * i + 1], w = rects_xywh[4 * i + 2], h = rects_xywh[4 * i + 3];
        const LevelStripLayout L = level_strip_layout(b->num_bands, b->float_mode, w, h);
        FeedRec f;
        f.iw = w; f.ih = h; f.left = 0; f.top = 0; f.g0_depth = depth; f.lvl8 = depth == SSP_U8;
        for (int l = 0; l <= nb; ++l) {
            f.pw[l] = w >> l; f.ph[l] = h >> l; f.rx[l] = x0 >> l; f.ry[l] = y0 >> l;
            const int gb = l == 0 ? 3 * depth_size(depth) : f.level_bytes(), wb = l == 0 ? 1 : 4;
            f.G[l].alloc = nullptr; f.G[l].pitch = L.gp[l]; f.G[l].base = (char *)bufs[i] + L.g[l] + (size_t)A * L.gp[l] + level_strip_phase(x0, origins_x[i], l, gb) + (size_t)A * gb;
            f.W[l].alloc = nullptr; f.W[l].pitch = L.wp[l]; f.W[l].base = (char *)bufs[i] + L.w[l] + (size_t)A * L.wp[l] + level_strip_phase(x0, origins_x[i], l, wb) + (size_t)A * wb;
        }
        b->feeds.push_back(f);
    }
    return 0;
}

// reorder the fed images: the float weight sums of the level kernels run in list order, which must be the global image order
int mb_order_feeds(ssp_blender *b, const int *keys, int n)
{
    SSP_TRY(mb_flush(b));
    SSP_REQUIRE(n == (int)b->feeds.size() && b->pending == 0, "order_feeds: %d keys for %d fed images", n, (int)b->feeds.size());
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return keys[a] < keys[c]; });
    std::vector<FeedRec> sorted;
    sorted.reserve(n);
    for (int i = 0; i < n; ++i) sorted.push_back(b->feeds[idx[i]]);
    b->feeds.swap(sorted);
    return 0;
}

int mb_feed_images(ssp_blender *b, int n, ssp_image *const *imgs, ssp_image *const *masks, const int *tls)
{
    static const bool eager = getenv("SSP_EAGER_FEED") != nullptr;      // (A/B switch: round-3 behaviour -- int16 levels, one pyramid chain per feed)
    for (int i = 0; i < n; ++i) {
        // an int16 image that is astype(np.int16) of an unmodified 8-bit image (sde.py:1755: every image the reference feeds) IS that image,
        // sample for sample: feed the 8-bit original -- 8-bit Gaussian levels, the packed blend paths (bit-identical: the Gaussian levels of an
        // 8-bit image never leave [0, 255])
        const ssp_image *im = imgs[i];
        if (!eager && !b->float_mode && im->depth == SSP_S16 && im->origin && im->version == im->self_ver && im->origin->version == im->origin_ver && im->origin->depth == SSP_U8 &&
            im->origin->w == im->w && im->origin->h == im->h && im->origin->cn == 3)
            im = im->origin;
        const int size[2] = {im->w, im->h};
        FeedSlot slot;
        SSP_TRY(mb_feed_begin(b, 1, tls + 2 * i, size, im->depth, &slot, true));
        CopyDesc c;
        c.simg = (const char *)im->data; c.sip = im->pitch; c.smask = (const uint8_t *)masks[i]->data; c.smp = masks[i]->pitch;
        c.dimg = (char *)slot.img; c.dip = slot.ipitch; c.dmask = slot.mask; c.dmp = slot.mpitch;
        c.w = im->w; c.h = im->h; c.bpp = 3 * depth_size(im->depth);
        ProfileScope ps("feed_copy", 2.0 * c.w * c.h * (c.bpp + 1));
        hipLaunchKernelGGL(k_copy_interior, dim3((c.w + 255) / 256, c.h), dim3(256), 0, stream(), c);
    }
    SSP_HIP(hipGetLastError());
    return eager ? mb_flush(b) : 0;
}

// Run the per-level gather kernels.
//   region: level-0 rectangle (pano-relative, multiples of 2^nb) to compute, or null for the whole padded pano.
//   export_level >= 0: only write the raw sums of that level's part of the region into exp_lap/exp_w (tightly packed).
//   outputs (result/rmask/mosaic) have their pixel (0,0) at the region origin.
int mb_run_levels(ssp_blender *b, ssp_image *result, ssp_image *rmask, ssp_image *mosaic, int export_level, const int *region, void *exp_lap,
                      float *exp_w)
{
    SSP_TRY(mb_flush(b));
    const int nb = b->num_bands, n = (int)b->feeds.size();
    const int esz = b->float_mode ? 4 : 2;
    int reg[4] = {0, 0, b->lw[0], b->lh[0]};
    if (region) memcpy(reg, region, sizeof reg);
    const int m = 1 << nb;
    SSP_REQUIRE(reg[0] % m == 0 && reg[1] % m == 0 && reg[2] % m == 0 && reg[3] % m == 0 && reg[0] >= 0 && reg[1] >= 0 && reg[2] > 0 && reg[3] > 0 &&
                    reg[0] + reg[2] <= b->lw[0] && reg[1] + reg[3] <= b->lh[0],
                "blend region (%d,%d %dx%d) must be inside the padded pano and aligned to %d", reg[0], reg[1], reg[2], reg[3], m);
    // image descriptors for every level.  Looked up by content first (ssp_blender::desc_cache): with fixed geometry the pool hands the same
    // planes out panorama after panorama and the table is already on the device; otherwise built in pinned staging and uploaded asynchronously
    const size_t cnt = (size_t)std::max(1, n) * (nb + 1);
    std::vector<char> hbuf(sizeof(LevelImg) * cnt);
    LevelImg *h_imgs = (LevelImg *)hbuf.data(), *d_imgs = nullptr;
    memset(h_imgs, 0, hbuf.size());
    for (int l = 0; l <= nb; ++l)
        for (int i = 0; i < n; ++i) {
            const FeedRec &f = b->feeds[i];
            LevelImg &li = h_imgs[(size_t)l * n + i];
            li.g = f.G[l].base; li.gp = f.G[l].pitch;
            li.gn = l < nb ? f.G[l + 1].base : nullptr; li.gnp = l < nb ? f.G[l + 1].pitch : 0;
            li.w = f.W[l].base; li.wp = f.W[l].pitch;
            li.rx = f.rx[l]; li.ry = f.ry[l]; li.pw = f.pw[l]; li.ph = f.ph[l];
            li.pwn = l < nb ? f.pw[l + 1] : 0; li.phn = l < nb ? f.ph[l + 1] : 0;
            li.src_depth = f.g0_depth;
            li.lvl8 = f.lvl8 ? 1 : 0;
        }
    // key of the cache: the descriptor table and the region; behind it (built on a miss only) the tile masks of the levels the 4x2 kernel runs
    const size_t key_bytes = hbuf.size() + sizeof reg;
    hbuf.resize(key_bytes);
    memcpy(hbuf.data() + key_bytes - sizeof reg, reg, sizeof reg);
    const int tgroups = (std::max(1, n) + 31) / 32;
    size_t tm_off[MAX_BANDS + 1] = {0}, tm_words = 0;
    int tm_gx[MAX_BANDS + 1] = {0}, tm_grid[MAX_BANDS + 1] = {0};
    for (int l = 0; l <= nb - 2; ++l) {
        const int cw = reg[2] >> l, ch = reg[3] >> l;
        tm_gx[l] = (cw + 255) / 256;
        const int n_tiles = tm_gx[l] * ((ch + 7) / 8), super_chunk = 8 * 4 * tm_gx[l];
        tm_grid[l] = (n_tiles + super_chunk - 1) / super_chunk * super_chunk;
        tm_off[l] = tm_words;
        tm_words += (size_t)tm_grid[l] * tgroups;
    }
    const size_t tm_base = align_up(key_bytes, 16);
    ssp_blender::DescCache *dc = nullptr;
    for (auto &c : b->desc_cache)
        if (c.dev && c.key_bytes == key_bytes && c.host.size() >= key_bytes && !memcmp(c.host.data(), hbuf.data(), key_bytes)) dc = &c;
    if (!dc) {
        dc = &b->desc_cache[0];
        for (auto &c : b->desc_cache)
            if (c.stamp < dc->stamp) dc = &c;          // least recently used
        if (dc->last_use) SSP_HIP(hipEventSynchronize(dc->last_use));      // its last readers (several blends ago) are done
        else SSP_HIP(hipEventCreateWithFlags(&dc->last_use, hipEventDisableTiming));
        hbuf.resize(tm_base + tm_words * sizeof(uint32_t), 0);
        h_imgs = (LevelImg *)hbuf.data();
        uint32_t *tm = (uint32_t *)(hbuf.data() + tm_base);
        for (int l = 0; l <= nb - 2; ++l) {
            const int cx0 = reg[0] >> l, cy0 = reg[1] >> l, cw = reg[2] >> l, ch = reg[3] >> l, gx = tm_gx[l], gy = (ch + 7) / 8;
            uint32_t *lm = tm + tm_off[l];
            for (int i = 0; i < n; ++i) {
                const LevelImg &li = h_imgs[(size_t)l * n + i];
                // tile (bx, by) covers [cx0 + 256 bx, +256) x [cy0 + 8 by, +8): the tiles the image's rectangle meets
                const int x0 = std::max(li.rx, cx0), x1 = std::min(li.rx + li.pw, cx0 + cw) - 1, y0 = std::max(li.ry, cy0), y1 = std::min(li.ry + li.ph, cy0 + ch) - 1;
                if (x0 > x1 || y0 > y1) continue;
                const int bx_lo = (x0 - cx0) / 256, bx_hi = std::min((x1 - cx0) / 256, gx - 1), by_lo = (y0 - cy0) / 8, by_hi = std::min((y1 - cy0) / 8, gy - 1);
                for (int by = by_lo; by <= by_hi; ++by)
                    for (int bx = bx_lo; bx <= bx_hi; ++bx) lm[((size_t)by * gx + bx) * tgroups + (i >> 5)] |= 1u << (i & 31);
            }
        }
        if (dc->host.size() != hbuf.size()) {
            pool_free(dc->dev); dc->dev = nullptr;
            SSP_TRY(pool_alloc(hbuf.size(), &dc->dev));
        }
        dc->host = hbuf;
        dc->key_bytes = key_bytes;
        // dc->host stays untouched until this entry is replaced, which waits for last_use first: it can serve as the source of the async copy
        SSP_HIP(hipMemcpyAsync(dc->dev, dc->host.data(), hbuf.size(), hipMemcpyHostToDevice, stream()));
    }
    else if (dc->used_on != stream() && dc->last_use) {
        // a hit on a table that another stream uploaded / last read: order this stream behind it (the event sits behind that upload)
        SSP_HIP(hipStreamWaitEvent(stream(), dc->last_use, 0));
    }
    dc->used_on = stream();
    dc->stamp = ++b->desc_stamp;
    d_imgs = (LevelImg *)dc->dev;
    const uint32_t *d_tm = (const uint32_t *)((const char *)dc->dev + tm_base);

    void *coll[MAX_BANDS + 1] = {nullptr};
    size_t cp[MAX_BANDS + 1] = {0};
    int rc = 0;
    bool oct_ok = !b->float_mode;  // the 4x2 kernel covers the integer pyramids with 8-bit or int16 level-0 images
    bool pk_ok = !b->float_mode;   // packed int16 paths: every fed image is 8-bit, so all Gaussian levels stay within [0, 255]
    for (int i = 0; i < n; ++i) {
        oct_ok = oct_ok && b->feeds[i].g0_depth != SSP_F32;
        pk_ok = pk_ok && b->feeds[i].g0_depth == SSP_U8;
    }
    const int l_first = export_level >= 0 ? export_level : nb, l_last = export_level >= 0 ? export_level : 0;
    for (int l = l_first; l >= l_last && !rc; --l) {
        LevelArgs a;
        memset(&a, 0, sizeof a);
        a.imgs = d_imgs + (size_t)l * n;
        a.n_imgs = n;
        a.lw = b->lw[l]; a.lh = b->lh[l];
        a.cx0 = reg[0] >> l; a.cy0 = reg[1] >> l; a.cw = reg[2] >> l; a.ch = reg[3] >> l;
        a.top = l == nb;
        a.eager = (long long)a.cw * a.ch <= 512 * 512;
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
        double in_b = 0;
        for (int i = 0; i < n; ++i) {
            const FeedRec &f = b->feeds[i];
            const double cover = (double)f.pw[l] * f.ph[l];
            in_b += l == 0 ? (double)f.iw * f.ih * (3.0 * depth_size(f.g0_depth) + 1) : cover * (f.level_bytes() + 4);
            if (l < nb) in_b += cover / 4 * f.level_bytes();
        }
        double px = (double)a.cw * a.ch;
        if (l < nb) in_b += px / 4 * 3 * esz;
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
                if (b->float_mode) mb_f32_blend_level(true, grid, block, stream(), &a);
                else hipLaunchKernelGGL((k_blend_level<true, false>), grid, block, 0, stream(), a);
            } else {
                if (b->float_mode) mb_f32_blend_level(false, grid, block, stream(), &a);
                else hipLaunchKernelGGL((k_blend_level<false, false>), grid, block, 0, stream(), a);
            }
        } else if (oct_ok && l <= nb - 2 && !a.export_mode) {
            // 4x2 pixels per lane: rectangles, regions and level sizes are multiples of 4 here
            a.gx = (a.cw + 255) / 256;
            const int n_tiles = a.gx * ((a.ch + 7) / 8), super_chunk = 8 * 4 * a.gx;   // grid padded to whole super-chunks; surplus groups fall outside
            dim3 grid((n_tiles + super_chunk - 1) / super_chunk * super_chunk), block(256);
            a.tmask = d_tm + tm_off[l]; a.tgroups = tgroups;       // (sized for exactly this padded grid above)
            if (pk_ok) {
                if (l == 0) hipLaunchKernelGGL((k_blend_oct<true, true>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_oct<false, true>), grid, block, 0, stream(), a);
            } else {
                if (l == 0) hipLaunchKernelGGL((k_blend_oct<true, false>), grid, block, 0, stream(), a);
                else hipLaunchKernelGGL((k_blend_oct<false, false>), grid, block, 0, stream(), a);
            }
        } else {
            dim3 grid((a.cw + 63) / 64, (a.ch + 15) / 16), block(256);
            if (l == 0) {
                if (b->float_mode) mb_f32_blend_quad(true, grid, block, stream(), &a);
                else hipLaunchKernelGGL((k_blend_quad<true, false>), grid, block, 0, stream(), a);
            } else {
                if (b->float_mode) mb_f32_blend_quad(false, grid, block, stream(), &a);
                else hipLaunchKernelGGL((k_blend_quad<false, false>), grid, block, 0, stream(), a);
            }
        }
    }
    for (int l = 1; l <= nb; ++l) pool_free(coll[l]);
    SSP_HIP(hipEventRecord(dc->last_use, stream()));  // the kernels above are the last readers of this table
    if (rc) return rc;
    SSP_HIP(hipGetLastError());
    return 0;
}


int mb_import_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, const void *lap, const void *wgt)
{
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

}  // namespace ssp
#endif   // SSP_MB_F32_TU
