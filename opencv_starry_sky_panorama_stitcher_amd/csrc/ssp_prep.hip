// ssp_prep.hip -- the per-frame prologue of the compose loop: decimate the full-resolution frame to compose scale and
// stretch its black / white point, in ONE pass over the frame.
//
// Replaces (stitching_detailed_enhanced.py / image_processors.py):
//   sde.py:1699-1707        cv.resize(full_img, None, fx=compose_scale, fy=compose_scale, interpolation=INTER_AREA)
//   sde.py:1711             adjust_black_and_white_point(img, config.black_and_white_point_adjustment["final_panorama"])
//   image_processors.py:32-41   ((np.clip(img, bp, wp) - bp) * (255 / (wp - bp))).astype(np.uint8)
//
// cv::resize(INTER_AREA) for decimation has two arithmetic variants (imgproc/resize.cpp), both kept bit for bit:
//   * non-integer factor: fractional-coverage tables (computeResizeAreaTab), binary32 accumulation -- horizontally per source
//     row in table order, then vertically -- and saturate_cast<uchar> (round half to even);
//   * integer factor (resizeAreaFast_): integer block sums; 2x2 -> (s+2)>>2, else cvRound(sum * (1.f/area)); cells that
//     stick out of the source average what exists.
// The stretch is a 256-entry table built on the host in binary64 exactly as numpy evaluates the expression.
//
// HBM-bound: the frame is read once (3 B per source pixel), the result is 3 B per destination pixel.  A lane owns one
// destination pixel and walks down its source rows reading its ~scale*3 bytes per row with 12-byte loads (4 pixels), so a
// wave reads one contiguous run per row.
#include "ssp_internal.hpp"

#include <cfloat>
#include <cmath>

using namespace ssp;

// ---- tables ------------------------------------------------------------------------------------------------------------
// One thread per destination index: first source index, entry count and `stride` weights (zero padded).
__global__ void k_area_tab(int ssize, int dsize, double scale, int stride, int *si0, int *cnt, float *alpha)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dsize) return;
    double fsx1 = d * scale, fsx2 = fsx1 + scale;
    double cell = fmin(scale, ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    float *a = alpha + (size_t)d * stride;
    int k = 0, first = sx1;
    if (sx1 - fsx1 > 1e-3) {
        first = sx1 - 1;
        a[k++] = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2; ++sx) a[k++] = (float)(1.0 / cell);
    if (fsx2 - sx2 > 1e-3) a[k++] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell);
    si0[d] = first;
    cnt[d] = k;
    for (int j = k; j < stride; ++j) a[j] = 0.f;
}

__device__ inline uint8_t sat_round_u8(float v)
{
    float r = __builtin_rintf(v);
    return (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}

typedef uint32_t u32x3_u __attribute__((ext_vector_type(3), aligned(1)));
typedef uint32_t u32x4_a __attribute__((ext_vector_type(4), aligned(4)));

// ---- fractional factor, 3 channels, <= 4*NCH source pixels per destination pixel per row ---------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void k_resize_area_c3(const uint8_t *__restrict__ src, size_t sp, int sw, uint8_t *__restrict__ dst, size_t dp, int dw,
                                                         int dh, const int *__restrict__ xsi, const float *__restrict__ xalpha, const int *__restrict__ ysi,
                                                         const int *__restrict__ ycnt, const float *__restrict__ yalpha, int ystride,
                                                         const uint8_t *__restrict__ lut)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
    if (dy >= dh || dx >= dw) return;
    float a[4 * NCH];
    const float4 *ap = (const float4 *)(xalpha + (size_t)dx * (4 * NCH));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float4 v = ap[c];
        a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
    }
    const int x0 = xsi[dx];
    // the aligned 16-byte loads may run past the weights' support; they must stay inside the row
    const bool wide = (x0 + 4 * NCH) * 3 + 4 <= sw * 3;
    const int y0 = ysi[dy], ny = ycnt[dy];
    const float *ya = yalpha + (size_t)dy * ystride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < ny; ++j) {
        const uint8_t *row = src + (size_t)(y0 + j) * sp + (size_t)x0 * 3;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        if (wide) {
            // 4-byte aligned 16-byte reads + v_alignbyte: a misaligned 12-byte read costs 2.5x as much (tools/ta_microbench.hip)
            const uint32_t sh = (uint32_t)((uintptr_t)row & 3u);
            const uint8_t *arow = row - sh;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const u32x4_a q = *(const u32x4_a *)(arow + 12 * c);
                u32x3_u v;
                v.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh); v.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh); v.z = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
                // bytes: p0 = v.x[0..2], p1 = v.x[3] v.y[0..1], p2 = v.y[2..3] v.z[0], p3 = v.z[1..3]
                b0 = b0 + (float)(v.x & 255u) * a[4 * c];
                b1 = b1 + (float)((v.x >> 8) & 255u) * a[4 * c];
                b2 = b2 + (float)((v.x >> 16) & 255u) * a[4 * c];
                b0 = b0 + (float)(v.x >> 24) * a[4 * c + 1];
                b1 = b1 + (float)(v.y & 255u) * a[4 * c + 1];
                b2 = b2 + (float)((v.y >> 8) & 255u) * a[4 * c + 1];
                b0 = b0 + (float)((v.y >> 16) & 255u) * a[4 * c + 2];
                b1 = b1 + (float)(v.y >> 24) * a[4 * c + 2];
                b2 = b2 + (float)(v.z & 255u) * a[4 * c + 2];
                b0 = b0 + (float)((v.z >> 8) & 255u) * a[4 * c + 3];
                b1 = b1 + (float)((v.z >> 16) & 255u) * a[4 * c + 3];
                b2 = b2 + (float)(v.z >> 24) * a[4 * c + 3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4 * NCH; ++k) {
                int o = min(x0 + k, sw - 1) * 3 - x0 * 3;  // weight is 0 beyond the support, the clamp only keeps the read legal
                b0 = b0 + (float)row[o] * a[k];
                b1 = b1 + (float)row[o + 1] * a[k];
                b2 = b2 + (float)row[o + 2] * a[k];
            }
        }
        const float beta = ya[j];
        s0 = s0 + beta * b0;
        s1 = s1 + beta * b1;
        s2 = s2 + beta * b2;
    }
    uint8_t r0 = sat_round_u8(s0), r1 = sat_round_u8(s1), r2 = sat_round_u8(s2);
    if (lut) { r0 = lut[r0]; r1 = lut[r1]; r2 = lut[r2]; }
    uint8_t *d = dst + (size_t)dy * dp + (size_t)dx * 3;
    d[0] = r0; d[1] = r1; d[2] = r2;
}

// ---- fractional factor, any channel count / any factor (one thread per destination sample) ---------------------------------
__global__ void k_resize_area_generic(const uint8_t *__restrict__ src, size_t sp, int cn, uint8_t *__restrict__ dst, size_t dp, int dw, int dh,
                                      const int *__restrict__ xsi, const int *__restrict__ xcnt, const float *__restrict__ xalpha, int xstride,
                                      const int *__restrict__ ysi, const int *__restrict__ ycnt, const float *__restrict__ yalpha, int ystride,
                                      const uint8_t *__restrict__ lut)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (i >= dw * cn || dy >= dh) return;
    int dx = i / cn, c = i - dx * cn;
    const int x0 = xsi[dx], nx = xcnt[dx], y0 = ysi[dy], ny = ycnt[dy];
    const float *xa = xalpha + (size_t)dx * xstride, *ya = yalpha + (size_t)dy * ystride;
    float s = 0.f;
    for (int j = 0; j < ny; ++j) {
        const uint8_t *row = src + (size_t)(y0 + j) * sp + (size_t)x0 * cn + c;
        float b = 0.f;
        for (int k = 0; k < nx; ++k) b = b + (float)row[(size_t)k * cn] * xa[k];
        s = s + ya[j] * b;
    }
    uint8_t r = sat_round_u8(s);
    dst[(size_t)dy * dp + i] = lut ? lut[r] : r;
}

// ---- integer factor (resizeAreaFast_) ------------------------------------------------------------------------------------
__global__ void k_resize_area_int(const uint8_t *__restrict__ src, size_t sp, int sw, int sh, int cn, uint8_t *__restrict__ dst, size_t dp, int dw, int dh,
                                  int isx, int isy, const uint8_t *__restrict__ lut)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (i >= dw * cn || dy >= dh) return;
    int dx = i / cn, c = i - dx * cn;
    const int sy0 = dy * isy, sx0 = dx * isx;
    int sum = 0, count = 0;
    for (int yy = 0; yy < isy && sy0 + yy < sh; ++yy) {
        const uint8_t *row = src + (size_t)(sy0 + yy) * sp + c;
        for (int xx = 0; xx < isx && sx0 + xx < sw; ++xx) {
            sum += row[(size_t)(sx0 + xx) * cn];
            ++count;
        }
    }
    const int wfull = sy0 + isy <= sh ? sw / isx : 0;
    uint8_t v;
    if (count == 0) v = 0;
    else if (dx < wfull) v = (isx == 2 && isy == 2) ? (uint8_t)((sum + 2) >> 2) : sat_round_u8((float)sum * (1.f / (float)(isx * isy)));
    else v = sat_round_u8((float)sum / (float)count);
    dst[(size_t)dy * dp + i] = lut ? lut[v] : v;
}

// ---- table look-up (adjust_black_and_white_point on its own) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut_u8(const uint8_t *__restrict__ src, size_t sp, uint8_t *__restrict__ dst, size_t dp, int wbytes, int h,
                                                 const uint8_t *__restrict__ lut)
{
    __shared__ uint8_t t[256];
    t[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (y >= h || x >= wbytes) return;
    const uint8_t *s = src + (size_t)y * sp + x;
    uint8_t *d = dst + (size_t)y * dp + x;
    if (x + 4 <= wbytes) {  // pitches are multiples of 4: aligned dword access
        uint32_t v = *(const uint32_t *)s;
        uint32_t r = (uint32_t)t[v & 255u] | ((uint32_t)t[(v >> 8) & 255u] << 8) | ((uint32_t)t[(v >> 16) & 255u] << 16) | ((uint32_t)t[v >> 24] << 24);
        *(uint32_t *)d = r;
    } else {
        for (int k = 0; x + k < wbytes; ++k) d[k] = t[s[k]];
    }
}

namespace {
struct AreaTab {
    int *si = nullptr, *cnt = nullptr;
    float *alpha = nullptr;
    int stride = 0;
    void *mem = nullptr;
};

int make_tab(int ssize, int dsize, double scale, int stride, AreaTab *t)
{
    size_t bytes = sizeof(int) * 2 * (size_t)dsize + sizeof(float) * (size_t)dsize * stride + 64;
    SSP_TRY(pool_alloc(bytes, &t->mem));
    // weights first: 16-byte aligned rows for the float4 loads
    t->alpha = (float *)t->mem;
    t->si = (int *)(t->alpha + (size_t)dsize * stride);
    t->cnt = t->si + dsize;
    t->stride = stride;
    hipLaunchKernelGGL(k_area_tab, dim3((dsize + 255) / 256), dim3(256), 0, stream(), ssize, dsize, scale, stride, t->si, t->cnt, t->alpha);
    return 0;
}
}  // namespace

// LUT of adjust_black_and_white_point (image_processors.py:32-41)
SSP_API int ssp_bw_point_lut(int black, int white, uint8_t lut[256])
{
    SSP_REQUIRE(lut, "bw_point_lut: null table");
    SSP_REQUIRE(0 <= black && black < white && white <= 255, "bw_point_lut: need 0 <= black < white <= 255");
    const double k = 255.0 / (double)(white - black);  // numpy evaluates the python float 255 / (wp - bp) in binary64
    for (int v = 0; v < 256; ++v) {
        int c = v < black ? black : (v > white ? white : v);
        lut[v] = (uint8_t)((double)(c - black) * k);  // astype(uint8) truncates
    }
    return 0;
}

namespace {
int upload_lut(const uint8_t *lut, uint8_t **dev)
{
    *dev = nullptr;
    if (!lut) return 0;
    SSP_TRY(pool_alloc(256, (void **)dev));
    // 256 bytes by value through a kernel-less path: hipMemcpyAsync from pageable memory stages the bytes before returning
    SSP_HIP(hipMemcpyAsync(*dev, lut, 256, hipMemcpyHostToDevice, stream()));
    return 0;
}
}  // namespace

SSP_API int ssp_apply_lut(const ssp_image *src, const uint8_t lut[256], ssp_image **out)
{
    SSP_REQUIRE(src && lut && out, "apply_lut: null argument");
    SSP_REQUIRE(src->depth == SSP_U8, "apply_lut: needs an 8-bit image (adjust_black_and_white_point returns uint8)");
    ssp_image *d = nullptr;
    SSP_TRY(image_new(src->w, src->h, src->cn, SSP_U8, &d));
    uint8_t *dl = nullptr;
    int rc = upload_lut(lut, &dl);
    if (rc) { image_unref(d); return rc; }
    const int wb = src->w * src->cn;
    {
        ProfileScope ps("bw_point_lut", 2.0 * wb * src->h);
        hipLaunchKernelGGL(k_lut_u8, dim3((wb + 1023) / 1024, src->h), dim3(256), 0, stream(), (const uint8_t *)src->data, src->pitch, (uint8_t *)d->data,
                           d->pitch, wb, src->h, dl);
    }
    pool_free(dl);
    SSP_HIP(hipGetLastError());
    *out = d;
    return 0;
}

SSP_API int ssp_resize_area(const ssp_image *src, double fx, double fy, const uint8_t *lut, ssp_image **out)
{
    SSP_REQUIRE(src && out, "resize(INTER_AREA): null argument");
    SSP_REQUIRE(src->depth == SSP_U8, "resize(INTER_AREA): needs an 8-bit image");
    SSP_REQUIRE(fx > 0 && fy > 0 && fx <= 1.0 && fy <= 1.0, "resize(INTER_AREA): only decimation (0 < fx, fy <= 1; sde.py:1701-1707) is implemented");
    const int dw = (int)std::nearbyint(src->w * fx), dh = (int)std::nearbyint(src->h * fy);  // saturate_cast<int>(double) = cvRound
    SSP_REQUIRE(dw > 0 && dh > 0, "resize(INTER_AREA): empty destination");
    const double sx = 1.0 / fx, sy = 1.0 / fy;
    const int isx = (int)std::nearbyint(sx), isy = (int)std::nearbyint(sy);
    ssp_image *d = nullptr;
    SSP_TRY(image_new(dw, dh, src->cn, SSP_U8, &d));
    uint8_t *dl = nullptr;
    int rc = upload_lut(lut, &dl);
    if (rc) { image_unref(d); return rc; }
    const double algo = (double)src->w * src->h * src->cn + (double)dw * dh * src->cn;
    if (std::fabs(sx - isx) < DBL_EPSILON && std::fabs(sy - isy) < DBL_EPSILON) {
        ProfileScope ps("resize_area", algo);
        hipLaunchKernelGGL(k_resize_area_int, dim3((dw * src->cn + 255) / 256, dh), dim3(256), 0, stream(), (const uint8_t *)src->data, src->pitch, src->w,
                           src->h, src->cn, (uint8_t *)d->data, d->pitch, dw, dh, isx, isy, dl);
    } else {
        const int nmax = (int)std::floor(sx) + 2;  // entries per destination column
        const int nch = (nmax + 3) / 4;
        const bool fast = src->cn == 3 && nch <= 4;
        AreaTab xt, yt;
        rc = make_tab(src->w, dw, sx, fast ? 4 * nch : nmax, &xt);
        if (!rc) rc = make_tab(src->h, dh, sy, (int)std::floor(sy) + 2, &yt);
        if (rc) { pool_free(xt.mem); pool_free(dl); image_unref(d); return rc; }
        {
            ProfileScope ps("resize_area", algo);
            if (fast) {
                dim3 grid((dw + 63) / 64, (dh + 3) / 4), block(256);
#define SSP_AREA_LAUNCH(N)                                                                                                                              \
    hipLaunchKernelGGL(k_resize_area_c3<N>, grid, block, 0, stream(), (const uint8_t *)src->data, src->pitch, src->w, (uint8_t *)d->data, d->pitch, dw, \
                       dh, xt.si, xt.alpha, yt.si, yt.cnt, yt.alpha, yt.stride, dl)
                switch (nch) {
                case 1: SSP_AREA_LAUNCH(1); break;
                case 2: SSP_AREA_LAUNCH(2); break;
                case 3: SSP_AREA_LAUNCH(3); break;
                default: SSP_AREA_LAUNCH(4); break;
                }
#undef SSP_AREA_LAUNCH
            } else {
                hipLaunchKernelGGL(k_resize_area_generic, dim3((dw * src->cn + 255) / 256, dh), dim3(256), 0, stream(), (const uint8_t *)src->data,
                                   src->pitch, src->cn, (uint8_t *)d->data, d->pitch, dw, dh, xt.si, xt.cnt, xt.alpha, xt.stride, yt.si, yt.cnt, yt.alpha,
                                   yt.stride, dl);
            }
        }
        pool_free(xt.mem);
        pool_free(yt.mem);
    }
    pool_free(dl);
    SSP_HIP(hipGetLastError());
    *out = d;
    return 0;
}
