// ssp_util.hip -- the small image operators between warp and feed (mask preparation, type conversion).
//
// Replaces (stitching_detailed_enhanced.py):
//   :1755       image_warped.astype(np.int16)                   -> ssp_image_convert
//   :1760-1764  cv.dilate(mask, None)                            -> ssp_dilate3x3
//   :1767-1768  cv.resize(mask, size, INTER_LINEAR_EXACT)        -> ssp_resize_linear_exact
//   :1772       cv.bitwise_and(seam_mask, mask_warped)           -> ssp_bitwise_and
// All are streaming byte kernels (HBM-bound, a few bytes per pixel).
#include "ssp_internal.hpp"

using namespace ssp;

// ---- fill / convert ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_fill(T *p, size_t pitch, int wcn, int h, T v)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < wcn && y < h) ((T *)((char *)p + (size_t)y * pitch))[x] = v;
}

template <typename S, typename D>
__device__ inline D convert_one(S v);
template <> __device__ inline int16_t convert_one<uint8_t, int16_t>(uint8_t v) { return (int16_t)v; }
template <> __device__ inline float convert_one<uint8_t, float>(uint8_t v) { return (float)v; }
template <> __device__ inline uint8_t convert_one<int16_t, uint8_t>(int16_t v) { return (uint8_t)min(max((int)v, 0), 255); }
template <> __device__ inline float convert_one<int16_t, float>(int16_t v) { return (float)v; }
template <> __device__ inline uint8_t convert_one<float, uint8_t>(float v)
{
    float r = __builtin_rintf(v);  // saturate_cast<uchar>(float) = cvRound then clamp
    return (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}
template <> __device__ inline int16_t convert_one<float, int16_t>(float v)
{
    float r = __builtin_rintf(v);
    return (int16_t)(r < -32768.f ? -32768 : (r > 32767.f ? 32767 : (int)r));
}

template <typename S, typename D>
__global__ void k_convert(const S *s, size_t sp, D *d, size_t dp, int wcn, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < wcn && y < h) ((D *)((char *)d + (size_t)y * dp))[x] = convert_one<S, D>(((const S *)((const char *)s + (size_t)y * sp))[x]);
}

SSP_API int ssp_image_fill(ssp_image *im, double value)
{
    SSP_REQUIRE(im, "fill: null image");
    int wcn = im->w * im->cn;
    dim3 grid((wcn + 255) / 256, im->h), block(256);
    if (im->depth == SSP_U8) hipLaunchKernelGGL(k_fill<uint8_t>, grid, block, 0, stream(), (uint8_t *)im->data, im->pitch, wcn, im->h, (uint8_t)value);
    else if (im->depth == SSP_S16) hipLaunchKernelGGL(k_fill<int16_t>, grid, block, 0, stream(), (int16_t *)im->data, im->pitch, wcn, im->h, (int16_t)value);
    else hipLaunchKernelGGL(k_fill<float>, grid, block, 0, stream(), (float *)im->data, im->pitch, wcn, im->h, (float)value);
    SSP_HIP(hipGetLastError());
    ++im->version;
    return 0;
}

// is every sample of an 8-bit image equal to `value`?  (the all-255 mask of sde.py:1739: the deferred object API checks it once per array
// before it lets the fused warp's validity mask stand in for warp(mask, INTER_NEAREST, BORDER_CONSTANT))
__global__ void k_all_equal_u8(const uint8_t *s, size_t sp, int wcn, int h, uint32_t v4, int *differs)
{
    const int y = blockIdx.y;
    const uint8_t *r = s + (size_t)y * sp;
    bool diff = false;
    for (int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x); x < wcn; x += 4 * gridDim.x * blockDim.x) {
        if (x + 4 <= wcn) diff = diff || *(const uint32_t *)(r + x) != v4;        // rows are 4-byte aligned (16-byte pitch; wrapped images: checked by the caller)
        else for (int q = x; q < wcn; ++q) diff = diff || r[q] != (uint8_t)v4;
    }
    if (diff) *differs = 1;
}
SSP_API int ssp_image_all_equal(const ssp_image *img, int value, int *flag)
{
    SSP_REQUIRE(img && flag && img->depth == SSP_U8 && value >= 0 && value <= 255, "all_equal: an 8-bit image and a value in 0..255");
    SSP_REQUIRE(img->pitch % 4 == 0 && (uintptr_t)img->data % 4 == 0, "all_equal: rows must be 4-byte aligned");
    int *d = nullptr, h_flag = 0;
    SSP_TRY(pool_alloc(sizeof(int), (void **)&d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(int), stream());
    const int wcn = img->w * img->cn;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_all_equal_u8, dim3(std::max(1, std::min(64, (wcn + 1023) / 1024)), img->h), dim3(256), 0, stream(), (const uint8_t *)img->data, img->pitch, wcn, img->h,
                           0x01010101u * (uint32_t)value, d);
        e = hipMemcpyAsync(&h_flag, d, sizeof(int), hipMemcpyDeviceToHost, stream());
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream());
    pool_free(d);
    if (e != hipSuccess) SSP_FAIL(SSP_ERR_DEVICE, "all_equal failed: %s", hipGetErrorString(e));
    *flag = h_flag ? 0 : 1;
    return 0;
}

SSP_API int ssp_image_convert(const ssp_image *src, int depth, ssp_image **out)
{
    SSP_REQUIRE(src && out, "convert: null argument");
    ssp_image *d = nullptr;
    SSP_TRY(image_new(src->w, src->h, src->cn, depth, &d));
    int wcn = src->w * src->cn;
    dim3 grid((wcn + 255) / 256, src->h), block(256);
#define CV(S, D) hipLaunchKernelGGL((k_convert<S, D>), grid, block, 0, stream(), (const S *)src->data, src->pitch, (D *)d->data, d->pitch, wcn, src->h)
    if (src->depth == depth) {
        hipError_t e = hipMemcpy2DAsync(d->data, d->pitch, src->data, src->pitch, (size_t)wcn * depth_size(depth), src->h, hipMemcpyDeviceToDevice, stream());
        if (e != hipSuccess) { image_unref(d); SSP_FAIL(SSP_ERR_DEVICE, "convert copy failed: %s", hipGetErrorString(e)); }
    } else if (src->depth == SSP_U8 && depth == SSP_S16) CV(uint8_t, int16_t);
    else if (src->depth == SSP_U8 && depth == SSP_F32) CV(uint8_t, float);
    else if (src->depth == SSP_S16 && depth == SSP_U8) CV(int16_t, uint8_t);
    else if (src->depth == SSP_S16 && depth == SSP_F32) CV(int16_t, float);
    else if (src->depth == SSP_F32 && depth == SSP_U8) CV(float, uint8_t);
    else if (src->depth == SSP_F32 && depth == SSP_S16) CV(float, int16_t);
    else { image_unref(d); SSP_FAIL(SSP_ERR_ARG, "convert: unsupported depth pair %d -> %d", src->depth, depth); }
#undef CV
    SSP_HIP(hipGetLastError());
    if (src->depth == SSP_U8 && depth == SSP_S16 && src->owned) {      // astype(np.int16) of an 8-bit image (sde.py:1755): see ssp_image::origin
        ssp_image *o = const_cast<ssp_image *>(src);
        ++o->refs;
        d->origin = o; d->origin_ver = o->version; d->self_ver = d->version;
    }
    *out = d;
    return 0;
}

// ---- dilate 3x3 ---------------------------------------------------------------------------------------------------
__global__ void k_dilate3(const uint8_t *s, size_t sp, uint8_t *d, size_t dp, int w, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    int m = 0;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = y + dy;
        if (yy < 0 || yy >= h) continue;  // pixels outside the image are ignored (morphologyDefaultBorderValue)
        const uint8_t *r = s + (size_t)yy * sp;
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = x + dx;
            if (xx < 0 || xx >= w) continue;
            m = max(m, (int)r[xx]);
        }
    }
    d[(size_t)y * dp + x] = (uint8_t)m;
}

SSP_API int ssp_dilate3x3(const ssp_image *mask, ssp_image **out)
{
    SSP_REQUIRE(mask && out && mask->depth == SSP_U8 && mask->cn == 1, "dilate: needs an 8UC1 mask");
    ssp_image *d = nullptr;
    SSP_TRY(image_new(mask->w, mask->h, 1, SSP_U8, &d));
    ProfileScope ps("mask_dilate", 2.0 * mask->w * mask->h);
    hipLaunchKernelGGL(k_dilate3, dim3((mask->w + 255) / 256, mask->h), dim3(256), 0, stream(), (const uint8_t *)mask->data, mask->pitch, (uint8_t *)d->data,
                       d->pitch, mask->w, mask->h);
    SSP_HIP(hipGetLastError());
    *out = d;
    return 0;
}

// ---- resize INTER_LINEAR_EXACT, 8UC1 -----------------------------------------------------------------------------
// Per-axis tables (offset, coefficient in 8.8 fixed point; -1 = copy the edge sample), then
// dst = (h0*(256-cy) + h1*cy + 2^15) >> 16 with h = p[o]*(256-cx) + p[o+1]*cx  (resize.cpp, ufixedpoint16/32).
__global__ void k_lin_exact_tab(int ssize, int dsize, int *ofs, int *coef)
{
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dsize) return;
    double scale = 1.0 / ((double)dsize / (double)ssize);
    double fval = scale * ((double)d + 0.5) - 0.5;
    int ival = (int)floor(fval);
    if (ival >= 0 && ssize > 1) {
        if (ival < ssize - 1) {
            ofs[d] = ival;
            coef[d] = (int)rint((fval - (double)ival) * 256.0);
        } else {
            ofs[d] = ssize - 1;
            coef[d] = -1;
        }
    } else {
        ofs[d] = 0;
        coef[d] = -1;
    }
}

// 4 destination pixels per lane: the x tables come as two 16-byte loads, every pixel's tap pair as one 2-byte read per row, the
// AND operand and the result as 4-byte accesses (image rows are 16-byte aligned)
typedef uint16_t u16_r1 __attribute__((aligned(1)));
typedef uint32_t u32_r1 __attribute__((aligned(1)));
// RPL rows per lane (rows y, y + 4, y + 8, ...: a wave's rows stay uniform): the column tables -- 32 bytes per lane, more than the 4 + 4 + 4
// bytes of a row's payload -- are loaded once for all of them (config 5 prepares a 33 MPix mask per frame: 61.6 -> 4 rows per lane)
template <int RPL>
__global__ __launch_bounds__(256) void k_resize_lin_exact(const uint8_t *s, size_t sp, int sw, uint8_t *d, size_t dp, int dw, int dh, const int *xo, const int *xc,
                                                          const int *yo, const int *yc, const uint8_t *and_with, size_t ap)
{
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, ybase = blockIdx.y * 4 * RPL + (threadIdx.x >> 6);
    if (x0 >= dw || ybase >= dh) return;
    const int4 o4 = *(const int4 *)(xo + x0), c4 = *(const int4 *)(xc + x0);   // tables are padded to a multiple of 4 entries
    const int o[4] = {o4.x, o4.y, o4.z, o4.w}, cx[4] = {c4.x, c4.y, c4.z, c4.w};
    // Upscaling by 3 or more (the seam-scale mask to compose scale is ~18x): the four pixels of a lane read source columns within
    // o[0] .. o[0] + 2, i.e. one 4-byte window per source row instead of eight 2-byte gathers.  Wave-uniform choice.
    const int ws = min(o[0], sw - 4);
    const bool window = sw >= 4 && o[3] + 1 - ws <= 3 && o[1] >= ws && o[2] >= ws && o[3] >= ws && x0 + 4 <= dw;
    const bool all_window = __all(window);
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
    const int y = ybase + 4 * rr;
    if (y >= dh) break;
    const uint8_t *r0 = s + (size_t)yo[y] * sp;
    const int cyv = yc[y];
    const uint8_t *r1 = cyv >= 0 ? r0 + sp : r0;
    const uint32_t cy1 = cyv >= 0 ? (uint32_t)cyv : 0, cy0 = 256 - cy1;
    uint32_t out = 0;
    if (all_window) {
        const uint32_t w0 = *(const u32_r1 *)(r0 + ws), w1 = *(const u32_r1 *)(r1 + ws);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool pair = cx[k] >= 0;
            const int sh = (o[k] - ws) * 8;
            const uint32_t p0 = w0 >> sh, p1 = w1 >> sh;           // low byte: the sample, next byte: its right neighbour (unused when !pair)
            const uint32_t cx1 = pair ? (uint32_t)cx[k] : 0u, cx0 = 256 - cx1;
            const uint32_t h0 = (p0 & 0xffu) * cx0 + ((p0 >> 8) & 0xffu) * cx1, h1 = (p1 & 0xffu) * cx0 + ((p1 >> 8) & 0xffu) * cx1;
            out |= ((h0 * cy0 + h1 * cy1 + (1u << 15)) >> 16) << (8 * k);
        }
        if (and_with) out &= *(const u32_r1 *)(and_with + (size_t)y * ap + x0);
        *(u32_r1 *)(d + (size_t)y * dp + x0) = out;
        continue;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool pair = cx[k] >= 0;           // -1: copy the edge sample (the pair read would leave the row)
        const uint32_t p0 = pair ? *(const u16_r1 *)(r0 + o[k]) : r0[o[k]], p1 = pair ? *(const u16_r1 *)(r1 + o[k]) : r1[o[k]];
        const uint32_t cx1 = pair ? (uint32_t)cx[k] : 0u, cx0 = 256 - cx1;
        const uint32_t h0 = (p0 & 0xffu) * cx0 + (p0 >> 8) * cx1, h1 = (p1 & 0xffu) * cx0 + (p1 >> 8) * cx1;
        out |= ((h0 * cy0 + h1 * cy1 + (1u << 15)) >> 16) << (8 * k);
    }
    if (x0 + 4 <= dw) {
        // (byte-aligned types: the in-place form works on a view at any offset inside a blender plane)
        if (and_with) out &= *(const u32_r1 *)(and_with + (size_t)y * ap + x0);
        *(u32_r1 *)(d + (size_t)y * dp + x0) = out;
    } else {
        for (int k = 0; x0 + k < dw; ++k) {
            uint32_t v = (out >> (8 * k)) & 0xffu;
            if (and_with) v &= and_with[(size_t)y * ap + x0 + k];
            d[(size_t)y * dp + x0 + k] = (uint8_t)v;
        }
    }
    }   // rows of the lane
}

namespace ssp {
// resize (+ optional fused bitwise_and with a mask of the destination size).  out == nullptr: in place, and_with &= resize(src)
// (and_with may be a view into a larger plane).
int resize_linear_exact(const ssp_image *src, int dw, int dh, const ssp_image *and_with, ssp_image **out)
{
    SSP_REQUIRE(src && src->depth == SSP_U8 && src->cn == 1, "resize(INTER_LINEAR_EXACT): needs an 8UC1 image");
    SSP_REQUIRE(out || and_with, "resize(INTER_LINEAR_EXACT): no destination");
    SSP_REQUIRE(dw > 0 && dh > 0, "resize: empty destination size");
    SSP_REQUIRE(!and_with || (and_with->w == dw && and_with->h == dh && and_with->depth == SSP_U8 && and_with->cn == 1), "resize+and: mask size mismatch");
    ssp_image *d = nullptr;
    if (out) SSP_TRY(image_new(dw, dh, 1, SSP_U8, &d));
    uint8_t *dptr = out ? (uint8_t *)d->data : (uint8_t *)and_with->data;
    const size_t dpitch = out ? d->pitch : and_with->pitch;
    int *tab = nullptr;
    const size_t dw4 = align_up((size_t)dw, 4);   // the x tables are read four entries at a time
    int rc = pool_alloc(sizeof(int) * 2 * (dw4 + dh), (void **)&tab);
    if (rc) { image_unref(d); return rc; }
    int *xo = tab, *xc = tab + dw4, *yo = tab + 2 * dw4, *yc = yo + dh;
    if (dw4 != (size_t)dw) (void)hipMemsetAsync(tab, 0, sizeof(int) * 2 * dw4, stream());   // defined values in the padding (offset 0, coefficient 0)
    hipLaunchKernelGGL(k_lin_exact_tab, dim3((dw + 255) / 256), dim3(256), 0, stream(), src->w, dw, xo, xc);
    hipLaunchKernelGGL(k_lin_exact_tab, dim3((dh + 255) / 256), dim3(256), 0, stream(), src->h, dh, yo, yc);
    {
        ProfileScope ps("mask_resize_and", (and_with ? 2.0 : 1.0) * dw * dh + (double)src->w * src->h);
        // large destinations: four rows per lane (the column tables are loaded once per lane); small ones keep the finer grid
        if ((long long)dw * dh >= (1 << 20))
            hipLaunchKernelGGL(k_resize_lin_exact<4>, dim3((dw + 255) / 256, (dh + 15) / 16), dim3(256), 0, stream(), (const uint8_t *)src->data, src->pitch, src->w,
                               dptr, dpitch, dw, dh, xo, xc, yo, yc, and_with ? (const uint8_t *)and_with->data : nullptr, and_with ? and_with->pitch : 0);
        else
            hipLaunchKernelGGL(k_resize_lin_exact<1>, dim3((dw + 255) / 256, (dh + 3) / 4), dim3(256), 0, stream(), (const uint8_t *)src->data, src->pitch, src->w,
                               dptr, dpitch, dw, dh, xo, xc, yo, yc, and_with ? (const uint8_t *)and_with->data : nullptr, and_with ? and_with->pitch : 0);
    }
    pool_free(tab);
    SSP_HIP(hipGetLastError());
    if (out) *out = d;
    return 0;
}
}  // namespace ssp

SSP_API int ssp_resize_linear_exact(const ssp_image *mask, int dw, int dh, ssp_image **out) { return resize_linear_exact(mask, dw, dh, nullptr, out); }

// ---- bitwise and ------------------------------------------------------------------------------------------------------
__global__ void k_and(const uint8_t *a, size_t ap, const uint8_t *b, size_t bp, uint8_t *d, size_t dp, int wbytes, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < wbytes && y < h) d[(size_t)y * dp + x] = a[(size_t)y * ap + x] & b[(size_t)y * bp + x];
}

SSP_API int ssp_bitwise_and(const ssp_image *a, const ssp_image *b, ssp_image **out)
{
    SSP_REQUIRE(a && b && out, "bitwise_and: null argument");
    SSP_REQUIRE(a->w == b->w && a->h == b->h && a->cn == b->cn && a->depth == b->depth, "bitwise_and: operands differ in size or type");
    ssp_image *d = nullptr;
    SSP_TRY(image_new(a->w, a->h, a->cn, a->depth, &d));
    int wb = a->w * a->cn * depth_size(a->depth);
    ProfileScope ps("mask_and", 3.0 * wb * a->h);
    hipLaunchKernelGGL(k_and, dim3((wb + 255) / 256, a->h), dim3(256), 0, stream(), (const uint8_t *)a->data, a->pitch, (const uint8_t *)b->data, b->pitch,
                       (uint8_t *)d->data, d->pitch, wb, a->h);
    SSP_HIP(hipGetLastError());
    *out = d;
    return 0;
}

// ---- PMC calibration: stream a buffer with a known byte count at 4 / 8 / 16 bytes per lane --------------------------------
// FETCH_SIZE / WRITE_SIZE are only calibrated for some access widths on gfx950 (MI355X_MICROARCH.md, HBM section); bench.py
// runs these kernels under rocprofv3 --pmc to measure the factor that applies to 8- and 16-byte-per-lane accesses.
template <typename V>
__global__ __launch_bounds__(256) void k_calib_read(const V *src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v = src[i];
        const uint32_t *w = (const uint32_t *)&v;
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive; practically never true
}
template <typename V>
__global__ __launch_bounds__(256) void k_calib_write(V *dst, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V v;
        uint32_t *w = (uint32_t *)&v;
        for (unsigned k = 0; k < sizeof(V) / 4; ++k) w[k] = seed + (uint32_t)i;
        dst[i] = v;
    }
}

SSP_API int ssp_calibrate_stream(int bytes_per_lane, size_t total_bytes, int reps)
{
    SSP_TRY(ensure_init());
    SSP_REQUIRE((bytes_per_lane == 4 || bytes_per_lane == 8 || bytes_per_lane == 16) && total_bytes >= (1u << 20) && reps > 0, "calibrate: bad arguments");
    void *buf = nullptr, *buf2 = nullptr;
    uint32_t *sink = nullptr;
    SSP_TRY(pool_alloc(total_bytes, &buf));
    SSP_TRY(pool_alloc(total_bytes, &buf2));
    SSP_TRY(pool_alloc(256, (void **)&sink));
    SSP_HIP(hipMemsetAsync(buf, 1, total_bytes, stream()));
    const size_t n = total_bytes / bytes_per_lane;
    dim3 grid(256 * 8), block(256);
    for (int r = 0; r < reps; ++r) {
        if (bytes_per_lane == 4) {
            hipLaunchKernelGGL(k_calib_read<uint32_t>, grid, block, 0, stream(), (const uint32_t *)buf, n, sink);
            hipLaunchKernelGGL(k_calib_write<uint32_t>, grid, block, 0, stream(), (uint32_t *)buf2, n, (uint32_t)r);
        } else if (bytes_per_lane == 8) {
            hipLaunchKernelGGL(k_calib_read<uint2>, grid, block, 0, stream(), (const uint2 *)buf, n, sink);
            hipLaunchKernelGGL(k_calib_write<uint2>, grid, block, 0, stream(), (uint2 *)buf2, n, (uint32_t)r);
        } else {
            hipLaunchKernelGGL(k_calib_read<uint4>, grid, block, 0, stream(), (const uint4 *)buf, n, sink);
            hipLaunchKernelGGL(k_calib_write<uint4>, grid, block, 0, stream(), (uint4 *)buf2, n, (uint32_t)r);
        }
    }
    SSP_HIP(hipStreamSynchronize(stream()));
    pool_free(buf); pool_free(buf2); pool_free(sink);
    return 0;
}
