// ssp_blend.hip -- cv.detail.Blender (NO) / FeatherBlender kernels and the C ABI of all three blenders.
//
// Replaces (stitching_detailed_enhanced.py):
//   :1806-1819  Blender_createDefault(NO) / detail_MultiBandBlender().setNumBands / detail_FeatherBlender().setSharpness
//   :1820       blender.prepare(resultRoi)
//   :1886/:1889 blender.feed(image_warped_s, mask_warped, corner)
//   :1930       blender.blend(None, None) -> (result int16, result_mask)
// The multiband implementation lives in ssp_multiband.hip.
#include "ssp_blender.hpp"

using namespace ssp;

#define WEIGHT_EPS 1e-5f
#define MAX_BANDS SSP_MAX_BANDS

// static_cast<short>(float) as on x86-64: cvttss2si, then the low 16 bits
__device__ inline int trunc16(float f)
{
    int t = (f > -2147483648.0f && f < 2147483648.0f) ? (int)f : INT32_MIN;
    return (int)(int16_t)(uint16_t)(t & 0xffff);
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


namespace ssp {
static void release_state(ssp_blender *b)
{
    mb_release(b);
    image_unref(b->dst); image_unref(b->dst_mask); image_unref(b->dst_weight);
    b->dst = b->dst_mask = b->dst_weight = nullptr;
    b->prepared = false;
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
        for (auto &c : b->desc_cache) {
            if (c.last_use) { (void)hipEventSynchronize(c.last_use); (void)hipEventDestroy(c.last_use); }
            ssp::pool_free(c.dev);
        }
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
        {
            const int tl[2] = {tlx, tly};
            return mb_feed_images(b, 1, &img, &mask, tl);
        }
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
    return mb_feed_images(b, n, imgs, masks, tls_xy);
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
            rc = mb_run_levels(b, res, rm, mo, -1, nullptr, nullptr, nullptr);
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
    return mb_run_levels(b, nullptr, nullptr, nullptr, level, rect, lap, (float *)wgt);
}

SSP_API int ssp_blender_import_partial(ssp_blender *b, int level, int x0, int y0, int w, int h, const void *lap, const void *wgt)
{
    SSP_REQUIRE(b && lap && wgt, "import_partial: null argument");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "import_partial needs a prepared multiband blender");
    return mb_import_partial(b, level, x0, y0, w, h, lap, wgt);
}

// ---- multi-GPU strip exchange (parallel.py: plan_strips) ---------------------------------------------------------------------------
SSP_API int ssp_blender_export_strips(ssp_blender *b, int n, const int *feed_indices, const int *rects_xywh, void *const *imgs_u8c3, void *const *masks_u8)
{
    SSP_REQUIRE(b && n > 0 && feed_indices && rects_xywh && imgs_u8c3 && masks_u8, "export_strips: bad arguments");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "export_strips needs a prepared multiband blender");
    return mb_export_strips(b, n, feed_indices, rects_xywh, imgs_u8c3, masks_u8);
}

SSP_API int ssp_blender_feed_strips(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs_u8c3, const void *const *masks_u8)
{
    SSP_REQUIRE(b && n > 0 && rects_xywh && imgs_u8c3 && masks_u8, "feed_strips: bad arguments");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "feed_strips needs a prepared multiband blender");
    return mb_feed_strips(b, n, rects_xywh, imgs_u8c3, masks_u8);
}

SSP_API int ssp_blender_export_level_strips(ssp_blender *b, int n, const int *feed_indices, const int *rects_xywh, void *const *bufs, int num_bands)
{
    SSP_REQUIRE(b && n > 0 && feed_indices && rects_xywh && bufs, "export_level_strips: bad arguments");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "export_level_strips needs a prepared multiband blender");
    SSP_REQUIRE(num_bands == b->num_bands, "export_level_strips: the buffers were sized for %d bands, the prepared blender has %d", num_bands, b->num_bands);
    return mb_export_level_strips(b, n, feed_indices, rects_xywh, bufs);
}

SSP_API int ssp_blender_feed_level_strips(ssp_blender *b, int n, const int *rects_xywh, const int *origins_x, const void *const *bufs, int num_bands)
{
    SSP_REQUIRE(b && n > 0 && rects_xywh && origins_x && bufs, "feed_level_strips: bad arguments");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "feed_level_strips needs a prepared multiband blender");
    SSP_REQUIRE(num_bands == b->num_bands, "feed_level_strips: the buffers were sized for %d bands, the prepared blender has %d", num_bands, b->num_bands);
    return mb_feed_level_strips(b, n, rects_xywh, origins_x, bufs);
}

SSP_API int ssp_level_strip_buffer_bytes(int w, int h, int num_bands, int float_pyramids, size_t *bytes)
{
    SSP_REQUIRE(bytes && w > 0 && h > 0 && num_bands >= 0 && num_bands <= SSP_MAX_BANDS, "level_strip_buffer_bytes: bad arguments");
    SSP_REQUIRE(w % (1 << num_bands) == 0 && h % (1 << num_bands) == 0, "level_strip_buffer_bytes: %dx%d is not a multiple of %d", w, h, 1 << num_bands);
    *bytes = mb_level_strip_buffer_bytes(num_bands, float_pyramids != 0, w, h);
    return 0;
}

SSP_API int ssp_blender_set_strip_layout(ssp_blender *b, int planes)
{
    SSP_REQUIRE(b, "set_strip_layout: null blender");
    b->strip_planes = planes != 0;
    return 0;
}

SSP_API int ssp_strip_buffer_bytes(int w, int h, int bytes_per_px, int planes, size_t *bytes)
{
    SSP_REQUIRE(bytes && w > 0 && h > 0 && (bytes_per_px == 1 || bytes_per_px == 3 || bytes_per_px == 12), "strip_buffer_bytes: bad arguments");
    *bytes = mb_strip_buffer_bytes(w, h, bytes_per_px, planes != 0);
    return 0;
}

SSP_API int ssp_blender_feed_strips_begin(ssp_blender *b, int n, const int *rects_xywh, const void *const *imgs_u8c3, const void *const *masks_u8)
{
    SSP_REQUIRE(b && n > 0 && rects_xywh && imgs_u8c3 && masks_u8, "feed_strips_begin: bad arguments");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "feed_strips_begin needs a prepared multiband blender");
    return mb_feed_strips(b, n, rects_xywh, imgs_u8c3, masks_u8, true);
}

SSP_API int ssp_blender_feed_end_pair(ssp_blender *a, ssp_blender *b)
{
    SSP_REQUIRE(a, "feed_end_pair: null argument");
    for (ssp_blender *q : {a, b})
        if (q && (!q->prepared || q->type != SSP_BLEND_MULTIBAND)) SSP_FAIL(SSP_ERR_STATE, "feed_end_pair needs prepared multiband blenders");
    return mb_feed_end_pair(a, b);
}

SSP_API int ssp_blender_order_feeds(ssp_blender *b, const int *keys, int n)
{
    SSP_REQUIRE(b && keys, "order_feeds: null argument");
    if (!b->prepared || b->type != SSP_BLEND_MULTIBAND) SSP_FAIL(SSP_ERR_STATE, "order_feeds needs a prepared multiband blender");
    return mb_order_feeds(b, keys, n);
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
    if (!rc) rc = mb_run_levels(b, res, rm, mo, -1, rect, nullptr, nullptr);
    if (rc) { image_unref(res); image_unref(rm); image_unref(mo); return rc; }
    release_state(b);
    if (result) *result = res;
    if (result_mask) *result_mask = rm;
    if (mosaic) *mosaic = mo;
    return 0;
}
